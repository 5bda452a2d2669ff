"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): discrete actions bit-exact, continuous actions within 1e-4; hidden
states / recurrent state within fp32 tolerance stated per test."""
import pytest
import torch

from lram_amd import init_state_dict, preset
from oracle import dt_ref
from tests.helpers import (Fp64Oracle, assert_actions_match, assert_close_or_as_close_as_fp32_oracle, elem_rel_err,
                           make_inputs, rel_err, relaxed_rows_fraction, relaxed_rows_reset)

pytestmark = pytest.mark.gpu


def _engine(spec, sd, B):
    from lram_amd.engine import Engine
    return Engine(spec, sd, B, device="cuda:0")


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (96, 2192, 512), (300, 80, 1536), (1000, 1408, 512),
                                   (37, 204, 204), (4096, 512, 1024), (5, 8, 4), (3, 2048, 512), (6, 513, 1024),
                                   (8, 80, 1536), (9, 80, 1536)])
def test_gemm_f32_matches_fp64(hip_lib, m, n, k):
    from lram_amd.engine import gemm_f32
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g)
    bias = torch.randn(n, generator=g)
    ref = (a.double() @ w.double().t() + bias.double())
    out = gemm_f32(a.cuda(), w.cuda(), bias.cuda())
    torch.cuda.synchronize()
    # fp32 fma chain over k: error bound ~ k * eps * |a||w|
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-6 * k ** 0.5 * 16, (m, n, k, err)
    # accumulate (residual) form, asymmetric check of row/col mapping: C += A W^T
    base = torch.randn(m, n, generator=g)
    out2 = gemm_f32(a.cuda(), w.cuda(), None, out=base.clone().cuda(), accumulate=True)
    torch.cuda.synchronize()
    ref2 = base.double() + a.double() @ w.double().t()
    assert (out2.cpu().double() - ref2).abs().max().item() < 2e-6 * k ** 0.5 * 16


@pytest.mark.parametrize("m,n,k", [(9, 80, 1536), (96, 2192, 512), (36, 2048, 512), (96, 512, 1024), (192, 2816, 512),
                                   (33, 130, 36), (100, 513, 1344), (24, 1280, 2560), (63, 5120, 1280), (17, 32, 32)])
def test_gemm_few_row_kernel_matches_fp64(hip_lib, m, n, k):
    """gemm_skinny_kernel (9 .. 192 operand rows in the engine): exact fp32 products on the matrix cores, K split over eight
    lane groups (ragged ranges, clamped loads), ragged 32 x 32 tiles, bias and residual (in place) -- same bar as the
    k-ordered fp32 kernel."""
    from lram_amd.engine import gemm_f32
    g = torch.Generator().manual_seed(m * 11 + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g)
    bias = torch.randn(n, generator=g)
    ref = (a.double() @ w.double().t() + bias.double())
    out = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel="skinny")
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-6 * k ** 0.5 * 16, (m, n, k, err)
    base = torch.randn(m, n, generator=g)
    out2 = gemm_f32(a.cuda(), w.cuda(), None, out=base.clone().cuda(), accumulate=True, kernel="skinny")
    torch.cuda.synchronize()
    ref2 = base.double() + a.double() @ w.double().t()
    assert (out2.cpu().double() - ref2).abs().max().item() < 2e-6 * k ** 0.5 * 16
    # a row range of a wider operand (lda > k) and of a wider output (ldc > n), as the engine's slices address them
    wide_a = torch.randn(m, k + 8, generator=g)
    wide_c = torch.zeros(m, n + 4).cuda()
    out3 = gemm_f32(wide_a.cuda()[:, :k], w.cuda(), bias.cuda(), out=wide_c[:, :n], kernel="skinny")
    torch.cuda.synchronize()
    ref3 = wide_a[:, :k].double() @ w.double().t() + bias.double()
    assert (out3.cpu().double() - ref3).abs().max().item() < 2e-6 * k ** 0.5 * 16
    assert float(wide_c[:, n:].abs().max()) == 0.0


@pytest.mark.parametrize("m,n,k", [(3072, 80, 1536), (300, 80, 1536), (257, 33, 256), (1000, 96, 320), (17, 5, 256), (4100, 80, 2560),
                                   (64, 16, 64 * 9), (40, 48, 64 * 5)])
@pytest.mark.parametrize("kern", ["narrow", "narrow16"])
def test_gemm_narrow_output_kernel_matches_fp64(hip_lib, m, n, k, kern):
    """gemm_narrow_kernel / gemm_narrow16_kernel (Mamba's x_proj in the engine: N = 80, K = 1536, from 256 operand rows): exact fp32
    products ("narrow") or f16x2 split products ("narrow16": the engine's default beside f16x2 projections) on the matrix
    cores, 16 rows x all columns per workgroup, K chunks of 64 dealt to 4 waves (fewer chunks than waves, a chunk count that 4
    does not divide), ragged last row block, column counts that are not multiples of 16, bias, a row range of a wider operand
    and of a wider output -- the bar of the k-ordered fp32 kernel."""
    from lram_amd.engine import gemm_f32
    g = torch.Generator().manual_seed(m * 7 + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g)
    bias = torch.randn(n, generator=g)
    if kern == "narrow16":   # rows of varied scale, one all-zero row: what the power-of-two row scales are for
        a = a * torch.exp(torch.randn(m, 1, generator=g))
        a[m // 2] = 0.0
    ref = a.double() @ w.double().t()
    out = gemm_f32(a.cuda(), w.cuda(), None, kernel=kern)
    torch.cuda.synchronize()
    scale = a.double().abs() @ w.double().abs().t() + 1e-300
    if kern == "narrow16":   # the f16x2 bar: within 1.25 x the exact fp32 kernel's error on the same data (sum |a||w| metric)
        e1 = ((gemm_f32(a.cuda(), w.cuda(), None, kernel="narrow").cpu().double() - ref).abs() / scale).max().item()
        e2 = ((out.cpu().double() - ref).abs() / scale).max().item()
        assert e2 < 1.25 * e1, (m, n, k, e2, e1)
        assert float(out[m // 2].abs().max()) == 0.0
    assert ((out.cpu().double() - ref).abs() / scale.clamp_min(1.0)).max().item() < 2e-6 * 16, (m, n, k)
    assert (out.cpu().double() - ref).abs().max().item() < 2e-6 * k ** 0.5 * 16 * max(1.0, float(a.abs().max())), (m, n, k)
    out_b = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel=kern)
    torch.cuda.synchronize()
    assert (out_b.cpu().double() - ref - bias.double()).abs().max().item() < 2e-6 * k ** 0.5 * 16 * max(1.0, float(a.abs().max()))
    assert torch.equal(out_b, gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel=kern))       # deterministic
    wide_a = torch.randn(m, k + 8, generator=g)
    wide_c = torch.zeros(m, n + 4).cuda()
    out3 = gemm_f32(wide_a.cuda()[:, :k], w.cuda(), bias.cuda(), out=wide_c[:, :n], kernel=kern)
    torch.cuda.synchronize()
    ref3 = wide_a[:, :k].double() @ w.double().t() + bias.double()
    assert (out3.cpu().double() - ref3).abs().max().item() < 2e-6 * k ** 0.5 * 16
    assert float(wide_c[:, n:].abs().max()) == 0.0


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (96, 2192, 512), (300, 80, 1536), (1000, 1408, 512),
                                   (4096, 512, 1024), (5, 8, 8), (257, 129, 48)])
def test_gemm_bf16x3_matches_fp64(hip_lib, m, n, k):
    """The projection kernel (3 x bf16 split operands, 6 MFMA products) is fp32-accurate: its error against
    fp64 stays within a small factor of the exact fp32-MFMA kernel's on the same data."""
    from lram_amd.engine import gemm_f32
    g = torch.Generator().manual_seed(m * 11 + n)
    a = torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g))   # rows of varied scale
    w = torch.randn(n, k, generator=g)
    bias = torch.randn(n, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    scale = (a.double().abs() @ w.double().abs().t())                                # sum |a||w| per output
    out3 = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel="bf16x3")
    out1 = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel="f32")
    torch.cuda.synchronize()
    e3 = ((out3.cpu().double() - ref).abs() / scale).max().item()
    e1 = ((out1.cpu().double() - ref).abs() / scale).max().item()
    # e1: the exact fp32 fma-chain kernel on the same data.  No additive slack (round 4 removed it from the f16x2 gate, round 5
    # here): a single 8-deep K tile, where the chain is all but exact, gets the measured factor instead
    # (floor of a third of an fp32 ulp on e1: where the fma chain happens to be exact on a shape the ratio would be against 0)
    assert e3 < (1.5 if k >= 32 else 3.0) * max(e1, 2e-8), (m, n, k, e3, e1)
    base = torch.randn(m, n, generator=g)
    out2 = gemm_f32(a.cuda(), w.cuda(), None, out=base.clone().cuda(), accumulate=True, kernel="bf16x3")
    torch.cuda.synchronize()
    ref2 = base.double() + a.double() @ w.double().t()
    # (accumulating into a base tensor adds one fp32 rounding of |base| ~ 1 per output: 6e-8 on the (scale + 1) metric)
    assert ((out2.cpu().double() - ref2).abs() / (scale + 1)).max().item() < (1.5 if k >= 32 else 3.0) * max(e1, 2e-8) + 6e-8


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (96, 2192, 512), (300, 80, 1536), (1000, 1408, 512),
                                   (4096, 512, 1024), (6144, 3072, 768), (65, 8, 8), (257, 129, 48)])
@pytest.mark.parametrize("spread", ["rows", "elements", "tiny"])
def test_gemm_f16x2_matches_fp64(hip_lib, m, n, k, spread):
    """The f16x2 projection kernel (operand rows scaled by a power of two, split into two binary16 pieces, 3 MFMA
    products, exact un-scaling) is fp32-accurate: its error against fp64 stays within 1.5 x the exact fp32-MFMA kernel's
    on the same data -- for rows of varied scale, for elements spread over many binades inside a row (where the low
    piece of small elements leaves binary16's normal range), and for rows far below binary16's range."""
    from lram_amd.engine import gemm_f32
    g = torch.Generator().manual_seed(m * 13 + n)
    a = torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g))
    w = torch.randn(n, k, generator=g) * torch.exp(torch.randn(n, 1, generator=g) * 0.5)
    if spread == "elements":
        a = a * torch.exp(torch.randn(m, k, generator=g) * 3.0)
        w = w * torch.exp(torch.randn(n, k, generator=g) * 2.0)
    if spread == "tiny":
        a, w = a * 1e-9, w * 1e-7
        a[0] = 0.0                                                                    # an all-zero row
    bias = torch.randn(n, generator=g) * (1e-16 if spread == "tiny" else 1.0)
    ref = a.double() @ w.double().t() + bias.double()
    scale = (a.double().abs() @ w.double().abs().t()) + 1e-300
    out2 = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel="f16x2")
    out1 = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel="f32")
    torch.cuda.synchronize()
    assert torch.isfinite(out2).all()
    e2 = ((out2.cpu().double() - ref).abs() / scale)[1:].max().item()
    e1 = ((out1.cpu().double() - ref).abs() / scale)[1:].max().item()
    # measured (profiles/r04_f16x2_accuracy.txt, scripts/f16x2_accuracy.py): on every shape with K >= 32 the split kernel
    # is CLOSER to fp64 than the fp32 fma chain (0.41 ... 0.97 x its error: it rounds 22-bit products into an fp32
    # accumulator, the chain rounds after every fma); with a single 8-deep K tile the chain is all but exact
    # (e1 ~ 1e-7) and the ratio is 1.4 ... 1.8.  No additive slack: round 3's `+ 1e-7` let it be 2-4 x worse unnoticed.
    assert e2 < (1.25 if k >= 32 else 2.5) * e1, (m, n, k, spread, e2, e1, e2 / e1)
    if spread == "tiny":
        assert torch.equal(out2[0].cpu(), bias)                                        # zero row: bias only, exactly
    base = torch.randn(m, n, generator=g)
    out3 = gemm_f32(a.cuda(), w.cuda(), None, out=base.clone().cuda(), accumulate=True, kernel="f16x2")
    torch.cuda.synchronize()
    ref3 = base.double() + a.double() @ w.double().t()
    # (accumulating into a base tensor adds one fp32 rounding of |base| ~ 1 per output: 6e-8 on the (scale + 1) metric)
    assert ((out3.cpu().double() - ref3).abs() / (scale + 1)).max().item() < (1.25 if k >= 32 else 2.5) * e1 + 6e-8


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (300, 80, 1536), (4096, 512, 1024), (3072, 3072, 768), (6144, 2048, 512),
                                   (65, 8, 64), (257, 129, 96)])
def test_gemm_f16x2_presplit_operands_are_bit_identical(hip_lib, m, n, k):
    """gemm_f16x2p.hip (A split once by its producer -- here the row-split kernel --, both operands staged global -> LDS by
    DMA, MFMA-only loop) forms the same pieces and the same products in the same order as the kernel that splits A while
    staging it: results are bit-identical, for ragged M / N tiles, with bias, and when accumulating into the output."""
    from lram_amd.engine import gemm_f32
    g = torch.Generator().manual_seed(m * 17 + n)
    a = torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g))
    a[m // 2] = 0.0                                                                    # an all-zero row
    w = torch.randn(n, k, generator=g) * torch.exp(torch.randn(n, 1, generator=g) * 0.5)
    bias = torch.randn(n, generator=g)
    want = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel="f16x2")
    got = gemm_f32(a.cuda(), w.cuda(), bias.cuda(), kernel="f16x2p")
    torch.cuda.synchronize()
    assert torch.equal(want, got), float((want - got).abs().max())
    assert torch.equal(got[m // 2].cpu(), bias)
    base = torch.randn(m, n, generator=g).cuda()
    want2 = gemm_f32(a.cuda(), w.cuda(), None, out=base.clone(), accumulate=True, kernel="f16x2")
    got2 = gemm_f32(a.cuda(), w.cuda(), None, out=base.clone(), accumulate=True, kernel="f16x2p")
    torch.cuda.synchronize()
    assert torch.equal(want2, got2)


@pytest.mark.parametrize("tile,stages", [(64, 1), (64, 2), (128, 1), (128, 2), (256, 0)])
def test_gemm_f16x2_presplit_every_tile_is_bit_identical(hip_lib, monkeypatch, tile, stages):
    """The pre-split kernel's workgroup tiles (64 x 128, 128 x 128) in their one- and two-stage forms: every wave runs the same
    K-ordered product sequence whatever the tile, so each form equals the on-the-fly-split kernel bit for bit -- ragged tiles on
    both edges, a single K tile, bias, accumulation.  (Round 5's 256 x 128 / 256 x 256 tiles and 3 / 4 stage rings passed the
    same test before they were removed for being no faster: profiles/r05_gemm_presplit_*_sweep.txt.)"""
    from lram_amd.engine import gemm_f32
    monkeypatch.setenv("LRAM_GEMM_TILE", str(tile))
    monkeypatch.setenv("LRAM_F16P_STAGES", str(stages))
    # (tile 256 = the 8-phase kernel of gemm_f16x2_8p.hip: 256 x 256 outputs per workgroup, 8 staggered waves, a two-tile LDS
    # ring filled six phases ahead with counted waits -- one, two, three and an odd number of K tiles, K splits, ragged edges)
    for m, n, k in [(257, 129, 96), (300, 80, 1536), (1000, 700, 32), (3072, 3072, 768), (6144, 2048, 512), (513, 300, 64),
                    (700, 520, 160), (768, 1280, 2560)]:
        g = torch.Generator().manual_seed(m * 13 + n + tile)
        a = (torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g))).cuda()
        w = (torch.randn(n, k, generator=g) * torch.exp(torch.randn(n, 1, generator=g) * 0.5)).cuda()
        bias = torch.randn(n, generator=g).cuda()
        want = gemm_f32(a, w, bias, kernel="f16x2")
        got = gemm_f32(a, w, bias, kernel="f16x2p")
        torch.cuda.synchronize()
        assert torch.equal(want, got), (m, n, k, float((want - got).abs().max()))
        base = torch.randn(m, n, generator=g).cuda()
        want2 = gemm_f32(a, w, None, out=base.clone(), accumulate=True, kernel="f16x2")
        got2 = gemm_f32(a, w, None, out=base.clone(), accumulate=True, kernel="f16x2p")
        torch.cuda.synchronize()
        assert torch.equal(want2, got2), (m, n, k)


@pytest.mark.parametrize("panel", [1, 5, 6, 99])
def test_gemm_tile_order_is_a_bijection_on_the_device(hip_lib, monkeypatch, panel):
    """The workgroup id -> output tile map (XCD-aware: a 2-D split of the tile grid over the 8 L2s where one fits, else column
    panels; csrc/common.h gemm_tile_of) only renames workgroups: every order gives the same bits as every other, on tile grids
    8 divides and on ones it does not, with a ragged last panel, for both f16x2 kernels."""
    from lram_amd.engine import gemm_f32
    outs = {}
    for order in (None, panel):
        if order is None:
            monkeypatch.delenv("LRAM_GEMM_PANEL", raising=False)
        else:
            monkeypatch.setenv("LRAM_GEMM_PANEL", str(order))
        for m, n, k in [(1536, 5120, 1280), (700, 900, 64), (3072, 3072, 768), (130, 1000, 96)]:
            g = torch.Generator().manual_seed(m + n)
            a = torch.randn(m, k, generator=g).cuda()
            w = (torch.randn(n, k, generator=g) * 0.05).cuda()
            for kern in ("f16x2", "f16x2p"):
                outs[(order, m, n, kern)] = gemm_f32(a, w, kernel=kern)
    torch.cuda.synchronize()
    for (order, m, n, kern), got in outs.items():
        if order is not None:
            assert torch.equal(got, outs[(None, m, n, kern)]), (order, m, n, kern)
            assert torch.equal(got, outs[(None, m, n, "f16x2")]), (order, m, n, kern)


def test_gemm_8phase_kernel_race_screen(hip_lib):
    """The 8-phase kernel orders its LDS-DMA against its fragment reads by counted vmcnt waits and raw barriers only
    (gemm_f16x2_8p.hip header): an early read or an early re-fill would show as rare wrong tiles that come and go with timing.
    200 launches over shapes with 1 ... 80 K tiles and 1 ... 288 workgroups, each compared bit for bit with the first result
    of the on-the-fly-split kernel, with a bandwidth hog running beside half of them to move the DMA timing."""
    from lram_amd.engine import gemm_f32, stream_copy
    hog_src = torch.empty(64 * 1024 * 1024, device="cuda")
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    for m, n, k in [(3072, 3072, 768), (6144, 3072, 96), (300, 300, 2560), (1024, 512, 32), (2048, 2192, 512)]:
        g = torch.Generator().manual_seed(m + n + k)
        a = (torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g))).cuda()
        w = (torch.randn(n, k, generator=g) * 0.05).cuda()
        want = gemm_f32(a, w, kernel="f16x2")
        torch.cuda.synchronize()
        for rep in range(40):
            if rep % 2:
                with torch.cuda.stream(side):
                    stream_copy(hog_dst, hog_src)
            got = gemm_f32(a, w, kernel="f16x2p8")
            assert torch.equal(want, got), (m, n, k, rep, float((want - got).abs().max()))
        torch.cuda.synchronize()


def _run_parity(name, B, steps, seed=0, discrete=False, graph=False, hidden_tol=2e-4, state_tol=2e-4, spec=None,
                sd=None, cond_aware=False, scheme="exercise", reset_prob=0.15, obs_gain=None, pooled=False):
    """scheme: weight distribution (lram_amd/weights.py::init_state_dict); reset_prob: per-env, per-step restart probability;
    obs_gain: (channel, factor) scales one observation channel (un-normalised outlier).
    cond_aware: where the engine is further than the tolerance from the fp32 oracle, accept it if it is as close to
    the float64 evaluation as the fp32 oracle itself is (tests/helpers.py::Fp64Oracle) -- deep stacks meet inputs that
    are ill-conditioned for any fp32 evaluation."""
    spec = preset(name) if spec is None else spec
    sd = init_state_dict(spec, seed=seed, scheme=scheme) if sd is None else sd
    eng = _engine(spec, sd, B)
    if graph:
        eng.set_graph_mode(True)
    ora = dt_ref.OraclePolicy(spec, sd)
    o64 = Fp64Oracle(spec, sd) if cond_aware else None
    ties = 0
    relaxed_rows_reset()
    worst_elem = 0.0
    # fixed device buffers so that graph mode sees stable pointers
    d_obs = torch.empty(B, spec.state_dim, device="cuda:0")
    d_rtg = torch.empty(B, device="cuda:0")
    d_rew = torch.empty(B, device="cuda:0")
    d_mask = torch.empty(B, dtype=torch.uint8, device="cuda:0")
    for t, (obs, rtg, rew, mask) in enumerate(make_inputs(spec, B, steps, seed=1234 + seed, reset_prob=reset_prob)):
        if obs_gain is not None:
            obs[:, obs_gain[0]] *= obs_gain[1]
        d_obs.copy_(obs), d_rtg.copy_(rtg), d_rew.copy_(rew), d_mask.copy_(mask)
        a_gpu, tok = eng.step(d_obs, d_rtg, d_rew, d_mask, discrete=discrete)
        a_ref, dbg = ora.step(obs, rtg, rew, mask, discrete=discrete, return_debug=True)
        torch.cuda.synchronize()
        tokens, hidden, logits = eng.taps()
        assert rel_err(tokens, dbg["tokens"]) < 1e-5, f"{name} step {t}: embed tokens"
        if cond_aware:
            _, d64 = o64.step(obs, rtg, rew, mask, discrete=discrete, return_debug=True)
            assert_close_or_as_close_as_fp32_oracle(hidden, dbg["hidden"], d64["hidden"], tol=hidden_tol,
                                                    what=f"{name} step {t}: hidden", pooled=pooled)
        else:
            assert rel_err(hidden, dbg["hidden"]) < hidden_tol, f"{name} step {t}: hidden {rel_err(hidden, dbg['hidden'])}"
        a_cmp = a_gpu[:, :1] if discrete else a_gpu
        ties += assert_actions_match(a_cmp, a_ref, dbg["logits"], spec, discrete, what=f"{name} step {t}")
    # final recurrent state against the oracle's, in the reference's past_key_values layout
    pkv = eng.export_past_key_values()
    if spec.backbone == "mamba":
        for i in range(spec.n_blocks):
            assert rel_err(pkv[i][0], ora.state[i][0]) < state_tol
            assert rel_err(pkv[i][1], ora.state[i][1]) < state_tol
    else:
        def close(got, want, want64, what):
            if cond_aware:   # per env: flatten everything behind the env axis
                nb = want.shape[0]
                assert_close_or_as_close_as_fp32_oracle(got.reshape(nb, 1, -1), want.reshape(nb, 1, -1),
                                                        want64.reshape(nb, 1, -1), tol=state_tol, what=what, pooled=pooled)
            else:
                assert rel_err(got, want) < state_tol, what
        for i in range(spec.n_blocks):
            blk, ref = pkv[f"block_{i}"], ora.state[f"block_{i}"]
            r64 = o64.ora.state[f"block_{i}"] if cond_aware else ref
            close(blk["conv_state"][0], ref["conv_state"][0], r64["conv_state"][0], f"conv {i}")
            if "mlstm_state" in blk:
                for j in range(3):
                    close(blk["mlstm_state"][j], ref["mlstm_state"][j], r64["mlstm_state"][j], f"mlstm state {i}.{j}")
                if not cond_aware:
                    # C and n entries span orders of magnitude: besides max-error / max-value, every entry must be right
                    # relative to ITSELF (entries below 1e-3 of the largest are measured against that floor)
                    for j in range(2):
                        e = elem_rel_err(blk["mlstm_state"][j], ref["mlstm_state"][j])
                        worst_elem = max(worst_elem, e)
                        assert e < ELEM_STATE_TOL, f"mlstm state {i}.{j}: per-element relative error {e:.2e}"
            else:   # [4, B, D] -> env-major
                close(blk["slstm_state"].transpose(0, 1), ref["slstm_state"].transpose(0, 1),
                      r64["slstm_state"].transpose(0, 1), f"slstm state {i}")
    eng.close()
    # the fp64-aware escape hatch must stay the exception: at most 5 % of the compared rows may need it
    assert relaxed_rows_fraction() <= 0.05, f"{name}: {relaxed_rows_fraction():.1%} of the rows needed the fp64 rule"
    import os
    if os.environ.get("LRAM_TEST_REPORT"):
        print(f"[report] {name} B={B}: worst per-element state error {worst_elem:.2e}, relaxed rows "
              f"{relaxed_rows_fraction():.2%}")
    return ties


ELEM_STATE_TOL = 5e-3   # per-element |err| / (|ref| + 1e-3 max|ref|) of C / n after a trajectory; measured worst 8.9e-4 (16M, B = 12)


def test_xlstm_tiny_trajectory(hip_lib):
    assert _run_parity("xlstm_tiny", B=8, steps=12) == 0


def test_xlstm_c1_b32(hip_lib):
    # BASELINE config 1: xLSTM[1:0] 2-layer d_model=128, batch 32
    assert _run_parity("xlstm_c1", B=32, steps=16) == 0


def test_xlstm_16m_shapes(hip_lib):
    # BASELINE config 2 shapes (xLSTM[7:1] 16M) at an oracle-sized batch
    assert _run_parity("xlstm_16m", B=12, steps=6) == 0


def test_xlstm_206m_shapes_two_blocks(hip_lib):
    # config 4/5 geometry (D=1280, inner 2560, DH 640, sLSTM head dim 320) cut to 3 blocks to stay small
    from lram_amd.config import ModelSpec
    spec = ModelSpec(backbone="xlstm", d_model=1280, n_blocks=3, slstm_at=[1])
    assert _run_parity("xlstm_206m_cut", B=3, steps=4, spec=spec) == 0


@pytest.mark.parametrize("d_model,n_blocks,B", [(128, 3, 11), (512, 3, 40), (768, 2, 19), (1024, 2, 17), (1280, 3, 21)])
def test_slstm_token_kernel_matches_oracle_and_the_gemm_path(hip_lib, monkeypatch, d_model, n_blocks, B):
    """Slices of 9 .. 256 envs run the sLSTM recurrence as ONE launch per token (slstm_token_kernel: recurrent projection in
    exact fp32 + pointwise cell; sLSTM head dims 32 / 128 / 192 / 256 / 320 here, ragged 16-env tiles) instead of a batched
    GEMM plus the pointwise kernel (LRAM_SLSTM_FUSED_ROWS=0): both against the oracle, and against each other after
    env-steps with resets, a 1-token encoder step (the launch that reads the state's h plane must not write it) and a
    9-timestep prefill (12-token passes)."""
    from lram_amd.config import ModelSpec
    from lram_amd.engine import Engine
    spec = ModelSpec(backbone="xlstm", d_model=d_model, n_blocks=n_blocks, slstm_at=[1]) if d_model != 128 else preset("xlstm_tiny")
    for rows in ("256", "0"):
        monkeypatch.setenv("LRAM_SLSTM_FUSED_ROWS", rows)
        assert _run_parity(f"slstm_tok_{d_model}_{rows}", B=B, steps=4, spec=spec) == 0
    sd = init_state_dict(spec, seed=7)
    seq = make_inputs(spec, B, 9, seed=21, reset_prob=0.2)
    engines = {}
    for rows in ("256", "0"):
        monkeypatch.setenv("LRAM_SLSTM_FUSED_ROWS", rows)
        engines[rows] = Engine(spec, sd, B, device="cuda:0")
    for obs, rtg, rew, mask in seq[:5]:
        a1, _ = engines["256"].step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        a0, _ = engines["0"].step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        assert float((a0 - a1).abs().max()) <= 1e-4
    x = torch.randn(B, 1, spec.d_model, generator=torch.Generator().manual_seed(3)).cuda()
    for _ in range(2):   # two single-token passes: the second reads the h plane the first one left
        y1, y0 = engines["256"].encoder_step(x), engines["0"].encoder_step(x)
        assert rel_err(y1, y0) < 1e-5
    obs_seq = torch.stack([t[0] for t in seq], 1).contiguous().cuda()
    rtg_seq = torch.stack([t[1] for t in seq], 1).contiguous().cuda()
    rew_seq = torch.stack([t[2] for t in seq], 1).contiguous().cuda()
    p1, _ = engines["256"].prefill(obs_seq, rtg_seq, rew_seq)
    p0, _ = engines["0"].prefill(obs_seq, rtg_seq, rew_seq)
    torch.cuda.synchronize()
    assert float((p0 - p1).abs().max()) <= 1e-4
    assert rel_err(engines["256"].export_state_tensor(1, 0), engines["0"].export_state_tensor(1, 0)) < 1e-5
    for e in engines.values():
        e.close()


@pytest.mark.parametrize("d_model,slstm_at", [(704, [1]), (1064, []), (1432, []), (1792, [1])])
def test_reference_half_presets_geometry(hip_lib, d_model, slstm_at):
    """The widths of the reference's xlstm_*_half presets (configs/agent_params/huggingface): head dims 352 / 544 / 720
    (not multiples of 64: 16-column cell slices) and 896 with inner 3584; two blocks each, with an sLSTM block where
    d_model / 4 is a multiple of 4."""
    from lram_amd.config import ModelSpec
    spec = ModelSpec(backbone="xlstm", d_model=d_model, n_blocks=2, slstm_at=slstm_at)
    assert spec.head_dim in (352, 544, 720, 896)
    assert _run_parity(f"half_{d_model}", B=3, steps=4, spec=spec) == 0
    # stored contexts fall back to the token-sequential kernels on these head dims
    sd = init_state_dict(spec, seed=0)
    eng = _engine(spec, sd, 2)
    ora = dt_ref.OraclePolicy(spec, sd)
    seq = make_inputs(spec, 2, 6, seed=3, reset_prob=0.0)
    obs = torch.stack([x[0] for x in seq], 1).cuda()
    rtg = torch.stack([x[1] for x in seq], 1).cuda()
    act, _ = eng.prefill(obs.contiguous(), rtg.contiguous(), torch.zeros(2, 6, device="cuda"),
                         torch.ones(2, dtype=torch.uint8, device="cuda"))
    for t, x in enumerate(seq):
        ref, dbg = ora.step(x[0], x[1], x[2], x[3] if t == 0 else None, return_debug=True)
    torch.cuda.synchronize()
    assert assert_actions_match(act, ref, dbg["logits"], spec, what=f"half_{d_model} prefill") == 0
    eng.close()


@pytest.mark.parametrize("d_model", [512, 1792])
def test_reference_mamba_preset_widths(hip_lib, d_model):
    """mamba_medium (d_model 512, dt_rank 32) and mamba_huge_half (1792, d_inner 3584, dt_rank 112), two layers each."""
    from lram_amd.config import ModelSpec
    spec = ModelSpec(backbone="mamba", kind="MDDMamba", d_model=d_model, n_blocks=2)
    assert _run_parity(f"mamba_{d_model}", B=3, steps=4, spec=spec) == 0


def test_mamba_tiny_trajectory(hip_lib):
    assert _run_parity("mamba_tiny", B=8, steps=12) == 0


def test_mamba_48m_shapes(hip_lib):
    assert _run_parity("mamba_48m", B=6, steps=4) == 0


@pytest.mark.parametrize("scheme", ["reference", "trained_like"])
@pytest.mark.parametrize("name,B,steps", [("xlstm_16m", 12, 40), ("mamba_48m", 6, 24)])
def test_weight_distributions_the_reference_actually_runs(hip_lib, scheme, name, B, steps):
    """Live oracle parity on the two weight distributions every other test leaves out (they all use scheme="exercise", which
    forgets within tens of steps): "reference" = what `post_init -> reset_parameters` leaves in a freshly built model
    (src/algos/models/decision_xlstm.py:170-171,210-213: mLSTM forget bias linspace(3, 6), zero gate weights, R = 0, N(0, 0.02)
    projections; Mamba A_log = log(1..16), dt bias from log-uniform [1e-3, 1e-1]); "trained_like" = the long-memory corner of a
    loaded checkpoint (src/algos/decision_transformer_sb3.py:1120-1184): f ~ 0.95-0.998, input-gate pre-activations of +-15,
    one observation channel 30 x the rest, Mamba dt bias at both ends of its range, A_log up to log 16 + 2.  Few resets, so
    the state integrates (the 1000-step horizon on these distributions: tests/test_gpu_horizon.py)."""
    gain = (3, 30.0) if scheme == "trained_like" else None
    ties = _run_parity(name, B=B, steps=steps, scheme=scheme, reset_prob=0.03, obs_gain=gain, cond_aware=True, pooled=True)
    # (a "tie" = an action that differs where the ORACLE's own top-2 logits are within 2e-4 of each other -- the stated rule.  The
    # ill-conditioned long-memory regime, where both fp32 evaluations sit 1e-4 ... 8e-4 from float64, produces a handful in
    # 12 envs x 40 steps x 8 action dims = 3840 choices; the fresh-model distribution none)
    assert ties <= (4 if scheme == "trained_like" else 0), ties


@pytest.mark.parametrize("B", [77, 130])
def test_mamba_lane_state_update_at_ragged_env_counts(hip_lib, B):
    """From 64 env slots Mamba-48M's selective state update runs lane = channel (mamba_ssm_lane_kernel: 8 env slots per wave,
    the 4 KB state block of a wave moved coalesced and transposed through LDS, softplus / SiLU / decay on the hardware
    transcendentals).  Env counts that are not multiples of 8 leave a short last wave, make_inputs' resets take the wave-uniform
    zero-state branch; hidden states, actions and every layer's conv / ssm state against the oracle at the usual bars."""
    from lram_amd.config import ModelSpec
    spec = ModelSpec(backbone="mamba", kind="MDDMamba", d_model=768, n_blocks=3)  # Mamba-48M's widths (d_inner 1536, dt_rank 48)
    assert spec.d_inner == 1536 and spec.dt_rank in (0, 48)
    assert _run_parity(f"mamba_768_b{B}", B=B, steps=5, spec=spec) == 0


@pytest.mark.parametrize("scheme", ["reference", "trained_like"])
@pytest.mark.parametrize("name,B,steps,discrete", [("xlstm_c1", 32, 24, False), ("xlstm_tiny", 16, 24, True), ("mamba_tiny", 8, 24, False)])
def test_small_configurations_on_the_weight_distributions_the_reference_runs(hip_lib, scheme, name, B, steps, discrete):
    """BASELINE config 1 (xLSTM[1:0], 2 layers, D = 128, 32 envs: the materialised small-batch kernels, few-row projections), the
    18-way discrete head, and the tiny Mamba stack on the fresh-model and trained-like weight distributions (as
    test_weight_distributions_the_reference_actually_runs does for the 16M / 48M stacks)."""
    gain = (3, 30.0) if scheme == "trained_like" else None
    ties = _run_parity(name, B=B, steps=steps, scheme=scheme, reset_prob=0.03, obs_gain=gain, cond_aware=True, pooled=True,
                       discrete=discrete)
    assert ties <= (2 if scheme == "trained_like" else 0), ties


def test_mamba_x_proj_narrow_kernel_against_the_tile_gemm_path(hip_lib, monkeypatch):
    """x_proj through the narrow-output kernel (default from 256 operand rows: one launch; f16x2 split products fed with the conv
    kernel's row maxima, or exact fp32 with LRAM_GEMM_NARROW=2) and through the f16x2 tile GEMM + split-K reduce it replaced
    (LRAM_GEMM_NARROW=0): all meet the oracle bars at 130 envs (390 rows, ragged 16-row blocks) and agree with each other to fp32
    rounding."""
    from lram_amd.engine import Engine
    spec = preset("mamba_48m")
    sd = init_state_dict(spec, seed=0)
    B = 130
    seq = make_inputs(spec, B, 3, seed=77)
    outs = {}
    for on in ("1", "2", "0"):
        monkeypatch.setenv("LRAM_GEMM_NARROW", on)
        eng = Engine(spec, sd, B, device="cuda:0")
        for obs, rtg, rew, mask in seq:
            a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        torch.cuda.synchronize()
        _, hidden, logits = eng.taps()
        counts = eng.gemm_counts()
        outs[on] = (a.clone(), hidden.clone(), counts)
        eng.close()
    # (the exact-fp32 form is counted with the fp32 family, the f16x2 form with the f16x2 family)
    assert outs["2"][2]["f32"]["launches"] >= 3 * spec.n_blocks > outs["0"][2]["f32"]["launches"]
    assert outs["1"][2]["f32"]["launches"] == outs["0"][2]["f32"]["launches"]
    assert rel_err(outs["1"][1], outs["0"][1]) < 1e-5 and rel_err(outs["2"][1], outs["0"][1]) < 1e-5
    ora = dt_ref.OraclePolicy(spec, sd)
    for obs, rtg, rew, mask in seq:
        a_ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
    for on in ("1", "2", "0"):
        assert rel_err(outs[on][1], dbg["hidden"]) < 2e-4, on
        assert_actions_match(outs[on][0], a_ref, dbg["logits"], spec, what=f"narrow={on}")


@pytest.mark.parametrize("name,B", [("mamba_48m", 6), ("mamba_tiny", 7)])
def test_mamba_dt_proj_as_its_own_gemm_stays_correct(hip_lib, monkeypatch, name, B):
    """Default: dt_proj (K = dt_rank) is evaluated inside the selective-state-update kernel (d_state 16, dt_rank <= 64; every
    other Mamba test runs that).  LRAM_MAMBA_DT_FUSE=0 keeps it a projection launch -- the path wider presets (dt_rank 112)
    take: same parity bars, and the two agree step by step and after a 9-timestep prefill (12-token state passes)."""
    from lram_amd.engine import Engine
    monkeypatch.setenv("LRAM_MAMBA_DT_FUSE", "0")
    assert _run_parity(name, B=B, steps=4) == 0
    spec = preset(name)
    sd = init_state_dict(spec, seed=5)
    seq = make_inputs(spec, B, 9, seed=11, reset_prob=0.15)
    engines = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("LRAM_MAMBA_DT_FUSE", fuse)
        engines[fuse] = Engine(spec, sd, B, device="cuda:0")
    for obs, rtg, rew, mask in seq:
        a0, _ = engines["0"].step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        a1, _ = engines["1"].step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        assert float((a0 - a1).abs().max()) <= 1e-4
    obs_seq = torch.stack([x[0] for x in seq], 1).contiguous().cuda()
    rtg_seq = torch.stack([x[1] for x in seq], 1).contiguous().cuda()
    rew_seq = torch.stack([x[2] for x in seq], 1).contiguous().cuda()
    ones = torch.ones(B, dtype=torch.uint8).cuda()
    p0, _ = engines["0"].prefill(obs_seq, rtg_seq, rew_seq, reset_mask=ones)
    p1, _ = engines["1"].prefill(obs_seq, rtg_seq, rew_seq, reset_mask=ones)
    torch.cuda.synchronize()
    assert float((p0 - p1).abs().max()) <= 1e-4
    for blk in (0, spec.n_blocks - 1):
        for which in (0, 3):
            assert rel_err(engines["0"].export_state_tensor(blk, which), engines["1"].export_state_tensor(blk, which)) < 1e-5
    for e in engines.values():
        e.close()


@pytest.mark.parametrize("kind", ["f16x2", "bf16x3", "f32"])
@pytest.mark.parametrize("name,B,steps", [("xlstm_16m", 12, 4), ("mamba_48m", 6, 4)])
def test_every_projection_kernel_kind_meets_the_parity_bars(hip_lib, monkeypatch, kind, name, B, steps):
    """LRAM_GEMM selects the projection kernel at lram_finalize.  The dispatcher keeps bf16x3 below 1024 operand rows, so at
    oracle-sized batches the default build would never run f16x2: LRAM_F16_MIN_ROWS=9 forces it here (its K tails, split-K,
    producer-handed row maxima and the standalone row-maximum launch included), and the two alternatives stay covered."""
    monkeypatch.setenv("LRAM_GEMM", kind)
    monkeypatch.setenv("LRAM_F16_MIN_ROWS", "9")
    assert _run_parity(name, B=B, steps=steps) == 0


def test_discrete_head_bit_exact(hip_lib):
    assert _run_parity("xlstm_tiny", B=16, steps=8, seed=3, discrete=True) == 0


def test_graph_replay_matches(hip_lib):
    assert _run_parity("xlstm_tiny", B=8, steps=10, seed=5, graph=True) == 0


@pytest.mark.parametrize("name", ["xlstm_tiny", "mamba_tiny"])
def test_micro_batch_pipeline_matches_single_slice(hip_lib, name):
    """Env slices on separate streams (cell kernels serialised on their own stream; Mamba: free-running slices, slice j
    enqueued j stages behind slice 0) == one slice, eager and under
    hipGraph capture, including ragged splits.  At this tiny batch the per-slice kernel choices (GEMV vs tile GEMM,
    split-K, cell column slicing) differ with the slice size, so equality is to fp32 rounding; at 4096 envs it is
    bit for bit (tests/test_gpu_fullsize.py)."""
    from lram_amd.engine import Engine
    spec = preset(name)
    sd = init_state_dict(spec, seed=13)
    B = 11
    seq = make_inputs(spec, B, 6, seed=99)
    outs = {}
    for n, graph in ((1, False), (2, False), (3, False), (4, True)):
        eng = Engine(spec, sd, B, device="cuda:0")
        eng.set_micro_batches(n)
        eng.set_graph_mode(graph)
        d = [torch.empty_like(t).cuda() for t in seq[0]]
        acts, hids = [], []
        for inp in seq:
            for dst, src in zip(d, inp):
                dst.copy_(src)
            a, _ = eng.step(*d)
            torch.cuda.synchronize()
            acts.append(a.clone())
            hids.append(eng.taps()[1])
        x = torch.randn(B, 2, spec.d_model, generator=torch.Generator().manual_seed(1)).cuda()
        enc = eng.encoder_step(x)
        torch.cuda.synchronize()
        outs[n] = (torch.stack(acts), torch.stack(hids), enc.clone(), eng.export_state_tensor(0, 0))
        eng.close()
    for n in (2, 3, 4):
        assert torch.equal(outs[1][0], outs[n][0]), n                 # actions (argmax'ed) identical
        for a, b in zip(outs[1][1:], outs[n][1:]):
            assert rel_err(a, b) < 1e-4, n


def test_rms_norm_and_ln_bias_variants(hip_lib):
    from lram_amd.config import ModelSpec
    for kw in (dict(rms_norm=True), dict(ln_bias=True)):
        spec = ModelSpec(backbone="xlstm", d_model=128, n_blocks=3, slstm_at=[2], state_dim=20, act_dim=4, **kw)
        assert _run_parity("variant", B=4, steps=5, spec=spec) == 0


def test_encoder_step_operator(hip_lib):
    """`self.encoder(inputs_embeds, use_cache=True)` plug point, T = 1..4 tokens per call."""
    from oracle import xlstm_ref
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=7)
    B = 5
    eng = _engine(spec, sd, B)
    state = None
    g = torch.Generator().manual_seed(9)
    for T in (3, 1, 4, 2):
        x = torch.randn(B, T, spec.d_model, generator=g)
        ref, state = xlstm_ref.encoder_forward_cached(spec, {k: v for k, v in sd.items()}, x, state)
        out = eng.encoder_step(x.cuda())
        torch.cuda.synchronize()
        assert rel_err(out, ref) < 2e-4, T
    eng.close()


def test_state_import_export_roundtrip_and_reset(hip_lib):
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=11)
    B = 4
    eng = _engine(spec, sd, B)
    seq = make_inputs(spec, B, 3)
    for obs, rtg, rew, mask in seq:
        eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
    pkv = eng.export_past_key_values()
    a1, _ = eng.step(seq[0][0].cuda(), seq[0][1].cuda(), seq[0][2].cuda(), None)
    a1 = a1.clone()
    eng.import_past_key_values(pkv)
    a2, _ = eng.step(seq[0][0].cuda(), seq[0][1].cuda(), seq[0][2].cuda(), None)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2)
    # masked reset == fresh engine for those envs
    m = torch.tensor([1, 0, 1, 0], dtype=torch.uint8).cuda()
    eng.reset(m)
    st = eng.export_past_key_values()
    for blk in st.values():
        for key, val in blk.items():
            for t in (val if isinstance(val, tuple) else (val,)):
                rows = t[:, [0, 2]] if key == "slstm_state" else t[[0, 2]]
                assert float(rows.abs().max()) == 0.0
    eng.close()


def test_slstm_hidden_plane_import_is_range_checked(hip_lib, monkeypatch):
    """The f16x2 form of the sLSTM step keeps h in LDS as binary16 planes of 2^12 h (fine for |h| < 1, which is all the
    recurrence produces): lram_state_import refuses a foreign hidden plane with |h| >= 16 or NaN where that form is active,
    accepts it where the exact-fp32 recurrence runs (LRAM_SLSTM_SEQ=2), and always accepts what an engine exported."""
    from lram_amd.engine import Engine, LramError
    spec = preset("xlstm_16m")
    sd = init_state_dict(spec, seed=0)
    B = 1024                                   # (f16x2 projections, hence the f16x2 step form, from 1024 operand rows)
    eng = Engine(spec, sd, B, device="cuda:0")
    for obs, rtg, rew, mask in make_inputs(spec, B, 2):
        eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
    blk = spec.slstm_at[0]
    good = eng.export_state_tensor(blk, 0).clone()
    assert float(good[0].abs().max()) < 1.0
    eng.import_state_tensor(blk, 0, good)      # what the model produced: fine
    for poison in (20.0, float("nan")):
        bad = good.clone()
        bad[0, 17, 5] = poison
        with pytest.raises(LramError, match="sLSTM hidden plane"):
            eng.import_state_tensor(blk, 0, bad)
    bad = good.clone()
    bad[1, 17, 5] = 1e6                        # the c / n / m planes stay fp32 in registers: no limit on them
    eng.import_state_tensor(blk, 0, bad)
    eng.close()
    monkeypatch.setenv("LRAM_SLSTM_SEQ", "2")  # exact-fp32 recurrence: no range limit, no check
    eng = Engine(spec, sd, B, device="cuda:0")
    bad = good.clone()
    bad[0, 17, 5] = 20.0
    eng.import_state_tensor(blk, 0, bad)
    a, _ = eng.step(*[v.cuda() for v in make_inputs(spec, B, 1)[0][:3]], None)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(a).all())
    eng.close()


def test_sampled_profile_counts_every_entry_point(hip_lib):
    """lram_profile_begin_sampled(n): every public entry that launches the stack (lram_step, lram_prefill, lram_encoder_step) is
    one call of the sample, timed or not by ITS position -- not by whatever lram_step ran before it."""
    from lram_amd.engine import Engine
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=0)
    B = 8
    eng = Engine(spec, sd, B, device="cuda:0")
    n_rec = spec.n_blocks - len(spec.slstm_at)
    obs, rtg, rew, _ = [v.cuda() if v is not None else None for v in make_inputs(spec, B, 1)[0]]
    x = torch.randn(B, 1, spec.d_model, device="cuda:0")
    seq = (obs.unsqueeze(1).repeat(1, 2, 1).contiguous(), rtg.unsqueeze(1).repeat(1, 2).contiguous(),
           rew.unsqueeze(1).repeat(1, 2).contiguous())
    def launches(fn):
        eng.profile_begin()
        fn()
        torch.cuda.synchronize()
        return eng.profile_end_split()[1]

    n_step, n_enc, n_pre = launches(lambda: eng.step(obs, rtg, rew, None)), launches(lambda: eng.encoder_step(x)), \
        launches(lambda: eng.prefill(*seq))
    assert min(n_step, n_enc, n_pre) >= n_rec
    eng.profile_begin_sampled(2)
    eng.step(obs, rtg, rew, None)          # call 0: timed
    eng.encoder_step(x)                    # call 1: not timed
    eng.encoder_step(x)                    # call 2: timed
    eng.step(obs, rtg, rew, None)          # call 3: not timed
    eng.prefill(*seq)                      # call 4: timed
    torch.cuda.synchronize()
    _, n_main, _, _ = eng.profile_end_split()
    assert n_main == n_step + n_enc + n_pre, (n_main, n_step, n_enc, n_pre)
    eng.profile_begin_sampled(2)
    eng.encoder_step(x)                    # call 0 of a new sample: timed, whatever the last step of the old one was
    torch.cuda.synchronize()
    _, n_one, _, _ = eng.profile_end_split()
    assert n_one > 0
    eng.close()


def test_errors_are_loud(hip_lib):
    from lram_amd.engine import Engine, LramError
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=0)
    bad = dict(sd)
    del bad["encoder.layers.blocks.0.xlstm.proj_up.weight"]
    with pytest.raises(KeyError):
        Engine(spec, bad, 2, device="cuda:0")
    eng = Engine(spec, sd, 2, device="cuda:0")
    with pytest.raises(ValueError):
        eng.step(torch.zeros(3, spec.state_dim).cuda(), torch.zeros(2).cuda(), torch.zeros(2).cuda())
    eng.close()


@pytest.mark.parametrize("name,B", [("xlstm_16m", 400), ("mamba_48m", 352), ("xlstm_206m", 176)])
def test_presplit_projection_operands_change_nothing(hip_lib, name, B, monkeypatch):
    """LRAM_GEMM_PRESPLIT (default on): the norms ahead of proj_up / in_proj write the f16x2 GEMM's operand planes instead of
    fp32 rows + row maxima and the projection runs on gemm_f16x2p.hip.  Same pieces, same products, same order: the engine's
    actions, hidden states and recurrent state are BIT-identical with the knob off (>= 1024 operand rows per launch; the 206M
    stack's wide weights take the f16x2 kernels from 512 rows -- the other clause of the shared predicate `f16x2_rows`)."""
    from lram_amd.engine import Engine
    spec = preset(name)
    sd = init_state_dict(spec, seed=91)
    seq = make_inputs(spec, B, 3, seed=92, reset_prob=0.1)
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("LRAM_GEMM_PRESPLIT", knob)
        eng = Engine(spec, sd, B, device="cuda:0")
        eng.set_micro_batches(1)
        eng.gemm_counts(reset=True)
        acts = []
        for obs, rtg, rew, mask in seq:
            a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
            acts.append(a.clone())
        torch.cuda.synchronize()
        ran = eng.gemm_counts()
        assert ran["f16x2"]["launches"] > 0                      # the big projections did take the f16x2 family
        _, hidden, _ = eng.taps()
        outs[knob] = (torch.stack(acts), hidden.clone(), eng.export_state_tensor(spec.n_blocks - 1, 0).clone())
        eng.close()
    monkeypatch.delenv("LRAM_GEMM_PRESPLIT")
    for x, y in zip(outs["1"], outs["0"]):
        assert torch.equal(x, y)


