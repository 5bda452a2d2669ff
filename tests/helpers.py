"""Shared helpers of the parity tests (oracle = checker, engine = thing under test)."""
import torch


def make_inputs(spec, B, steps, seed=1234, reset_prob=0.15, image=False):
    """Seeded trajectory inputs: obs U(-1,1) (zero beyond a random native dim, as after pad_inputs),
    rtg decreasing by 1/scale per step, reward token 0, random per-env resets."""
    g = torch.Generator().manual_seed(seed)
    seq = []
    rtg = torch.full((B,), 4.5)
    for t in range(steps):
        if image:
            obs = torch.randint(0, 256, (B, *spec.image_shape), generator=g, dtype=torch.uint8)
        else:
            obs = torch.rand(B, spec.state_dim, generator=g) * 2 - 1
            obs[:, spec.state_dim * 3 // 4:] = 0.0
        mask = (torch.rand(B, generator=g) < reset_prob).to(torch.uint8) if t > 0 else torch.ones(B, dtype=torch.uint8)
        rtg = torch.where(mask.bool(), torch.full_like(rtg, 4.5), rtg - 0.01)
        seq.append((obs, rtg.clone(), torch.zeros(B), mask))
    return seq


def assert_actions_match(a_gpu, a_ref, logits_ref, spec, discrete=False, gap_tol=2e-4, what=""):
    """Discrete actions bit-exact, continuous within 1e-4 (= identical bins, bin width 2/256).  A mismatch
    is tolerated only where the oracle's own top-2 logit gap is below `gap_tol` (a numerical tie); the
    number of such ties is returned so tests can assert it is zero on the committed seeds."""
    a_gpu = a_gpu.detach().cpu()
    if discrete:
        bad = (a_gpu.reshape(-1).long() != a_ref.reshape(-1).long())
        lg = logits_ref.reshape(a_ref.shape[0], -1)[:, : spec.n_discrete]
        top2 = lg.topk(2, dim=-1).values
        gap = (top2[:, 0] - top2[:, 1])
    else:
        bad = (a_gpu - a_ref).abs() > 1e-4
        top2 = logits_ref.topk(2, dim=-1).values
        gap = (top2[..., 0] - top2[..., 1])
    bad = bad.reshape(gap.shape)
    real = bad & (gap >= gap_tol)
    assert not bool(real.any()), f"{what}: {int(real.sum())} action mismatches with a clear oracle margin " \
                                 f"(min gap among them {float(gap[real].min()):.3e})"
    return int(bad.sum())


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
