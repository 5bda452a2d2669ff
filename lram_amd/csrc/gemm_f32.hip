// fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = A[M,K] * W[N,K]^T (+ bias) (+ residual)
//
// Replaces the cuBLAS `nn.Linear` calls of the reference hot path: mLSTM proj_up /
// proj_down, sLSTM gate / recurrent / FFN projections, Mamba in/x/dt/out projections, embed_state,
// action_net.  fp32 inputs and `v_mfma_f32_32x32x2_f32` accumulation: bit-for-bit a k-ordered fmaf
// chain, so logits keep full fp32 accuracy (actions are argmax'ed and must match the fp32 CPU path).
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile 128 x 128 x 32, wave tile 64 x 64 = 2 x 2 MFMA
// 32x32 accumulators (64 VGPR).  Both operands are K-contiguous, staged global -> registers -> LDS
// (row pitch 36 floats: ds_read_b128 of 16 consecutive rows hits 16 distinct 4-bank slots), next
// K-tile's global loads are issued before the MFMAs of the current one.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace lram {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BM = 128, BN = 128, BK = 32, PITCH = 36;

template <bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * PITCH];
  float* As = lds;
  float* Bs = lds + BM * PITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // batch pointers
  const int z = blockIdx.y;
  const int z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const float* A = g.a + z1 * g.sA1 + z2 * g.sA2;
  const float* W = g.w + z1 * g.sW1 + z2 * g.sW2;
  float* C = g.c + z1 * g.sC1 + z2 * g.sC2;
  const float* R = HAS_RES ? g.residual + z1 * g.sC1 + z2 * g.sC2 : nullptr;
  const float* bias = HAS_BIAS ? g.bias + z1 * g.sBias1 + z2 * g.sBias2 : nullptr;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous run
  // of tiles so that neighbouring N-tiles of one M-tile (same A rows) hit the same L2.
  const int tiles_n = (g.n + BN - 1) / BN;
  const int tiles_m = (g.m + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x;
  if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
  const int tm_idx = bid / tiles_n;
  const int tn_idx = bid - tm_idx * tiles_n;
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;

  // global -> register staging: 4 float4 of A and 4 of W per thread per K-tile
  const int lr = tid >> 3;        // 0..31 row within a 32-row slab
  const int lc = (tid & 7) << 2;  // k offset 0,4,..,28
  float4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = lr + 32 * i;
      const int kk = k0 + lc;
      const int gm = m0 + r, gn = n0 + r;
      ra[i] = (gm < g.m && kk < g.k) ? *reinterpret_cast<const float4*>(A + (int64_t)gm * g.lda + kk)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[i] = (gn < g.n && kk < g.k) ? *reinterpret_cast<const float4*>(W + (int64_t)gn * g.ldw + kk)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = lr + 32 * i;
      *reinterpret_cast<float4*>(As + r * PITCH + lc) = ra[i];
      *reinterpret_cast<float4*>(Bs + r * PITCH + lc) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31;
  const int lh = lane >> 5;
  const float* a_base = As + (64 * wm + li) * PITCH + 4 * lh;
  const float* b_base = Bs + (64 * wn + li) * PITCH + 4 * lh;

  const int nk_all = (g.k + BK - 1) / BK;
  const int kt0 = g.split_k > 1 ? blockIdx.z * g.k_tiles_per_split : 0;
  const int nk = g.split_k > 1 ? min(nk_all, kt0 + g.k_tiles_per_split) : nk_all;
  load_tile(kt0 * BK);
  for (int kt = kt0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int kc = 0; kc < BK / 8; ++kc) {
      float4 af[2], bf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = *reinterpret_cast<const float4*>(a_base + 32 * t * PITCH + 8 * kc);
        bf[t] = *reinterpret_cast<const float4*>(b_base + 32 * t * PITCH + 8 * kc);
      }
      // lane half h supplies k = 8*kc + 4*h + j to MFMA j: A and W use the same k for the same (h, j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (g.split_k > 1) {  // raw partial sums into this split's slab [M][N]; bias / residual are applied by the reduce
    float* S = g.splitk_ws + (int64_t)blockIdx.z * g.m * g.n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + 64 * wn + 32 * j + li;
        if (col >= g.n) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < g.m) S[(int64_t)row * g.n + col] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + li;
      if (col >= g.n) continue;
      const float bv = HAS_BIAS ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < g.m) {
          float v = acc[i][j][r] + bv;
          if (HAS_RES) v += R[(int64_t)row * g.ldc + col];
          C[(int64_t)row * g.ldc + col] = v;
        }
      }
    }
}
}  // namespace

namespace {
// Small-M path (M <= 8: one or two envs x 3 tokens, the reference's own operating point): a tile GEMM would
// spend a whole 128x128 MFMA tile and a serial K loop on 3 rows.  Here every wave owns two output columns, its
// lanes stride over K with 16-byte loads of the weight rows (read exactly once, coalesced), the few activation
// rows come from L1/L2, and a wave reduction finishes the dot products: exact fp32 fma arithmetic, ~2 dependent
// memory round trips per launch.
constexpr int kGemvMaxM = 8;
constexpr int kGemvCols = 2;  // columns per wave

__global__ __launch_bounds__(256) void gemv_small_m_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int z = blockIdx.y;
  const int z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const float* A = g.a + z1 * g.sA1 + z2 * g.sA2;
  const float* W = g.w + z1 * g.sW1 + z2 * g.sW2;
  float* C = g.c + z1 * g.sC1 + z2 * g.sC2;
  const float* R = g.residual ? g.residual + z1 * g.sC1 + z2 * g.sC2 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.sBias1 + z2 * g.sBias2 : nullptr;
  const int n0 = (blockIdx.x * 4 + wave) * kGemvCols;
  if (n0 >= g.n) return;
  float acc[kGemvMaxM][kGemvCols];
#pragma unroll
  for (int m = 0; m < kGemvMaxM; ++m)
#pragma unroll
    for (int c = 0; c < kGemvCols; ++c) acc[m][c] = 0.f;
  // bias and residual are requested with the first weight rows, not after the reduction; lane 0 is the only one that
  // uses them
  float bv[kGemvCols], rv[kGemvMaxM][kGemvCols];
#pragma unroll
  for (int c = 0; c < kGemvCols; ++c) {
    const bool ok = lane == 0 && n0 + c < g.n;
    bv[c] = (ok && bias) ? bias[n0 + c] : 0.f;
#pragma unroll
    for (int m = 0; m < kGemvMaxM; ++m) rv[m][c] = (ok && R && m < g.m) ? R[(int64_t)m * g.ldc + n0 + c] : 0.f;
  }
#pragma unroll 4
  for (int k = lane * 4; k < g.k; k += 256) {
    float4 w[kGemvCols];
#pragma unroll
    for (int c = 0; c < kGemvCols; ++c)
      w[c] = (n0 + c < g.n) ? *reinterpret_cast<const float4*>(W + (int64_t)(n0 + c) * g.ldw + k)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < kGemvMaxM; ++m) {
      if (m < g.m) {
        const float4 a = *reinterpret_cast<const float4*>(A + (int64_t)m * g.lda + k);
#pragma unroll
        for (int c = 0; c < kGemvCols; ++c) {
          acc[m][c] = fmaf(a.x, w[c].x, acc[m][c]);
          acc[m][c] = fmaf(a.y, w[c].y, acc[m][c]);
          acc[m][c] = fmaf(a.z, w[c].z, acc[m][c]);
          acc[m][c] = fmaf(a.w, w[c].w, acc[m][c]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < kGemvMaxM; ++m) {
    if (m < g.m) {
#pragma unroll
      for (int c = 0; c < kGemvCols; ++c) {
        float v = acc[m][c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && n0 + c < g.n) C[(int64_t)m * g.ldc + n0 + c] = (v + bv[c]) + rv[m][c];
      }
    }
  }
}

// Few-row path (9 .. a few hundred rows: tens of envs x 3 tokens).  A 128 x 128 tile kernel leaves most of its tile empty
// there, needs split-K plus a reduce launch to reach more than a handful of CUs, and pays a serial K loop of global -> LDS ->
// MFMA round trips.  Here one workgroup owns a 32 x 32 output tile as ONE accumulator of the exact fp32 matrix instruction
// (v_mfma_f32_32x32x2_f32), K is split over the four waves and the two lane halves, and every lane requests its operands --
// its A row and its W row over the lane's K range, contiguous 16-byte runs -- straight from global memory into registers:
// kSkQ float4 per operand per round, the next round's requests issued before the current round's matrix instructions.
// No LDS staging, no split-K workspace, no second launch; the four partial tiles meet in LDS and leave as 16-byte stores
// with bias / residual applied.  Same arithmetic class as gemm_f32_kernel (fp32 products, fp32 accumulation).
// Instances: NW waves x Q float4 per operand per lane and round.  ONE round where it fits -- K <= 256: <4, 8>, K <= 512:
// <4, 16> (+1..2 % over two rounds) -- else <4, 8> with a second register set for the next round.
template <bool HAS_BIAS, bool HAS_RES, int NW, int Q, bool MULTI, bool NORM = false>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(GemmArgs g) {
  static_assert(!(NORM && MULTI), "the norm prologue needs the whole row slice in registers");
  __shared__ float part[NW][32][33];
  __shared__ float stat[NORM ? 2 * NW : 1][32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int z = blockIdx.z;
  const int z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const bool tab = g.a_tab[0] != nullptr;
  const float* A = tab ? g.a_tab[z2] + z1 * g.sA1 : g.a + z1 * g.sA1 + z2 * g.sA2;
  const float* W = tab ? g.w_tab[z2] + z1 * g.sW1 : g.w + z1 * g.sW1 + z2 * g.sW2;
  float* C = tab ? g.c_tab[z2] + z1 * g.sC1 : g.c + z1 * g.sC1 + z2 * g.sC2;
  const float* R = HAS_RES ? g.residual + z1 * g.sC1 + z2 * g.sC2 : nullptr;
  const float* bias = HAS_BIAS ? g.bias + z1 * g.sBias1 + z2 * g.sBias2 : nullptr;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  // lane group (w, lh) owns the float4 indices [q0, q1) of the K axis
  const int nq = g.k >> 2;
  const int per = (nq + 2 * NW - 1) / (2 * NW);
  const int grp = 2 * w + lh;
  const int q0 = min(grp * per, nq), q1 = min(q0 + per, nq);
  const float4* ap = reinterpret_cast<const float4*>(A + (int64_t)min(m0 + li, g.m - 1) * g.lda);
  const float4* bp = reinterpret_cast<const float4*>(W + (int64_t)min(n0 + li, g.n - 1) * g.ldw);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 av0[Q], bv0[Q];
  float4 av1[MULTI ? Q : 1], bv1[MULTI ? Q : 1];  // second register set, named apart: a run-time set index would put both in scratch
  // unconditional loads at clamped indices (a load under a lane condition costs a branch and a full wait); what lies
  // beyond the lane's range is multiplied by zero
  auto request = [&](auto& av, auto& bv, int q) {
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int qi = min(q + i, nq - 1);
      av[i] = ap[qi];
      bv[i] = bp[qi];
    }
  };
  auto products = [&](const auto& av, const auto& bv, int q) {
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const float live = (q + i < q1) ? 1.f : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x * live, bv[i].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y * live, bv[i].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z * live, bv[i].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w * live, bv[i].w, acc, 0, 0, 0);
    }
  };
  request(av0, bv0, q0);
  // bias / residual of this thread's outputs, requested with the first operands: 256 threads x 4 outputs cover the tile
  // (an 8-wave workgroup's upper half only takes part in the products)
  const int et = tid & 255;
  const int er = et >> 3, ec = (et & 7) << 2;
  const int row = m0 + er, col = n0 + ec;
  const bool ok = tid < 256 && row < g.m && col < g.n;
  float4 bz = make_float4(0.f, 0.f, 0.f, 0.f), rz = bz;
  // (the residual shares C's row pitch; its base must be 16-byte aligned too for the float4 load below)
  const bool vec = ok && col + 3 < g.n && (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                   (!HAS_RES || (reinterpret_cast<uintptr_t>(R) & 15) == 0);
  if (HAS_BIAS && ok) {
    bz.x = bias[col];
    if (col + 1 < g.n) bz.y = bias[col + 1];
    if (col + 2 < g.n) bz.z = bias[col + 2];
    if (col + 3 < g.n) bz.w = bias[col + 3];
  }
  if (HAS_RES && ok) {
    const float* rp = R + (int64_t)row * g.ldc + col;
    if (vec) {
      rz = *reinterpret_cast<const float4*>(rp);
    } else {
      rz.x = rp[0];
      if (col + 1 < g.n) rz.y = rp[1];
      if (col + 2 < g.n) rz.z = rp[2];
      if (col + 3 < g.n) rz.w = rp[3];
    }
  }
  if (!MULTI) {
    float4 gv[NORM ? Q : 1], ov[NORM ? Q : 1];
    if (NORM) {  // norm weights of the lane's K range (the same for every row: 32 lanes share an address)
      const float4* gp = reinterpret_cast<const float4*>(g.norm_g);
      const float4* op = reinterpret_cast<const float4*>(g.norm_b);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const int qi = min(q0 + i, nq - 1);
        gv[i] = gp[qi];
        ov[i] = op != nullptr ? op[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // every request above is issued before the first use
    if (NORM) {
      // row statistics over the 2 * NW lane groups: two passes (mean, then centred squares), as row_norm_kernel
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < Q; ++i)
        if (q0 + i < q1) s1 += (av0[i].x + av0[i].y) + (av0[i].z + av0[i].w);
      stat[grp][li] = s1;
      __syncthreads();
      float mean = 0.f;
      if (!g.norm_rms) {
#pragma unroll
        for (int x = 0; x < 2 * NW; ++x) mean += stat[x][li];
        mean /= (float)g.k;
      }
      __syncthreads();
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < Q; ++i)
        if (q0 + i < q1) {
          const float dx = av0[i].x - mean, dy = av0[i].y - mean, dz = av0[i].z - mean, dw = av0[i].w - mean;
          s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
      stat[grp][li] = s2;
      __syncthreads();
      float var = 0.f;
#pragma unroll
      for (int x = 0; x < 2 * NW; ++x) var += stat[x][li];
      var /= (float)g.k;
      const float rstd = g.norm_rms ? rsqrtf(var + g.norm_eps) : 1.f / sqrtf(var + g.norm_eps);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        av0[i].x = (av0[i].x - mean) * rstd * gv[i].x + ov[i].x;
        av0[i].y = (av0[i].y - mean) * rstd * gv[i].y + ov[i].y;
        av0[i].z = (av0[i].z - mean) * rstd * gv[i].z + ov[i].z;
        av0[i].w = (av0[i].w - mean) * rstd * gv[i].w + ov[i].w;
      }
    }
    products(av0, bv0, q0);
  } else {
    // every lane group runs the same number of rounds (the matrix instruction is wave-wide; the groups' ranges differ by
    // at most one float4)
    const int rounds = (per + Q - 1) / Q;
    for (int r = 0; r < rounds; r += 2) {
      const int q = q0 + r * Q;
      __builtin_amdgcn_sched_barrier(0);
      if (r + 1 < rounds) request(av1, bv1, q + Q);
      __builtin_amdgcn_sched_barrier(0);
      products(av0, bv0, q);
      if (r + 1 < rounds) {
        __builtin_amdgcn_sched_barrier(0);
        if (r + 2 < rounds) request(av0, bv0, q + 2 * Q);
        __builtin_amdgcn_sched_barrier(0);
        products(av1, bv1, q + Q);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[w][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
  __syncthreads();
  if (!ok) return;
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = (part[0][er][ec + j] + part[1][er][ec + j]) + (part[2][er][ec + j] + part[3][er][ec + j]);
    if (NW == 8) v += (part[4][er][ec + j] + part[5][er][ec + j]) + (part[6][er][ec + j] + part[7][er][ec + j]);
    o[j] = v;
  }
  o[0] = (o[0] + bz.x) + rz.x, o[1] = (o[1] + bz.y) + rz.y, o[2] = (o[2] + bz.z) + rz.z, o[3] = (o[3] + bz.w) + rz.w;
  float* cp = C + (int64_t)row * g.ldc + col;
  if (vec) {
    *reinterpret_cast<float4*>(cp) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    cp[0] = o[0];
    if (col + 1 < g.n) cp[1] = o[1];
    if (col + 2 < g.n) cp[2] = o[2];
    if (col + 3 < g.n) cp[3] = o[3];
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
  const int64_t total = (int64_t)g.m * g.n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / g.n), col = (int)(i - (int64_t)row * g.n);
    float v = 0.f;
    for (int s = 0; s < g.split_k; ++s) v += g.splitk_ws[(int64_t)s * total + i];
    if (g.bias != nullptr) v += g.bias[col];
    if (g.residual != nullptr) v += g.residual[(int64_t)row * g.ldc + col];
    g.c[(int64_t)row * g.ldc + col] = v;
  }
}

// The same sums (slab 0 first, then 1, ...: bit-identical to the scalar form) with four columns per thread and every slab's
// request issued before the first add: the scalar form walks S dependent round trips per element (round 5: 1536 x 512, S = 6
// beside a read pass 20.9 us per launch, on the chain of every 16M block at 1024 slots; Mamba's x_proj 11.6 us).
constexpr int kReduceMaxS = 16;
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(GemmArgs g) {
  const int n4 = g.n >> 2;
  const int64_t total4 = (int64_t)g.m * n4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int row = (int)(i / n4), col = 4 * (int)(i - (int64_t)row * n4);
  const float4* ws = reinterpret_cast<const float4*>(g.splitk_ws) + i;
  float4 p[kReduceMaxS];
#pragma unroll
  for (int s = 0; s < kReduceMaxS; ++s)
    if (s < g.split_k) p[s] = ws[(int64_t)s * total4];
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f), b = r;
  if (g.residual != nullptr) r = *reinterpret_cast<const float4*>(g.residual + (int64_t)row * g.ldc + col);
  if (g.bias != nullptr) b = *reinterpret_cast<const float4*>(g.bias + col);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < kReduceMaxS; ++s)
    if (s < g.split_k) v.x += p[s].x, v.y += p[s].y, v.z += p[s].z, v.w += p[s].w;
  if (g.bias != nullptr) v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
  if (g.residual != nullptr) v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
  *reinterpret_cast<float4*>(g.c + (int64_t)row * g.ldc + col) = v;
}
}  // namespace

namespace {
GemmKnobs g_knobs;
bool g_knobs_read = false;
int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : dflt;
}
}  // namespace

void gemm_knobs_reload() {
  GemmKnobs k;
  k.tile = env_int("LRAM_GEMM_TILE", 0);
  k.stages = env_int("LRAM_F16P_STAGES", 0);
  k.panel = env_int("LRAM_GEMM_PANEL", 0);
  k.splitk_tiles = env_int("LRAM_SPLITK_TILES", 56);
  g_knobs = k;
  g_knobs_read = true;
}

const GemmKnobs& gemm_knobs() {
  if (!g_knobs_read) gemm_knobs_reload();
  return g_knobs;
}

// Split-K when the output has too few 128x128 tiles to fill the 256 CUs and K is deep enough to split.
int gemm_choose_split_k(GemmArgs& g) {
  g.split_k = 1;
  g.k_tiles_per_split = 0;
  if (g.splitk_ws == nullptr || g.nb1 * g.nb2 != 1) return 1;
  const int tiles = ((g.m + BM - 1) / BM) * ((g.n + BN - 1) / BN);
  const int nk = (g.k + BK - 1) / BK;
  // LRAM_SPLITK_TILES (measurement knob): outputs with fewer 128 x 128 tiles than this are split along K
    // (round 4: 128 -> 48.  With the faster front end the 1024-slot step is bound by its chain, whose proj_down -- 48 tiles of
    // 128 x 128 per 512-env slice -- ran as 5 K splits + a reduce launch: 313k -> 320k env-steps/s unsplit; 512 / 2048 / 4096 slots,
    // Mamba-48M and the 206M stack within +- 0.5 %: profiles/r04_ab_splitk_threshold.txt.
    // Round 5: 48 -> 56.  With two read-pass workgroups per CU for chain-bound slices the split launches find slots: the 48-tile
    // proj_down / ffn_down of 1024 slots split again, +2-3 % (361.1k / 364.8k -> 374.2k / 367.3k at a threshold of 64); the 206M
    // stack's 60-tile proj_down (K = 2560, five slabs of 3.9 MB) stays unsplit: split it loses 1.1 % (32.16k / 32.06k ->
    // 31.74k / 31.75k); 128: -6 % at 512 slots.  profiles/r05_ab_splitk_threshold.txt)
  const int min_tiles = gemm_knobs().splitk_tiles;
  if (tiles >= min_tiles || nk < 4) return 1;
  int S = std::min(std::min(nk / 2, 16), (256 + tiles - 1) / tiles);
  while (S > 1 && (int64_t)S * g.m * g.n > g.splitk_ws_elems) --S;
  if (S < 2) return 1;
  const int per = (nk + S - 1) / S;
  S = (nk + per - 1) / per;
  if (S < 2) return 1;
  g.split_k = S;
  g.k_tiles_per_split = per;
  return S;
}

void gemm_choose_xcd_split(GemmArgs& g, int bm, int bn, int bytes_per_elem) {
  g.xcd_gm = g.xcd_gn = 0;
  g.panel_w = 0;
  if (g.nb1 * g.nb2 != 1) return;
  const int tiles_m = (g.m + bm - 1) / bm, tiles_n = (g.n + bn - 1) / bn;
  // LRAM_GEMM_PANEL (measurement knob; gemm_knobs(): the bit-identity test walks through the orders via the standalone entries):
  // > 0 forces the panel order with that width (a width >= the grid's = the one-dimensional map: each XCD a contiguous run of
  // the row-major order)
  const int pw = gemm_knobs().panel;
  if (pw > 0) {
    g.panel_w = std::min(pw, tiles_n);
    return;
  }
  double best = 0.0;
  for (int gm = 8; gm >= 1; gm >>= 1) {
    const int gn = 8 / gm;
    if (tiles_m % gm != 0 || tiles_n % gn != 0) continue;
    const double w_band = (double)(tiles_n / gn) * bn * g.k * bytes_per_elem;
    if (w_band > 3.0e6) continue;                       // the W band must stay resident beside the A tiles in flight
    const double cost = (double)g.m * gn + (double)g.n * gm;
    if (g.xcd_gm == 0 || cost < best) best = cost, g.xcd_gm = gm, g.xcd_gn = gn;
  }
  // No such split (the 206M stack's projections: no W band fits an L2; tile grids that 8 does not divide): column panels of 6
  // tiles, each XCD a contiguous eighth of the walk.  Round 5, standalone launches, beyond-L2 reads / operand bytes: 206M
  // proj_up 6.5 -> 2.8, proj_down 4.2 -> 3.2, its 16128-row prefill launches 30.6 -> 11.1 and 9.2 -> 3.6 (widths 6 / 8 / 10 within
  // 10 % of each other; profiles/r05_gemm_tile_order.txt).  The launches themselves get 0 - 3 % shorter and no step time moves
  // (the re-reads were served by the 256 MB memory-side cache, not by HBM), so this is traffic hygiene, not speed.
  if (g.xcd_gm == 0) g.panel_w = std::min(6, tiles_n);
}

void launch_splitk_reduce(const GemmArgs& g, hipStream_t stream) {
  const int64_t total = (int64_t)g.m * g.n;
  const bool vec = (g.n & 3) == 0 && (g.ldc & 3) == 0 && g.split_k <= kReduceMaxS &&
                   ((reinterpret_cast<uintptr_t>(g.c) | reinterpret_cast<uintptr_t>(g.splitk_ws) | reinterpret_cast<uintptr_t>(g.residual) |
                     reinterpret_cast<uintptr_t>(g.bias)) & 15) == 0;
  if (vec) {
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream, g);
    LRAM_HIP_CHECK(hipGetLastError());
    return;
  }
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g);
  LRAM_HIP_CHECK(hipGetLastError());
}

bool gemm_small_m(const GemmArgs& g) { return g.m <= kGemvMaxM; }

bool gemm_skinny_supported(const GemmArgs& g) {
  return g.m >= 1 && g.k >= 32 && (g.k & 3) == 0 && (g.lda & 3) == 0 && (g.ldw & 3) == 0 &&
         ((g.sA1 | g.sA2 | g.sW1 | g.sW2) & 3) == 0 && g.gate == nullptr && g.act_silu_from < 0 &&
         (reinterpret_cast<uintptr_t>(g.a) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.w) & 15) == 0;
}

template <int NW, int Q, bool MULTI>
void launch_gemm_skinny_inst(const GemmArgs& g, hipStream_t stream) {
  dim3 grid((g.n + 31) / 32, (g.m + 31) / 32, g.nb1 * g.nb2), block(64 * NW);
  const bool hb = g.bias != nullptr, hr = g.residual != nullptr;
  if constexpr (!MULTI) {
    if (g.norm_g != nullptr) {  // (the projections that follow a norm have neither bias nor residual)
      LRAM_REQUIRE(!hb && !hr && g.nb1 * g.nb2 == 1, "gemm: the norm prologue serves un-batched GEMMs without bias / residual");
      hipLaunchKernelGGL((gemm_skinny_kernel<false, false, NW, Q, false, true>), grid, block, 0, stream, g);
      LRAM_HIP_CHECK(hipGetLastError());
      return;
    }
  }
  if (hb && hr)
    hipLaunchKernelGGL((gemm_skinny_kernel<true, true, NW, Q, MULTI>), grid, block, 0, stream, g);
  else if (hb)
    hipLaunchKernelGGL((gemm_skinny_kernel<true, false, NW, Q, MULTI>), grid, block, 0, stream, g);
  else if (hr)
    hipLaunchKernelGGL((gemm_skinny_kernel<false, true, NW, Q, MULTI>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<false, false, NW, Q, MULTI>), grid, block, 0, stream, g);
  LRAM_HIP_CHECK(hipGetLastError());
}

bool gemm_skinny_norm_supported(const GemmArgs& g) {
  return gemm_skinny_supported(g) && (((g.k >> 2) + 7) / 8) <= 16 && g.bias == nullptr && g.residual == nullptr &&
         g.nb1 * g.nb2 == 1;
}

void launch_gemm_skinny(const GemmArgs& g, hipStream_t stream) {
  LRAM_REQUIRE(gemm_skinny_supported(g), "gemm: shape not supported by the few-row kernel");
  LRAM_REQUIRE(g.norm_g == nullptr || gemm_skinny_norm_supported(g), "gemm: norm prologue needs K <= 512, no bias / residual / batch");
  if (g.a_tab[0] != nullptr) {
    LRAM_REQUIRE(g.nb2 >= 1 && g.nb2 <= 4 && g.bias == nullptr && g.residual == nullptr,
                 "gemm: operand tables serve up to four projections without bias / residual");
    for (int i = 0; i < g.nb2; ++i)
      LRAM_REQUIRE(g.a_tab[i] != nullptr && g.w_tab[i] != nullptr && g.c_tab[i] != nullptr &&
                       ((reinterpret_cast<uintptr_t>(g.a_tab[i]) | reinterpret_cast<uintptr_t>(g.w_tab[i])) & 15) == 0,
                   "gemm: operand tables need 16-byte aligned operands for every entry");
  }
  const int nq = g.k >> 2;
  if ((nq + 7) / 8 <= 8) return launch_gemm_skinny_inst<4, 8, false>(g, stream);
  if ((nq + 7) / 8 <= 16) return launch_gemm_skinny_inst<4, 16, false>(g, stream);
  // (an 8-wave instance with one round up to K = 1024 measured slower: Mamba-48M at 16 envs 0.887 -> 0.948 ms)
  launch_gemm_skinny_inst<4, 8, true>(g, stream);
}

void launch_gemm_f32(const GemmArgs& g_in, hipStream_t stream) {
  GemmArgs g = g_in;
  LRAM_REQUIRE(g.m > 0 && g.n > 0 && g.k > 0, "gemm: empty problem");
  if (gemm_small_m(g)) {
    LRAM_REQUIRE((g.k & 3) == 0 && (g.lda & 3) == 0 && (g.ldw & 3) == 0, "gemm: K, lda, ldw must be multiples of 4");
    LRAM_REQUIRE(((g.sA1 | g.sA2 | g.sW1 | g.sW2) & 3) == 0, "gemm: batch strides of A/W must be multiples of 4");
    dim3 grid((g.n + 4 * kGemvCols - 1) / (4 * kGemvCols), g.nb1 * g.nb2);
    hipLaunchKernelGGL(gemv_small_m_kernel, grid, dim3(256), 0, stream, g);
    LRAM_HIP_CHECK(hipGetLastError());
    return;
  }
  const int S = gemm_choose_split_k(g);
  LRAM_REQUIRE((g.k & 3) == 0 && (g.lda & 3) == 0 && (g.ldw & 3) == 0, "gemm: K, lda, ldw must be multiples of 4");
  LRAM_REQUIRE(((g.sA1 | g.sA2 | g.sW1 | g.sW2) & 3) == 0, "gemm: batch strides of A/W must be multiples of 4");
  const int tiles = ((g.m + BM - 1) / BM) * ((g.n + BN - 1) / BN);
  dim3 grid(tiles, g.nb1 * g.nb2, S);
  dim3 block(256);
  const bool hb = g.bias != nullptr, hr = g.residual != nullptr;
  if (hb && hr)
    hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, stream, g);
  else if (hb)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, stream, g);
  else if (hr)
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, stream, g);
  LRAM_HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce(g, stream);
}

}  // namespace lram
