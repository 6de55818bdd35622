// Exact-GP conditioning on gfx950.
//
// Replaces what gpytorch/linear_operator do lazily on the first posterior call inside
// botorch.optimize_acqf (/root/reference/Algorithms/BayesianOptimization/PCA_BO.py:535-545, 607):
//   K = k(Zn,Zn) + s2 I  ->  L = chol(K)  ->  R = L^-1 (root-inverse cache)  ->  alpha = K^-1 y_s.
//
// Storage: K/L and R are NP x NP row-major with leading dimension ld, NP = n rounded up to 64.
// The padding block is the identity, so every kernel works on whole 64 x 64 tiles and the
// padded rows/columns never influence the leading n x n part.
#include "pcabo_internal.h"
#include <mutex>
#include <cstdlib>

#define BS PCABO_BS
#define TLD PCABO_TLD

// ---------------------------------------------------------------------------------------------
// Gram matrix.  One 64x64 tile per work-group (lower triangle of tiles only), wave w owns 16 rows.
// The cross term a_i.a_j runs on v_mfma_f64_16x16x4_f64 (gpytorch computes the squared distance
// as one GEMM of [-2a, |a|^2, 1] x [b, 1, |b|^2]^T; the norms are added in the epilogue here),
// followed by the fused Matern-5/2 (or RBF) map and the noise on the diagonal.
// Operands are read straight from AT (KP x ld, point index contiguous): each MFMA operand load is
// four 128-byte rows, fully coalesced; the whole AT (<= 36 x 512 doubles) is L2 resident.
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ AT, const double* __restrict__ nrm, int n,
                                              int KP, int ld, double noise, int kernel, double* __restrict__ K,
                                              const int* __restrict__ k_dev, double* __restrict__ K2,
                                              int* __restrict__ info_reset, size_t zs) {
  const XcdTile xt_ = xcd_tile();                    // (the tiles of a run read the same rows of AT: one XCD)
  ZRUNX(AT); ZRUNX(nrm); ZRUNX(K); ZRUNX(k_dev); ZRUNX(K2); ZRUNX(info_reset);
  const int ti = (int)xt_.x, tj = (int)xt_.y;
  // K2: the copy the factorisation works on in place; info_reset: its failure flag (saves a copy and a fill launch).
  // K itself is written only when somebody asks for it (pcabo_get_gram): a second full-size store doubled the kernel's
  // HBM traffic for a matrix that a retry can just as well build again.
  if (info_reset && ti == 0 && tj == 0 && threadIdx.x == 0) *info_reset = 0;
  if (tj > ti) return;
  if (k_dev) KP = (*k_dev + 3) & ~3;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int i0 = ti * BS + 16 * w, j0 = tj * BS;
  double4_t acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < KP; kk += 4) {
    const double* row = AT + (size_t)(kk + (l >> 4)) * ld;
    double a = row[i0 + (l & 15)];
    for (int t = 0; t < 4; ++t) {
      double b = row[j0 + 16 * t + (l & 15)];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  const double s5 = 2.23606797749979;   // sqrt(5)
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + 16 * t + (l & 15);
    const double nj = nrm[j];
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + (l >> 4) + 4 * r;
      double v;
      if (i >= n || j >= n) {
        v = (i == j) ? 1.0 : 0.0;
      } else {
        double sq = (i == j) ? 0.0 : (nrm[i] + nj) - 2.0 * acc[t][r];
        sq = fmax(sq, 0.0);
        if (kernel == 1) {
          v = exp(-0.5 * sq);
        } else {
          double dist = sqrt(fmax(sq, 1e-30));
          v = ((s5 * dist + 1.0) + (5.0 / 3.0) * (dist * dist)) * exp(-s5 * dist);
        }
        if (i == j) v += noise;
      }
      if (K) K[(size_t)i * ld + j] = v;
      if (K2) K2[(size_t)i * ld + j] = v;
    }
  }
}

__global__ void k_add_jitter(double* __restrict__ K, int n, int ld, double jitter) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) K[(size_t)i * ld + i] += jitter;
}

// ---------------------------------------------------------------------------------------------
// Blocked left-looking Cholesky, panel width 64: per panel k_chol_panel_m (kernels_gpw.hip: two waves per 64 x 64
// block, a matrix row per lane) factors the diagonal block and solves the off-diagonal blocks, k_chol_lookback below
// brings a block column up to date on MFMA before it is factored (left-looking).
__device__ inline void load_tile(const double* __restrict__ src, int ld, double* s_t) {
  for (int idx = threadIdx.x; idx < BS * BS; idx += 256) {
    int r = idx >> 6, c = idx & 63;
    s_t[r * TLD + c] = src[(size_t)r * ld + c];
  }
}

// The panel kernel does not store the factored diagonal block over its input inside its own launch: the other groups of
// that launch read the block as INPUT at their start, and nothing orders their start before block 0's end (on a GPU
// shared with other processes a group can start tens of microseconds late - seen as a spurious "not positive
// definite").  With more than one group the factor goes to `diag_scratch`; one extra group of the next panel's
// look-back launch copies it into place (that launch does not touch the diagonal block otherwise).
// LEFT-LOOKING update on f64 MFMA: before panel J is factored, block row I >= J of block column J receives the
// contributions of ALL earlier panels at once,  A[I][J] -= sum_{p<J} L[I][p] L[J][p]^T.
// One work-group per 64x64 tile keeps the tile's accumulators in registers over the J steps (the right-looking form of
// round 1 re-read and re-wrote every trailing tile once per panel: 128 KB of traffic per 0.5 MFLOP, bound by the
// L2 / Infinity-Cache bandwidth at ~3 TB/s with many runs side by side); operand tiles go through LDS with a leading
// dimension of 66 doubles (conflict-free ds_read_b64), the next step's tiles travel in registers while the MFMAs of
// the current one run.  The extra last work-group puts the previous panel's diagonal factor in place (see above).
// (round 4) The operand tiles travel in HALVES of 32 columns: 35 KB of LDS and 155 registers per work-group instead of 68 KB and
// 190, so THREE work-groups share a CU (12 waves) instead of two - while one waits for its loads or at a barrier, the others have
// MFMAs to issue (120 runs at n = 1050: Cholesky 2 318 -> 2 209 us; four per CU would need 128 registers: 28 spills).  The products of a panel still accumulate from zero over ascending k (first half, then second half) before they
// are subtracted: same bits.
#define TLH 34                                       // leading dimension of a half tile (64 x 32): 272-byte rows, conflict-free reads
__global__ __launch_bounds__(256, 3) void k_chol_lookback(double* __restrict__ A, int J, int nblk, int ld,
                                                          const double* __restrict__ diag_scratch, size_t zs) {
  const XcdTile xt_ = xcd_tile();                    // (all tiles of a run on one XCD: they share the block row L[J][.])
  ZRUNX(A); ZRUNX(diag_scratch);
  __shared__ __attribute__((aligned(16))) double s_a[BS * TLH];
  __shared__ __attribute__((aligned(16))) double s_b[BS * TLH];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  if (xt_.x == gridDim.x - 1) {                      // panel J-1's diagonal factor: scratch -> its place
    double* Add = A + (size_t)((J - 1) * BS) * ld + (J - 1) * BS;
    for (int idx = tid; idx < BS * BS; idx += 256) Add[(size_t)(idx >> 6) * ld + (idx & 63)] = diag_scratch[idx];
    return;
  }
  const int I = J + (int)xt_.x;
  double* dst = A + (size_t)(I * BS) * ld + J * BS;
  const double* Arow = A + (size_t)(I * BS) * ld;    // L[I][p] tiles
  const double* Brow = A + (size_t)(J * BS) * ld;    // L[J][p] tiles
  // the 16 elements of the target tile this thread owns, fetched up front
  double cold[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) cold[q][r] = dst[(size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15)];
  // 16 bytes per lane and load; a half tile is 64 rows x 16 pairs: 4 pairs per thread
  typedef double d2_t __attribute__((ext_vector_type(2)));
  d2_t pa[4], pb[4];
  auto fetch = [&](int hs) {                         // hs = 2 p + half
    const int col0 = (hs >> 1) * BS + (hs & 1) * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u, r = idx >> 4, c = (idx & 15) * 2;
      pa[u] = *reinterpret_cast<const d2_t*>(Arow + (size_t)r * ld + col0 + c);
      pb[u] = *reinterpret_cast<const d2_t*>(Brow + (size_t)r * ld + col0 + c);          // (the same tile on the diagonal: an L2 hit)
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u, r = idx >> 4, c = (idx & 15) * 2;
      *reinterpret_cast<d2_t*>(s_a + r * TLH + c) = pa[u];
      *reinterpret_cast<d2_t*>(s_b + r * TLH + c) = pb[u];
    }
  };
  fetch(0);
  double4_t acc[4];
  for (int hs = 0; hs < 2 * J; ++hs) {
    // per panel: the 64-term products accumulate from zero and are then subtracted from the tile - the arithmetic (and the
    // bits) of a right-looking update applied panel by panel, without the tile leaving the registers in between
    if ((hs & 1) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    __syncthreads();                                  // the previous sub-step's MFMAs have read the LDS tiles
    put();
    if (hs + 1 < 2 * J) fetch(hs + 1);
    __syncthreads();
    for (int kk = 0; kk < 32; kk += 4) {
      const double a = s_a[(16 * w + (l & 15)) * TLH + kk + (l >> 4)];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bb = s_b[(16 * q + (l & 15)) * TLH + kk + (l >> 4)];     // B[k][j] = L[J-row j][k]
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
      }
    }
    if (hs & 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[q][r] -= acc[q][r];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15)] = cold[q][r];
}

// Look-back over SEVERAL block columns at once (round 4; the form launch_cholesky uses when a batch oversubscribes the chip).
// With many runs side by side the look-back is bound by the bytes it pulls through the fabric (3.2-3.8 TB/s in 512-byte row
// segments, profiles/r03/pmc_summary.txt), and the bytes that cannot be shared are the tiles L[I][p] of the row being updated:
// every block column J > p reads them again.  A work-group therefore updates its block row I in NC neighbouring block columns
// Jc .. Jc + ncols - 1 per step: the tile L[I][p] goes into the LDS once, the tiles L[Jc + c][p] (shared by the whole launch,
// L2 hits on the run's XCD) follow one after the other, each with 64 MFMAs per wave - the unshared traffic per flop falls by
// ncols.  Steps p0 <= p < p1: the schedule (launch_cholesky) runs it once per group of NC columns over all earlier panels and
// once after each panel inside the group for the columns still to come (p1 = p0 + 1).  Per element nothing changes - the
// products of panel p accumulate from zero over ascending k and are subtracted from the tile in ascending p, the value merely
// passes through memory between two launches - so the factors are those of k_chol_lookback / k_chol_step bit for bit
// (tests/golden/gp_factor_hashes.json).  copy_panel >= 0: the extra last work-group puts that panel's diagonal factor in place.
template <int NC>
__global__ __launch_bounds__(256 * NC) void k_chol_lookn(double* __restrict__ A, int Jc, int ncols, int p0, int p1, int ld,
                                                         const double* __restrict__ diag_scratch, int copy_panel, size_t zs) {
  const XcdTile xt_ = xcd_tile();
  ZRUNX(A); ZRUNX(diag_scratch);
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // the row tile, then one tile per column group
  double* s_a = s_dyn;
  const int tid = threadIdx.x, c = __builtin_amdgcn_readfirstlane(tid >> 8), lt = tid & 255, w = lt >> 6, l = lt & 63;
  double* s_b = s_dyn + (size_t)(1 + c) * BS * TLD;
  if (xt_.x == gridDim.x - 1) {
    if (copy_panel >= 0) {
      double* Add = A + (size_t)(copy_panel * BS) * ld + copy_panel * BS;
      for (int idx = tid; idx < BS * BS; idx += 256 * NC) Add[(size_t)(idx >> 6) * ld + (idx & 63)] = diag_scratch[idx];
    }
    return;
  }
  // four waves per block column of the group (wave group c: column Jc + c), every wave group its own tile L[Jc + c][p]; the
  // row tile L[I][p] is loaded once by all of them together
  const int I = Jc + (int)xt_.x;
  const bool live = c < ncols && Jc + c <= I;          // (a column right of the diagonal: nothing to update in this block row)
  const double* Arow = A + (size_t)(I * BS) * ld;
  const double* Brow = A + (size_t)((Jc + c) * BS) * ld;
  double* dst = A + (size_t)(I * BS) * ld + (size_t)(Jc + c) * BS;
  double cold[4][4];
  if (live) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) cold[q][r] = dst[(size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15)];
  }
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr int NA = 8 / NC;                           // 16-byte pieces of the row tile per thread
  d2_t pa[NA], pb[8];
  auto fetch = [&](int p) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * NC * u;
      pa[u] = *reinterpret_cast<const d2_t*>(Arow + (size_t)(idx >> 5) * ld + p * BS + (idx & 31) * 2);
    }
    if (live) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = lt + 256 * u;
        pb[u] = *reinterpret_cast<const d2_t*>(Brow + (size_t)(idx >> 5) * ld + p * BS + (idx & 31) * 2);
      }
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int idx = tid + 256 * NC * u; *reinterpret_cast<d2_t*>(s_a + (idx >> 5) * TLD + (idx & 31) * 2) = pa[u]; }
    if (live) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int idx = lt + 256 * u; *reinterpret_cast<d2_t*>(s_b + (idx >> 5) * TLD + (idx & 31) * 2) = pb[u]; }
    }
  };
  fetch(p0);
  for (int p = p0; p < p1; ++p) {
    __syncthreads();                                  // the previous step's MFMAs have read the LDS tiles
    put();
    if (p + 1 < p1) fetch(p + 1);
    __syncthreads();
    if (live) {
      double4_t acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
      for (int kk = 0; kk < BS; kk += 4) {
        const double a = s_a[(16 * w + (l & 15)) * TLD + kk + (l >> 4)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double bb = s_b[(16 * q + (l & 15)) * TLD + kk + (l >> 4)];
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[q][r] -= acc[q][r];
    }
  }
  if (live) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15)] = cold[q][r];
  }
}
#define CHOL_LOOKN_NC 2
#define CHOL_LOOKN_LDS ((size_t)(1 + CHOL_LOOKN_NC) * BS * TLD * sizeof(double))

// ---------------------------------------------------------------------------------------------
// Root inverse R = L^-1 (what gpytorch caches as `covar_cache`, stored here un-transposed).
// Step 1 (k_trinv_diag_w, kernels_gpw.hip): invert every 64x64 diagonal block, a column of the inverse per lane.
// Step 2: one work-group per chunk of 16 columns of R.  Going down the block rows I = J+1..nblk-1
// (J = block holding the chunk; its diagonal block is already done):
//   S   = - sum_{K=J}^{I-1} L[I][K] X_K          (64x16 accumulators, f64 MFMA)
//   X_I = Rdiag_I S
// Column chunks are independent: NP/16 work-groups, no inter-group synchronisation.  X_K tiles
// written earlier by this same work-group are re-read from global memory after a barrier.
// The operand tiles of the NEXT step (a 64x64 tile of L or of the diagonal inverses, a 64x16 block of X) travel in
// registers while the MFMAs of the current step run: the walk is a chain of ~35 short steps for the first block column,
// and each used to pay a full global-load latency (2 us per step, 62 us per call on average).
__global__ __launch_bounds__(256) void k_trinv_cols(const double* __restrict__ L, int nblk, int ld,
                                                    double* R, size_t zs) {
  const XcdTile xt_ = xcd_tile();                    // (the column chunks of a run walk the same L tiles: one XCD, one L2 copy)
  ZRUNX(L); ZRUNX(R);
  __shared__ __attribute__((aligned(16))) double s_t[BS * TLD];   // 64x64 operand tile
  __shared__ __attribute__((aligned(16))) double s_xk[BS * 16];   // 64x16 block of X (or S)
  const int tid = threadIdx.x;
  const int c0 = (int)xt_.x * 16;
  const int J = c0 / BS;
  const int w = tid >> 6, l = tid & 63;
  double pt[16], px[4];                                            // prefetched tile / X block (this thread's share)
  auto fetch_tile = [&](const double* src) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int idx = tid + 256 * u; pt[u] = src[(size_t)(idx >> 6) * ld + (idx & 63)]; }
  };
  auto fetch_x = [&](int Kb) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int idx = tid + 256 * u; px[u] = R[(size_t)(Kb * BS + (idx >> 4)) * ld + c0 + (idx & 15)]; }
  };
  auto put_tile = [&]() {
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int idx = tid + 256 * u; s_t[(idx >> 6) * TLD + (idx & 63)] = pt[u]; }
  };
  auto put_x = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) s_xk[tid + 256 * u] = px[u];
  };
  if (J + 1 < nblk) { fetch_tile(L + (size_t)((J + 1) * BS) * ld + J * BS); fetch_x(J); }
  for (int I = J + 1; I < nblk; ++I) {
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int Kb = J; Kb < I; ++Kb) {
      __syncthreads();                         // the previous step's MFMAs have read the LDS tiles
      put_tile(); put_x();
      if (Kb + 1 < I) { fetch_tile(L + (size_t)(I * BS) * ld + (Kb + 1) * BS); fetch_x(Kb + 1); }
      else fetch_tile(R + (size_t)(I * BS) * ld + I * BS);         // Rdiag_I from step 1, for the closing product
      __syncthreads();
      for (int kk = 0; kk < BS; kk += 4) {
        double a = s_t[(16 * w + (l & 15)) * TLD + kk + (l >> 4)];
        double b = s_xk[(kk + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) s_xk[(16 * w + (l >> 4) + 4 * r) * 16 + (l & 15)] = -acc[r];
    put_tile();
    if (I + 1 < nblk) { fetch_tile(L + (size_t)((I + 1) * BS) * ld + J * BS); fetch_x(J); }
    __syncthreads();
    double4_t x = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < BS; kk += 4) {
      double a = s_t[(16 * w + (l & 15)) * TLD + kk + (l >> 4)];
      double b = s_xk[(kk + (l >> 4)) * 16 + (l & 15)];
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r)
      R[(size_t)(I * BS + 16 * w + (l >> 4) + 4 * r) * ld + c0 + (l & 15)] = x[r];
    __threadfence_block();
  }
}

// ---------------------------------------------------------------------------------------------
// alpha = K^-1 y_s = R^T (R y_s): two matrix-vector products with the root inverse.
__global__ __launch_bounds__(256) void k_rmatvec(const double* __restrict__ R, const double* __restrict__ y, int n,
                                                 int ld, double* __restrict__ t, size_t zs) {
  ZRUN(R); ZRUN(y); ZRUN(t);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int rr = 0; rr < 4; ++rr) {
    const int i = blockIdx.x * 16 + w * 4 + rr;
    double s = 0.0;
    if (i < n)
      for (int j = l; j <= i; j += 64) s += R[(size_t)i * ld + j] * y[j];
    s = wave_sum(s);
    if (l == 0) t[i] = (i < n) ? s : 0.0;
  }
}

// out = R^T t, 64 columns per group, 16 waves walking down the rows: every wave reads whole 512-byte row segments
// (the same row for all its lanes; rows above the group's first column hold only zeros of the lower-triangular R and are
// skipped group-wise).  Fixed summation order: 4 interleaved accumulators per wave, then the 16 waves in order.
#define RTM_WAVES 16
__global__ __launch_bounds__(64 * RTM_WAVES) void k_rtmatvec(const double* __restrict__ R, const double* __restrict__ t,
                                                             int n, int ld, double* __restrict__ out, size_t zs) {
  ZRUN(R); ZRUN(t); ZRUN(out);
  __shared__ double s_p[RTM_WAVES][64];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int j0 = blockIdx.x * 64, j = j0 + l;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int i = j0 + w;
  for (; i + 3 * RTM_WAVES < n; i += 4 * RTM_WAVES) {
    a0 = fma(R[(size_t)i * ld + j], t[i], a0);
    a1 = fma(R[(size_t)(i + RTM_WAVES) * ld + j], t[i + RTM_WAVES], a1);
    a2 = fma(R[(size_t)(i + 2 * RTM_WAVES) * ld + j], t[i + 2 * RTM_WAVES], a2);
    a3 = fma(R[(size_t)(i + 3 * RTM_WAVES) * ld + j], t[i + 3 * RTM_WAVES], a3);
  }
  for (; i < n; i += RTM_WAVES) a0 = fma(R[(size_t)i * ld + j], t[i], a0);
  s_p[w][l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (w == 0) {
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < RTM_WAVES; ++u) s += s_p[u][l];
    out[j] = (j < n) ? s : 0.0;
  }
}

// ---- launchers --------------------------------------------------------------------------------
void launch_gram(hipStream_t s, const double* AT, const double* nrm, int n, int NP, int KP, int ld, double noise,
                 int kernel, double* K, const int* k_dev, double* K2, int* info_reset, ZB zb) {
  int nb = NP / BS;
  hipLaunchKernelGGL(k_gram, dim3(nb, nb, zb.B), dim3(256), 0, s, AT, nrm, n, KP, ld, noise, kernel, K, k_dev, K2, info_reset,
                     zb.zs);
}
void launch_add_jitter(hipStream_t s, double* K, int n, int ld, double jitter) {
  hipLaunchKernelGGL(k_add_jitter, dim3((n + 255) / 256), dim3(256), 0, s, K, n, ld, jitter);
}
void launch_cholesky(hipStream_t s, double* L, int NP, int ld, int* info, double* diag_scratch, ZB zb) {
  const int nblk = NP / BS;
  // Two forms of the same left-looking factorisation, bit-identical (tests/golden/gp_factor_hashes.json):
  //  * k_chol_step (kernels_gpw.hip): ONE launch per panel - final update + panel of column J beside the look-ahead of
  //    column J+1.  Its groups are heavy (the panel's registers, two sets of operand tiles: one group per CU), which is
  //    right while a launch's groups fit the chip at once: one run at any size, 30 runs up to n = 512
  //    (measured round 3, us: n=450: 223 -> 161 (1 run), 248 -> 221 (30 runs); n=1050: 709 -> 514 (1 run));
  //  * look-back launch + panel launch per panel (round 2) where the chip is oversubscribed anyway and the lighter
  //    look-back groups (2-3 per CU) keep the matrix cores busier: 30 runs at n = 1050: 983 us against 1132 fused.
  //  * round 4, the oversubscribed case in groups of NC = 4 block columns (k_chol_lookn): one look-back over all earlier panels
  //    for the four columns together (the row tiles L[I][p] read once instead of four times), then per panel of the group the
  //    panel launch and one single-step update of the group's remaining columns - the same number of launches, the same bits.
  if (zb.B * nblk <= 256) { launch_chol_steps(s, L, NP, ld, info, diag_scratch, zb); return; }
  if (zb.B * nblk > 1024) {
    // several waves of work-groups per launch: two independent 4-wave groups per CU overlap one's loads with the other's MFMAs
    // better than one 8-wave group in lock-step (measured round 4, us: 120 runs at n = 1050: 2 468 against 2 602 grouped; the
    // grouped form wins below: 30 runs at n = 1050 947 -> 913, 120 runs at n = 450 444 -> 431)
    for (int p = 0; p < nblk; ++p) {                   // bring block column p up to date, then factor it
      if (p > 0)
        hipLaunchKernelGGL(k_chol_lookback, dim3(nblk - p + 1, 1, zb.B), dim3(256), 0, s, L, p, nblk, ld, diag_scratch, zb.zs);
      launch_chol_panel(s, L, p, nblk - p, ld, info, diag_scratch, zb);
    }
    return;
  }
  constexpr int NC = CHOL_LOOKN_NC;
  {
    static std::mutex attr_mu;
    static bool attr_done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!attr_done[dev]) {
      if (hipFuncSetAttribute((const void*)k_chol_lookn<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_LOOKN_LDS) != hipSuccess) return;
      attr_done[dev] = true;
    }
  }
  int last_panel = -1;                               // the panel whose diagonal factor still sits in the scratch tile
  for (int Jc = 0; Jc < nblk; Jc += NC) {
    const int ncols = nblk - Jc < NC ? nblk - Jc : NC;
    if (Jc > 0) {
      hipLaunchKernelGGL(k_chol_lookn<NC>, dim3(nblk - Jc + 1, 1, zb.B), dim3(256 * NC), CHOL_LOOKN_LDS, s, L, Jc, ncols, 0, Jc, ld,
                         diag_scratch, last_panel, zb.zs);
      last_panel = -1;
    }
    for (int c = 0; c < ncols; ++c) {
      const int p = Jc + c;
      launch_chol_panel(s, L, p, nblk - p, ld, info, diag_scratch, zb);
      last_panel = nblk - p > 1 ? p : -1;            // (the last panel has one group and stores its factor directly)
      if (c + 1 < ncols) {                           // columns p + 1 .. of the group: the contribution of panel p
        hipLaunchKernelGGL(k_chol_lookn<NC>, dim3(nblk - (p + 1) + 1, 1, zb.B), dim3(256 * NC), CHOL_LOOKN_LDS, s, L, p + 1,
                           ncols - c - 1, p, p + 1, ld, diag_scratch, last_panel, zb.zs);
        last_panel = -1;
      }
    }
  }
}
void launch_trinv(hipStream_t s, const double* L, int NP, int ld, double* R, ZB zb) {
  const int nblk = NP / BS;
  // (blocks above the diagonal are never written by anything: they keep the zeros of pcabo_ctx_create)
  launch_trinv_diag_w(s, L, nblk, ld, R, zb);
  hipLaunchKernelGGL(k_trinv_cols, dim3(NP / 16, 1, zb.B), dim3(256), 0, s, L, nblk, ld, R, zb.zs);
}
void launch_alpha(hipStream_t s, const double* R, const double* ys, int n, int NP, int ld, double* tmp, double* alpha,
                  ZB zb) {
  hipLaunchKernelGGL(k_rmatvec, dim3(NP / 16, 1, zb.B), dim3(256), 0, s, R, ys, n, ld, tmp, zb.zs);
  hipLaunchKernelGGL(k_rtmatvec, dim3(NP / 64, 1, zb.B), dim3(64 * RTM_WAVES), 0, s, R, tmp, n, ld, alpha, zb.zs);
}
