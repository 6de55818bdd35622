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
  ZRUN(AT); ZRUN(nrm); ZRUN(K); ZRUN(k_dev); ZRUN(K2); ZRUN(info_reset);
  const int ti = blockIdx.x, tj = blockIdx.y;
  // K2: the copy the factorisation works on in place; info_reset: its failure flag (saves a copy and a fill launch)
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
      K[(size_t)i * ld + j] = v;
      if (K2) K2[(size_t)i * ld + j] = v;
    }
  }
}

__global__ void k_add_jitter(double* __restrict__ K, int n, int ld, double jitter) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) K[(size_t)i * ld + i] += jitter;
}

// ---------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, panel width 64.
// k_chol_panel, block b of the grid: every work-group factors the 64x64 diagonal block in LDS
// (redundantly - cheaper than a second launch), block 0 stores it, block b > 0 solves its 64x64
// off-diagonal block X L_dd^T = A_bd.
__device__ inline void load_tile(const double* __restrict__ src, int ld, double* s_t) {
  for (int idx = threadIdx.x; idx < BS * BS; idx += 256) {
    int r = idx >> 6, c = idx & 63;
    s_t[r * TLD + c] = src[(size_t)r * ld + c];
  }
}

// Factor the 64x64 SPD block held in s_d (lower triangle used) in place into its Cholesky factor (upper part
// zeroed).  256 threads; thread (ri = tid>>2, part = tid&3) owns the row elements c = part + 4u in registers.
// Right-looking with DELAYED scaling: at step j column j is final, its owner lanes capture it (fin) and publish
// it through a double-buffered LDS column (one barrier per step); every row then subtracts (a_ij / a_jj) a_cj
// from its not-yet-final columns.  Updates are unconditional (columns <= j are never read again; rows <= j use
// a zero factor), columns final for every lane are pruned at compile time, and L = fin / sqrt(a_jj) is applied
// once at the end.  Returns through *bad_pivot the 1-based index of the first non-positive pivot (0 = none).
__device__ inline void chol64_inplace(double* s_d, double (*s_col)[BS], int* bad_pivot) {
  const int tid = threadIdx.x;
  const int ri = tid >> 2, part = tid & 3;
  double a[16], fin[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) { a[u] = s_d[ri * TLD + part + 4 * u]; fin[u] = a[u]; }
  int bad = 0;
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    if (part == (j & 3)) { fin[j >> 2] = a[j >> 2]; s_col[j & 1][ri] = a[j >> 2]; }
    __syncthreads();
    double piv = s_col[j & 1][j];
    if (!(piv > 0.0)) { if (bad == 0) bad = j + 1; piv = 1.0; }
    const double f = (ri > j) ? s_col[j & 1][ri] * fast_rcp(piv) : 0.0;
#pragma unroll
    for (int u = j >> 2; u < 16; ++u) {
      a[u] -= f * s_col[j & 1][part + 4 * u];
      asm volatile("" : "+v"(a[u]));         // evaluate now: hipcc otherwise defers these FMAs and spills their inputs
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (part + 4 * u == ri) s_col[0][ri] = fin[u];     // pivots (static register index: a runtime one would spill)
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = part + 4 * u;
    const double pv = s_col[0][c];
    const double rs = fast_rsq(pv > 0.0 ? pv : 1.0);
    double v;
    if (c < ri) v = fin[u] * rs;
    else if (c == ri) v = (pv > 0.0 ? pv : 1.0) * rs;     // sqrt(pv)
    else v = 0.0;
    s_d[ri * TLD + c] = v;
  }
  *bad_pivot = bad;
  __syncthreads();
}

// Solve X L^T = B for a 64x64 block B (global, leading dimension ld) with the factor L in s_d; rows in
// registers, column by column: x_c = b_c / L[c][c] is broadcast inside the 4-lane row group by a shuffle, then
// b_m -= x_c L[m][c] (unconditional for columns certainly beyond c; the boundary group is predicated).
__device__ inline void trsm64_right_lt(const double* s_d, const double* s_rdiag, double* Bt, int ld) {
  const int tid = threadIdx.x;
  const int ri = tid >> 2, part = tid & 3;
  double bb[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) bb[u] = Bt[(size_t)ri * ld + part + 4 * u];
#pragma unroll
  for (int c = 0; c < BS; ++c) {
    const double own = bb[c >> 2] * s_rdiag[c];
    double xc;                                   // broadcast inside the quad by DPP (quad_perm [m,m,m,m]), no LDS
    switch (c & 3) {
      case 0: xc = dpp_get<0x00, 0xf>(own); break;
      case 1: xc = dpp_get<0x55, 0xf>(own); break;
      case 2: xc = dpp_get<0xAA, 0xf>(own); break;
      default: xc = dpp_get<0xFF, 0xf>(own); break;
    }
    if (part == (c & 3)) bb[c >> 2] = xc;
    if (part > (c & 3)) bb[c >> 2] -= xc * s_d[(part + 4 * (c >> 2)) * TLD + c];
#pragma unroll
    for (int u = (c >> 2) + 1; u < 16; ++u) {
      bb[u] -= xc * s_d[(part + 4 * u) * TLD + c];
      asm volatile("" : "+v"(bb[u]));
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) Bt[(size_t)ri * ld + part + 4 * u] = bb[u];
}

// Block 0 must not store the factor over the diagonal block inside this launch: the other groups read that block as
// INPUT at their start, and nothing orders their start before block 0's end (on a GPU shared with other processes a
// group can start tens of microseconds late - seen as a spurious "not positive definite").  With more than one group
// the factor goes to `diag_scratch`; the trailing-update launch that follows copies it into place (it does not touch
// the diagonal block otherwise).
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ A, int p, int ld, int* __restrict__ info,
                                                    double* __restrict__ diag_scratch) {
  __shared__ __attribute__((aligned(16))) double s_d[BS * TLD];
  __shared__ __attribute__((aligned(16))) double s_col[2][BS];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  double* Add = A + (size_t)(p * BS) * ld + p * BS;
  load_tile(Add, ld, s_d);
  __syncthreads();
  int bad = 0;
  chol64_inplace(s_d, s_col, &bad);
  if (bad && b == 0 && tid == 0) atomicCAS(info, 0, p * BS + bad);
  if (b == 0) {
    const bool direct = gridDim.x == 1;              // last panel: nobody else reads the block
    for (int idx = tid; idx < BS * BS; idx += 256) {
      int r = idx >> 6, c = idx & 63;
      if (direct) Add[(size_t)r * ld + c] = s_d[r * TLD + c];
      else diag_scratch[idx] = s_d[r * TLD + c];
    }
    return;
  }
  if (tid < BS) s_col[1][tid] = fast_rcp(s_d[tid * TLD + tid]);
  __syncthreads();
  trsm64_right_lt(s_d, s_col[1], A + (size_t)((p + b) * BS) * ld + p * BS, ld);
}

// Trailing update A[I][J] -= L[I][p] L[J][p]^T for p < J <= I on f64 MFMA.
// Both 64x64 operand tiles are staged in LDS with coalesced loads; fragments are read with a
// leading dimension of 66 doubles (conflict-free for ds_read_b64).
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ A, int p, int nblk, int ld,
                                                     const double* __restrict__ diag_scratch, size_t zs) {
  ZRUN(A); ZRUN(diag_scratch);
  __shared__ __attribute__((aligned(16))) double s_a[BS * TLD];
  __shared__ __attribute__((aligned(16))) double s_b[BS * TLD];
  if (blockIdx.x == gridDim.x - 1) {                 // one extra group puts panel p's diagonal factor in place (see
    double* Add = A + (size_t)(p * BS) * ld + p * BS;  // k_chol_panel): off the critical path of the tile groups
    for (int idx = threadIdx.x; idx < BS * BS; idx += 256) Add[(size_t)(idx >> 6) * ld + (idx & 63)] = diag_scratch[idx];
    return;
  }
  // linear block id -> (I, J) in the lower triangle of the trailing (nblk-p-1)^2 tiles
  const int m = nblk - p - 1;
  int t = blockIdx.x, I = 0;
  while (t >= I + 1) { t -= I + 1; ++I; }
  const int J = t;
  if (I >= m) return;
  const int gi = p + 1 + I, gj = p + 1 + J;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  // the 16 elements of the target tile this thread updates are fetched up front, together with the operand tiles: read
  // after the MFMAs in a load-subtract-store loop they cost 16 dependent global round trips (the compiler cannot
  // hoist the loads over the stores to the same array), which was most of the kernel's 15.8 us
  double* dst = A + (size_t)(gi * BS) * ld + gj * BS;
  double cold[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) cold[q][r] = dst[(size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15)];
  load_tile(A + (size_t)(gi * BS) * ld + p * BS, ld, s_a);
  load_tile(A + (size_t)(gj * BS) * ld + p * BS, ld, s_b);
  __syncthreads();
  double4_t acc[4];
  for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < BS; kk += 4) {
    double a = s_a[(16 * w + (l & 15)) * TLD + kk + (l >> 4)];
    for (int q = 0; q < 4; ++q) {
      double bb = s_b[(16 * q + (l & 15)) * TLD + kk + (l >> 4)];     // B[k][j] = L[J-row j][k]
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      size_t off = (size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15);
      dst[off] = cold[q][r] - acc[q][r];
    }
}

// ---------------------------------------------------------------------------------------------
// Root inverse R = L^-1 (what gpytorch caches as `covar_cache`, stored here un-transposed).
// Step 1: invert every 64x64 diagonal block by forward substitution (one thread per column).
__global__ __launch_bounds__(256) void k_trinv_diag(const double* __restrict__ L, int ld, double* __restrict__ R) {
  __shared__ __attribute__((aligned(16))) double s_l[BS * TLD];
  __shared__ __attribute__((aligned(16))) double s_x[BS * TLD];   // s_x[c][r]: column c of the inverse, contiguous in r
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c = tid >> 2, part = tid & 3;                         // four lanes (same wave) per column
  load_tile(L + (size_t)(b * BS) * ld + b * BS, ld, s_l);
  __syncthreads();
  for (int r = 0; r < BS; ++r) {
    double s = 0.0;
    if (r > c)
      for (int m = c + part; m < r; m += 4) s += s_l[r * TLD + m] * s_x[c * TLD + m];
    s += dpp_get<0xB1, 0xf>(s);      // quad_perm [1,0,3,2]
    s += dpp_get<0x4E, 0xf>(s);      // quad_perm [2,3,0,1]
    if (part == 0) s_x[c * TLD + r] = (r < c) ? 0.0 : (((r == c) ? 1.0 : 0.0) - s) / s_l[r * TLD + r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
  double* dst = R + (size_t)(b * BS) * ld + b * BS;
  for (int idx = tid; idx < BS * BS; idx += 256) {
    int r = idx >> 6, cc = idx & 63;
    dst[(size_t)r * ld + cc] = s_x[cc * TLD + r];
  }
}

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
  ZRUN(L); ZRUN(R);
  __shared__ __attribute__((aligned(16))) double s_t[BS * TLD];   // 64x64 operand tile
  __shared__ __attribute__((aligned(16))) double s_xk[BS * 16];   // 64x16 block of X (or S)
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * 16;
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
  static const bool lanes4 = getenv("PCABO_GP_FOUR_LANE_ROWS") != nullptr;      // A/B: the earlier panel / inverse kernels
  for (int p = 0; p < nblk; ++p) {
    if (lanes4 && zb.B == 1) hipLaunchKernelGGL(k_chol_panel, dim3(nblk - p), dim3(256), 0, s, L, p, ld, info, diag_scratch);
    else launch_chol_panel_w(s, L, p, nblk - p, ld, info, diag_scratch, zb);
    int m = nblk - p - 1;
    if (m > 0)
      hipLaunchKernelGGL(k_chol_update, dim3(m * (m + 1) / 2 + 1, 1, zb.B), dim3(256), 0, s, L, p, nblk, ld, diag_scratch,
                         zb.zs);
  }
}
void launch_trinv(hipStream_t s, const double* L, int NP, int ld, double* R, ZB zb) {
  const int nblk = NP / BS;
  // (blocks above the diagonal are never written by anything: they keep the zeros of pcabo_ctx_create)
  static const bool lanes4 = getenv("PCABO_GP_FOUR_LANE_ROWS") != nullptr;
  if (lanes4 && zb.B == 1) hipLaunchKernelGGL(k_trinv_diag, dim3(nblk), dim3(256), 0, s, L, ld, R);
  else launch_trinv_diag_w(s, L, nblk, ld, R, zb);
  hipLaunchKernelGGL(k_trinv_cols, dim3(NP / 16, 1, zb.B), dim3(256), 0, s, L, nblk, ld, R, zb.zs);
}
void launch_alpha(hipStream_t s, const double* R, const double* ys, int n, int NP, int ld, double* tmp, double* alpha,
                  ZB zb) {
  hipLaunchKernelGGL(k_rmatvec, dim3(NP / 16, 1, zb.B), dim3(256), 0, s, R, ys, n, ld, tmp, zb.zs);
  hipLaunchKernelGGL(k_rtmatvec, dim3(NP / 64, 1, zb.B), dim3(64 * RTM_WAVES), 0, s, R, tmp, n, ld, alpha, zb.zs);
}
