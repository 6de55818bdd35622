// Wave-per-block variants of the two latency-bound steps of the GP conditioning (panel factorisation + solve,
// inverse of the diagonal blocks).  See DESIGN.md section 4.
#include "pcabo_internal.h"

#define BS PCABO_BS
#define WLD 65   // LDS leading dimension of the transposing tile (row reads and column writes both conflict-free)

// ---- Cholesky panel: one wave per 64x64 block, a matrix ROW per lane ------------------------------------------------
// The panel step is a chain of 64 dependent pivots; with four lanes per row (k_chol_panel) every link of the chain
// costs a work-group barrier and two LDS round trips (44-48 us per panel).  Here lane r keeps row r of the diagonal
// block D AND row r of its own off-diagonal block A in registers, the factor column of a step is broadcast lane by
// lane with v_readlane (SGPR operand of the FMAs) and the triangular solve A <- A L^-T rides along in the same loop:
// no barrier and no LDS between the loads and the stores.  Sub-panels of 16 columns: after each one the register
// arrays shift down by 16 so that the loop body (static register indices) is the same for all four.
__device__ inline double lane_get(double v, int lane) {          // lane: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <bool TRSM>
__device__ inline void panel_rows(double (&dr)[BS], double (&ar)[BS], double* s_d, double* s_a, int r, int& bad) {
  for (int jb = 0; jb < BS / 16; ++jb) {
    const int base = 16 * jb;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double piv = lane_get(dr[j], base + j);
      if (!(piv > 0.0)) { if (bad == 0) bad = base + j + 1; piv = 1.0; }      // uniform
      const double rs = fast_rsq(piv);
      const double dl = dr[j] * rs;                // lanes r >= base+j: L[r][base+j] (lane base+j: sqrt(piv))
      dr[j] = dl;
      double al = 0.0;
      if (TRSM) { al = ar[j] * rs; ar[j] = al; }
#pragma unroll
      for (int c = j + 1; c < 16; ++c) {
        const double s = lane_get(dl, base + c);   // L[base+c][base+j]
        dr[c] = fma(-dl, s, dr[c]);
        if (TRSM) ar[c] = fma(-al, s, ar[c]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the 16 finished columns go to the LDS tile (they leave through it anyway) ...
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      s_d[r * WLD + base + j] = (base + j <= r) ? dr[j] : 0.0;
      if (TRSM) s_a[r * WLD + base + j] = ar[j];
    }
    __syncthreads();
    // ... and the trailing columns of the panel read their multipliers L[base+c][base+j] from there as LDS broadcasts
    // (one address for the whole wave): no v_readlane / SGPR hazard on this, the larger, part.  16 columns at a time,
    // 4 independent accumulation chains between scheduling barriers.
    // Batches of 4 columns x 4 multiplier columns (16 LDS values, 4..8 independent FMA chains), double-buffered by
    // hand: the reads of batch t+1 are issued before the FMAs of batch t (left to itself the compiler emits
    // read - wait - use, one full LDS latency per pair of FMAs).
#pragma unroll
    for (int g = 1; g < BS / 16; ++g) {
      if (jb + g < BS / 16) {                      // uniform
        const double* ltile = s_d + (base + 16 * g) * WLD + base;        // rows base+16g.., columns base..
        double sb[2][16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) sb[0][u * 4 + jj] = ltile[u * WLD + jj];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int c4 = 4 * (t >> 2), j4 = 4 * (t & 3);
          if (t + 1 < 16) {
            const int nc4 = 4 * ((t + 1) >> 2), nj4 = 4 * ((t + 1) & 3);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) sb[(t + 1) & 1][u * 4 + jj] = ltile[(nc4 + u) * WLD + nj4 + jj];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int c = 16 * g + c4 + u, j = j4 + jj;
              const double sv = sb[t & 1][u * 4 + jj];
              dr[c] = fma(-dr[j], sv, dr[c]);
              if (TRSM) ar[c] = fma(-ar[j], sv, ar[c]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // everything else moves down
#pragma unroll
    for (int c = 0; c < BS - 16; ++c) {
      dr[c] = dr[c + 16];
      if (TRSM) ar[c] = ar[c + 16];
    }
  }
}

__global__ __launch_bounds__(64) void k_chol_panel_w(double* __restrict__ A, int p, int ld, int* __restrict__ info,
                                                     double* __restrict__ diag_scratch) {
  __shared__ double s_d[BS * WLD];
  __shared__ double s_a[BS * WLD];
  const int r = threadIdx.x, b = blockIdx.x;
  double* Add = A + (size_t)(p * BS) * ld + p * BS;
  double* Abd = A + (size_t)((p + b) * BS) * ld + p * BS;
  double dr[BS], ar[BS];
  // coalesced (lane = column), every load of the tile(s) in flight before the first use
#pragma unroll
  for (int i = 0; i < BS; ++i) dr[i] = Add[(size_t)i * ld + r];
  if (b > 0) {
#pragma unroll
    for (int i = 0; i < BS; ++i) ar[i] = Abd[(size_t)i * ld + r];
  }
#pragma unroll
  for (int i = 0; i < BS; ++i) s_d[i * WLD + r] = dr[i];
  if (b > 0) {
#pragma unroll
    for (int i = 0; i < BS; ++i) s_a[i * WLD + r] = ar[i];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < BS; ++c) dr[c] = s_d[r * WLD + c];
  int bad = 0;
  if (b == 0) {
    __syncthreads();
    panel_rows<false>(dr, ar, s_d, s_a, r, bad);
    if (bad && r == 0) atomicCAS(info, 0, p * BS + bad);
    __syncthreads();
    const bool direct = gridDim.x == 1;              // last panel: nobody else reads the block (see k_chol_panel)
#pragma unroll 8
    for (int i = 0; i < BS; ++i) {
      if (direct) Add[(size_t)i * ld + r] = s_d[i * WLD + r];
      else diag_scratch[i * BS + r] = s_d[i * WLD + r];
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < BS; ++c) ar[c] = s_a[r * WLD + c];
  __syncthreads();
  panel_rows<true>(dr, ar, s_d, s_a, r, bad);
  __syncthreads();
#pragma unroll 8
  for (int i = 0; i < BS; ++i) Abd[(size_t)i * ld + r] = s_a[i * WLD + r];
}

// Inverse of the 64x64 diagonal blocks of L, a COLUMN of the inverse per lane: right-looking forward substitution,
//   x[m] = acc[m] / L[m][m];  acc[r] -= L[r][m] x[m]  (r > m),
// all of x in registers, the L[r][m] are LDS broadcasts (same address for every lane, contiguous in r), no cross-lane
// traffic and no barrier inside the chain.
// step M: x[M] final, then x[r] -= L[r][M] x[M] for r > M.  The multipliers of step M+1 are read from LDS into the other
// half of `lb` BEFORE the FMAs of step M (explicit double buffering, see panel_rows).
template <int M>
struct InvSteps {
  static __device__ inline void run(double (&x)[BS], double (&lb)[2][BS], const double* s_lt, const double* s_rd) {
    if (M + 1 < BS) {
#pragma unroll
      for (int r = M + 2; r < BS; ++r) lb[(M + 1) & 1][r] = s_lt[(M + 1) * BS + r];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double xm = x[M] * s_rd[M];
    x[M] = xm;
#pragma unroll
    for (int r = M + 1; r < BS; ++r) x[r] = fma(-lb[M & 1][r], xm, x[r]);
    __builtin_amdgcn_sched_barrier(0);
    InvSteps<M + 1>::run(x, lb, s_lt, s_rd);
  }
};
template <>
struct InvSteps<BS> {
  static __device__ inline void run(double (&)[BS], double (&)[2][BS], const double*, const double*) {}
};

__global__ __launch_bounds__(64) void k_trinv_diag_w(const double* __restrict__ L, int ld, double* __restrict__ R) {
  __shared__ __attribute__((aligned(16))) double s_lt[BS * BS];    // s_lt[m][r] = L[r][m]
  __shared__ double s_rd[BS];
  const int c = threadIdx.x, b = blockIdx.x;
  const double* Lbb = L + (size_t)(b * BS) * ld + b * BS;
  // lane = column m of L, i = row: the write s_lt[m*64 + i] has a 64-double stride across lanes (bank conflicts, once)
  double x[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) x[i] = Lbb[(size_t)i * ld + c];        // all loads in flight
#pragma unroll
  for (int i = 0; i < BS; ++i) s_lt[c * BS + i] = x[i];
  s_rd[c] = 1.0 / Lbb[(size_t)c * ld + c];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < BS; ++r) x[r] = (r == c) ? 1.0 : 0.0;
  double lb[2][BS];
#pragma unroll
  for (int r = 1; r < BS; ++r) lb[0][r] = s_lt[r];
  InvSteps<0>::run(x, lb, s_lt, s_rd);
  double* dst = R + (size_t)(b * BS) * ld + b * BS;
#pragma unroll
  for (int r = 0; r < BS; ++r) dst[(size_t)r * ld + c] = (r >= c) ? x[r] : 0.0;
}


void launch_chol_panel_w(hipStream_t s, double* L, int p, int nblocks, int ld, int* info, double* diag_scratch) {
  hipLaunchKernelGGL(k_chol_panel_w, dim3(nblocks), dim3(64), 0, s, L, p, ld, info, diag_scratch);
}
void launch_trinv_diag_w(hipStream_t s, const double* L, int nblk, int ld, double* R) {
  hipLaunchKernelGGL(k_trinv_diag_w, dim3(nblk), dim3(64), 0, s, L, ld, R);
}
