// Wave-per-block variants of the two latency-bound steps of the GP conditioning (panel factorisation + solve,
// inverse of the diagonal blocks).  See DESIGN.md section 4.
#include "pcabo_internal.h"
#include <mutex>

#define BS PCABO_BS
#ifdef PCABO_ACQ_TIMING
__device__ unsigned long long g_panel_stamps[16];
extern "C" int pcabo_debug_panel_stamps(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_panel_stamps), sizeof(g_panel_stamps)) == hipSuccess ? 0 : -3;
}
#define PSTAMP(i) do { if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_panel_stamps[i] = wall_clock64(); } while (0)
#else
#define PSTAMP(i)
#endif
#define WLD 65   // LDS leading dimension of the transposing tile (row reads and column writes both conflict-free)

// ---- Cholesky panel: two waves per 64x64 block, a matrix ROW per lane in the sequential parts -------------------------
// The panel step is a chain of 64 dependent pivots; with four lanes per row (the first version of this kernel) every link
// of the chain cost a work-group barrier and two LDS round trips (44-48 us per panel).  Here, inside a 16-column sub-panel,
// a lane holds its row's 16 entries in registers:
//   wave 0, lane r: row r of the diagonal block D.  The factor column of a step is broadcast lane by lane with v_readlane
//     (SGPR operand of the FMAs) for the columns the chain needs next, through an LDS column for the others; the finished
//     16 columns go to the LDS tile, one barrier.
//   wave 1, lane r: row r of the work-group's off-diagonal block A.  It solves A <- A L^-T one sub-panel BEHIND wave 0
//     with every multiplier an LDS broadcast - no cross-lane traffic at all.
// Four barriers per panel instead of 64, and the two chains run on two SIMDs side by side.  LDS multipliers are
// double-buffered by hand: left to itself the compiler emits read - wait - use, a full LDS latency per pair of FMAs.
// The columns still to come live as 16 x 16 accumulator tiles of the matrix cores and take their rank-16 updates there.
__device__ inline double lane_get(double v, int lane) {          // lane: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}


// ---- Trailing updates on the matrix cores ---------------------------------------------------------------------------
// With a whole row per lane (round 1 / early round 2: k_chol_panel_w) the 16-wide rank updates of the columns still to come
// were the largest item of the timeline: 8.7 of 23 us, bound by the LDS broadcasts of the multipliers that BOTH waves pulled.
// v_mfma_f64_16x16x4 accumulates its four products as a chain of fused multiply-adds in ascending k ON TOP of the
// accumulator (profiles/tools/mfma_f64_order.hip: 512 000 of 512 000 elements equal the fma chain bit for bit), which is
// exactly what the VALU loop does per element - so the updates move to the matrix cores without changing a single bit:
// a wave keeps its 64 x 64 block as 16 accumulator tiles (the same 64 doubles per lane), and only the 16 columns of the
// current sub-panel take the trip through the LDS tile into the row-per-lane form that the pivot / substitution chain needs.
#define PANEL_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")     // LDS hand-over inside one wave
// acc[rt][ct][q] = T[16 rt + (l >> 4) + 4 q][16 ct + (l & 15)]
template <int JB, bool LOWER>
__device__ inline void tiles_colblock_to_lds(const double4_t (&acc)[4][4], double* s_t, int l) {
#pragma unroll
  for (int rt = LOWER ? JB : 0; rt < 4; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) s_t[(16 * rt + (l >> 4) + 4 * q) * WLD + 16 * JB + (l & 15)] = acc[rt][JB][q];
}
// acc[rt][ct] -= X[rows of rt][16 JB ..] * L[rows of ct][16 JB ..]^T for the column tiles ct > JB (LOWER: row tiles rt >= ct).
// s_x: the tile that holds the wave's own finished sub-panel, s_l: the factor's.
template <int JB, bool LOWER>
__device__ inline void trailing_mfma(double4_t (&acc)[4][4], const double* s_x, const double* s_l, int l) {
  const int base = 16 * JB;
  double bl[4][4];                                  // bl[ct][kk] = L[16 ct + (l & 15)][base + 4 kk + (l >> 4)]
#pragma unroll
  for (int ct = JB + 1; ct < 4; ++ct)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bl[ct][kk] = s_l[(16 * ct + (l & 15)) * WLD + base + 4 * kk + (l >> 4)];
#pragma unroll
  for (int rt = LOWER ? JB + 1 : 0; rt < 4; ++rt) {
    double ax[4];                                   // - X[16 rt + (l & 15)][base + 4 kk + (l >> 4)]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ax[kk] = LOWER ? -bl[rt][kk] : -s_x[(16 * rt + (l & 15)) * WLD + base + 4 * kk + (l >> 4)];
#pragma unroll
    for (int ct = JB + 1; ct < (LOWER ? rt + 1 : 4); ++ct)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[kk], bl[ct][kk], acc[rt][ct], 0, 0, 0);
  }
}

// wave 0, sub-panel JB of the diagonal block.
// The chain pivot -> 1/sqrt -> column -> next pivot is what the panel waits for, so only the two columns the NEXT two
// pivots need take their multiplier the quick way (v_readlane -> SGPR operand); the other columns' multipliers of pivot j
// go through an LDS column (one write, broadcast reads) and are applied one pivot LATER, after the chain of pivot j+1 has
// been issued - their LDS round trip runs under that chain.  Per element the updates still arrive in ascending pivot order
// with the same operands: same bits as the all-readlane loop (28 v_readlane per pivot, issue-bound: 110 ns per pivot).
template <int JB>
__device__ inline void panel_m_diag_step(double4_t (&acc)[4][4], double* s_d, double* s_rs, double* s_col, int r, int& bad) {
  constexpr int base = 16 * JB;
  double dr[16];
  if (JB > 0) { tiles_colblock_to_lds<JB, true>(acc, s_d, r); PANEL_LDS_SYNC(); }     // (JB = 0: the loaded tile is still there)
#pragma unroll
  for (int j = 0; j < 16; ++j) dr[j] = s_d[r * WLD + base + j];
  PSTAMP(2 + 3 * JB);
  double dl_prev = 0.0, mp[16];                    // pivot j-1's column and its multipliers for the deferred columns
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    // chain of pivot j
    double piv = lane_get(dr[j], base + j);
    if (!(piv > 0.0)) { if (bad == 0) bad = base + j + 1; piv = 1.0; }      // uniform
    const double rs = fast_rsq(piv);
    if (r == 0) s_rs[base + j] = rs;
    const double dl = dr[j] * rs;                // lanes r >= base+j: L[r][base+j] (lane base+j: sqrt(piv))
    dr[j] = dl;
    if (j + 1 < 16) dr[j + 1] = fma(-dl, lane_get(dl, base + j + 1), dr[j + 1]);
    __builtin_amdgcn_sched_barrier(0);
    // pivot j-1's deferred columns (their multipliers have had the time of the chain above to arrive)
    if (j > 0) {
#pragma unroll
      for (int c = j + 2; c < 16; ++c) dr[c] = fma(-dl_prev, mp[c], dr[c]);
    }
    // pivot j: the column after next, then its own deferred multipliers are requested
    if (j + 2 < 16) dr[j + 2] = fma(-dl, lane_get(dl, base + j + 2), dr[j + 2]);
    if (j + 3 < 16) {
      s_col[(j & 1) * BS + r] = dl;
      asm volatile("" ::: "memory");             // (a wave's LDS instructions execute in order: the reads below see the write)
#pragma unroll
      for (int c = j + 3; c < 16; ++c) mp[c] = s_col[(j & 1) * BS + base + c];
    }
    dl_prev = dl;
    __builtin_amdgcn_sched_barrier(0);
  }
  PSTAMP(3 + 3 * JB);
#pragma unroll
  for (int j = 0; j < 16; ++j) s_d[r * WLD + base + j] = (base + j <= r) ? dr[j] : 0.0;
  __syncthreads();                               // sub-panel JB is published
  PSTAMP(4 + 3 * JB);
  if (JB < 3) trailing_mfma<JB, true>(acc, s_d, s_d, r);
}

// wave 1, sub-panel JB of the work-group's off-diagonal block: A <- A L^-T one sub-panel behind wave 0
template <int JB>
__device__ inline void panel_m_solve_step(double4_t (&acc)[4][4], const double* s_d, const double* s_rs, double* s_a, int r) {
  constexpr int base = 16 * JB;
  double ar[16];
  if (JB > 0) { tiles_colblock_to_lds<JB, false>(acc, s_a, r); PANEL_LDS_SYNC(); }
#pragma unroll
  for (int j = 0; j < 16; ++j) ar[j] = s_a[r * WLD + base + j];
  __syncthreads();                               // wait for sub-panel JB of the factor
  const double* tri = s_d + base * WLD + base;   // the 16x16 triangle: L[base+c][base+j] = tri[c*WLD + j]
  double rsv[16], tb[3][16];                     // tb[j % 3][c]: column j of the triangle, prefetched two steps ahead
#pragma unroll
  for (int j = 0; j < 16; ++j) rsv[j] = s_rs[base + j];
#pragma unroll
  for (int c = 1; c < 16; ++c) tb[0][c] = tri[c * WLD];
#pragma unroll
  for (int c = 2; c < 16; ++c) tb[1][c] = tri[c * WLD + 1];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j + 2 < 15) {
#pragma unroll
      for (int c = j + 3; c < 16; ++c) tb[(j + 2) % 3][c] = tri[c * WLD + j + 2];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double al = ar[j] * rsv[j];
    ar[j] = al;
#pragma unroll
    for (int c = j + 1; c < 16; ++c) ar[c] = fma(-al, tb[j % 3][c], ar[c]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) s_a[r * WLD + base + j] = ar[j];
  if (JB < 3) { PANEL_LDS_SYNC(); trailing_mfma<JB, false>(acc, s_a, s_d, r); }
}

__global__ __launch_bounds__(128) void k_chol_panel_m(double* __restrict__ A, int p, int ld, int* __restrict__ info,
                                                      double* __restrict__ diag_scratch, size_t zs) {
  const XcdTile xt_ = xcd_tile();                    // (every group of a run reads the run's diagonal block: one XCD)
  ZRUNX(A); ZRUNX(info); ZRUNX(diag_scratch);
  __shared__ double s_d[BS * WLD];
  __shared__ double s_a[BS * WLD];
  __shared__ double s_rs[BS];
  __shared__ double s_col[2 * BS];
  const int r = threadIdx.x & 63, role = threadIdx.x >> 6, b = (int)xt_.x;
  PSTAMP(0);
  double* Add = A + (size_t)(p * BS) * ld + p * BS;
  double* Abd = A + (size_t)((p + b) * BS) * ld + p * BS;
  double4_t acc[4][4];
  {
    // coalesced (lane = column), every load of the tile in flight before the first use; into the LDS tile
    double a[BS];
    if (role == 0) {
#pragma unroll
      for (int i = 0; i < BS; ++i) a[i] = Add[(size_t)i * ld + r];
#pragma unroll
      for (int i = 0; i < BS; ++i) s_d[i * WLD + r] = a[i];
    } else if (b > 0) {
#pragma unroll
      for (int i = 0; i < BS; ++i) a[i] = Abd[(size_t)i * ld + r];
#pragma unroll
      for (int i = 0; i < BS; ++i) s_a[i * WLD + r] = a[i];
    }
  }
  __syncthreads();
  if (role == 0) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 1; ct <= rt; ++ct)               // (column tile 0 is read row-wise by the first step)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rt][ct][q] = s_d[(16 * rt + (r >> 4) + 4 * q) * WLD + 16 * ct + (r & 15)];
    PSTAMP(1);
    int bad = 0;
    panel_m_diag_step<0>(acc, s_d, s_rs, s_col, r, bad);
    panel_m_diag_step<1>(acc, s_d, s_rs, s_col, r, bad);
    panel_m_diag_step<2>(acc, s_d, s_rs, s_col, r, bad);
    panel_m_diag_step<3>(acc, s_d, s_rs, s_col, r, bad);
    if (bad && b == 0 && r == 0) atomicCAS(info, 0, p * BS + bad);
  } else if (b > 0) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 1; ct < 4; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rt][ct][q] = s_a[(16 * rt + (r >> 4) + 4 * q) * WLD + 16 * ct + (r & 15)];
    panel_m_solve_step<0>(acc, s_d, s_rs, s_a, r);
    panel_m_solve_step<1>(acc, s_d, s_rs, s_a, r);
    panel_m_solve_step<2>(acc, s_d, s_rs, s_a, r);
    panel_m_solve_step<3>(acc, s_d, s_rs, s_a, r);
  } else {
    for (int jb = 0; jb < BS / 16; ++jb) __syncthreads();      // block 0 has no off-diagonal block: keep the barriers paired
  }
  PSTAMP(14);
  __syncthreads();
  // the finished block leaves through both waves, 32 rows each
  const int i0 = 32 * role;
  if (b == 0) {
    const bool direct = gridDim.x == 1;              // last panel: nobody else reads the block (see kernels_gp.hip)
    double t[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i] = s_d[(i0 + i) * WLD + r];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (direct) Add[(size_t)(i0 + i) * ld + r] = t[i];
      else diag_scratch[(i0 + i) * BS + r] = t[i];
    }
  } else {
    double t[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i] = s_a[(i0 + i) * WLD + r];
#pragma unroll
    for (int i = 0; i < 32; ++i) Abd[(size_t)(i0 + i) * ld + r] = t[i];
  }
  PSTAMP(15);
}

// ---- One launch per panel: final update + panel of block column J, look-ahead of block column J+1 ----------------------
// Round 2 ran the left-looking factorisation as two launches per panel (k_chol_lookback over ALL earlier panels, then
// k_chol_panel_m), and the look-back's chain of J tile steps sat in front of every panel.  Here the chain leaves the
// critical path: launch J holds
//   * panel groups (tile (J+b, J), b = 0 .. nblk-J-1): apply the ONE update that could not be known earlier,
//       A[J+b][J] -= L[J+b][J-1] L[J][J-1]^T   (and the same for the diagonal tile, which every group factors for itself),
//     on the matrix cores with all four waves, then factor / solve exactly as k_chol_panel_m did (waves 0 and 1);
//   * look-ahead groups (tile (I, J+1), I = J+1 .. nblk-1): A[I][J+1] -= sum_{p<J} L[I][p] L[J+1][p]^T - everything block
//     column J+1 needs EXCEPT panel J, which is being factored in this very launch - they run beside the panel groups and
//     the next launch's panel groups add the missing term;
//   * one group that puts panel J-1's diagonal factor into its place (the panel groups of a launch all read their
//     diagonal tile as INPUT, so the factor leaves through a scratch tile; two scratch tiles alternate because the
//     writer of panel J and the copier of panel J-1 now share a launch).
// Per element the arithmetic is what it was: the products of panel p accumulate from zero over ascending k on
// v_mfma_f64_16x16x4 and are then subtracted from the tile, panels in ascending order (the value just passes through
// memory between p = J-2 and p = J-1) - tests/golden/gp_factor_hashes.json holds bit for bit.
#define SLD PCABO_TLD
__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ A, int J, int nblk, int ld, int* __restrict__ info,
                                                   double* __restrict__ diag_scratch, size_t zs) {
  ZRUN(A); ZRUN(info); ZRUN(diag_scratch);
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int nP = nblk - J, nLA = J >= 1 ? nblk - J - 1 : 0;
  const int bx = blockIdx.x;
  if (bx == nP + nLA) {                                // (only launched for J >= 1) panel J-1's diagonal factor -> its place
    double* Add = A + (size_t)((J - 1) * BS) * ld + (J - 1) * BS;
    const double* src = diag_scratch + (size_t)((J - 1) & 1) * BS * BS;
    for (int idx = tid; idx < BS * BS; idx += 256) Add[(size_t)(idx >> 6) * ld + (idx & 63)] = src[idx];
    return;
  }
  double* s_x = s_mem;                                 // operand tiles of the matrix-core updates, leading dimension 66
  double* s_y = s_mem + BS * SLD;
  if (bx >= nP) {
    // ---- look-ahead: tile (I, J+1) takes the panels p < J --------------------------------------------------------
    const int Jc = J + 1, I = Jc + (bx - nP);
    double* dst = A + (size_t)(I * BS) * ld + Jc * BS;
    const double* Arow = A + (size_t)(I * BS) * ld;    // L[I][p] tiles
    const double* Brow = A + (size_t)(Jc * BS) * ld;   // L[J+1][p] tiles
    double cold[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) cold[q][r] = dst[(size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15)];
    double pa[16], pb[16];
    auto fetch = [&](int p) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
        pa[u] = Arow[(size_t)r * ld + p * BS + c];
        pb[u] = Brow[(size_t)r * ld + p * BS + c];
      }
    };
    // two sets of operand tiles in the LDS: while the matrix cores work on panel p, panel p+1's tiles (fetched one step
    // earlier) go into the other set and panel p+2's loads are issued - one barrier per step, a whole step to hide a load
    auto put = [&](int set) {
      double* bx_ = s_mem + (size_t)set * 2 * BS * SLD;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
        bx_[r * SLD + c] = pa[u];
        bx_[BS * SLD + r * SLD + c] = pb[u];
      }
    };
    fetch(0);
    put(0);
    if (J > 1) fetch(1);
    __syncthreads();
    for (int p = 0; p < J; ++p) {
      const double* cx = s_mem + (size_t)(p & 1) * 2 * BS * SLD;
      const double* cy = cx + BS * SLD;
      double4_t acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
      for (int kk = 0; kk < BS; kk += 4) {
        const double a = cx[(16 * w + (l & 15)) * SLD + kk + (l >> 4)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double bb = cy[(16 * q + (l & 15)) * SLD + kk + (l >> 4)];
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
        }
      }
      if (p + 1 < J) {
        put((p + 1) & 1);
        if (p + 2 < J) fetch(p + 2);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[q][r] -= acc[q][r];
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15)] = cold[q][r];
    return;
  }
  // ---- panel group b: tile (J+b, J) ----------------------------------------------------------------------------------
  const int b = bx;
  double* s_d = s_mem;                                 // the panel's two tiles (leading dimension 65) share the operand
  double* s_a = s_mem + BS * WLD;                      // tiles' memory: a barrier separates the two uses
  double* s_rs = s_mem + 4 * BS * SLD;
  double* s_col = s_rs + BS;
  double* Add = A + (size_t)(J * BS) * ld + J * BS;
  double* Abd = A + (size_t)((J + b) * BS) * ld + J * BS;
  if (J == 0) {
    // nothing to add: the tiles go straight into the LDS (coalesced: lane = column, all loads in flight first)
    if (w < 2 && (w == 0 || b > 0)) {
      const double* src = w == 0 ? Add : Abd;
      double* dstl = w == 0 ? s_d : s_a;
      double a[BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) a[i] = src[(size_t)i * ld + l];
#pragma unroll
      for (int i = 0; i < BS; ++i) dstl[i * WLD + l] = a[i];
    }
  } else {
    // the one update that had to wait for panel J-1:  D -= Y Y^T,  T -= X Y^T  with  X = L[J+b][J-1], Y = L[J][J-1]
    const double* X = A + (size_t)((J + b) * BS) * ld + (J - 1) * BS;
    const double* Y = A + (size_t)(J * BS) * ld + (J - 1) * BS;
    double told[4][4], dold[4][4], pa[16], pb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
      pb[u] = Y[(size_t)r * ld + c];
      pa[u] = b > 0 ? X[(size_t)r * ld + c] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t off = (size_t)(16 * w + (l >> 4) + 4 * r) * ld + 16 * q + (l & 15);
        dold[q][r] = Add[off];
        told[q][r] = b > 0 ? Abd[off] : 0.0;
      }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
      s_y[r * SLD + c] = pb[u];
      if (b > 0) s_x[r * SLD + c] = pa[u];
    }
    __syncthreads();
    double4_t accd[4], acct[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { accd[q] = (double4_t){0.0, 0.0, 0.0, 0.0}; acct[q] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
    for (int kk = 0; kk < BS; kk += 4) {
      const double ay = s_y[(16 * w + (l & 15)) * SLD + kk + (l >> 4)];
      const double ax = b > 0 ? s_x[(16 * w + (l & 15)) * SLD + kk + (l >> 4)] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bb = s_y[(16 * q + (l & 15)) * SLD + kk + (l >> 4)];
        accd[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, bb, accd[q], 0, 0, 0);
        if (b > 0) acct[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bb, acct[q], 0, 0, 0);
      }
    }
    __syncthreads();                                   // every wave has read the operand tiles: their memory becomes s_d / s_a
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + (l >> 4) + 4 * r, col = 16 * q + (l & 15);
        s_d[row * WLD + col] = dold[q][r] - accd[q][r];
        if (b > 0) s_a[row * WLD + col] = told[q][r] - acct[q][r];
      }
  }
  __syncthreads();
  const int r = l;
  double4_t acc[4][4];
  if (w == 0) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 1; ct <= rt; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rt][ct][q] = s_d[(16 * rt + (r >> 4) + 4 * q) * WLD + 16 * ct + (r & 15)];
    int bad = 0;
    panel_m_diag_step<0>(acc, s_d, s_rs, s_col, r, bad);
    panel_m_diag_step<1>(acc, s_d, s_rs, s_col, r, bad);
    panel_m_diag_step<2>(acc, s_d, s_rs, s_col, r, bad);
    panel_m_diag_step<3>(acc, s_d, s_rs, s_col, r, bad);
    if (bad && b == 0 && r == 0) atomicCAS(info, 0, J * BS + bad);
  } else if (w == 1 && b > 0) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 1; ct < 4; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rt][ct][q] = s_a[(16 * rt + (r >> 4) + 4 * q) * WLD + 16 * ct + (r & 15)];
    panel_m_solve_step<0>(acc, s_d, s_rs, s_a, r);
    panel_m_solve_step<1>(acc, s_d, s_rs, s_a, r);
    panel_m_solve_step<2>(acc, s_d, s_rs, s_a, r);
    panel_m_solve_step<3>(acc, s_d, s_rs, s_a, r);
  } else {
    for (int jb = 0; jb < BS / 16; ++jb) __syncthreads();      // the other waves keep the four barriers of the sub-panels paired
  }
  __syncthreads();
  // the finished block leaves through all four waves, 16 rows each
  const int i0 = 16 * w;
  const double* srcl = b == 0 ? s_d : s_a;
  double t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = srcl[(i0 + i) * WLD + r];
  if (b == 0) {
    const bool direct = nP == 1;                       // last panel: nobody else reads the block
    double* scr = diag_scratch + (size_t)(J & 1) * BS * BS;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (direct) Add[(size_t)(i0 + i) * ld + r] = t[i];
      else scr[(i0 + i) * BS + r] = t[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) Abd[(size_t)(i0 + i) * ld + r] = t[i];
  }
}
#define CHOL_STEP_LDS ((4 * BS * SLD + 3 * BS) * sizeof(double))     // look-ahead: two sets of two operand tiles

// Inverse of the 64x64 diagonal blocks of L, a COLUMN of the inverse per lane: right-looking forward substitution,
//   x[m] = acc[m] / L[m][m];  acc[r] -= L[r][m] x[m]  (r > m),
// all of x in registers, the L[r][m] are LDS broadcasts (same address for every lane, contiguous in r), no cross-lane
// traffic and no barrier inside the chain.
// Step M: x[M] final, then x[r] -= L[r][M] x[M] for r > M, in chunks of 16 rows.  The chunks of all steps form one
// compile-time sequence; the multipliers of the chunk after the next are read from LDS into the free third of `lb` before
// the FMAs of the current one (explicit double buffering, see panel_rows; 16-value buffers keep everything in the 256
// architectural VGPRs - whole-column buffers ended up in AGPRs with a v_accvgpr_read per operand).
constexpr bool inv_last(int m, int ch) { return m + 1 + 16 * ch + 16 >= BS; }
constexpr int inv_next_m(int m, int ch) { return inv_last(m, ch) ? m + 1 : m; }
constexpr int inv_next_ch(int m, int ch) { return inv_last(m, ch) ? 0 : ch + 1; }
template <int M, int CH, int P>          // P: buffer (of 3) holding this chunk's multipliers; prefetch distance 2
struct InvChunk {
  static constexpr int r0 = M + 1 + 16 * CH;
  static constexpr int NM = inv_next_m(M, CH), NCH = inv_next_ch(M, CH);
  static constexpr int N2M = inv_next_m(NM, NCH), N2CH = inv_next_ch(NM, NCH), n2r0 = N2M + 1 + 16 * N2CH;
  static __device__ inline void run(double (&x)[BS], double (&lb)[3][16], const double* s_lt, const double* s_rd,
                                    double& xm) {
    if (N2M < BS - 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (n2r0 + i < BS) lb[(P + 2) % 3][i] = s_lt[N2M * BS + n2r0 + i];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (CH == 0) { xm = x[M] * s_rd[M]; x[M] = xm; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (r0 + i < BS) x[r0 + i] = fma(-lb[P][i], xm, x[r0 + i]);
    __builtin_amdgcn_sched_barrier(0);
    InvChunk<NM, NCH, (P + 1) % 3>::run(x, lb, s_lt, s_rd, xm);
  }
};
template <int CH, int P>
struct InvChunk<BS - 1, CH, P> {
  static __device__ inline void run(double (&x)[BS], double (&)[3][16], const double*, const double* s_rd, double&) {
    x[BS - 1] *= s_rd[BS - 1];
  }
};

__global__ __launch_bounds__(64) void k_trinv_diag_w(const double* __restrict__ L, int ld, double* __restrict__ R,
                                                     size_t zs) {
  ZRUN(L); ZRUN(R);
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
  double lb[3][16], xm = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { lb[0][i] = s_lt[1 + i]; lb[1][i] = s_lt[17 + i]; }     // chunks (0,0) and (0,1)
  InvChunk<0, 0, 0>::run(x, lb, s_lt, s_rd, xm);
  double* dst = R + (size_t)(b * BS) * ld + b * BS;
#pragma unroll
  for (int r = 0; r < BS; ++r) dst[(size_t)r * ld + c] = (r >= c) ? x[r] : 0.0;
}


void launch_chol_panel(hipStream_t s, double* L, int p, int nblocks, int ld, int* info, double* diag_scratch, ZB zb) {
  hipLaunchKernelGGL(k_chol_panel_m, dim3(nblocks, 1, zb.B), dim3(128), 0, s, L, p, ld, info, diag_scratch, zb.zs);
}
// diag_scratch: TWO 64 x 64 tiles
void launch_chol_steps(hipStream_t s, double* L, int NP, int ld, int* info, double* diag_scratch, ZB zb) {
  static std::mutex attr_mu;
  static bool attr_done[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
  {
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!attr_done[dev]) {
      if (hipFuncSetAttribute((const void*)k_chol_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_STEP_LDS) != hipSuccess) return;
      attr_done[dev] = true;
    }
  }
  const int nblk = NP / BS;
  for (int J = 0; J < nblk; ++J) {
    const int groups = (nblk - J) + (J >= 1 ? nblk - J - 1 : 0) + (J >= 1 ? 1 : 0);
    hipLaunchKernelGGL(k_chol_step, dim3(groups, 1, zb.B), dim3(256), CHOL_STEP_LDS, s, L, J, nblk, ld, info, diag_scratch, zb.zs);
  }
}
void launch_trinv_diag_w(hipStream_t s, const double* L, int nblk, int ld, double* R, ZB zb) {
  hipLaunchKernelGGL(k_trinv_diag_w, dim3(nblk, 1, zb.B), dim3(64), 0, s, L, ld, R, zb.zs);
}
