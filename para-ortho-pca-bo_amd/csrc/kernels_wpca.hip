// Rank-weighted PCA on gfx950.
//
// Replaces numpy + sklearn.decomposition.PCA as used by
// /root/reference/Algorithms/BayesianOptimization/PCA_BO.py:316-434.
// All matrices here are small (d <= 128, n <= ~1k): the work is latency-bound, so the design is
// "few launches, one work-group where a phase is sequential, MFMA for the one GEMM-like
// contraction (the d x d covariance W^T W)".
#include "pcabo_internal.h"
#include <cstdlib>

#define WP_THREADS 1024

// ---- deterministic block-wide sum (tree in LDS), result broadcast to every thread -----------
// The sum is DEFINED as the stride-halving tree  s[t] += s[t + off], off = 512, 256, .., 1  (round 1 executed it literally:
// eleven work-group barriers of sixteen waves, ~5 us per sum).  Same tree, two barriers: thread t = 64 j + l holds leaf (j, l);
// the levels 512 .. 64 pair wave j with wave j + 8, 4, 2, 1 at a fixed lane - wave 0 adds them for its lane from the LDS in
// exactly that order - and the levels 32 .. 1 pair lane l with lane l + off inside wave 0.  Bit-identical to the literal tree.
__device__ inline double block_sum_1024(double v, double* s_red) {
  const int tid = threadIdx.x, l = tid & 63;
  s_red[tid] = v;
  __syncthreads();
  if (tid < 64) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = s_red[64 * j + l] + s_red[64 * (j + 8) + l];      // level 512
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = a[j] + a[j + 4];                                   // level 256
    a[0] = a[0] + a[2]; a[1] = a[1] + a[3];                                              // level 128
    double x = a[0] + a[1];                                                              // level 64
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {                                             // levels 32 .. 1
      const double y = __shfl_down(x, off, 64);
      if (l < off) x = x + y;
    }
    if (l == 0) s_red[0] = x;
  }
  __syncthreads();
  const double r = s_red[0];
  __syncthreads();
  return r;
}

// Row A (device variant): 1-based rank of each f, best first, ties broken by index.
__global__ void k_rank(const double* __restrict__ f, int n, int maximize, long long* __restrict__ ranks) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double fi = maximize ? -f[i] : f[i];
  int r = 1;
  for (int j = 0; j < n; ++j) {
    double fj = maximize ? -f[j] : f[j];
    r += (fj < fi) || (fj == fi && j < i);
  }
  ranks[i] = r;
}

// Rows A+B: weights, data mean, weighted+noised matrix W, its column mean, centred Wc (n4 x DP,
// zero padded so the MFMA covariance kernel can run whole 16x4 fragments).
// One work-group; thread (c = tid % CP, g = tid / CP) owns column c of the rows i = g (mod G), so
// every pass touches only the thread's own elements (no inter-thread global dependencies).
__global__ __launch_bounds__(WP_THREADS) void k_wpca_prep(
    const double* __restrict__ X, const long long* __restrict__ ranks, const double* __restrict__ noise,
    int n, int d, int DP, double* __restrict__ weights, double* __restrict__ data_mean,
    double* __restrict__ pca_mean, double* __restrict__ Wc, size_t zs) {
  ZRUN(X); ZRUN(ranks); ZRUN(noise); ZRUN(weights); ZRUN(data_mean); ZRUN(pca_mean); ZRUN(Wc);
  __shared__ double s_red[WP_THREADS];
  __shared__ double s_col[PCABO_MAXD];
  const int tid = threadIdx.x;
  const int CP = DP <= 64 ? 64 : 128;
  const int G = WP_THREADS / CP;
  const int c = tid % CP, g = tid / CP;
  const int n4 = (n + 3) & ~3;

  // pre-weights ln n - ln r_i and their sum (PCA_BO.py:336-339)
  double loc = 0.0;
  const double logn = log((double)n);
  for (int i = tid; i < n; i += WP_THREADS) {
    double pw = logn - log((double)ranks[i]);
    weights[i] = pw;
    loc += pw;
  }
  const double tot = block_sum_1024(loc, s_red);
  for (int i = tid; i < n; i += WP_THREADS) weights[i] = weights[i] / tot;
  __syncthreads();   // weights[] written by other threads are read below

  // column means of X (PCA_BO.py:364)
  // Up to WP_ROWS rows per thread the thread's elements stay in registers from here to the single store of Wc: one
  // pass over X with all loads in flight instead of three dependent passes (the last one a load-subtract-store loop
  // on Wc, i.e. one global round trip per row).  Same operations in the same order, so the same bits.
  constexpr int WP_ROWS = 32;
  const bool in_regs = (n4 + G - 1) / G <= WP_ROWS;
  double xr[WP_ROWS];
  double acc = 0.0;
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < WP_ROWS; ++u) {
      const int i = g + u * G;
      xr[u] = (c < d && i < n) ? X[(size_t)i * d + c] : 0.0;
      if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);       // 8 loads in flight at a time (128-VGPR budget)
    }
    if (c < d) {
#pragma unroll
      for (int u = 0; u < WP_ROWS; ++u)
        if (g + u * G < n) acc += xr[u];
    }
  } else if (c < d) {
    for (int i = g; i < n; i += G) acc += X[(size_t)i * d + c];
  }
  s_red[tid] = acc;
  __syncthreads();
  if (tid < CP) {
    double s = 0.0;
    for (int gg = 0; gg < G; ++gg) s += s_red[gg * CP + tid];
    s_col[tid] = s / (double)n;
    if (tid < d) data_mean[tid] = s_col[tid];
  }
  __syncthreads();
  const double mu = (c < d) ? s_col[c] : 0.0;
  __syncthreads();

  // W = (X - mu) * sqrt(w_i) + noise (PCA_BO.py:365-377), column sums of W
  acc = 0.0;
  if (in_regs) {
    if (c < DP) {
#pragma unroll
      for (int u = 0; u < WP_ROWS; ++u) {
        const int i = g + u * G;
        double w = 0.0;
        if (c < d && i < n) {
          w = (xr[u] - mu) * sqrt(weights[i]);
          if (noise) w += noise[(size_t)i * d + c];
        }
        xr[u] = w;
        if (i < n4) acc += w;
        if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);     // 8 rows' loads in flight at a time (128-VGPR budget)
      }
    }
  } else if (c < DP) {
    for (int i = g; i < n4; i += G) {
      double w = 0.0;
      if (c < d && i < n) {
        w = (X[(size_t)i * d + c] - mu) * sqrt(weights[i]);
        if (noise) w += noise[(size_t)i * d + c];
      }
      Wc[(size_t)i * DP + c] = w;
      acc += w;
    }
  }
  s_red[tid] = acc;
  __syncthreads();
  if (tid < CP) {
    double s = 0.0;
    for (int gg = 0; gg < G; ++gg) s += s_red[gg * CP + tid];
    s_col[tid] = s / (double)n;                       // sklearn: mean_ = mean(W, axis=0)
    if (tid < d) pca_mean[tid] = s_col[tid];
  }
  __syncthreads();
  const double mw = (c < d) ? s_col[c] : 0.0;
  if (in_regs) {
    if (c < DP) {
#pragma unroll
      for (int u = 0; u < WP_ROWS; ++u) {
        const int i = g + u * G;
        if (i < n4) Wc[(size_t)i * DP + c] = (c < d && i < n) ? xr[u] - mw : 0.0;
        if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (c < d) {
    for (int i = g; i < n; i += G) Wc[(size_t)i * DP + c] -= mw;
  }
}

// Row C (covariance): C = Wc^T Wc / (n-1), DP x DP, one 16x16 tile per work-group; the four waves
// split the n-long contraction and are summed in a fixed order (deterministic).
// v_mfma_f64_16x16x4_f64: lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15]; D holds
// rows (l>>4)+4r, column l&15.
__global__ __launch_bounds__(256) void k_cov(const double* __restrict__ Wc, int n4, int DP, double inv_nm1,
                                             double* __restrict__ C, size_t zs) {
  ZRUN(Wc); ZRUN(C);
  __shared__ double s_acc[4][4][64];
  const int ti = blockIdx.x, tj = blockIdx.y;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int steps = n4 >> 2;
  for (int s = w; s < steps; s += 4) {
    const size_t row = (size_t)(4 * s + (l >> 4)) * DP;
    double a = Wc[row + ti * 16 + (l & 15)];
    double b = Wc[row + tj * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) s_acc[w][r][l] = acc[r];
  __syncthreads();
  if (w == 0) {
    for (int r = 0; r < 4; ++r) {
      double s = ((s_acc[0][r][l] + s_acc[1][r][l]) + s_acc[2][r][l]) + s_acc[3][r][l];
      C[(size_t)(ti * 16 + (l >> 4) + 4 * r) * DP + tj * 16 + (l & 15)] = s * inv_nm1;
    }
  }
}

// Row C (eigen-decomposition): one-sided (Hestenes) Jacobi on G = C (symmetric PSD, d x d) in LDS.
// Plane rotations are applied to column pairs until all columns are mutually orthogonal; then
// G = C V = V diag(lambda): column norms are the eigenvalues and the normalised columns the
// eigenvectors.  Round-robin ordering gives d/2 independent pairs per round, each handled by a
// group of LP lanes (shuffle reductions, no LDS traffic for the dot products).
#define JAC_THREADS 1024
__device__ inline double rcp_1nr(double x) { double r = __builtin_amdgcn_rcp(x); return fma(fma(-x, r, 1.0), r, r); }
__device__ inline double rsq_1nr(double x) { double y = __builtin_amdgcn_rsq(x); return y * fma(-0.5 * x * y, y, 1.5); }
#ifdef PCABO_ACQ_TIMING
__device__ unsigned long long g_jac_stamps[8];
extern "C" int pcabo_debug_jacobi_stamps(unsigned long long* out4) {
  return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_jac_stamps), sizeof(g_jac_stamps)) == hipSuccess ? 0 : -3;
}
#endif
// Warm start: if the eigenvectors V0 of the previous BO iteration are given (and orthonormal), iterate on
// G0 = C V0 instead of C: still G = C V with V orthogonal, but the columns start almost orthogonal, so two or
// three sweeps suffice instead of ~8.  The result does not depend on the start beyond rounding.
struct PcaSelect {        // arguments of the selection step that ends the kernel (rows C, PCA_BO.py:389-399)
  int n; double var_threshold; int n_components; double* comps; double* evr; int* k_dev; HostMirror* hm;
};
// NT: rows of a column pair that a lane keeps in registers between the dot products and the rotation (16 lanes per pair: NT = 4 covers
// d <= 64, NT = 8 the whole range d <= 128 - configs[4]'s d = 100 used to take the second loop, two LDS reads per element and round)
template <int NT>
__global__ __launch_bounds__(JAC_THREADS) void k_jacobi(const double* __restrict__ C, int d, int DP,
                                                        const double* __restrict__ V0, double* __restrict__ Gout,
                                                        double* __restrict__ lam, int* __restrict__ sweeps_out,
                                                        PcaSelect sel, size_t zs, size_t hzs) {
  ZRUN(C); ZRUN(V0); ZRUN(Gout); ZRUN(lam); ZRUN(sweeps_out); ZRUN(sel.comps); ZRUN(sel.evr); ZRUN(sel.k_dev);
  sel.hm = zrun(sel.hm, hzs, blockIdx.z);
  extern __shared__ __attribute__((aligned(16))) double s_g[];   // d columns of length d, column-major, stride LD
  __shared__ double s_lam[PCABO_MAXD], s_sgn[PCABO_MAXD];
  __shared__ int s_order[PCABO_MAXD];
  const int tid = threadIdx.x;
  const int LD = d | 1;                    // odd stride: column p and q of a pair never share banks systematically
  volatile int& s_rot = *reinterpret_cast<volatile int*>(s_g + (size_t)d * LD);   // flag lives after the matrix
  if (tid == 0) s_rot = 0;
  __syncthreads();
  // Warm start from LDS copies of C and V0 when they fit (3 d^2 doubles; d <= 64): read once, coalesced - the d-long
  // loops below otherwise walk global memory (50 us of the kernel at d = 40).
  const bool staged = V0 && d <= 64;           // the launcher sizes the dynamic LDS for it
  double* s_c = s_g + (size_t)d * LD + 2;      // C row-major, stride LD      (after the matrix and the flag)
  double* s_v = s_c + (size_t)d * LD;          // V0: column `col` contiguous, stride LD
  if (staged) {
    for (int idx = tid; idx < d * d; idx += (int)blockDim.x) {
      const int a = idx / d, b = idx % d;
      s_c[a * LD + b] = C[(size_t)a * DP + b];
      s_v[a * LD + b] = V0[(size_t)a * d + b];
    }
    __syncthreads();
  }
  if (V0) {                                  // usable only if every column has unit norm (none collapsed to zero)
    if (staged) {
      for (int col = tid; col < d; col += (int)blockDim.x) {
        double a = 0.0;
        for (int r = 0; r < d; ++r) { double v = s_v[col * LD + r]; a += v * v; }
        if (!(fabs(a - 1.0) < 1e-8)) s_rot = 1;
      }
    } else {                                 // from global memory: a wave per column, coalesced (a thread per column walked d lines one by one)
      for (int col = tid >> 6; col < d; col += (int)blockDim.x >> 6) {
        double a = 0.0;
        for (int r = tid & 63; r < d; r += 64) { const double v = V0[(size_t)col * d + r]; a += v * v; }
        a = wave_sum(a);
        if ((tid & 63) == 0 && !(fabs(a - 1.0) < 1e-8)) s_rot = 1;
      }
    }
  }
  __syncthreads();
  const bool warm = V0 && !s_rot;
  __syncthreads();
  // Range guard.  The sweeps square the entries of G = C V (column norms, a b in the convergence test): with |x| ~ 1e80 -
  // reached by the reference's own dynamics on some functions, where every out-of-box candidate widens the next search
  // box (PCA_BO.py:253,260-263,558-573) - C ~ 1e160 and those squares leave the double range, while LAPACK's eigh
  // behind sklearn scales such a matrix and carries on.  So C is scaled by a power of FOUR when its largest entry (on the
  // diagonal: C is a covariance) lies outside 2^+-100.  Eigenvectors and variance ratios do not depend on the scale, and
  // every operation of the sweep (products, sums, rcp / rsq estimates and their Newton steps, the square roots at the
  // end) commutes with a power of four exactly - inside that range the factor is 1 and nothing changes.
  if (tid < d) s_lam[tid] = fabs(C[(size_t)tid * DP + tid]);
  __syncthreads();
  double cscale = 1.0;
  {
    double mx = 0.0;
    for (int j = 0; j < d; ++j) mx = s_lam[j] > mx ? s_lam[j] : mx;      // (LDS broadcast reads; NaN / inf: left alone)
    if (mx > 0.0 && mx < INFINITY) {
      const int e = ilogb(mx);
      if (e > 100 || e < -100) cscale = ldexp(1.0, -(e & ~1));
    }
  }
  __syncthreads();                           // (s_lam is written again below)
  for (int idx = tid; idx < d * d; idx += (int)blockDim.x) {
    int col = idx / d, row = idx % d;
    double v;
    if (warm && staged) {
      v = 0.0;
      const double* crow = s_c + row * LD;
      const double* vcol = s_v + col * LD;
      for (int j = 0; j < d; ++j) v += (crow[j] * cscale) * vcol[j];
    } else if (warm) {
      // C is symmetric BIT FOR BIT (k_cov forms C[i][j] and C[j][i] from the same products in the same order), so row `row` is read as
      // column `row`: consecutive lanes, consecutive addresses (reading the row itself put every lane on a line of its own)
      v = 0.0;
      const double* ccol = C + row;
      const double* vcol = V0 + (size_t)col * d;
      for (int j = 0; j < d; ++j) v += (ccol[(size_t)j * DP] * cscale) * vcol[j];
    } else {
      v = C[(size_t)row * DP + col] * cscale;
    }
    s_g[col * LD + row] = v;
  }
  if (tid == 0) s_rot = 0;
  __syncthreads();

  const int de = (d + 1) & ~1;             // even number of players (a virtual empty column if d is odd)
  const int npairs = de / 2;
  const int LP = (int)blockDim.x >= npairs * 16 ? 16 : 8;      // lanes per column pair (launcher sizes the block)
  const int grp = tid / LP, lane = tid % LP;
  const double tol = 1e-15;
  // round-robin schedule: group g > 0 plays p = (round + g) mod m against q = (round - g) mod m, group 0 plays
  // `round` against the fixed last player; both indices advance by one (mod m) per round - kept incrementally
  // (two runtime modulo operations per round were ~15 % of the round)
  const int m_rr = de - 1;
  int pr = grp < npairs ? grp % m_rr : 0, qr = grp < npairs ? (m_rr - grp % m_rr) % m_rr : 0;
  int sweep = 0;
#ifdef PCABO_ACQ_TIMING
  const unsigned long long jc0 = clock64(), jw0 = wall_clock64();
  unsigned long long ph_dot = 0, ph_rot = 0, ph_wr = 0, ph_bar = 0, ph_t = 0;
#define JSTAMP(acc) do { if (tid == 0) { __builtin_amdgcn_s_waitcnt(0); unsigned long long n_ = clock64(); acc += n_ - ph_t; ph_t = n_; } } while (0)
#else
#define JSTAMP(acc) do {} while (0)
#endif
  for (; sweep < 40; ++sweep) {
    for (int round = 0; round < de - 1; ++round) {
#ifdef PCABO_ACQ_TIMING
      if (tid == 0) ph_t = clock64();
#endif
      if (grp < npairs) {
        int p = pr, q = grp == 0 ? de - 1 : qr;
        pr = pr + 1 == m_rr ? 0 : pr + 1;
        qr = qr + 1 == m_rr ? 0 : qr + 1;
        if (p > q) { int t = p; p = q; q = t; }
        if (q < d) {
          double* gp = s_g + p * LD;
          double* gq = s_g + q * LD;
          // the lane's rows stay in registers between the dot products and the rotation (d <= 4 LP: 4 rows at most
          // on the fast path; longer columns take the second loop)
          double xr[NT], yr[NT];
          double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int r = lane + t * LP;
            xr[t] = r < d ? gp[r] : 0.0;
            yr[t] = r < d ? gq[r] : 0.0;
            a += xr[t] * xr[t]; b += yr[t] * yr[t]; g += xr[t] * yr[t];
          }
          for (int r = lane + NT * LP; r < d; r += LP) {
            double x = gp[r], y = gq[r];
            a += x * x; b += y * y; g += x * y;
          }
          if (LP == 16) {                       // the pair's 16 lanes are one DPP row: no LDS traffic, no waits
            a = row_sum16(a); b = row_sum16(b); g = row_sum16(g);
          } else if (LP == 8) {                 // half a DPP row: quad_perm x2 + row_half_mirror
            a += dpp_get<0xB1, 0xf>(a); b += dpp_get<0xB1, 0xf>(b); g += dpp_get<0xB1, 0xf>(g);
            a += dpp_get<0x4E, 0xf>(a); b += dpp_get<0x4E, 0xf>(b); g += dpp_get<0x4E, 0xf>(g);
            a += dpp_get<0x141, 0xf>(a); b += dpp_get<0x141, 0xf>(b); g += dpp_get<0x141, 0xf>(g);
          } else {
            for (int off = LP >> 1; off > 0; off >>= 1) {
              a += __shfl_xor(a, off, 64);
              b += __shfl_xor(b, off, 64);
              g += __shfl_xor(g, off, 64);
            }
          }
          const double ab = a * b;
          JSTAMP(ph_dot);
          if (g * g > tol * tol * ab && fabs(g) > 1e-300) {
            // rotation from hardware rcp/rsq estimates + Newton steps (c^2 + s^2 = 1 to ~1 ulp is what matters)
            // (the ANGLE only needs ~1e-12: an error there is removed by the next rotation of the pair - one Newton
            // step on the estimates; cs below keeps two)
            const double zeta = (b - a) * 0.5 * rcp_1nr(g);
            const double hyp = 1.0 + zeta * zeta;
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) * rcp_1nr(fabs(zeta) + hyp * rsq_1nr(hyp));
            const double cs = fast_rsq(1.0 + t * t);
            const double sn = cs * t;
            if (cs == 123.456) s_rot = 2;      // (timing build only) keeps cs live before the stamp
            JSTAMP(ph_rot);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
              const int r = lane + u * LP;
              if (r < d) { gp[r] = cs * xr[u] - sn * yr[u]; gq[r] = sn * xr[u] + cs * yr[u]; }
            }
            for (int r = lane + NT * LP; r < d; r += LP) {
              double x = gp[r], y = gq[r];
              gp[r] = cs * x - sn * y;
              gq[r] = sn * x + cs * y;
            }
            if (lane == 0 && g * g > 1e-16 * ab) s_rot = 1;   // a further sweep is needed
            JSTAMP(ph_wr);
          }
        }
      }
      __syncthreads();
      JSTAMP(ph_bar);
    }
    int rot = s_rot;
    __syncthreads();
    if (tid == 0) s_rot = 0;
    __syncthreads();
    if (!rot) { ++sweep; break; }
  }
#ifdef PCABO_ACQ_TIMING
  if (tid == 0) { g_jac_stamps[0] = clock64() - jc0; g_jac_stamps[1] = wall_clock64() - jw0; g_jac_stamps[2] = (unsigned long long)sweep; g_jac_stamps[3] = (unsigned long long)(de - 1);
    g_jac_stamps[4] = ph_dot; g_jac_stamps[5] = ph_rot; g_jac_stamps[6] = ph_wr; g_jac_stamps[7] = ph_bar; }
#endif
  // eigenvalues = column norms; write normalised columns (eigenvectors), column-major d x d
  for (int col = tid / 64; col < d; col += (int)blockDim.x / 64) {
    int l = tid & 63;
    double a = 0.0;
    for (int r = l; r < d; r += 64) { double x = s_g[col * LD + r]; a += x * x; }
    a = wave_sum(a);
    double nrm = sqrt(a);
    double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
    for (int r = l; r < d; r += 64) {
      const double v = s_g[col * LD + r] * inv;
      Gout[(size_t)col * d + r] = v;
      s_g[col * LD + r] = v;                                     // the selection below works on the normalised columns
    }
    if (l == 0) { lam[col] = nrm / cscale; s_lam[col] = nrm; }      // (the ratios below do not see the scale)
  }
  if (tid == 0) *sweeps_out = sweep;
  // ---- selection (was a launch of its own: k_pca_finalize): eigenpairs by decreasing variance, explained-variance
  // ratios, k from the variance threshold (PCA_BO.py:389-394), sign rule svd_flip(u_based_decision=False)
  __syncthreads();
  if (tid < d) {
    const double me = s_lam[tid];
    int r = 0;
    for (int j = 0; j < d; ++j) r += (s_lam[j] > me) || (s_lam[j] == me && j < tid);
    s_order[r] = tid;
  }
  __syncthreads();
  const int rcount = sel.n < d ? sel.n : d;           // sklearn keeps min(n, d) components
  if (tid == 0) {
    double tot = 0.0;
    for (int r = 0; r < rcount; ++r) tot += s_lam[s_order[r]];
    int k;
    double cum = 0.0;
    int cnt = 0;
    for (int r = 0; r < rcount; ++r) {
      double e = s_lam[s_order[r]] / tot;
      sel.evr[r] = e;
      cum += e;
      cnt += (cum <= sel.var_threshold);
    }
    if (sel.n_components > 0) k = sel.n_components < rcount ? sel.n_components : rcount;
    else {
      k = cnt + 1;
      if (k > rcount) k = rcount;
      if (k < 1) k = 1;
    }
    *sel.k_dev = k;
    sel.hm->k = k;
  }
  if (tid < rcount) {                                   // sign of each component: its max-|.| entry positive
    const double* v = s_g + s_order[tid] * LD;
    double best = -1.0, sgn = 1.0;
    for (int j = 0; j < d; ++j) {
      double a = fabs(v[j]);
      if (a > best) { best = a; sgn = v[j] < 0.0 ? -1.0 : 1.0; }
    }
    s_sgn[tid] = sgn;
  }
  __syncthreads();
  for (int idx = tid; idx < rcount * d; idx += (int)blockDim.x) {
    const int r = idx / d, j = idx % d;
    sel.comps[idx] = s_sgn[r] * s_g[s_order[r] * LD + j];
  }
}

// Row C (projection): Z = (X - mu_x) Ck^T - m_w Ck^T  (PCA_BO.py:407; sklearn _base.py _transform).
// Eight points per work-group; Z is n x k row-major, k read from device memory.
__global__ __launch_bounds__(256) void k_project(const double* __restrict__ X, const double* __restrict__ data_mean,
                                                 const double* __restrict__ pca_mean, const double* __restrict__ comps,
                                                 const int* __restrict__ k_dev, int n, int d, double* __restrict__ Z,
                                                 size_t zs) {
  ZRUN(X); ZRUN(data_mean); ZRUN(pca_mean); ZRUN(comps); ZRUN(k_dev); ZRUN(Z);
  __shared__ double s_x[8][PCABO_MAXD];
  __shared__ double s_m[PCABO_MAXD];
  const int k = *k_dev;
  const int i0 = blockIdx.x * 8;
  for (int idx = threadIdx.x; idx < 8 * d; idx += 256) {
    int r = idx / d, j = idx % d;
    s_x[r][j] = (i0 + r < n) ? X[(size_t)(i0 + r) * d + j] - data_mean[j] : 0.0;
  }
  if (threadIdx.x < d) s_m[threadIdx.x] = pca_mean[threadIdx.x];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 8 * k; idx += 256) {
    int r = idx / k, c = idx % k;
    if (i0 + r >= n) continue;
    const double* ck = comps + (size_t)c * d;
    double s = 0.0, sm = 0.0;
    for (int j = 0; j < d; ++j) { s += s_x[r][j] * ck[j]; sm += s_m[j] * ck[j]; }
    Z[(size_t)(i0 + r) * k + c] = s - sm;
  }
}

// Rows D, F, J: per-component min/max of Z -> Normalize bounds (PCA_BO.py:514-518) and optimiser
// box (PCA_BO.py:558-573); mean of the normalised inputs (gpytorch centres before the distance);
// Standardize statistics of y and y_s.
__global__ __launch_bounds__(WP_THREADS) void k_zstats(const double* __restrict__ Z, const double* __restrict__ y,
                                                       int n, int k, const double* __restrict__ user_nb,
                                                       double* __restrict__ bounds4, double* __restrict__ zn_mean,
                                                       double* __restrict__ ystats, double* __restrict__ ys,
                                                       HostMirror* hm, const int* __restrict__ k_dev, size_t zs,
                                                       size_t hzs) {
  ZRUN(Z); ZRUN(y); ZRUN(user_nb); ZRUN(bounds4); ZRUN(zn_mean); ZRUN(ystats); ZRUN(ys); ZRUN(k_dev);
  hm = zrun(hm, hzs, blockIdx.z);
  if (k_dev) k = *k_dev;          // enqueued behind the wPCA: the reduced dimension is not on the host yet
  __shared__ double s_red[WP_THREADS];
  __shared__ double s_min[WP_THREADS];
  __shared__ double s_max[WP_THREADS];
  const int tid = threadIdx.x;
  const int CP = k <= 64 ? 64 : 128;
  const int G = WP_THREADS / CP;
  const int c = tid % CP, g = tid / CP;
  double mn = INFINITY, mx = -INFINITY, sm = 0.0;
  if (c < k) {
    // (eight loads in flight per trip: left to itself the loop waited for every load in turn - 28 L2 round trips at n = 450
    // were most of this kernel's 28 us; the min / max / sum chains keep their order)
    int i = g;
    for (; i + 7 * G < n; i += 8 * G) {
      double z[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = Z[(size_t)(i + u * G) * k + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) { mn = fmin(mn, z[u]); mx = fmax(mx, z[u]); sm += z[u]; }
    }
    for (; i < n; i += G) {
      double z = Z[(size_t)i * k + c];
      mn = fmin(mn, z); mx = fmax(mx, z); sm += z;
    }
  }
  s_min[tid] = mn; s_max[tid] = mx; s_red[tid] = sm;
  __syncthreads();
  if (tid < k) {
    double a = INFINITY, b = -INFINITY, s = 0.0;
    for (int gg = 0; gg < G; ++gg) {
      a = fmin(a, s_min[gg * CP + tid]); b = fmax(b, s_max[gg * CP + tid]); s += s_red[gg * CP + tid];
    }
    double rng = b - a;
    double nlo = a - 0.1 * rng, nhi = b + 0.1 * rng;
    if (user_nb) { nlo = user_nb[tid]; nhi = user_nb[k + tid]; }
    double alo = a - 0.5 * rng, ahi = b + 0.5 * rng;
    if (ahi - alo < 0.1) { double mid = (ahi + alo) / 2; alo = mid - 0.1 / 2; ahi = mid + 0.1 / 2; }
    bounds4[tid] = nlo; bounds4[PCABO_MAXD + tid] = nhi;
    bounds4[2 * PCABO_MAXD + tid] = alo; bounds4[3 * PCABO_MAXD + tid] = ahi;
    hm->norm_lo[tid] = nlo; hm->norm_hi[tid] = nhi; hm->acq_lo[tid] = alo; hm->acq_hi[tid] = ahi;
    zn_mean[tid] = (s / (double)n - nlo) / (nhi - nlo);
  }
  __syncthreads();
  // Standardize(m=1): mean, unbiased std (>= 1e-8 else 1)
  double loc = 0.0;
  for (int i = tid; i < n; i += WP_THREADS) loc += y[i];
  const double ym = block_sum_1024(loc, s_red) / (double)n;
  loc = 0.0;
  for (int i = tid; i < n; i += WP_THREADS) { double t = y[i] - ym; loc += t * t; }
  double var = block_sum_1024(loc, s_red) / (double)(n > 1 ? n - 1 : 1);
  double sd = sqrt(var);
  if (!(sd >= 1e-8)) sd = 1.0;
  for (int i = tid; i < n; i += WP_THREADS) ys[i] = (y[i] - ym) / sd;
  if (tid == 0) { ystats[0] = ym; ystats[1] = sd; hm->y_mean = ym; hm->y_std = sd; }
}

// Row E: Normalize the training inputs and lay them out for the Gram / acquisition kernels:
//   ZnT[c][i] = (Z[i][c]-lo_c)/(hi_c-lo_c)          (KP x ld, point index contiguous)
//   AT[c][i]  = (ZnT[c][i] - mean_c)/lengthscale    (what gpytorch feeds its distance GEMM)
//   nrm[i]    = sum_c AT[c][i]^2
// Points i >= n and components c >= k are zero so whole tiles can be processed.
__global__ __launch_bounds__(256) void k_znorm(const double* __restrict__ Z, int n, int k, int NP, int KP, int ld,
                                               const double* __restrict__ bounds4, const double* __restrict__ zn_mean,
                                               double inv_ls, double* __restrict__ ZnT, double* __restrict__ AT,
                                               double* __restrict__ nrm, const int* __restrict__ k_dev, size_t zs) {
  ZRUN(Z); ZRUN(bounds4); ZRUN(zn_mean); ZRUN(ZnT); ZRUN(AT); ZRUN(nrm); ZRUN(k_dev);
  if (k_dev) { k = *k_dev; KP = (k + 3) & ~3; }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NP) return;
  double s = 0.0;
  const size_t row = (size_t)(i < n ? i : n - 1) * k;
  for (int c0 = 0; c0 < KP; c0 += 4) {                 // (KP is a multiple of 4; four loads in flight, from clamped addresses)
    double zr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) zr[u] = Z[row + (c0 + u < k ? c0 + u : k - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u;
      double zn = 0.0, a = 0.0;
      if (i < n && c < k) {
        double lo = bounds4[c], hi = bounds4[PCABO_MAXD + c];
        zn = (zr[u] - lo) / (hi - lo);
        a = (zn - zn_mean[c]) * inv_ls;
      }
      ZnT[(size_t)c * ld + i] = zn;
      AT[(size_t)c * ld + i] = a;
      s += a * a;
    }
  }
  nrm[i] = s;
}

// Row O: x = z Ck + m_w + mu_x (PCA_BO.py:427).
__global__ __launch_bounds__(128) void k_inverse_map(const double* __restrict__ z, const double* __restrict__ comps,
                                                     const double* __restrict__ data_mean,
                                                     const double* __restrict__ pca_mean, int k, int d,
                                                     double* __restrict__ x, const int* __restrict__ k_dev, size_t zs,
                                                     double* __restrict__ host_x, HostMirror* hm, unsigned long long seq) {
  ZRUN(z); ZRUN(comps); ZRUN(data_mean); ZRUN(pca_mean); ZRUN(x); ZRUN(k_dev);
  if (k_dev) k = *k_dev;
  const int j = threadIdx.x;
  if (j < d) {
    double s = 0.0;
    int c = 0;
    for (; c + 8 <= k; c += 8) {                       // eight loads in flight; the sum keeps its order
      double cv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cv[u] = comps[(size_t)(c + u) * d + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += z[c + u] * cv[u];
    }
    for (; c < k; ++c) s += z[c] * comps[(size_t)c * d + j];
    const double xj = (s + pca_mean[j]) + data_mean[j];
    x[j] = xj;
    if (host_x) host_x[j] = xj;                        // single context: straight to pinned host memory, then the flag below
  }
  if (hm) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(const_cast<unsigned long long*>(&hm->qflag[0]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- launchers --------------------------------------------------------------------------------
void launch_rank(hipStream_t s, const double* f, int n, int maximize, long long* ranks) {
  hipLaunchKernelGGL(k_rank, dim3((n + 255) / 256), dim3(256), 0, s, f, n, maximize, ranks);
}
void launch_wpca_prep(hipStream_t s, const double* X, const long long* ranks, const double* noise, int n, int d,
                      int DP, double* weights, double* data_mean, double* pca_mean, double* Wc, ZB zb) {
  hipLaunchKernelGGL(k_wpca_prep, dim3(1, 1, zb.B), dim3(WP_THREADS), 0, s, X, ranks, noise, n, d, DP, weights, data_mean,
                     pca_mean, Wc, zb.zs);
}
void launch_cov(hipStream_t s, const double* Wc, int n, int DP, double* C, ZB zb) {
  int n4 = (n + 3) & ~3;
  hipLaunchKernelGGL(k_cov, dim3(DP / 16, DP / 16, zb.B), dim3(256), 0, s, Wc, n4, DP, 1.0 / (double)(n - 1), C, zb.zs);
}
void launch_jacobi(hipStream_t s, const double* C, int d, int DP, const double* V0, double* G, double* lam, int* sweeps,
                   int n, double var_threshold, int n_components, double* comps, double* evr, int* k_dev, HostMirror* hm,
                   ZB zb) {
  size_t lds = ((size_t)d * (d | 1) + 2) * sizeof(double);
  if (V0 && d <= 64) lds += (size_t)2 * d * (d | 1) * sizeof(double);      // LDS copies of C and V0 for the warm start
  static bool attr_set = false;
  if (!attr_set) {
    // 160 KB per work-group minus the kernel's static arrays (2.5 KB); d = 128 needs 132 KB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)k_jacobi<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)k_jacobi<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    attr_set = true;
  }
  const int npairs = ((d + 1) & ~1) / 2;
  int threads = npairs * 16 <= JAC_THREADS ? npairs * 16 : npairs * 8;      // 16 lanes per column pair while they fit
  threads = (threads + 63) & ~63;
  if (threads < d) threads = (d + 63) & ~63;          // the selection step at the end uses one thread per component
  PcaSelect sel{n, var_threshold, n_components, comps, evr, k_dev, hm};
  if (d <= 64) hipLaunchKernelGGL(k_jacobi<4>, dim3(1, 1, zb.B), dim3(threads), lds, s, C, d, DP, V0, G, lam, sweeps, sel, zb.zs, zb.hzs);
  else hipLaunchKernelGGL(k_jacobi<8>, dim3(1, 1, zb.B), dim3(threads), lds, s, C, d, DP, V0, G, lam, sweeps, sel, zb.zs, zb.hzs);
}
void launch_project(hipStream_t s, const double* X, const double* data_mean, const double* pca_mean,
                    const double* comps, const int* k_dev, int n, int d, double* Z, ZB zb) {
  hipLaunchKernelGGL(k_project, dim3((n + 7) / 8, 1, zb.B), dim3(256), 0, s, X, data_mean, pca_mean, comps, k_dev, n, d, Z,
                     zb.zs);
}
void launch_zstats(hipStream_t s, const double* Z, const double* y, int n, int k, const double* user_norm_bounds,
                   double* bounds4, double* zn_mean, double* ystats, double* ys, HostMirror* hm, const int* k_dev, ZB zb) {
  hipLaunchKernelGGL(k_zstats, dim3(1, 1, zb.B), dim3(WP_THREADS), 0, s, Z, y, n, k, user_norm_bounds, bounds4, zn_mean,
                     ystats, ys, hm, k_dev, zb.zs, zb.hzs);
}
void launch_znorm(hipStream_t s, const double* Z, int n, int k, int NP, int KP, int ld, const double* bounds4,
                  const double* zn_mean, double inv_ls, double* ZnT, double* AT, double* nrm, const int* k_dev, ZB zb) {
  hipLaunchKernelGGL(k_znorm, dim3((NP + 255) / 256, 1, zb.B), dim3(256), 0, s, Z, n, k, NP, KP, ld, bounds4, zn_mean, inv_ls,
                     ZnT, AT, nrm, k_dev, zb.zs);
}
void launch_inverse_map(hipStream_t s, const double* z, const double* comps, const double* data_mean,
                        const double* pca_mean, int k, int d, double* x, const int* k_dev, ZB zb, double* host_x,
                        HostMirror* hm, unsigned long long seq) {
  hipLaunchKernelGGL(k_inverse_map, dim3(1, 1, zb.B), dim3(128), 0, s, z, comps, data_mean, pca_mean, k, d, x, k_dev, zb.zs,
                     host_x, hm, seq);
}
