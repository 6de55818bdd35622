// Device-resident L-BFGS-B for the restart groups of a batch (first option; the reference reaches
// the algorithm through botorch.optimize_acqf -> gen_candidates_scipy -> scipy's L-BFGS-B,
// /root/reference/Algorithms/BayesianOptimization/PCA_BO.py:607-614).
//
// One work-group of 1024 threads per restart group (<= 5 joint query points, <= 200 variables) runs the whole
// optimisation without the host: evaluate (all 16 waves) -> L-BFGS-B step (wave 0, wave 1 helping) -> next point, until the group has
// converged or reached its limits.  No work-group waits for another one, nothing polls: a plain kernel whose every loop is
// bounded (maxiter iterations of <= 20 line-search evaluations, a hard cap on evaluations on top).
//
//  * The optimiser state (S, Y of the last 10 pairs, the compact-representation matrices, the iterate, gradient, bounds,
//    index sets) lives in LDS; the step is csrc/lbfgsb.cpp restated for ONE WAVE: loops over the variables run a variable
//    per lane; the sums over the variables (d'd, g'd, r'r, the 2m accumulations W'd, f1 of the Cauchy search) are formed in
//    the 64-LANE TREE ORDER - lane l adds the terms l, l + 64, ..., the lane sums meet in a balanced tree of adjacent pairs
//    (DPP row operations; several sums at once through an LDS tile) - which csrc/lbfgsb.cpp implements behind
//    Lbfgsb::set_sum_order(1): the host class in that order is this kernel's twin.  The short ordered sums over the <= 2m
//    history columns and the small dense pieces (10 x 10 and 20 x 20 factorisations and triangular solves) keep the
//    published order: a column or a right-hand side per lane, pivots broadcast with v_readlane.  Every number goes
//    through the same operations in the same order as in the twin (this file is compiled with -ffp-contract=off; IEEE
//    divide and square root), so that fed the same f / g values the device takes the twin's iterates bit for bit
//    (tests/test_gpu_device_lbfgsb.py compares the two through the evaluation-only mode of the same kernel).  Wave 1 helps:
//    it runs the routines whose results wave 0 does not need at once (second half of matupd + formt beside the head of the
//    Cauchy search, cmprlb + the head of subsm beside formk) - same routines, same data, handed over through two LDS words.
//  * The evaluation (rows I of SURVEY.md 8a for the group's points: kernel vectors, v = R ks, |v|^2, mu, log-EI / PI chain,
//    w = R' v, gradient contraction) is thread-per-output with coalesced reads in the two triangular passes: pass 1 reads
//    the TRANSPOSED root inverse RT (built once per conditioning by k_rt_build), pass 2 reads R itself; a thread loads 16
//    bytes (two neighbouring outputs) and the half-waves split the summation index by parity - one exchange per query
//    joins them (a CU issues a wave's load instruction every ~11 ns whatever its width: lb_eval); the work of a pass is
//    dealt out to the 16 waves by a plan built once per launch (lb_build_plan) whose partial sums are added in a fixed order.  Same formulas as k_acq_group / k_acq_fast, another (fixed) summation order: a third arithmetic mode, selected
//    per batch (PCABO_OPT_DEVICE_LBFGSB), never mixed within a run.
#include "pcabo_internal.h"
#include "lbfgsb.h"
#include "lb_plan.h"
#include <cfloat>
#include <cmath>
#include <mutex>
#include <vector>
#include "../../include/pcabo.h"      // PCABO_ERR_NAN

#define LB_M 10
#define LB_GQ 5
#define LB_QS 6             // stride of a point's per-query values in LDS (16-byte aligned pairs)
#define LB_MAXK 40
#define LB_MAXEVAL 20000    // hard cap on the evaluations of one group (the host's limits stop it long before)

#define LSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// In-kernel clocks of the timing build (make timing; tools/gpu_device_lbfgsb_phases.py): ticks (100 MHz) and calls per phase,
// accumulated by work-group 0 (its thread 0 / lane 0 of wave 0)
#ifdef PCABO_ACQ_TIMING
__device__ unsigned long long g_lb_ticks[64], g_lb_calls[64];
extern "C" int pcabo_debug_lb_ticks(unsigned long long* ticks64, unsigned long long* calls64, int reset) {
  if (hipMemcpyFromSymbol(ticks64, HIP_SYMBOL(g_lb_ticks), sizeof(g_lb_ticks)) != hipSuccess) return -3;
  if (hipMemcpyFromSymbol(calls64, HIP_SYMBOL(g_lb_calls), sizeof(g_lb_calls)) != hipSuccess) return -3;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lb_ticks), z, sizeof(z)) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_lb_calls), z, sizeof(z)) != hipSuccess) return -3;
  }
  return 0;
}
#define LBT_BEGIN() unsigned long long lbt_t0_ = wall_clock64()
#define LBT_NEXT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); g_lb_ticks[i] += t_ - lbt_t0_; g_lb_calls[i] += 1; lbt_t0_ = t_; } } while (0)
#else
#define LBT_BEGIN()
#define LBT_NEXT(i)
#endif
// LDS pointers carry their address space in the type: ds_read / ds_write whether or not a helper is inlined
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) double ldsd;
typedef __attribute__((address_space(3))) int ldsi;
typedef const __attribute__((address_space(1))) double gcd;      // global memory, read only: global_load even behind a call
typedef double lb_v2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) lb_v2 ldsv2;     // two neighbouring doubles of an LDS array: ds_read_b128
#else
typedef double ldsd;
typedef int ldsi;
typedef const double gcd;
typedef double lb_v2 __attribute__((ext_vector_type(2)));
typedef const lb_v2 ldsv2;
#endif

// ---- persistent scalars of a group (LDS) ----------------------------------------------------------------------------
enum { S_THETA, S_FOLD, S_DNORM, S_GD, S_STPMX, S_SBGNRM, S_STP, S_GDOLD, S_DTD, S_F, S_FC, S_DR,
       LS_GINIT, LS_GTEST, LS_GX, LS_GY, LS_FINIT, LS_FX, LS_FY, LS_STX, LS_STY, LS_STMIN, LS_STMAX, LS_WIDTH, LS_WIDTH1,
       S_COUNT };
enum { I_COL, I_HEAD, I_ITAIL, I_ITER, I_IUPDAT, I_UPDATD, I_WRK, I_NFREE, I_ILEAVE, I_NENTER, I_INFO, I_IFUN, I_IBACK,
       I_NFGV, I_IWORD, I_TASK, I_PHASE, LS_TASK, LS_BRACKT, LS_STAGE, I_NITER, I_NFEV, I_HAVE_CACHE, I_ACTIVE, I_STATUS,
       I_TIES, I_EVALS, I_HCMD, I_HDONE, I_HPEND, I_INFO2, I_COUNT };

// LDS layout of a group: every array at a compile-time offset from the dynamic block (the optimiser's arrays are sized for the
// largest group, LB_GQ * LB_MAXK variables), the evaluation's three NP-sized arrays last.  The struct is three words and
// travels by value.
#define LB_NVCAP (LB_GQ * LB_MAXK)
#define LB_TILE_LD 66                         // leading dimension of a transposition tile: 16-byte aligned rows, 4 banks apart
#define LB_TILE (4 * LB_M * LB_TILE_LD)       // 40 sums x 64 lane terms
// work plan of the two triangular passes (lb_build_plan): per pass and wave [count, 2 x (unit, lo, hi, dest)], then per unit
// [first extra slot, extra slots]
#define LB_LDW (LB_NVCAP | 1)
#define LB_NVP ((LB_NVCAP + 3) & ~1)
#define LB_EVEN(x) (((x) + 1) & ~1)
enum {
  OFF_WS = 0, OFF_WY = OFF_WS + LB_EVEN(LB_M * LB_LDW), OFF_X = OFF_WY + LB_EVEN(LB_M * LB_LDW),
  OFF_G = OFF_X + LB_NVP, OFF_LO = OFF_G + LB_NVP, OFF_HI = OFF_LO + LB_NVP, OFF_Z = OFF_HI + LB_NVP, OFF_R = OFF_Z + LB_NVP,
  OFF_D = OFF_R + LB_NVP, OFF_T = OFF_D + LB_NVP, OFF_XP = OFF_T + LB_NVP, OFF_FULL = OFF_XP + LB_NVP, OFF_COEF = OFF_FULL + LB_NVP,
  OFF_XC = OFF_COEF + LB_NVP, OFF_GC = OFF_XC + LB_NVP, OFF_PROD = OFF_GC + LB_NVP,
  OFF_SY = OFF_PROD + LB_NVP, OFF_SS = OFF_SY + LB_M * LB_M, OFF_WT = OFF_SS + LB_M * LB_M, OFF_WN = OFF_WT + LB_M * LB_M,
  OFF_WN1 = OFF_WN + 4 * LB_M * LB_M, OFF_WA = OFF_WN1 + 4 * LB_M * LB_M, OFF_ACC = OFF_WA + 8 * LB_M, OFF_ACC2 = OFF_ACC + 64, OFF_SC = OFF_ACC2 + 64,
  OFF_VC = OFF_SC + LB_EVEN(S_COUNT + 2), OFF_INTS = OFF_VC + 8,
  OFF_XN = OFF_INTS + LB_EVEN((4 * LB_NVP + I_COUNT + 2) / 2 + 2), OFF_RED = OFF_XN + LB_EVEN(LB_GQ * (LB_MAXK + 2)),
  OFF_VALS = OFF_RED + 80, OFF_CQ = OFF_VALS + 8, OFF_NLO = OFF_CQ + 16, OFF_NHI = OFF_NLO + LB_MAXK, OFF_EV = OFF_NHI + LB_MAXK,
  OFF_PLAN = OFF_EV + 16, OFF_KS = OFF_PLAN + LB_PLAN_INTS / 2
};
struct LbLds {
  ldsd* base;
  int n, NP;                            // n: variables of the group (nq * k)
  __device__ ldsd* ws() const { return base + OFF_WS; }
  __device__ ldsd* wy() const { return base + OFF_WY; }
  __device__ ldsd* x() const { return base + OFF_X; }
  __device__ ldsd* g() const { return base + OFF_G; }
  __device__ ldsd* lo() const { return base + OFF_LO; }
  __device__ ldsd* hi() const { return base + OFF_HI; }
  __device__ ldsd* z() const { return base + OFF_Z; }
  __device__ ldsd* r() const { return base + OFF_R; }
  __device__ ldsd* d() const { return base + OFF_D; }
  __device__ ldsd* t() const { return base + OFF_T; }
  __device__ ldsd* xp() const { return base + OFF_XP; }
  __device__ ldsd* full() const { return base + OFF_FULL; }
  __device__ ldsd* coef() const { return base + OFF_COEF; }
  __device__ ldsd* xc() const { return base + OFF_XC; }
  __device__ ldsd* gc() const { return base + OFF_GC; }
  __device__ ldsd* prod() const { return base + OFF_PROD; }
  __device__ ldsd* sy() const { return base + OFF_SY; }
  __device__ ldsd* ss() const { return base + OFF_SS; }
  __device__ ldsd* wt() const { return base + OFF_WT; }
  __device__ ldsd* wn() const { return base + OFF_WN; }
  __device__ ldsd* wn1() const { return base + OFF_WN1; }
  __device__ ldsd* wa() const { return base + OFF_WA; }
  __device__ ldsd* acc() const { return base + OFF_ACC; }
  __device__ ldsd* acc2() const { return base + OFF_ACC2; }      // the helper wave's accumulations
  __device__ ldsd* sc() const { return base + OFF_SC; }
  __device__ ldsd* vc() const { return base + OFF_VC; }
  __device__ ldsi* index() const { return (ldsi*)(base + OFF_INTS); }
  __device__ ldsi* iwhere() const { return (ldsi*)(base + OFF_INTS) + LB_NVP; }
  __device__ ldsi* indx2() const { return (ldsi*)(base + OFF_INTS) + 2 * LB_NVP; }
  __device__ ldsi* rows() const { return (ldsi*)(base + OFF_INTS) + 3 * LB_NVP; }
  __device__ ldsi* isc() const { return (ldsi*)(base + OFF_INTS) + 4 * LB_NVP; }
  __device__ ldsd* xn() const { return base + OFF_XN; }
  __device__ ldsd* red() const { return base + OFF_RED; }
  __device__ ldsd* vals() const { return base + OFF_VALS; }
  __device__ ldsd* cq() const { return base + OFF_CQ; }
  __device__ ldsd* nlo() const { return base + OFF_NLO; }
  __device__ ldsd* nhi() const { return base + OFF_NHI; }
  __device__ ldsd* ev() const { return base + OFF_EV; }
  __host__ __device__ ldsi* plan() const { return (ldsi*)(base + OFF_PLAN); }
  __device__ ldsd* ks() const { return base + OFF_KS; }
  __device__ ldsd* vb() const { return base + OFF_KS + LB_QS * NP; }
  __device__ ldsd* slots() const { return base + OFF_KS + 2 * LB_QS * NP; }       // partial sums of a pass: [slot][64][LB_GQ]
  // the step's transposition tiles (lb_accum): the evaluation's arrays are dead while the optimiser steps; one tile per stepping wave
  __device__ ldsd* tile(int w) const { return base + OFF_KS + w * LB_TILE; }
};
// the triangular passes' partial slots: at most (waves + two-slab pairs - slabs) segments do not start their slab (lb_build_plan)
static inline size_t lb_lds_doubles(int NP) { return (size_t)OFF_KS + (size_t)2 * LB_QS * NP + (size_t)lb_max_slots(NP) * 64 * LB_GQ; }

#define SC(i) L.sc()[i]
#define ISC(i) L.isc()[i]
// The same scalars read as wave-uniform values (v_readfirstlane): what comes out of LDS is a per-lane value to the compiler -
// branches and loop bounds on it would be lane-masked, a lane index for v_readlane would need a search loop
#define ISR(i) __builtin_amdgcn_readfirstlane(L.isc()[i])
#define SR(i) uni(L.sc()[i])
#define WS_(i, p) L.ws()[(p) * LB_LDW + (i)]
#define WY_(i, p) L.wy()[(p) * LB_LDW + (i)]
#define SY_(i, j) L.sy()[(j) * LB_M + (i)]
#define SS_(i, j) L.ss()[(j) * LB_M + (i)]
#define WT_(i, j) L.wt()[(j) * LB_M + (i)]
#define WN_(i, j) L.wn()[(j) * 2 * LB_M + (i)]
#define WN1_(i, j) L.wn1()[(j) * 2 * LB_M + (i)]

__device__ inline int nxt(int p) { return p + 1 == LB_M ? 0 : p + 1; }
// e = a (a + 1) / 2 + r with r <= a, for e < 64: the pair (r <= a) of a packed triangle without a search loop.  8 e + 1 is the
// square of 2 a + 1 at the start of a row and at least 4 / (2 a + 3) below the next odd square at its end: the margins cover
// any rounding of the single-precision square root.
__device__ inline void tri_decode(int e, int& a, int& r) {
  a = (int)((sqrtf(8.0f * (float)e + 1.0f) - 0.999f) * 0.5f);
  r = e - a * (a + 1) / 2;
}
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ inline gcd* uni(gcd* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (gcd*)(((unsigned long long)hi << 32) | lo);
}
__device__ inline double bcast(double v, int lane) {      // value of `lane` (uniform index) in every lane
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ inline void st0(ldsd* p, double v, int lane) { if (lane == 0) *p = v; }
__device__ inline void sti0(ldsi* p, int v, int lane) { if (lane == 0) *p = v; }
__device__ inline unsigned long long lanes_below(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

// ---- sums over the variables in the 64-lane tree order (lbfgsb.cpp: tree_sum / tree_dot / tree_accum) -------------------------
// lane l adds its terms l, l + 64, l + 128 ... in that order (a lane without a term holds 0.0); wave_sum (pcabo_internal.h) joins
// the 64 lane sums in the balanced tree of adjacent pairs: quad_perm 1 + 1, 2 + 2, row_half_mirror 4 + 4, row_mirror 8 + 8,
// row_bcast15 16 + 16, row_bcast31 32 + 32 (a + b == b + a bit for bit, so which lane holds which operand does not matter).
__device__ inline double wave_tsum(const ldsd* p, int n, int lane) {
  double s = lane < n ? p[lane] : 0.0;
  for (int i = lane + 64; i < n; i += 64) s += p[i];
  return wave_sum(s);
}
__device__ inline double wave_ddot(const ldsd* a, const ldsd* b, int n, int lane) {
  double s = lane < n ? a[lane] * b[lane] : 0.0;
  for (int i = lane + 64; i < n; i += 64) s += a[i] * b[i];
  return wave_sum(s);
}
// ddot of two short LDS vectors (n <= 2 LB_M) in the published order: lane j forms product j, every lane adds them in order
__device__ inline double small_ddot(const ldsd* a, const ldsd* b, int n, int lane) {
  const int j = lane < n ? lane : 0;
  const double pr = a[j] * b[j];
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 2 * LB_M; ++u) if (u < n) s += bcast(pr, u);
  return s;
}

// balanced adjacent-pair tree over 16 neighbouring doubles of an LDS row (8 ds_read_b128)
__device__ inline double tree16(const ldsd* row) {
  lb_v2 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = *(ldsv2*)(row + 2 * u);
  double b[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) b[u] = v[u].x + v[u].y;
  const double c0 = b[0] + b[1], c1 = b[2] + b[3], c2 = b[4] + b[5], c3 = b[6] + b[7];
  return (c0 + c1) + (c2 + c3);
}
// "reduce over variables" (lbfgsb.cpp: tree_accum): out[c] = tree sum over ALL variables t of coefA[t] * W(t, c), c over the
// 2 LB_M physical columns of WY | WS - a variable outside the published algorithm's index list carries the coefficient 0.0 -
// and, with TWO, out[32 + c] the same with coefB.  Lane l forms the terms of variables l, l + 64, ... in registers; the 64 lane
// sums of an output meet through an LDS tile (a row of LB_TILE_LD doubles per output, conflict-free both ways), where lane o
// reduces output o in the tree's order, 16 values at a time.
template <bool TWO>
__device__ inline void lb_accum(const LbLds L, const ldsd* coefA, const ldsd* coefB, int lane, ldsd* out, ldsd* tile) {
  const int n = L.n;
  constexpr int NO = TWO ? 4 * LB_M : 2 * LB_M;
  double p[NO];
  {
    const bool in = lane < n;
    const int v = in ? lane : 0;
    const double ca = coefA[v], cb = TWO ? coefB[v] : 0.0;
#pragma unroll
    for (int c = 0; c < LB_M; ++c) {
      const double wy = L.wy()[c * LB_LDW + v], ws = L.ws()[c * LB_LDW + v];
      p[c] = in ? ca * wy : 0.0; p[LB_M + c] = in ? ca * ws : 0.0;
      if (TWO) { p[2 * LB_M + c] = in ? cb * wy : 0.0; p[3 * LB_M + c] = in ? cb * ws : 0.0; }
    }
  }
  for (int v = lane + 64; v < n; v += 64) {
    const double ca = coefA[v], cb = TWO ? coefB[v] : 0.0;
#pragma unroll
    for (int c = 0; c < LB_M; ++c) {
      const double wy = L.wy()[c * LB_LDW + v], ws = L.ws()[c * LB_LDW + v];
      p[c] += ca * wy; p[LB_M + c] += ca * ws;
      if (TWO) { p[2 * LB_M + c] += cb * wy; p[3 * LB_M + c] += cb * ws; }
    }
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) tile[o * LB_TILE_LD + lane] = p[o];
  LSYNC();
  if (lane < NO) {
    const ldsd* row = tile + lane * LB_TILE_LD;
    const double q0 = tree16(row), q1 = tree16(row + 16), q2 = tree16(row + 32), q3 = tree16(row + 48);
    out[(TWO && lane >= 2 * LB_M) ? 32 + lane - 2 * LB_M : lane] = (q0 + q1) + (q2 + q3);
  }
  LSYNC();
}

// LINPACK dpofa on an LDS matrix (upper factor, column-major, leading dimension lda), a column per lane, RIGHT-LOOKING: at step k
// every lane j > k forms  t = (a[k][j] - dot_k) / a[k][k]  where dot_k = sum_{i<k} a[i][k] a[i][j] has been accumulated as the
// rows i became final (ascending i, products rounded, then added from 0.0: the host's ddot, term for term) - so a step's chain
// is pivot -> square root -> divide, and the products for the later pivots issue beside the next step's square root.  Straight-line:
// steps beyond nn run on a unit pivot and zeros, a failed pivot goes on in NaNs (the caller resets the memory and reads nothing
// of the matrix).  Returns 0 or 1 + the index of the failing pivot.
__device__ inline int lb_dpofa(ldsd* A, int lda, int nn, int lane) {
  const int j = lane < LB_M ? lane : LB_M - 1;
  double a[LB_M], dot[LB_M];
#pragma unroll
  for (int i = 0; i < LB_M; ++i) { a[i] = (i <= j && j < nn) ? A[j * lda + i] : 0.0; dot[i] = 0.0; }
  double s = 0.0;
  int info = 0;
#pragma unroll
  for (int k = 0; k < LB_M; ++k) {
    const double dk0 = bcast(a[k] - s, k);
    const bool live = k < nn;
    if (live && info == 0 && dk0 <= 0.0) info = k + 1;
    const double akk = sqrt(live ? dk0 : 1.0);
    const double t = (a[k] - dot[k]) / akk;
    if (lane == k) a[k] = akk;
    else if (lane > k) { a[k] = t; s += t * t; }
#pragma unroll
    for (int m = k + 1; m < LB_M; ++m) dot[m] += bcast(a[k], m) * a[k];
  }
  if (info == 0 && lane < nn) {
#pragma unroll
    for (int i = 0; i < LB_M; ++i) if (i <= lane) A[lane * lda + i] = a[i];
  }
  LSYNC();
  return info;
}

__device__ inline int lb_trsl_zero_diag(double tdiag, int nn, int lane) {
  const unsigned long long m = __ballot(lane < nn && tdiag == 0.0);
  return m ? __ffsll((long long)m) : 0;
}

// LINPACK dtrsl with an upper-triangular T of order nn <= NN held in registers - lane l: tc[j] = T(j, l) (its column),
// tr[j] = T(l, j) (its row), rd = 1 / T(l, l) (the reciprocal of its pivot: lbfgsb.cpp dtrsl_recip - the pivots' reciprocals are
// formed side by side and multiplied in, the solve's chain holds no division) - and ONE right-hand side spread over the lanes
// (lane l holds b[l]).  job 11: T' x = b, job 1: T x = b; the twin's operation order.  Returns the solution in the same lanes.
template <int NN>
__device__ inline double lb_dtrsl_regs(const double (&tc)[NN], const double (&tr)[NN], double rd, int nn, double b, int job, int lane) {
  if (job == 11) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NN; ++j) {
      if (j < nn) {
        const double rjj = bcast(rd, j);
        const double cand = (j == 0) ? b * rjj : (b - s) * rjj;       // (b[0] r[0]: no subtraction on the host)
        const double xj = bcast(cand, j);
        if (lane == j) b = xj;
        if (lane > j && lane < nn) s += tc[j] * xj;
      }
    }
    return b;
  }
  // job 1: b[nn-1] *= r[nn-1]; for j = nn-2 .. 0: b[0..j] += -b[j+1] T(0..j, j+1); b[j] *= r[j]
  {
    const double last = bcast(b, nn - 1) * bcast(rd, nn - 1);
    if (lane == nn - 1) b = last;
  }
#pragma unroll
  for (int j = NN - 2; j >= 0; --j) {
    if (j <= nn - 2) {
      const double temp = -bcast(b, j + 1);
      if (lane <= j) b += temp * tr[j + 1];
      const double q = bcast(b, j) * bcast(rd, j);
      if (lane == j) b = q;
    }
  }
  return b;
}

// p = M v with the middle matrix of the compact representation (lbfgsb.cpp: bmv).  v, p: LDS vectors of 2 col entries.
// The 10 x 10 matrices come into registers in one LDS round trip (lane i: row i and column i of SY, row i and column i of the
// factor T); everything after that is register arithmetic and broadcasts.  Returns info (0 ok).
__device__ __noinline__ int lb_bmv(const LbLds L, const ldsd* v, ldsd* p, int lane) {
  const int col = ISR(I_COL);
  if (col == 0) return 0;
  const int i = lane < LB_M ? lane : LB_M - 1, ic = i < col ? i : 0;
  double syr[LB_M], syc[LB_M], tc[LB_M], tr[LB_M];
#pragma unroll
  for (int k = 0; k < LB_M; ++k) { syr[k] = SY_(i, k); syc[k] = SY_(k, i); tc[k] = WT_(k, i); tr[k] = WT_(i, k); }
  const double sydiag = SY_(i, i), tdiag = WT_(i, i);
  const double vin1 = v[ic], vin2 = v[col + ic];
  // the lane's reciprocals (lbfgsb.cpp bmv, tree / wave order): rs = 1 / SY(i, i), rq = 1 / sqrt(SY(i, i)), rt = 1 / T(i, i)
  const double rs = 1.0 / sydiag, rq = 1.0 / sqrt(sydiag), rt = 1.0 / tdiag;
  // p2[i] = v[col + i] + sum_{k < i} SY(i, k) v[k] rs[k]
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < LB_M - 1; ++k) {
    if (k + 1 < col) {
      const double vk = bcast(vin1, k), rk = bcast(rs, k);
      if (i > k && i < col) sum += syr[k] * vk * rk;
    }
  }
  double p2 = (i == 0) ? vin2 : vin2 + sum;
  const int info = lb_trsl_zero_diag(tdiag, col, lane);
  if (info != 0) return info;
  p2 = lb_dtrsl_regs<LB_M>(tc, tr, rt, col, p2, 11, lane);
  double p1 = vin1 * rq;
  p2 = lb_dtrsl_regs<LB_M>(tc, tr, rt, col, p2, 1, lane);
  p1 = -p1 * rq;
  double s2 = 0.0;
#pragma unroll
  for (int k = 1; k < LB_M; ++k) {
    if (k < col) {
      const double pk = bcast(p2, k);
      if (i < k) s2 += syc[k] * pk * rs;
    }
  }
  p1 += s2;
  LSYNC();                       // every lane has read v (p may alias it)
  if (lane < col) { p[lane] = p1; p[col + lane] = p2; }
  LSYNC();
  return 0;
}

// The two triangular solves of the subspace minimisation (lbfgsb.cpp: subsm - dtrsl job 11, sign change of the first col
// entries, dtrsl job 1) with the 2 col x 2 col factor K in registers.  b: lane l holds wv[l].  info through *info.
__device__ __noinline__ double lb_subsm_solves(const LbLds L, double b, int* info, int lane) {
  const int col = ISR(I_COL), col2 = 2 * col;
  const int l = lane < 2 * LB_M ? lane : 2 * LB_M - 1;
  double tc[2 * LB_M], tr[2 * LB_M];
#pragma unroll
  for (int j = 0; j < 2 * LB_M; ++j) { tc[j] = WN_(j, l); tr[j] = WN_(l, j); }
  const double td = WN_(l, l);
  *info = lb_trsl_zero_diag(td, col2, lane);
  if (*info != 0) return b;
  const double rd = 1.0 / td;
  b = lb_dtrsl_regs<2 * LB_M>(tc, tr, rd, col2, b, 11, lane);
  if (lane < col) b = -b;
  b = lb_dtrsl_regs<2 * LB_M>(tc, tr, rd, col2, b, 1, lane);
  return b;
}

// DPP reductions whose excluded lanes read `old` (the operation's neutral element: the lane's own value)
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_get_or(double v, double old) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_geti_or(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
#define LB_DPP_REDUCE(T, GET, v, OP)                                                                       \
  do {                                                                                                      \
    T o_;                                                                                                   \
    o_ = GET<0xB1, 0xf>(v, v); v = OP(o_, v); o_ = GET<0x4E, 0xf>(v, v); v = OP(o_, v);                     \
    o_ = GET<0x141, 0xf>(v, v); v = OP(o_, v); o_ = GET<0x140, 0xf>(v, v); v = OP(o_, v);                   \
    o_ = GET<0x142, 0xa>(v, v); v = OP(o_, v); o_ = GET<0x143, 0xc>(v, v); v = OP(o_, v);                   \
  } while (0)
#define LB_MAXOP(o, v) ((o) > (v) ? (o) : (v))
#define LB_MINOP(o, v) ((o) < (v) ? (o) : (v))
__device__ inline double wave_max(double v) {          // uniform (the comparisons of the shuffle form it replaces; no LDS traffic)
  LB_DPP_REDUCE(double, dpp_get_or, v, LB_MAXOP);
  return bcast(v, 63);
}
__device__ inline double wave_min(double v) {
  LB_DPP_REDUCE(double, dpp_get_or, v, LB_MINOP);
  return bcast(v, 63);
}
__device__ inline int wave_min_i(int v) {
  LB_DPP_REDUCE(int, dpp_geti_or, v, LB_MINOP);
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ inline void lb_projgr(const LbLds L, int lane) {
  double m = 0.0;
  for (int i = lane; i < L.n; i += 64) {
    const double gi = L.g()[i], hi = L.x()[i] - L.hi()[i], lo = L.x()[i] - L.lo()[i];
    const double a = hi > gi ? hi : gi, b = lo < gi ? lo : gi;
    const double pv = fabs(gi < 0.0 ? a : b);
    m = pv > m ? pv : m;
  }
  m = wave_max(m);
  st0(&SC(S_SBGNRM), m, lane);
  LSYNC();
}

__device__ inline void lb_reset_memory(const LbLds L, int lane) {
  if (lane == 0) { ISC(I_INFO) = 0; ISC(I_COL) = 0; ISC(I_HEAD) = 0; SC(S_THETA) = 1.0; ISC(I_IUPDAT) = 0; ISC(I_UPDATD) = 0; }
  LSYNC();
}

// ---- the helper wave -------------------------------------------------------------------------------------------------
// Two routines of an iteration do not depend on what wave 0 does next: formt (the factor T of the middle matrix: the Cauchy
// search needs it only at its first product with that matrix) and cmprlb (the reduced gradient: it needs the Cauchy point, not
// the factor that formk builds meanwhile).  Wave 1 runs them while wave 0 goes on: a command word in LDS (sequence number | op),
// an answer word, both sides sleeping between polls; every wait is bounded.  The routines and their data are what they were -
// the results are bit for bit those of one wave doing everything in turn.
#define LB_OP_END 1
#define LB_OP_FORMT 2
#define LB_OP_CMPRLB 3
#define LB_OP_DTD 4
__device__ inline int lds_peek(ldsi* p) { return __builtin_amdgcn_readfirstlane(*(volatile ldsi*)p); }
__device__ inline void lb_help_post(const LbLds L, int op, int lane) {          // wave 0; its LDS writes so far are visible first
  LSYNC();
  if (lane == 0) { const int c = (*(volatile ldsi*)&ISC(I_HCMD) & ~15) + 16; *(volatile ldsi*)&ISC(I_HCMD) = c | op; }
  LSYNC();
}
__device__ inline void lb_help_wait(const LbLds L, int lane) {                  // wave 0: the helper has answered the last command
  const int want = lds_peek(&ISC(I_HCMD));
  bool ok = false;
  for (int spin = 0; spin < (1 << 21); ++spin) {
    if (lds_peek(&ISC(I_HDONE)) == want) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (!ok && lane == 0) { ISC(I_STATUS) = PCABO_ERR_HIP; ISC(I_INFO) = -99; }   // (cannot happen: the helper's loop is bounded too)
  LSYNC();
}
// wave 0, wherever the factor T is needed (or its failure has to be known): a formt handed to the helper is awaited once
__device__ inline void lb_await_formt(const LbLds L, int lane) {
  if (ISR(I_HPEND)) { lb_help_wait(L, lane); sti0(&ISC(I_HPEND), 0, lane); LSYNC(); }
}

// Generalised Cauchy point (lbfgsb.cpp: cauchy, the branch for variables with both bounds)
__device__ __noinline__ void lb_cauchy(const LbLds L, int lane) {
  const int n = L.n, col = ISR(I_COL), head = ISR(I_HEAD), col2 = 2 * col;
  const double theta = SR(S_THETA);
  ldsd* p = L.wa(); ldsd* c = L.wa() + 2 * LB_M; ldsd* wbp = L.wa() + 4 * LB_M; ldsd* v = L.wa() + 6 * LB_M;
  ldsd* t = L.t(); ldsd* d = L.d(); ldsd* xcp = L.z();
  ldsi* iorder = L.indx2();
  if (SR(S_SBGNRM) <= 0.0) { for (int i = lane; i < n; i += 64) xcp[i] = L.x()[i]; LSYNC(); return; }
  LBT_BEGIN();
  int nbreak = 0;
  double bk_lane = INFINITY;                       // the lane's smallest breakpoint (the search below starts from the wave's)
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    double neggi = 0.0, tl = 0.0, tu = 0.0;
    bool moving = false;
    if (i < n) {
      neggi = -L.g()[i]; tl = L.x()[i] - L.lo()[i]; tu = L.hi()[i] - L.x()[i];
      int iw = L.iwhere()[i];
      if (iw != 3) {
        iw = 0;
        if (tl <= 0.0) { if (neggi <= 0.0) iw = 1; }
        else if (tu <= 0.0) { if (neggi >= 0.0) iw = 2; }
        else if (fabs(neggi) <= 0.0) iw = -3;
        L.iwhere()[i] = iw;
      }
      moving = iw == 0;
      d[i] = moving ? neggi : 0.0;
      L.prod()[i] = moving ? neggi * neggi : 0.0;
    }
    const bool brk = moving && neggi != 0.0;
    const unsigned long long mb = __ballot(brk);
    const int pb = nbreak + __popcll(mb & lanes_below(lane));
    if (brk) {
      const double ahead = neggi < 0.0 ? tl : tu, tb = ahead / fabs(neggi);
      iorder[pb] = i; t[pb] = tb;
      bk_lane = tb < bk_lane ? tb : bk_lane;
    }
    nbreak += __popcll(mb);
  }
  LSYNC();
  LBT_NEXT(32);
  double f1 = -wave_tsum(L.prod(), n, lane);       // f1 = -(sum of neggi^2 over the moving variables; the others add 0.0), tree order
  if (col > 0) {
    lb_accum<false>(L, d, nullptr, lane, L.acc(), L.tile(0));   // d[i] = -g[i] for the moving variables, 0.0 for the others
    if (lane < col) {
      int pointr = head + lane; if (pointr >= LB_M) pointr -= LB_M;
      p[lane] = L.acc()[pointr];
      p[col + lane] = L.acc()[LB_M + pointr] * theta;       // (x * 1.0 == x: the host's "if theta != 1" changes nothing)
    }
  }
  for (int i = lane; i < n; i += 64) xcp[i] = L.x()[i];
  if (lane < col2) c[lane] = 0.0;
  LSYNC();
  LBT_NEXT(33);
  if (nbreak == 0) return;                        // (with both bounds everywhere nfree stays n)
  double f2 = -theta * f1;
  const double f2_org = f2;
  if (col > 0) {
    lb_await_formt(L, lane);                        // T comes from the helper wave (lb_step)
    LBT_NEXT(34);
    if (ISR(I_INFO) != 0) return;                   // formt failed: the caller resets the memory and starts the iteration again
    const int info = lb_bmv(L, p, v, lane);
    if (info != 0) { sti0(&ISC(I_INFO), info, lane); LSYNC(); return; }
    f2 -= small_ddot(v, p, col2, lane);
  }
  double dtm = -f1 / f2;
  double tsum = 0.0;
  bool skip_to_999 = false;
  LBT_NEXT(35);
  // the minimiser lies in front of the first breakpoint (the usual case late in a run): the search's first test, taken on the
  // minimum the classification loop has kept in registers - nothing of the breakpoint set is read or reordered
  if (!(dtm < wave_min(bk_lane))) {
    int nleft = nbreak;
    double tj = 0.0;
    int ties = 0;
    int nleft_dbg = nbreak; (void)nleft_dbg;
    while (true) {
      const double tj0 = tj;
      // smallest remaining breakpoint.  The host pops a heap; the order of DISTINCT values does not depend on the heap's shape -
      // two bit-equal breakpoints do (counted in I_TIES; the position of the first one in the array decides here)
      double bv = INFINITY; int bp = 0x7fffffff;
      int mine = 0;                                // the lane's elements that equal its minimum
      for (int e = lane; e < nleft; e += 64) { const double tv = t[e]; if (tv < bv) { bv = tv; bp = e; mine = 1; } else if (tv == bv) ++mine; }
      const double gmin = wave_min(bv);
      const bool at = bv == gmin;
      bp = wave_min_i(at ? bp : 0x7fffffff);
      if (bp == 0x7fffffff) bp = 0;               // (all NaN cannot happen; keep the index in range)
      { const unsigned long long ma = __ballot(at && nleft > 0);
        if (__popcll(ma) > 1 || __ballot(at && mine > 1)) ++ties; }
      tj = t[bp];
      const int ibp = iorder[bp];
      LSYNC();
      if (lane == 0) { t[bp] = t[nleft - 1]; iorder[bp] = iorder[nleft - 1]; }      // remove it from the set
      LSYNC();
      const double dt = tj - tj0;
      if (dtm < dt) break;
      tsum += dt;
      --nleft; nleft_dbg = nleft;

      const double dibp = d[ibp];
      double zibp;
      const double xi = L.x()[ibp], ui = L.hi()[ibp], li = L.lo()[ibp];
      LSYNC();
      if (dibp > 0.0) { zibp = ui - xi; if (lane == 0) { d[ibp] = 0.0; xcp[ibp] = ui; L.iwhere()[ibp] = 2; } }
      else { zibp = li - xi; if (lane == 0) { d[ibp] = 0.0; xcp[ibp] = li; L.iwhere()[ibp] = 1; } }
      LSYNC();
      if (nleft == 0 && nbreak == n) { dtm = dt; skip_to_999 = true; break; }
      const double dibp2 = dibp * dibp;
      f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
      f2 = f2 - theta * dibp2;
      if (col > 0) {
        if (lane < col2) c[lane] += dt * p[lane];
        if (lane < col) {
          int pointr = head + lane; if (pointr >= LB_M) pointr -= LB_M;
          wbp[lane] = WY_(ibp, pointr);
          wbp[col + lane] = theta * WS_(ibp, pointr);
        }
        LSYNC();
        const int info = lb_bmv(L, wbp, v, lane);
        if (info != 0) { sti0(&ISC(I_INFO), info, lane); LSYNC(); return; }
        const double wmc = small_ddot(c, v, col2, lane);
        const double wmp = small_ddot(p, v, col2, lane);
        const double wmw = small_ddot(wbp, v, col2, lane);
        LSYNC();
        if (lane < col2) p[lane] -= dibp * wbp[lane];
        LSYNC();
        f1 += dibp * wmc;
        f2 = f2 + 2.0 * dibp * wmp - dibp2 * wmw;
      }
      f2 = fmax(DBL_EPSILON * f2_org, f2);
      if (nleft > 0) { dtm = -f1 / f2; continue; }
      // (bnded: every variable has both bounds)
      f1 = 0.0; f2 = 0.0; dtm = 0.0;
      break;
    }
    if (ties && lane == 0) ISC(I_TIES) += ties;
#ifdef PCABO_ACQ_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) g_lb_calls[40] += nbreak - nleft_dbg;       // breakpoints crossed
#endif
  }
  LBT_NEXT(36);
  if (!skip_to_999) {
    if (dtm <= 0.0) dtm = 0.0;
    tsum += dtm;
    for (int i = lane; i < n; i += 64) xcp[i] += tsum * d[i];
  }
  if (col > 0 && lane < col2) c[lane] += dtm * p[lane];
  LSYNC();
}

__device__ __noinline__ void lb_freev(const LbLds L, int lane) {
  const int n = L.n, nfree_old = ISR(I_NFREE), iter = ISR(I_ITER);
  int nenter = 0, ileave = n;
  if (iter > 0) {
    for (int base = 0; base < nfree_old; base += 64) {
      const int i = base + lane;
      int k = 0; bool f = false;
      if (i < nfree_old) { k = L.index()[i]; f = L.iwhere()[k] > 0; }
      const unsigned long long m = __ballot(f);
      if (f) L.indx2()[ileave - 1 - __popcll(m & lanes_below(lane))] = k;
      ileave -= __popcll(m);
    }
    for (int base = nfree_old; base < n; base += 64) {
      const int i = base + lane;
      int k = 0; bool f = false;
      if (i < n) { k = L.index()[i]; f = L.iwhere()[k] <= 0; }
      const unsigned long long m = __ballot(f);
      if (f) L.indx2()[nenter + __popcll(m & lanes_below(lane))] = k;
      nenter += __popcll(m);
    }
  }
  LSYNC();
  const int wrk = (ileave < n) || (nenter > 0) || ISR(I_UPDATD);
  int nfree = 0, iact = n;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool in = i < n;
    const bool fr = in && L.iwhere()[i] <= 0;
    const unsigned long long mf = __ballot(fr), ma = __ballot(in && !fr);
    if (fr) L.index()[nfree + __popcll(mf & lanes_below(lane))] = i;
    else if (in) L.index()[iact - 1 - __popcll(ma & lanes_below(lane))] = i;
    nfree += __popcll(mf);
    iact -= __popcll(ma);
  }
  if (lane == 0) { ISC(I_NENTER) = nenter; ISC(I_ILEAVE) = ileave; ISC(I_WRK) = wrk; ISC(I_NFREE) = nfree; }
  LSYNC();
}

// LEL' factorisation of the indefinite subspace matrix (lbfgsb.cpp: formk)
__device__ __noinline__ void lb_formk(const LbLds L, int lane) {
  const int n = L.n, m = LB_M, col = ISR(I_COL), head = ISR(I_HEAD);
  const int nenter = ISR(I_NENTER), ileave = ISR(I_ILEAVE);
  const double theta = SR(S_THETA);
  const ldsi* indx2 = L.indx2();
  int upcl;
  LBT_BEGIN();
  if (ISR(I_UPDATD)) {
    if (ISR(I_IUPDAT) > m) {
      // shift the three blocks of WN1 one step up-left: every source is read before anything is written
      double src[3]; int dst[3];
      const int tri = m * (m - 1) / 2, tot = 2 * tri + (m - 1) * (m - 1);      // 171 elements: three per lane
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int e = lane + 64 * u;
        int di = 0, dj = 0, si = 0, sj = 0;
        if (e < 2 * tri) {
          // rows of m - 1, m - 2, ... 1 elements: the packed triangle read backwards
          int ap, rp;
          tri_decode(tri - 1 - (e < tri ? e : e - tri), ap, rp);
          const int jy = m - 2 - ap, tt = ap - rp;
          if (e < tri) { di = jy + tt; dj = jy; si = jy + 1 + tt; sj = jy + 1; }
          else { const int js = m + jy; di = js + tt; dj = js; si = js + 1 + tt; sj = js + 1; }
        } else if (e < tot) {
          const int ee = e - 2 * tri, jy = ee / (m - 1), tt = ee % (m - 1);
          di = m + tt; dj = jy; si = m + 1 + tt; sj = jy + 1;
        }
        src[u] = e < tot ? WN1_(si, sj) : 0.0;
        dst[u] = e < tot ? dj * 2 * m + di : -1;
      }
      LSYNC();
#pragma unroll
      for (int u = 0; u < 3; ++u) if (dst[u] >= 0) L.wn1()[dst[u]] = src[u];
      LSYNC();
    }
    const int ipntr = (head + col - 1) % m;
    const int iy = col - 1, is = m + col - 1;
    // free variables (increasing order) with WY(k, ipntr), active ones (the host's list runs downwards) with WS(k, ipntr)
    for (int k = lane; k < n; k += 64) {
      const bool fr = L.iwhere()[k] <= 0;
      L.coef()[k] = fr ? WY_(k, ipntr) : 0.0;
      L.prod()[k] = fr ? 0.0 : WS_(k, ipntr);
    }
    LSYNC();
    LBT_NEXT(18);
    lb_accum<true>(L, L.coef(), L.prod(), lane, L.acc(), L.tile(0));
    LBT_NEXT(19);
    if (lane < col) {
      int jp = head + lane; if (jp >= m) jp -= m;
      WN1_(iy, lane) = L.acc()[jp];                 // t1
      WN1_(is, m + lane) = L.acc()[32 + m + jp];    // t2
      WN1_(is, lane) = L.acc()[32 + jp];            // t3
    }
    LSYNC();
    if (lane < col) { int jp = head + lane; if (jp >= m) jp -= m; WN1_(m + lane, col - 1) = L.acc()[m + jp]; }    // t4 last
    LSYNC();
    upcl = col - 1;
  } else {
    upcl = col;
  }
  LBT_NEXT(20);
  // corrections for the variables that entered / left the free set
  {
    const int npair = upcl * (upcl + 1) / 2;
    for (int e = lane; e < npair; e += 64) {
      int iy, jy;
      tri_decode(e, iy, jy);
      const int is = m + iy, js = m + jy;
      int ipntr = head + iy; if (ipntr >= m) ipntr -= m;
      int jpntr = head + jy; if (jpntr >= m) jpntr -= m;
      double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0, temp4 = 0.0;
      for (int k = 0; k < nenter; ++k) { const int k1 = indx2[k]; temp1 += WY_(k1, ipntr) * WY_(k1, jpntr); temp2 += WS_(k1, ipntr) * WS_(k1, jpntr); }
      for (int k = ileave; k < n; ++k) { const int k1 = indx2[k]; temp3 += WY_(k1, ipntr) * WY_(k1, jpntr); temp4 += WS_(k1, ipntr) * WS_(k1, jpntr); }
      WN1_(iy, jy) = WN1_(iy, jy) + temp1 - temp3;
      WN1_(is, js) = WN1_(is, js) - temp2 + temp4;
    }
    for (int e = lane; e < LB_M * LB_M; e += 64) {          // (a fixed 10 x 10 grid, masked: no division by a runtime number)
      const int isr = e / LB_M, jy = e % LB_M, is = m + isr;
      if (isr >= upcl || jy >= upcl) continue;
      int ipntr = head + isr; if (ipntr >= m) ipntr -= m;
      int jpntr = head + jy; if (jpntr >= m) jpntr -= m;
      double temp1 = 0.0, temp3 = 0.0;
      for (int k = 0; k < nenter; ++k) { const int k1 = indx2[k]; temp1 += WS_(k1, ipntr) * WY_(k1, jpntr); }
      for (int k = ileave; k < n; ++k) { const int k1 = indx2[k]; temp3 += WS_(k1, ipntr) * WY_(k1, jpntr); }
      if (is <= jy + m) WN1_(is, jy) = WN1_(is, jy) + temp1 - temp3;
      else WN1_(is, jy) = WN1_(is, jy) - temp1 + temp3;
    }
    LSYNC();
  }
  LBT_NEXT(21);
  // upper triangle of WN
  for (int e = lane; e < LB_M * LB_M; e += 64) {
    const int iy = e / LB_M, jy = e % LB_M, is = col + iy, is1 = m + iy, js = col + jy, js1 = m + jy;
    if (iy >= col || jy >= col) continue;
    if (jy <= iy) {
      double w = WN1_(iy, jy) / theta;
      if (jy == iy) w += SY_(iy, iy);
      WN_(jy, iy) = w;
      WN_(js, is) = WN1_(is1, js1) * theta;
    }
    WN_(jy, is) = jy < iy ? -WN1_(is1, jy) : WN1_(is1, jy);
  }
  LSYNC();
  LBT_NEXT(22);
  int info = lb_dpofa(L.wn(), 2 * m, col, lane);
  LBT_NEXT(23);
  if (info != 0) { sti0(&ISC(I_INFO), -1, lane); LSYNC(); return; }
  // the col right-hand sides WN(0:col, js), js = col .. 2 col - 1: a system per lane (job 11, the host's order)
  {
    const int js = col + (lane < col ? lane : 0);
    double b[LB_M];
#pragma unroll
    for (int i = 0; i < LB_M; ++i) b[i] = i < col ? WN_(i, js) : 0.0;
    const double rl = 1.0 / WN_(lane < col ? lane : 0, lane < col ? lane : 0);      // the pivots' reciprocals, one per lane (dtrsl_recip)
#pragma unroll
    for (int j = 0; j < LB_M; ++j) {
      if (j < col) {
        const double rj = bcast(rl, j);
        if (j == 0) b[0] = b[0] * rj;
        else {
          double s = 0.0;
#pragma unroll
          for (int i = 0; i < j; ++i) s += WN_(i, j) * b[i];
          b[j] = b[j] - s;
          b[j] = b[j] * rj;
        }
      }
    }
    LSYNC();
    if (lane < col) {
#pragma unroll
      for (int i = 0; i < LB_M; ++i) if (i < col) WN_(i, js) = b[i];
    }
    LSYNC();
  }
  LBT_NEXT(24);
  for (int e = lane; e < col * (col + 1) / 2; e += 64) {
    int a, ee;
    tri_decode(e, a, ee);                          // pair (ee <= a)
    const int is = col + ee, js = col + a;
    double s = 0.0;
    for (int i = 0; i < col; ++i) s += WN_(i, is) * WN_(i, js);
    WN_(is, js) += s;
  }
  LSYNC();
  LBT_NEXT(25);
  info = lb_dpofa(&WN_(col, col), 2 * m, col, lane);
  LBT_NEXT(26);
  if (info != 0) { sti0(&ISC(I_INFO), -2, lane); LSYNC(); }
}

__device__ __noinline__ void lb_cmprlb(const LbLds L, int lane, int info_word) {
  const int n = L.n, col = ISR(I_COL), nfree = ISR(I_NFREE);
  const double theta = SR(S_THETA);
  for (int k = lane; k < n; k += 64) L.full()[k] = -theta * (L.z()[k] - L.x()[k]) - L.g()[k];
  LSYNC();
  const int info = lb_bmv(L, L.wa() + 2 * LB_M, L.wa(), lane);
  if (info != 0) { sti0(&ISC(info_word), -8, lane); LSYNC(); return; }
  {
    // a chain per variable over the history columns (the published order), the variable's value in a register meanwhile
    constexpr int NU = (LB_NVCAP + 63) / 64;
    double f[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { const int k = lane + 64 * u; f[u] = k < n ? L.full()[k] : 0.0; }
    int pointr = ISR(I_HEAD);
#pragma unroll
    for (int j = 0; j < LB_M; ++j) {
      if (j < col) {
        const double a1 = L.wa()[j], a2 = theta * L.wa()[col + j];
#pragma unroll
        for (int u = 0; u < NU; ++u) { const int k = lane + 64 * u; if (k < n) f[u] += WY_(k, pointr) * a1 + WS_(k, pointr) * a2; }
        pointr = nxt(pointr);
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) { const int k = lane + 64 * u; if (k < n) L.full()[k] = f[u]; }
  }
  LSYNC();
  for (int i = lane; i < nfree; i += 64) L.r()[i] = L.full()[L.index()[i]];
  LSYNC();
}

// the first part of subsm: it needs the reduced gradient (cmprlb), not formk's factor - the helper wave runs it beside formk
__device__ inline void lb_subsm_head(const LbLds L, int lane) {
  const int n = L.n, nsub = ISR(I_NFREE);
  const ldsi* ind = L.index();
  const ldsd* d = L.r();
  // full = d scattered to the variables' own places (zeros elsewhere; with every variable free it is d itself)
  for (int k = lane; k < n; k += 64) L.full()[k] = 0.0;
  LSYNC();
  for (int i = lane; i < nsub; i += 64) L.full()[ind[i]] = d[i];
  LSYNC();
  lb_accum<false>(L, L.full(), nullptr, lane, L.acc2(), L.tile(1));
}

__device__ __noinline__ void lb_subsm(const LbLds L, int lane) {
  const int n = L.n, m = LB_M, col = ISR(I_COL), nsub = ISR(I_NFREE), col2 = 2 * col;
  const double theta = SR(S_THETA);
  const ldsi* ind = L.index();
  ldsd* x = L.z(); ldsd* d = L.r(); ldsd* wv = L.wa();
  if (nsub <= 0) return;
  LBT_BEGIN();
  // (full = d scattered to the variables' own places and W' full: lb_subsm_head, run by the helper wave right after cmprlb)
  LBT_NEXT(27);
  LBT_NEXT(28);
  double b = 0.0;
  if (lane < col2) {
    const int i = lane < col ? lane : lane - col;
    int pointr = ISR(I_HEAD) + i; if (pointr >= m) pointr -= m;
    b = lane < col ? L.acc2()[pointr] : theta * L.acc2()[m + pointr];
  }
  int info = 0;
  b = lb_subsm_solves(L, b, &info, lane);
  if (info != 0) { sti0(&ISC(I_INFO), info, lane); LSYNC(); return; }
  LBT_NEXT(29);
  if (lane < col2) wv[lane] = b;
  LSYNC();
  const double inv_theta = 1.0 / theta;
  {
    constexpr int NU = (LB_NVCAP + 63) / 64;
    double f[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { const int k = lane + 64 * u; f[u] = k < n ? L.full()[k] : 0.0; }
    int pointr = ISR(I_HEAD);
#pragma unroll
    for (int jy = 0; jy < LB_M; ++jy) {
      if (jy < col) {
        const double a1 = wv[jy], a2 = wv[col + jy];
#pragma unroll
        for (int u = 0; u < NU; ++u) { const int k = lane + 64 * u; if (k < n) f[u] = f[u] + WY_(k, pointr) * a1 * inv_theta + WS_(k, pointr) * a2; }
        pointr = nxt(pointr);
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) { const int k = lane + 64 * u; if (k < n) L.full()[k] = f[u]; }
  }
  LSYNC();
  LBT_NEXT(30);
  for (int i = lane; i < nsub; i += 64) d[i] = L.full()[ind[i]] * inv_theta;
  for (int i = lane; i < n; i += 64) L.xp()[i] = x[i];
  LSYNC();
  bool hit = false;
  for (int i = lane; i < nsub; i += 64) {
    const int k = ind[i];
    const double dk = d[i], v = x[k] + dk, lk = L.lo()[k], uk = L.hi()[k];
    const double xk = lk > v ? lk : v;
    const double w = uk < xk ? uk : xk;
    x[k] = w;
    if (w == lk || w == uk) hit = true;
  }
  const int iword = __ballot(hit) ? 1 : 0;
  sti0(&ISC(I_IWORD), iword, lane);
  LSYNC();
  LBT_NEXT(31);
  if (iword == 0) return;
  for (int i = lane; i < n; i += 64) L.prod()[i] = (x[i] - L.x()[i]) * L.g()[i];
  LSYNC();
  const double dd_p = wave_tsum(L.prod(), n, lane);
  if (dd_p > 0.0) {
    for (int i = lane; i < n; i += 64) x[i] = L.xp()[i];
    LSYNC();
    double alpha = 1.0, temp1 = alpha;
    int ibd = 0;
    for (int i = 0; i < nsub; ++i) {                 // (every lane the same scan: rare path)
      const int k = ind[i];
      const double dk = d[i];
      if (dk < 0.0) {
        const double temp2 = L.lo()[k] - x[k];
        if (temp2 >= 0.0) temp1 = 0.0;
        else if (dk * alpha < temp2) temp1 = temp2 / dk;
      } else if (dk > 0.0) {
        const double temp2 = L.hi()[k] - x[k];
        if (temp2 <= 0.0) temp1 = 0.0;
        else if (dk * alpha > temp2) temp1 = temp2 / dk;
      }
      if (temp1 < alpha) { alpha = temp1; ibd = i; }
    }
    if (alpha < 1.0) {
      const double dk = d[ibd];
      const int k = ind[ibd];
      LSYNC();
      if (lane == 0) {
        if (dk > 0.0) { x[k] = L.hi()[k]; d[ibd] = 0.0; }
        else if (dk < 0.0) { x[k] = L.lo()[k]; d[ibd] = 0.0; }
      }
      LSYNC();
    }
    for (int i = lane; i < nsub; i += 64) { const int k = ind[i]; x[k] += alpha * d[i]; }
    LSYNC();
  }
}

// ---- More-Thuente line search (lbfgsb.cpp: dcstep / dcsrch), every lane the same scalars ------------------------------------------
struct LsState { int task, brackt, stage; double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1; };

__device__ inline void lb_dcstep(double* stx, double* fx, double* dx, double* sty, double* fy, double* dy, double* stp, double fp,
                          double dp, int* brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (*dx / fabs(*dx));
  double stpf, stpc, stpq, theta, s, gamma, p, q, r;
  if (fp > *fx) {
    theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
    gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    if (*stp < *stx) gamma = -gamma;
    p = (gamma - *dx) + theta;
    q = ((gamma - *dx) + gamma) + dp;
    r = p / q;
    stpc = *stx + r * (*stp - *stx);
    stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
    if (fabs(stpc - *stx) < fabs(stpq - *stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    *brackt = 1;
  } else if (sgnd < 0.0) {
    theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
    gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + *dx;
    r = p / q;
    stpc = *stp + r * (*stx - *stp);
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
    else stpf = stpq;
    *brackt = 1;
  } else if (fabs(dp) < fabs(*dx)) {
    theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
    gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (*dx - dp)) + gamma;
    r = p / q;
    if (r < 0.0 && gamma != 0.0) stpc = *stp + r * (*stx - *stp);
    else if (*stp > *stx) stpc = stpmax;
    else stpc = stpmin;
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (*brackt) {
      if (fabs(stpc - *stp) < fabs(stpq - *stp)) stpf = stpc;
      else stpf = stpq;
      if (*stp > *stx) stpf = fmin(*stp + 0.66 * (*sty - *stp), stpf);
      else stpf = fmax(*stp + 0.66 * (*sty - *stp), stpf);
    } else {
      if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
      else stpf = stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {
    if (*brackt) {
      theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
      s = fmax(fabs(theta), fmax(fabs(*dy), fabs(dp)));
      gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
      if (*stp > *sty) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + *dy;
      r = p / q;
      stpc = *stp + r * (*sty - *stp);
      stpf = stpc;
    } else if (*stp > *stx) stpf = stpmax;
    else stpf = stpmin;
  }
  if (fp > *fx) { *sty = *stp; *fy = fp; *dy = dp; }
  else {
    if (sgnd < 0.0) { *sty = *stx; *fy = *fx; *dy = *dx; }
    *stx = *stp; *fx = fp; *dx = dp;
  }
  *stp = stpf;
}

__device__ inline void lb_dcsrch(double f, double g, double* stp, double ftol, double gtol, double xtol, double stpmin,
                          double stpmax, LsState& s) {
  const double xtrapl = 1.1, xtrapu = 4.0, p5 = 0.5, p66 = 0.66;
  if (s.task == 0) {
    if (*stp < stpmin || *stp > stpmax || g >= 0.0) { s.task = 4; return; }
    s.brackt = 0; s.stage = 1; s.finit = f; s.ginit = g; s.gtest = ftol * s.ginit;
    s.width = stpmax - stpmin; s.width1 = s.width / p5;
    s.stx = 0.0; s.fx = s.finit; s.gx = s.ginit; s.sty = 0.0; s.fy = s.finit; s.gy = s.ginit;
    s.stmin = 0.0; s.stmax = *stp + xtrapu * *stp;
    s.task = 1;
    return;
  }
  const double ftest = s.finit + *stp * s.gtest;
  if (s.stage == 1 && f <= ftest && g >= 0.0) s.stage = 2;
  int task = 1;
  if (s.brackt && (*stp <= s.stmin || *stp >= s.stmax)) task = 3;
  if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) task = 3;
  if (*stp == stpmax && f <= ftest && g <= s.gtest) task = 3;
  if (*stp == stpmin && (f > ftest || g >= s.gtest)) task = 3;
  if (f <= ftest && fabs(g) <= gtol * (-s.ginit)) task = 2;
  if (task == 2 || task == 3) { s.task = task; return; }
  if (s.stage == 1 && f <= s.fx && f > ftest) {
    double fm = f - *stp * s.gtest, fxm = s.fx - s.stx * s.gtest, fym = s.fy - s.sty * s.gtest;
    double gm = g - s.gtest, gxm = s.gx - s.gtest, gym = s.gy - s.gtest;
    lb_dcstep(&s.stx, &fxm, &gxm, &s.sty, &fym, &gym, stp, fm, gm, &s.brackt, s.stmin, s.stmax);
    s.fx = fxm + s.stx * s.gtest; s.fy = fym + s.sty * s.gtest; s.gx = gxm + s.gtest; s.gy = gym + s.gtest;
  } else {
    lb_dcstep(&s.stx, &s.fx, &s.gx, &s.sty, &s.fy, &s.gy, stp, f, g, &s.brackt, s.stmin, s.stmax);
  }
  if (s.brackt) {
    if (fabs(s.sty - s.stx) >= p66 * s.width1) *stp = s.stx + p5 * (s.sty - s.stx);
    s.width1 = s.width;
    s.width = fabs(s.sty - s.stx);
  }
  if (s.brackt) { s.stmin = fmin(s.stx, s.sty); s.stmax = fmax(s.stx, s.sty); }
  else { s.stmin = *stp + xtrapl * (*stp - s.stx); s.stmax = *stp + xtrapu * (*stp - s.stx); }
  *stp = fmax(*stp, stpmin);
  *stp = fmin(*stp, stpmax);
  if ((s.brackt && (*stp <= s.stmin || *stp >= s.stmax)) || (s.brackt && s.stmax - s.stmin <= xtol * s.stmax))
    *stp = s.stx;
  s.task = 1;
}

__device__ inline void ls_load(const LbLds L, LsState& s) {
  s.task = ISR(LS_TASK); s.brackt = ISR(LS_BRACKT); s.stage = ISR(LS_STAGE);
  s.ginit = SR(LS_GINIT); s.gtest = SR(LS_GTEST); s.gx = SR(LS_GX); s.gy = SR(LS_GY); s.finit = SR(LS_FINIT);
  s.fx = SR(LS_FX); s.fy = SR(LS_FY); s.stx = SR(LS_STX); s.sty = SR(LS_STY); s.stmin = SR(LS_STMIN); s.stmax = SR(LS_STMAX);
  s.width = SR(LS_WIDTH); s.width1 = SR(LS_WIDTH1);
}
__device__ inline void ls_store(const LbLds L, const LsState& s, int lane) {
  if (lane == 0) {
    ISC(LS_TASK) = s.task; ISC(LS_BRACKT) = s.brackt; ISC(LS_STAGE) = s.stage;
    SC(LS_GINIT) = s.ginit; SC(LS_GTEST) = s.gtest; SC(LS_GX) = s.gx; SC(LS_GY) = s.gy; SC(LS_FINIT) = s.finit;
    SC(LS_FX) = s.fx; SC(LS_FY) = s.fy; SC(LS_STX) = s.stx; SC(LS_STY) = s.sty; SC(LS_STMIN) = s.stmin; SC(LS_STMAX) = s.stmax;
    SC(LS_WIDTH) = s.width; SC(LS_WIDTH1) = s.width1;
  }
}

// One call of the line-search driver (lbfgsb.cpp: lnsrlb).  Sets I_TASK to FG (x holds the trial point) or NEW_X.
__device__ __noinline__ void lb_lnsrlb(const LbLds L, int lane) {
  const int n = L.n;
  const double big = 1e10, ftol = 1e-3, gtol = 0.9, xtol = 0.1;
  const double f = SR(S_F);
  LsState ls;
  double stp, stpmx;
  const bool first_call = ISR(I_PHASE) != 2;
  LBT_BEGIN();
  if (first_call) {
    // d'd is not needed before the next matupd: the helper wave forms it (its own product buffer) while this wave goes on
    lb_help_post(L, LB_OP_DTD, lane);
    stpmx = big;
    if (ISR(I_ITER) == 0) stpmx = 1.0;
    else {
      // the host's scan  "if (a1 stpmx < a2) stpmx = a2 / a1"  in order: stpmx only falls and the products are monotone in it,
      // so within a chunk of 64 the first lane whose test holds is the next one to change it; the others test again after it
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        double a1 = 0.0, a2 = 0.0; int kind = 0;                  // 1: towards the lower bound, 2: towards the upper
        if (i < n) {
          a1 = L.d()[i];
          if (a1 < 0.0) { a2 = L.lo()[i] - L.x()[i]; kind = 1; }
          else if (a1 > 0.0) { a2 = L.hi()[i] - L.x()[i]; kind = 2; }
        }
        unsigned long long pending = ~0ull;
        for (;;) {
          bool upd = false;
          if (kind == 1) upd = (a2 >= 0.0) || (a1 * stpmx < a2);
          else if (kind == 2) upd = (a2 <= 0.0) || (a1 * stpmx > a2);
          const unsigned long long mk = __ballot(upd) & pending;
          if (!mk) break;
          const int first = __ffsll((long long)mk) - 1;
          double nv;
          if (kind == 1) nv = a2 >= 0.0 ? 0.0 : a2 / a1; else nv = a2 <= 0.0 ? 0.0 : a2 / a1;
          stpmx = bcast(nv, first);
          pending = first == 63 ? 0ull : (~0ull << (first + 1));
        }
      }
    }
    stp = 1.0;                                        // (boxed: never min(1 / dnorm, stpmx))
    for (int i = lane; i < n; i += 64) { L.t()[i] = L.x()[i]; L.r()[i] = L.g()[i]; }
    if (lane == 0) { SC(S_STPMX) = stpmx; SC(S_FOLD) = f; ISC(I_IFUN) = 0; ISC(I_IBACK) = 0; }
    ls.task = 0; ls.brackt = 0; ls.stage = 1;
    ls.ginit = ls.gtest = ls.gx = ls.gy = ls.finit = ls.fx = ls.fy = ls.stx = ls.sty = ls.stmin = ls.stmax = ls.width = ls.width1 = 0.0;
    LSYNC();
  } else {
    ls_load(L, ls);
    stp = SR(S_STP); stpmx = SR(S_STPMX);
  }
  LBT_NEXT(37);
  const double gd = wave_ddot(L.g(), L.d(), n, lane);
  LBT_NEXT(38);
  if (first_call) lb_help_wait(L, lane);              // (the helper's chain ran beside this one)
  int ifun = ISR(I_IFUN);
  LSYNC();
  if (lane == 0) SC(S_GD) = gd;
  if (ifun == 0) {
    if (lane == 0) SC(S_GDOLD) = gd;
    if (gd >= 0.0) { sti0(&ISC(I_INFO), -4, lane); ls_store(L, ls, lane); st0(&SC(S_STP), stp, lane); LSYNC(); return; }
  }
  lb_dcsrch(f, gd, &stp, ftol, gtol, xtol, 0.0, stpmx, ls);
  ls_store(L, ls, lane);
  if (lane == 0) SC(S_STP) = stp;
  if (ls.task != 2 && ls.task != 3) {
    if (lane == 0) { ISC(I_TASK) = LBFGSB_FG; ISC(I_IFUN) = ifun + 1; ISC(I_NFGV) = ISR(I_NFGV) + 1; ISC(I_IBACK) = ifun; }
    if (stp == 1.0) { for (int i = lane; i < n; i += 64) L.x()[i] = L.z()[i]; }
    else { for (int i = lane; i < n; i += 64) L.x()[i] = stp * L.d()[i] + L.t()[i]; }
  } else {
    if (lane == 0) ISC(I_TASK) = LBFGSB_NEW_X;
  }
  LSYNC();
  LBT_NEXT(39);
}

// lbfgsb.cpp: matupd, in two parts.  Part A (wave 0): the new columns of WS / WY, theta and the ring's pointers - what the Cauchy
// search reads first.  Part B (the helper wave, followed by formt): the new row of SY and column of SS - the step s_k is read from
// the WS column part A has just written (the search overwrites d) - after the shift of the old entries, in the host's order.
__device__ inline void lb_matupd_a(const LbLds L, double rr, double dr, int lane) {
  const int n = L.n, m = LB_M, iupdat = ISR(I_IUPDAT);
  int col = ISR(I_COL), head = ISR(I_HEAD), itail = ISR(I_ITAIL);
  if (iupdat <= m) { col = iupdat; itail = (head + iupdat - 1) % m; }
  else { itail = nxt(itail); head = nxt(head); }
  for (int i = lane; i < n; i += 64) { WS_(i, itail) = L.d()[i]; WY_(i, itail) = L.r()[i]; }
  const double theta = rr / dr;
  if (lane == 0) { ISC(I_COL) = col; ISC(I_HEAD) = head; ISC(I_ITAIL) = itail; SC(S_THETA) = theta; SC(S_DR) = dr; }
  LSYNC();
}
__device__ __noinline__ void lb_matupd_b(const LbLds L, int lane) {
  const int m = LB_M, iupdat = ISR(I_IUPDAT);
  const int col = ISR(I_COL), head = ISR(I_HEAD), itail = ISR(I_ITAIL);          // (part A's values)
  if (iupdat > m) {
    double src[2]; int dst[2];
    const int tri = (col - 1) * col / 2;               // 45 + 45 elements: two per lane
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u;
      src[u] = 0.0; dst[u] = 0;
      if (e < tri) {                                   // SS(t, j) = SS(t + 1, j + 1), t <= j < col - 1
        int ee, j;
        tri_decode(e, j, ee);
        src[u] = SS_(ee + 1, j + 1); dst[u] = j * m + ee + 1;
      } else if (e < 2 * tri) {                        // SY(j + t, j) = SY(j + 1 + t, j + 1), t < col - 1 - j: rows of col - 1 ... 1 elements
        int ap, rp;
        tri_decode(2 * tri - 1 - e, ap, rp);
        const int j = col - 2 - ap, ee = ap - rp;
        src[u] = SY_(j + 1 + ee, j + 1); dst[u] = -(j * m + j + ee) - 1;
      }
    }
    LSYNC();
#pragma unroll
    for (int u = 0; u < 2; ++u) { if (dst[u] > 0) L.ss()[dst[u] - 1] = src[u]; else if (dst[u] < 0) L.sy()[-dst[u] - 1] = src[u]; }
  }
  LSYNC();
  lb_accum<false>(L, &WS_(0, itail), nullptr, lane, L.acc2(), L.tile(1));
  if (lane < col - 1) {
    int pointr = head + lane; if (pointr >= m) pointr -= m;
    SY_(col - 1, lane) = L.acc2()[pointr];
    SS_(lane, col - 1) = L.acc2()[m + pointr];
  }
  if (lane == 0) {
    const double stp = SR(S_STP), dtd = SR(S_DTD);
    SS_(col - 1, col - 1) = stp == 1.0 ? dtd : stp * stp * dtd;
    SY_(col - 1, col - 1) = SR(S_DR);
  }
  LSYNC();
}

__device__ __noinline__ void lb_formt(const LbLds L, int lane) {
  const int col = ISR(I_COL);
  const double theta = SR(S_THETA);
  ldsd* rs = L.acc2() + 40;                            // 1 / SY(k, k), formed side by side (lbfgsb.cpp formt, tree / wave order)
  if (lane < col) rs[lane] = 1.0 / SY_(lane, lane);
  LSYNC();
  for (int e = lane; e < col * (col + 1) / 2; e += 64) {
    int j, i;
    tri_decode(e, j, i);                               // i <= j
    if (i == 0) WT_(0, j) = theta * SS_(0, j);
    else {
      double ddum = 0.0;
      for (int k = 0; k < i; ++k) ddum += SY_(i, k) * SY_(j, k) * rs[k];
      WT_(i, j) = ddum + theta * SS_(i, j);
    }
  }
  LSYNC();
  const int info = lb_dpofa(L.wt(), LB_M, col, lane);
  if (info != 0) { sti0(&ISC(I_INFO), -3, lane); LSYNC(); }
}

// lbfgsb.cpp: Lbfgsb::step (reverse communication), the state in LDS.  Returns the task.
__device__ int lb_step(const LbLds L, int lane) {
  const int n = L.n;
  LBT_BEGIN();                                         // (timing build: the step's own code between the routines, slots 41 ..)
  const double pgtol = 1e-5, factr = 1e7;
  if (ISR(I_TASK) >= LBFGSB_CONV_PG) return ISR(I_TASK);
  bool need_iteration_start = false, resume_linesearch = false;
  const int phase = ISR(I_PHASE);
  if (phase == 0) {
    // (init: the caller has zeroed the state, clamped x into the box and classified the variables)
    if (lane == 0) { ISC(I_PHASE) = 1; ISC(I_TASK) = LBFGSB_FG; }
    LSYNC();
    return LBFGSB_FG;
  }
  if (phase == 1) {
    sti0(&ISC(I_NFGV), 1, lane);
    lb_projgr(L, lane);
    if (SR(S_SBGNRM) <= pgtol) { sti0(&ISC(I_TASK), LBFGSB_CONV_PG, lane); LSYNC(); return LBFGSB_CONV_PG; }
    need_iteration_start = true;
  } else if (phase == 2) {
    resume_linesearch = true;
  } else {
    if (SR(S_SBGNRM) <= pgtol) { sti0(&ISC(I_TASK), LBFGSB_CONV_PG, lane); LSYNC(); return LBFGSB_CONV_PG; }
    const double f = SR(S_F), fold = SR(S_FOLD);
    const double ddum = fmax(fmax(fabs(fold), fabs(f)), 1.0);
    if ((fold - f) <= (factr * DBL_EPSILON) * ddum) {
      if (lane == 0) { ISC(I_TASK) = LBFGSB_CONV_F; if (ISR(I_IBACK) >= 10) ISC(I_INFO) = -5; }
      LSYNC();
      return LBFGSB_CONV_F;
    }
    for (int i = lane; i < n; i += 64) L.r()[i] = L.g()[i] - L.r()[i];
    LSYNC();
    const double rr = wave_ddot(L.r(), L.r(), n, lane);
    const double stp = SR(S_STP), gd = SR(S_GD), gdold = SR(S_GDOLD);
    double dr, ddum2;
    if (stp == 1.0) { dr = gd - gdold; ddum2 = -gdold; }
    else {
      dr = (gd - gdold) * stp;
      for (int i = lane; i < n; i += 64) L.d()[i] *= stp;
      LSYNC();
      ddum2 = -gdold * stp;
    }
    if (dr <= DBL_EPSILON * ddum2) {
      sti0(&ISC(I_UPDATD), 0, lane);
      LSYNC();
    } else {
      if (lane == 0) { ISC(I_UPDATD) = 1; ISC(I_IUPDAT) = ISR(I_IUPDAT) + 1; }
      LSYNC();
      // formt goes to the helper wave; the Cauchy search below waits for it where it first needs T.  (If it fails the search
      // returns with the failure set and the memory is reset there - the host resets it here and searches with an empty
      // memory: the same state either way, the search's first part depends on x, g and the bounds only)
      LBT_NEXT(41);                                  // tests, r = g - r, r'r, scaling of d
      { LBT_BEGIN(); lb_matupd_a(L, rr, dr, lane); LBT_NEXT(6); }
      sti0(&ISC(I_HPEND), 1, lane);
      lb_help_post(L, LB_OP_FORMT, lane);               // (the rest of matupd, then formt)
    }
    need_iteration_start = true;
  }
  while (true) {
    if (need_iteration_start) {
      need_iteration_start = false;
      sti0(&ISC(I_IWORD), -1, lane);
      LBT_NEXT(42);                                  // posts, flags in front of the Cauchy search
      { LBT_BEGIN(); lb_cauchy(L, lane); LBT_NEXT(0); }
#ifdef PCABO_ACQ_TIMING
      lbt_t0_ = wall_clock64();
#endif
      lb_await_formt(L, lane);                          // (a search that returned before it needed T)
      if (ISR(I_INFO) != 0) { lb_reset_memory(L, lane); need_iteration_start = true; continue; }
      { LBT_BEGIN(); lb_freev(L, lane); LBT_NEXT(1); }
      if (ISR(I_NFREE) != 0 && ISR(I_COL) != 0) {
        // cmprlb on the helper wave beside formk (it reads the Cauchy point and T, formk the index sets: disjoint data); its
        // failure comes back in a word of its own and counts only if formk did not fail first, as on the host
        sti0(&ISC(I_INFO2), 0, lane);
        lb_help_post(L, LB_OP_CMPRLB, lane);
        if (ISR(I_WRK)) { LBT_BEGIN(); lb_formk(L, lane); LBT_NEXT(2); }
        { LBT_BEGIN(); lb_help_wait(L, lane); LBT_NEXT(3); }
        if (ISR(I_INFO) != 0) { lb_reset_memory(L, lane); need_iteration_start = true; continue; }
        if (ISR(I_INFO2) != 0) { sti0(&ISC(I_INFO), ISR(I_INFO2), lane); LSYNC(); }
        if (ISR(I_INFO) == 0) { LBT_BEGIN(); lb_subsm(L, lane); LBT_NEXT(4); }
        if (ISR(I_INFO) != 0) { lb_reset_memory(L, lane); need_iteration_start = true; continue; }
      }
#ifdef PCABO_ACQ_TIMING
      lbt_t0_ = wall_clock64();
#endif
      for (int i = lane; i < n; i += 64) L.d()[i] = L.z()[i] - L.x()[i];
      sti0(&ISC(I_PHASE), 0, lane);
      LSYNC();
      LBT_NEXT(43);                                  // d = z - x
    }
#ifdef PCABO_ACQ_TIMING
    lbt_t0_ = wall_clock64();
#endif
    if (resume_linesearch) { resume_linesearch = false; sti0(&ISC(I_PHASE), 2, lane); LSYNC(); }
    { LBT_BEGIN(); lb_lnsrlb(L, lane); LBT_NEXT(5); }
#ifdef PCABO_ACQ_TIMING
    lbt_t0_ = wall_clock64();
#endif
    if (ISR(I_INFO) != 0 || ISR(I_IBACK) >= 20) {
      for (int i = lane; i < n; i += 64) { L.x()[i] = L.t()[i]; L.g()[i] = L.r()[i]; }
      if (lane == 0) SC(S_F) = SR(S_FOLD);
      LSYNC();
      if (ISR(I_COL) == 0) {
        if (lane == 0) {
          if (ISR(I_INFO) == 0) { ISC(I_INFO) = -9; ISC(I_NFGV) -= 1; ISC(I_IFUN) -= 1; ISC(I_IBACK) -= 1; }
          ISC(I_TASK) = LBFGSB_ABNORMAL; ISC(I_ITER) += 1;
        }
        LSYNC();
        return LBFGSB_ABNORMAL;
      }
      if (lane == 0 && ISC(I_INFO) == 0) ISC(I_NFGV) -= 1;
      LSYNC();
      lb_reset_memory(L, lane);
      need_iteration_start = true;
      continue;
    }
    if (ISR(I_TASK) == LBFGSB_FG) { sti0(&ISC(I_PHASE), 2, lane); LSYNC(); return LBFGSB_FG; }
    sti0(&ISC(I_ITER), ISR(I_ITER) + 1, lane);
    LSYNC();
    lb_projgr(L, lane);
    if (lane == 0) { ISC(I_PHASE) = 3; ISC(I_TASK) = LBFGSB_NEW_X; }
    LSYNC();
    LBT_NEXT(44);                                    // the tail of an accepted line search: iteration count, projected gradient
    return LBFGSB_NEW_X;
  }
}

// wave 1 while wave 0 advances: serves commands until LB_OP_END (bounded polling)
__device__ void lb_helper(const LbLds L, int lane, int& last) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    const int c = lds_peek(&ISC(I_HCMD));
    if (c == last) { __builtin_amdgcn_s_sleep(2); continue; }
    last = c;
    const int op = c & 15;
    if (op == LB_OP_END) return;
    if (op == LB_OP_FORMT) { lb_matupd_b(L, lane); lb_formt(L, lane); }
    else if (op == LB_OP_DTD) {
      const double dtd = wave_ddot(L.d(), L.d(), L.n, lane);
      if (lane == 0) { SC(S_DTD) = dtd; SC(S_DNORM) = sqrt(dtd); }
    }
    else if (op == LB_OP_CMPRLB) { lb_cmprlb(L, lane, I_INFO2); if (ISR(I_INFO2) == 0 && ISR(I_NFREE) > 0) lb_subsm_head(L, lane); }
    LSYNC();
    if (lane == 0) *(volatile ldsi*)&ISC(I_HDONE) = c;
    LSYNC();
  }
}

// RestartGroup::advance (pcabo_api.hip): step until the group needs f, g at x (I_TASK == FG on return) or has stopped (I_ACTIVE == 0)
__device__ void lb_advance(const LbLds L, int maxiter, int lane) {
  while (ISR(I_ACTIVE)) {
    const int task = lb_step(L, lane);
    if (task == LBFGSB_FG) {
      if (ISR(I_HAVE_CACHE)) {
        bool diff = false;
        for (int i = lane; i < L.n; i += 64) diff = diff || (__double_as_longlong(L.x()[i]) != __double_as_longlong(L.xc()[i]));
        if (!__ballot(diff)) {
          for (int i = lane; i < L.n; i += 64) L.g()[i] = L.gc()[i];
          st0(&SC(S_F), SR(S_FC), lane);
          LSYNC();
          continue;
        }
      }
      return;
    }
    if (task == LBFGSB_NEW_X) {
      const int niter = ISR(I_NITER) + 1;
      LSYNC();
      if (lane == 0) {
        ISC(I_NITER) = niter;
        if (niter >= maxiter) ISC(I_TASK) = LBFGSB_STOP_ITER;
        else if (ISR(I_NFEV) > 15000) ISC(I_TASK) = LBFGSB_STOP_FUN;
      }
      LSYNC();
      continue;
    }
    sti0(&ISC(I_ACTIVE), 0, lane);
    LSYNC();
  }
}

// =====================================================================================================================
// Evaluation: value and gradient of the acquisition at the group's nq points (all LB_THREADS threads)
// =====================================================================================================================
// (the work plan of the two triangular passes: lb_plan.h)
__host__ __device__ inline void lb_build_plan(const LbLds L, int n, int S) { lb_build_plan_t(L.plan(), n, S); }

struct LbEval {
  gcd *ZnT, *R, *RT, *alpha, *nlo, *nhi;
  int n, k, NP, ld, S;
  double best_f, ym, ysd, inv_ls;
  int maximize, acq, kernel;
};

__device__ inline void lb_log_ei_helper(double u, double* h, double* dh) {
  const double inv_sqrt2 = 0.7071067811865476, inv_sqrt_2pi = 0.3989422804014327, log2pi = 1.8378770664093453;
  if (u > -1.0) {
    const double phi = inv_sqrt_2pi * exp(-0.5 * u * u);
    const double Phi = 0.5 * erfc(-inv_sqrt2 * u);
    const double ei = phi + u * Phi;
    *h = log(ei);
    *dh = Phi / ei;
    return;
  }
  const double log_phi = -0.5 * (u * u + log2pi);
  if (u > -1e6) {
    const double ex = erfcx(-inv_sqrt2 * u);
    const double E = (ex * fabs(u)) * 1.2533141373155003;
    *h = log_phi + log1p(-E);
    const double dw = (u + 0.7978845608028654 / ex) + 1.0 / u;
    *dh = -u - dw * E / (1.0 - E);
  } else {
    *h = log_phi - 2.0 * log(fabs(u));
    *dh = -u - 2.0 / u;
  }
}
// scalar chain of one query (kernels_acq.hip: acq_scalar_core): value and the two coefficients of the gradient's chain rule
__device__ inline void lb_scalar_core(double vv, double mus, const LbEval& E, double* value, double* c_mu, double* c_sg) {
  const double mu = E.ym + E.ysd * mus;
  double var = (1.0 - vv) * (E.ysd * E.ysd);
  bool clamped = false;
  if (!(var >= 1e-10)) { var = 1e-10; clamped = true; }
  if (var < 1e-12) { var = 1e-12; clamped = true; }
  const double sigma = sqrt(var);
  double u = (mu - E.best_f) / sigma;
  const double sgn = E.maximize ? 1.0 : -1.0;
  u *= sgn;
  double val, dv_du, dv_dsig;
  if (E.acq == 0) {
    double h, dh;
    lb_log_ei_helper(u, &h, &dh);
    val = h + log(sigma);
    dv_du = dh;
    dv_dsig = 1.0 / sigma;
  } else {
    val = 0.5 * erfc(-0.7071067811865476 * u);
    dv_du = 0.3989422804014327 * exp(-0.5 * u * u);
    dv_dsig = 0.0;
  }
  *value = val;
  *c_mu = dv_du * sgn * E.ysd / sigma;
  *c_sg = clamped ? 0.0 : (dv_dsig - dv_du * u / sigma) * (-(E.ysd * E.ysd) / sigma);
}

// x in L.x() (nq * k) -> L.vals()[q] (acquisition values) and, with want_grad, L.g()[q * k + c] = -d value_q / d x_qc
// (the gradient of the minimised objective -sum_q value_q).  Ends with a work-group barrier.
// Every global load of a loop trip is issued before the first use: one CU has to pull ~0.8 MB per triangular pass through ~1 us
// of L2 / Infinity Cache latency.
// element `lane8 / 8` of a row whose base is wave-uniform: (scalar base) + (32-bit lane offset) - global_load with an SGPR base, no
// vector address arithmetic per load
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(1))) char gcc_;
#else
typedef const char gcc_;
#endif
__device__ inline double ld_row(gcd* row, unsigned lane8) { return *(gcd*)((gcc_*)row + lane8); }
// the same with 16 bytes per lane (two neighbouring elements of the row): a wave's load instruction costs the CU's address path
// ~11 ns whether a lane takes 8 or 16 bytes (profiles/r03/device_lbfgsb_pass_experiments.txt), so the triangular passes move two
// rows (columns) of the matrix per instruction - lanes 0 .. 31 the even one, lanes 32 .. 63 the odd one
typedef double lb_d2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(1))) lb_d2 gcd2;
#else
typedef const lb_d2 gcd2;
#endif
__device__ inline lb_d2 ld_row2(gcd* row, unsigned off) { return *(gcd2*)((gcc_*)row + off); }
#define LB_UB2 8            // 16-byte loads in flight per thread and trip (a trip covers 2 LB_UB2 rows or columns)
// The evaluation's arguments live in LDS (written once by the kernel): a by-value struct of this size would travel through scratch
// memory at every call, a round trip to memory before the first useful instruction.
__device__ inline unsigned long long lds_u64(const ldsd* p) {
  const double v = *p;
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(v)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(v));
}
__device__ inline void lb_store_eval_args(const LbLds L, const LbEval& E) {       // one thread
  ldsd* e = L.ev();
  e[0] = __longlong_as_double((long long)(unsigned long long)E.ZnT); e[1] = __longlong_as_double((long long)(unsigned long long)E.R);
  e[2] = __longlong_as_double((long long)(unsigned long long)E.RT); e[3] = __longlong_as_double((long long)(unsigned long long)E.alpha);
  e[4] = __hiloint2double(E.n, E.k); e[5] = __hiloint2double(E.NP, E.ld); e[6] = __hiloint2double(0, E.S);
  e[7] = __hiloint2double(E.maximize, E.acq); e[8] = __hiloint2double(E.kernel, 0);
  e[9] = E.best_f; e[10] = E.ym; e[11] = E.ysd; e[12] = E.inv_ls;
}
__device__ __noinline__ void lb_eval(const LbLds L_, int nq, bool want_grad) {
  // (arguments of a function that is not inlined arrive in vector registers: made wave-uniform again here, so that loop bounds,
  // row bases and LDS addresses of the broadcasts are scalar)
  const int tid = threadIdx.x, lane = tid & 63, w = uni(tid >> 6);
  LbLds L; L.base = (ldsd*)(__UINTPTR_TYPE__)(unsigned)uni((int)(unsigned)(__UINTPTR_TYPE__)L_.base); L.n = uni(L_.n); L.NP = uni(L_.NP);
  LbEval E;
  {
    const ldsd* e = L.ev();
    E.ZnT = (gcd*)lds_u64(e); E.R = (gcd*)lds_u64(e + 1); E.RT = (gcd*)lds_u64(e + 2); E.alpha = (gcd*)lds_u64(e + 3);
    E.nlo = nullptr; E.nhi = nullptr;
    unsigned long long v;
    v = lds_u64(e + 4); E.n = (int)(v >> 32); E.k = (int)(unsigned)v;
    v = lds_u64(e + 5); E.NP = (int)(v >> 32); E.ld = (int)(unsigned)v;
    v = lds_u64(e + 6); E.S = (int)(unsigned)v;
    v = lds_u64(e + 7); E.maximize = (int)(v >> 32); E.acq = (int)(unsigned)v;
    v = lds_u64(e + 8); E.kernel = (int)(v >> 32);
    E.best_f = e[9]; E.ym = e[10]; E.ysd = e[11]; E.inv_ls = uni(e[12]);
  }
  nq = uni(nq);
  const int n = E.n, k = E.k, NP = E.NP, ld = E.ld, S = E.S;
  const int XS = LB_MAXK + 2;
  LBT_BEGIN();
  for (int idx = tid; idx < LB_GQ * k; idx += LB_THREADS) {
    const int q = idx / k, c = idx - q * k, qq = q < nq ? q : 0;
    const double lo = L.nlo()[c], hi = L.nhi()[c];
    L.xn()[q * XS + c] = (L.x()[qq * k + c] - lo) / (hi - lo);
  }
  __syncthreads();
  LBT_NEXT(8);
  // ---- kernel vectors: a thread per training point (the threads of pass 2's first group), all queries per pass over ZnT
  double cf[LB_GQ];
  // waves 0 .. S-1 own the points of block w (kernel vectors, |v|^2 / mu, the weights u); the triangular passes follow the plan
  const bool owner = w < S;
  const int cbu = w;
  const int half = lane >> 5, l32 = lane & 31;
  const int jmine = 64 * cbu + 2 * l32 + half;    // the point this thread owns (owner waves): lanes 0 .. 31 the even ones
  const unsigned off2 = (unsigned)half * (unsigned)ld * 8u + (unsigned)l32 * 16u;   // (row + half, elements 2 l32 and 2 l32 + 1)
  if (owner) {
    // 16-byte loads here too: a thread takes points 2 l32 and 2 l32 + 1 of the wave's block and every second component (its
    // half-wave's parity); (even components) + (odd components) meet in the lane that keeps the point
    const int j = jmine;
    double s0[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0}, s1[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < LB_GQ; ++q) cf[q] = 0.0;
    double ksv[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0};
    constexpr int CB = 6;
    gcd* zp = E.ZnT + 64 * cbu;
    int c0 = 0;
    for (; c0 + 2 * CB <= k; c0 += 2 * CB) {              // whole chunks: CB loads in flight, no guards
      lb_d2 z[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) z[u] = ld_row2(zp + (size_t)(c0 + 2 * u) * ld, off2);
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const ldsd* xp = L.xn() + c0 + 2 * u + half;
#pragma unroll
        for (int q = 0; q < LB_GQ; ++q) {
          const double xv = xp[q * XS], d0 = xv - z[u].x, d1 = xv - z[u].y;
          s0[q] = fma(d0, d0, s0[q]); s1[q] = fma(d1, d1, s1[q]);
        }
      }
    }
    if (c0 < k) {
      lb_d2 z[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) { const int c = c0 + 2 * u + half, cl = c < k ? c : k - 1; z[u] = ld_row2(zp + (size_t)cl * ld, (unsigned)l32 * 16u); }
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const int c = c0 + 2 * u + half;
        if (c < k) {
          const ldsd* xp = L.xn() + c;
#pragma unroll
          for (int q = 0; q < LB_GQ; ++q) {
            const double xv = xp[q * XS], d0 = xv - z[u].x, d1 = xv - z[u].y;
            s0[q] = fma(d0, d0, s0[q]); s1[q] = fma(d1, d1, s1[q]);
          }
        }
      }
    }
    double sq[LB_GQ];
#pragma unroll
    for (int q = 0; q < LB_GQ; ++q) {
      const double got = __shfl_xor(half ? s0[q] : s1[q], 32, 64);
      sq[q] = half ? got + s1[q] : s0[q] + got;
    }
    if (j < n) {
      const double s5 = 2.23606797749979, il2 = E.inv_ls * E.inv_ls;
#pragma unroll
      for (int q = 0; q < LB_GQ; ++q) {
        const double sqq = sq[q] * il2;
        if (E.kernel == 1) {
          ksv[q] = exp(-0.5 * sqq);
          cf[q] = -ksv[q] * il2;
        } else {
          const double dist = sqrt(fmax(sqq, 1e-30));
          const double e = exp(-s5 * dist);
          ksv[q] = ((s5 * dist + 1.0) + (5.0 / 3.0) * (dist * dist)) * e;
          cf[q] = -(5.0 / 3.0) * (1.0 + s5 * dist) * e * il2;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < LB_GQ; ++q) L.ks()[j * LB_QS + q] = ksv[q];
  }
  __syncthreads();
  LBT_NEXT(9);
  // ---- pass 1: v_q[i] = sum_j RT[j][i] ks_q[j].  A thread holds rows 2 l32 and 2 l32 + 1 of the segment's slab and takes every second
  // column of the segment (its half-wave's parity); the two half-waves' sums meet at the end: (even columns) + (odd columns)
  {
    const ldsi* pl = L.plan() + w * LB_PLAN_WAVE;
    const int nseg = uni(pl[0]);
    for (int g = 0; g < nseg; ++g) {
      const int unit = uni(pl[1 + 4 * g]), J0 = uni(pl[2 + 4 * g]), J1 = uni(pl[3 + 4 * g]), dest = uni(pl[4 + 4 * g]);
      double p0[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0}, p1[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0};
      gcd* rp = E.RT + 64 * unit;            // (wave-uniform base + the lane's 32-bit offset: no address arithmetic per load)
      int j = J0;
      for (; j + 2 * LB_UB2 <= J1; j += 2 * LB_UB2) {     // whole trips: LB_UB2 loads in flight, no guards (the other waves cover the wait)
        lb_d2 rc[LB_UB2];
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) rc[u] = ld_row2(rp + (size_t)(j + 2 * u) * ld, off2);
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) {
          const ldsd* kp = L.ks() + (j + 2 * u + half) * LB_QS;
#pragma unroll
          for (int q = 0; q < LB_GQ; ++q) { const double kq = kp[q]; p0[q] = fma(rc[u].x, kq, p0[q]); p1[q] = fma(rc[u].y, kq, p1[q]); }
        }
      }
      if (j < J1) {                                       // the ragged end: one trip, every load issued before the first use
        lb_d2 rc[LB_UB2];
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) {
          const int jc = j + 2 * u + half, jl = jc < J1 ? jc : J1 - 1;
          rc[u] = ld_row2(rp + (size_t)jl * ld, (unsigned)l32 * 16u);
        }
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) {
          const int jc = j + 2 * u + half;
          if (jc < J1) {
            const ldsd* kp = L.ks() + jc * LB_QS;
#pragma unroll
            for (int q = 0; q < LB_GQ; ++q) { const double kq = kp[q]; p0[q] = fma(rc[u].x, kq, p0[q]); p1[q] = fma(rc[u].y, kq, p1[q]); }
          }
        }
      }
      // lanes 0 .. 31 finish row 2 l32 (their own even-column sum + the partner's odd-column sum), lanes 32 .. 63 row 2 l32 + 1
      const int rin = 2 * l32 + half;
      ldsd* dst = dest < 0 ? L.vb() + (64 * unit + rin) * LB_QS : L.slots() + ((size_t)dest * 64 + rin) * LB_GQ;
#pragma unroll
      for (int q = 0; q < LB_GQ; ++q) {
        const double got = __shfl_xor(half ? p0[q] : p1[q], 32, 64);
        dst[q] = half ? got + p1[q] : p0[q] + got;
      }
    }
  }
  __syncthreads();
  LBT_NEXT(10);
  if (owner) {
    const int i = 64 * w + lane;
    const ldsi* pu = L.plan() + LB_WAVES * LB_PLAN_WAVE + 2 * w;
    const int first = uni(pu[0]), extra = uni(pu[1]);
    double v[LB_GQ];
#pragma unroll
    for (int q = 0; q < LB_GQ; ++q) v[q] = L.vb()[i * LB_QS + q];
    for (int t = 0; t < extra; ++t)
#pragma unroll
      for (int q = 0; q < LB_GQ; ++q) v[q] += L.slots()[((size_t)(first + t) * 64 + lane) * LB_GQ + q];
    const double ai = i < n ? E.alpha[i] : 0.0;
#pragma unroll
    for (int q = 0; q < LB_GQ; ++q) {
      L.vb()[i * LB_QS + q] = v[q];
      const double vv = wave_sum(v[q] * v[q]);
      const double mu = wave_sum(ai * L.ks()[i * LB_QS + q]);
      if (lane == 0) { L.red()[w * 10 + q] = vv; L.red()[w * 10 + LB_GQ + q] = mu; }
    }
  }
  __syncthreads();
  LBT_NEXT(11);
  // ---- scalar chains (lanes 0 .. 4 of wave 0) beside pass 2 of the other waves
  if (tid < LB_GQ) {
    double vv = 0.0, mus = 0.0;
    for (int s = 0; s < S; ++s) { vv += L.red()[s * 10 + tid]; mus += L.red()[s * 10 + LB_GQ + tid]; }
    double value, cmu, csg;
    lb_scalar_core(vv, mus, E, &value, &cmu, &csg);
    L.vals()[tid] = value; L.cq()[2 * tid] = cmu; L.cq()[2 * tid + 1] = csg;
  }
  if (!want_grad) { __syncthreads(); return; }
  // ---- pass 2: w_q[j] = sum_{i >= j} R[i][j] v_q[i].  A thread holds columns 2 l32 and 2 l32 + 1 of the segment's block and takes
  // every second row of the segment; the half-waves meet as in pass 1: (even rows) + (odd rows).  The block's first segment
  // leaves its sums where the kernel vectors were (they are no longer needed), the others in the partial slots
  {
    const ldsi* pl = L.plan() + LB_PLAN_PASS + w * LB_PLAN_WAVE;
    const int nseg = uni(pl[0]);
    for (int g = 0; g < nseg; ++g) {
      const int unit = uni(pl[1 + 4 * g]), I0 = uni(pl[2 + 4 * g]), I1 = uni(pl[3 + 4 * g]), dest = uni(pl[4 + 4 * g]);
      double p0[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0}, p1[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0};
      gcd* rp = E.R + 64 * unit;
      int i = I0;
      for (; i + 2 * LB_UB2 <= I1; i += 2 * LB_UB2) {
        lb_d2 rc[LB_UB2];
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) rc[u] = ld_row2(rp + (size_t)(i + 2 * u) * ld, off2);
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) {
          const ldsd* vp = L.vb() + (i + 2 * u + half) * LB_QS;
#pragma unroll
          for (int q = 0; q < LB_GQ; ++q) { const double vq = vp[q]; p0[q] = fma(rc[u].x, vq, p0[q]); p1[q] = fma(rc[u].y, vq, p1[q]); }
        }
      }
      if (i < I1) {
        lb_d2 rc[LB_UB2];
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) {
          const int ic = i + 2 * u + half, il = ic < I1 ? ic : I1 - 1;
          rc[u] = ld_row2(rp + (size_t)il * ld, (unsigned)l32 * 16u);
        }
#pragma unroll
        for (int u = 0; u < LB_UB2; ++u) {
          const int ic = i + 2 * u + half;
          if (ic < I1) {
            const ldsd* vp = L.vb() + ic * LB_QS;
#pragma unroll
            for (int q = 0; q < LB_GQ; ++q) { const double vq = vp[q]; p0[q] = fma(rc[u].x, vq, p0[q]); p1[q] = fma(rc[u].y, vq, p1[q]); }
          }
        }
      }
      const int cin = 2 * l32 + half;
      ldsd* dst = dest < 0 ? L.ks() + (64 * unit + cin) * LB_QS : L.slots() + ((size_t)dest * 64 + cin) * LB_GQ;
#pragma unroll
      for (int q = 0; q < LB_GQ; ++q) {
        const double got = __shfl_xor(half ? p0[q] : p1[q], 32, 64);
        dst[q] = half ? got + p1[q] : p0[q] + got;
      }
    }
  }
  // the rows of ZnT the gradient contraction of this wave needs (components w, w + 16, w + 32) leave now: they arrive while
  // the barrier and the u phase pass
  constexpr int CW = (LB_MAXK + 15) / 16, NB = LB_MAXNP / 128;          // (a lane takes points 128 bq + 2 lane and + 1)
  lb_d2 z[CW][NB];
#pragma unroll
  for (int ci = 0; ci < CW; ++ci) {
    const int c = w + 16 * ci, cc = c < k ? c : k - 1;
#pragma unroll
    for (int bq = 0; bq < NB; ++bq) {
      const int j = 2 * lane + 128 * bq;
      z[ci][bq] = ld_row2(E.ZnT + (size_t)cc * ld, (unsigned)(j < NP ? j : 0) * 8u);
    }
  }
  __syncthreads();
  LBT_NEXT(12);
  if (owner) {
    const int j = jmine;
    const ldsi* pu = L.plan() + LB_PLAN_PASS + LB_WAVES * LB_PLAN_WAVE + 2 * w;
    const int first = uni(pu[0]), extra = uni(pu[1]);
    double wv[LB_GQ];
#pragma unroll
    for (int q = 0; q < LB_GQ; ++q) wv[q] = L.ks()[j * LB_QS + q];
    for (int t = 0; t < extra; ++t)
#pragma unroll
      for (int q = 0; q < LB_GQ; ++q) wv[q] += L.slots()[((size_t)(first + t) * 64 + (j - 64 * w)) * LB_GQ + q];
    const double aj = j < n ? E.alpha[j] : 0.0;
    // u_q[j] = c_mu (alpha_j cf) + c_sg (w cf): the point's weight in the contraction with (xn_c - zn_cj); stored [q][j] where v
    // was (pass 2 is over)
#pragma unroll
    for (int q = 0; q < LB_GQ; ++q) L.vb()[q * NP + j] = fma(L.cq()[2 * q], aj * cf[q], L.cq()[2 * q + 1] * (wv[q] * cf[q]));
  }
  __syncthreads();
  LBT_NEXT(13);
  // ---- gradient: wave per component (c = w, w + 16, w + 32), lanes over pairs of points; all rows of ZnT a wave needs were loaded above
  {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(3))) lb_d2 ldsd2;
#else
    typedef const lb_d2 ldsd2;
#endif
#pragma unroll
    for (int ci = 0; ci < CW; ++ci) {
      const int c = w + 16 * ci;
      if (c < k) {
        double acc[LB_GQ] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int bq = 0; bq < NB; ++bq) {
          const int j = 2 * lane + 128 * bq;
          if (128 * bq < n) {
            const double z0 = j < n ? z[ci][bq].x : 0.0, z1 = j + 1 < n ? z[ci][bq].y : 0.0;
#pragma unroll
            for (int q = 0; q < LB_GQ; ++q) {
              const lb_d2 uq = j < NP ? *(ldsd2*)(L.vb() + q * NP + j) : lb_d2{0.0, 0.0};
              const double xv = L.xn()[q * XS + c];
              acc[q] = fma(j < n ? uq.x : 0.0, xv - z0, acc[q]);
              acc[q] = fma(j + 1 < n ? uq.y : 0.0, xv - z1, acc[q]);
            }
          }
        }
        const double inv = L.nhi()[c] - L.nlo()[c];
        double mine = 0.0;                           // lane q keeps query q's sum: ONE division for the five
#pragma unroll
        for (int q = 0; q < LB_GQ; ++q) { const double s = wave_sum(acc[q]); if (lane == q) mine = s; }
        if (lane < nq) L.g()[lane * k + c] = -(mine / inv);
      }
    }
  }
  __syncthreads();
  LBT_NEXT(14);
}

// =====================================================================================================================
// The kernel.  grid = entries of the table (run << 16 | first query << 8 | count), block = LB_THREADS.
// mode 1: optimise from the initial conditions; mode 0: one evaluation at the given points (value and gradient to the run's
// dVal / dGrad: the host-paced twin of the tests and the end-point values).
// Per run (stride zs): Xq = [num_restarts x k initial points | lower[k] | upper[k]], out_x = candidates (num_restarts x k),
// out_v = [values (num_restarts) ... | 64 + 8 gi: niter, nfev, warnflag, task, status, evaluations, ties].
// =====================================================================================================================
__global__ __launch_bounds__(LB_THREADS) void k_lbfgsb_group(
    const unsigned* __restrict__ table, int mode, int num_restarts, int maxiter, int n, int NP, int ld,
    const double* __restrict__ Xq, const double* __restrict__ ZnT, const double* __restrict__ R, const double* __restrict__ RT,
    const double* __restrict__ alpha, const double* __restrict__ bounds4, const double* __restrict__ ystats,
    const double* __restrict__ bestf, const int* __restrict__ k_dev, double inv_ls, int maximize, int acq, int kernel,
    double* __restrict__ out_x, double* __restrict__ out_v, size_t zs) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  const unsigned ent = table[blockIdx.x];
  const unsigned run_ = ent >> 16;
  const int q0 = (int)((ent >> 8) & 0xffu), nq = (int)(ent & 0xffu), gi = q0 / LB_GQ;
  Xq = zrun(Xq, zs, run_); ZnT = zrun(ZnT, zs, run_); R = zrun(R, zs, run_); RT = zrun(RT, zs, run_);
  alpha = zrun(alpha, zs, run_); bounds4 = zrun(bounds4, zs, run_); ystats = zrun(ystats, zs, run_);
  bestf = zrun(bestf, zs, run_); k_dev = zrun(k_dev, zs, run_); out_x = zrun(out_x, zs, run_); out_v = zrun(out_v, zs, run_);
  const int k = *k_dev;
  const int tid = threadIdx.x, lane = tid & 63, w = uni(tid >> 6);
  const int nv = nq * k;
  LbLds L;
  L.base = (ldsd*)s_dyn; L.n = nv; L.NP = NP;
  LbEval E;
  E.ZnT = (gcd*)ZnT; E.R = (gcd*)R; E.RT = (gcd*)RT; E.alpha = (gcd*)alpha; E.nlo = (gcd*)bounds4; E.nhi = (gcd*)(bounds4 + PCABO_MAXD);
  E.n = n; E.k = k; E.NP = NP; E.ld = ld; E.S = NP / 64;
  E.best_f = *bestf; E.ym = ystats[0]; E.ysd = ystats[1]; E.inv_ls = inv_ls; E.maximize = maximize; E.acq = acq; E.kernel = kernel;
  // ---- initial state
  for (int i = tid; i < nv; i += LB_THREADS) {
    const int q = i / k, c = i - q * k;
    const double l = Xq[(size_t)num_restarts * k + c], h = Xq[(size_t)num_restarts * k + k + c], v = Xq[(size_t)(q0 + q) * k + c];
    L.lo()[i] = l; L.hi()[i] = h;
    L.x()[i] = mode == 1 ? (v < l ? l : (v > h ? h : v)) : v;
    L.g()[i] = 0.0;
    L.iwhere()[i] = (h - l <= 0.0) ? 3 : 0;
  }
  for (int i = tid; i < OFF_X; i += LB_THREADS) L.ws()[i] = 0.0;                     // (ws and wy are adjacent)
  for (int i = tid; i < LB_M * LB_M; i += LB_THREADS) { L.sy()[i] = 0.0; L.ss()[i] = 0.0; L.wt()[i] = 0.0; }
  for (int i = tid; i < 4 * LB_M * LB_M; i += LB_THREADS) { L.wn()[i] = 0.0; L.wn1()[i] = 0.0; }
  if (tid < 8 * LB_M) L.wa()[tid] = 0.0;
  if (tid < k) { L.nlo()[tid] = bounds4[tid]; L.nhi()[tid] = bounds4[PCABO_MAXD + tid]; }
  if (tid < S_COUNT) L.sc()[tid] = 0.0;
  if (tid < I_COUNT) L.isc()[tid] = 0;
  __syncthreads();
  if (tid == 0) { SC(S_THETA) = 1.0; ISC(I_NFREE) = nv; ISC(I_ACTIVE) = 1; ISC(I_TASK) = LBFGSB_START; lb_store_eval_args(L, E); }
  if (tid == 64) lb_build_plan(L, n, NP / 64);        // (a thread of another wave: beside the line above)
  __syncthreads();
  if (mode == 0) {
    lb_eval(L, nq, true);
    if (tid < nq) out_v[q0 + tid] = L.vals()[tid];
    for (int i = tid; i < nv; i += LB_THREADS) out_x[(size_t)q0 * k + i] = -L.g()[i];      // the acquisition's own gradient
    return;
  }
  // ---- the optimisation
  int helper_seen = 0;                                  // (wave 1: the last command it has served)
  for (int guard = 0; guard < LB_MAXEVAL; ++guard) {
    LBT_BEGIN();
    if (w == 0) { lb_advance(L, maxiter, lane); lb_help_post(L, LB_OP_END, lane); }
    else if (w == 1) lb_helper(L, lane, helper_seen);
    __syncthreads();
    LBT_NEXT(16);
    if (!ISR(I_ACTIVE)) break;
    lb_eval(L, nq, true);
    LBT_NEXT(17);
    // RestartGroup::absorb: f = -(sum of the values, in order), NaN check of the gradient, cache
    if (w == 0) {
      bool nan = false;
      for (int i = lane; i < nv; i += 64) { const double gv = L.g()[i]; nan = nan || (gv != gv); }
      if (__ballot(nan)) {
        if (lane == 0) { ISC(I_STATUS) = PCABO_ERR_NAN; ISC(I_ACTIVE) = 0; }
      } else {
        double fs = 0.0;
        for (int q = 0; q < nq; ++q) fs += L.vals()[q];
        for (int i = lane; i < nv; i += 64) { L.xc()[i] = L.x()[i]; L.gc()[i] = L.g()[i]; }
        if (lane < nq) L.vc()[lane] = L.vals()[lane];
        if (lane == 0) { SC(S_F) = -fs; SC(S_FC) = -fs; ISC(I_NFEV) += 1; ISC(I_HAVE_CACHE) = 1; ISC(I_EVALS) += 1; }
      }
    }
    __syncthreads();
    LBT_NEXT(46);                                    // absorb: NaN check, f, cache copies, barrier
    if (!ISR(I_ACTIVE)) break;
  }
  __syncthreads();
  // ---- end points (clamped), their values: the last evaluation's if that was the point, one more evaluation otherwise
  bool same = ISR(I_HAVE_CACHE) != 0;
  for (int i = tid; i < nv; i += LB_THREADS) {
    double v = L.x()[i];
    v = v < L.lo()[i] ? L.lo()[i] : (v > L.hi()[i] ? L.hi()[i] : v);
    L.x()[i] = v;
    if (__double_as_longlong(v) != __double_as_longlong(L.xc()[i])) same = false;
  }
  const int all_same = __syncthreads_and(same ? 1 : 0);
  const int status = ISR(I_STATUS);
  if (status == 0) {
    if (!all_same) lb_eval(L, nq, false);
    else { if (tid < nq) L.vals()[tid] = L.vc()[tid]; __syncthreads(); }
    if (tid < nq) out_v[q0 + tid] = L.vals()[tid];
  }
  for (int i = tid; i < nv; i += LB_THREADS) out_x[(size_t)q0 * k + i] = L.x()[i];
  if (tid == 0) {
    const int task = ISR(I_TASK);
    const int wf = (task == LBFGSB_CONV_PG || task == LBFGSB_CONV_F) ? 0 : ((task == LBFGSB_STOP_ITER || task == LBFGSB_STOP_FUN) ? 1 : 2);
    double* o = out_v + 64 + 8 * gi;
    o[0] = ISR(I_NITER); o[1] = ISR(I_NFEV); o[2] = wf; o[3] = task; o[4] = status; o[5] = ISR(I_EVALS); o[6] = ISR(I_TIES);
    o[7] = ISR(I_ACTIVE) ? 1.0 : 0.0;                 // 1: stopped by the evaluation cap (cannot happen within the host's limits)
  }
}

// ---- RT = transposed root inverse, clean (zeros above the diagonal and beyond n) --------------------------------------
__global__ __launch_bounds__(256) void k_rt_build(const double* __restrict__ R, int n, int ld, double* __restrict__ RT, size_t zs) {
  R = zrun(R, zs, blockIdx.z); RT = zrun(RT, zs, blockIdx.z);
  __shared__ double tile[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;        // source tile: rows 64 bi.., columns 64 bj..
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if (bi >= bj) {
    for (int r = ty; r < 64; r += 4) {
      const int i = 64 * bi + r, j = 64 * bj + tx;
      tile[r][tx] = (i >= j && i < n && j < n) ? R[(size_t)i * ld + j] : 0.0;
    }
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int j = 64 * bj + r, i = 64 * bi + tx;     // RT[j][i] = R[i][j]
    RT[(size_t)j * ld + i] = bi >= bj ? tile[tx][r] : 0.0;
  }
}

void launch_rt_build(hipStream_t s, const double* R, int n, int NP, int ld, double* RT, ZB zb) {
  hipLaunchKernelGGL(k_rt_build, dim3(NP / 64, NP / 64, zb.B), dim3(256), 0, s, R, n, ld, RT, zb.zs);
}

// The passes' work plan as the kernel builds it, for the host-side test of its invariants (tests/test_lbfgsb_plan.py: every
// column / row covered once, at most two segments per wave, partial slots in ascending order and within the LDS that
// launch_lbfgsb_group asks for).  out: LB_PLAN_INTS ints; sizes[0..3] = ints per pass, ints per wave, slots the launch reserves,
// dynamic LDS bytes of the launch.  Not part of the ABI in include/pcabo.h.
extern "C" int pcabo_debug_lbfgsb_plan(int n, int NP, int* out, int* sizes) {
  if (!out || !sizes || NP < 64 || NP > LB_MAXNP || (NP % 64) != 0 || n < 1 || n > NP || n <= NP - 64) return -1;
  std::vector<double> buf((size_t)OFF_KS, 0.0);
  LbLds L; L.base = (ldsd*)buf.data(); L.n = 0; L.NP = NP;
  lb_build_plan(L, n, NP / 64);
  const ldsi* p = L.plan();
  for (int i = 0; i < LB_PLAN_INTS; ++i) out[i] = p[i];
  sizes[0] = LB_PLAN_PASS; sizes[1] = LB_PLAN_WAVE; sizes[2] = lb_max_slots(NP); sizes[3] = (int)(lb_lds_doubles(NP) * sizeof(double));
  return 0;
}

extern "C" int pcabo_device_lbfgsb_limits(int* max_n, int* max_k, int* max_group) {
  if (max_n) *max_n = LB_MAXNP;
  if (max_k) *max_k = LB_MAXK;
  if (max_group) *max_group = LB_GQ;
  return 0;
}

bool lbfgsb_device_possible(int NP, int kmax, int batch_limit) {
  return NP <= LB_MAXNP && kmax <= LB_MAXK && batch_limit <= LB_GQ && NP >= 64;
}


int launch_lbfgsb_group(hipStream_t st, const unsigned* table, int entries, int mode, int num_restarts, int maxiter, int n, int NP,
                        int ld, const double* Xq, const double* ZnT, const double* R, const double* RT, const double* alpha,
                        const double* bounds4, const double* ystats, const double* bestf, const int* k_dev, double inv_ls,
                        int maximize, int acq, int kernel, double* out_x, double* out_v, size_t zs) {
  const size_t lds = lb_lds_doubles(NP) * sizeof(double);
  if (lds > 150 * 1024) return -1;
  if (lb_lds_doubles(NP) < (size_t)OFF_KS + 2 * LB_TILE) return -1;      // the step's two transposition tiles lie in the evaluation's arrays
  {
    static std::mutex attr_mu;
    static bool attr_done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!attr_done[dev]) {
      if (hipFuncSetAttribute((const void*)k_lbfgsb_group, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -1;
      attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL(k_lbfgsb_group, dim3(entries), dim3(LB_THREADS), lds, st, table, mode, num_restarts, maxiter, n, NP, ld, Xq,
                     ZnT, R, RT, alpha, bounds4, ystats, bestf, k_dev, inv_ls, maximize, acq, kernel, out_x, out_v, zs);
  return 0;
}
