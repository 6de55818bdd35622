// Batched acquisition value + analytic gradient on gfx950 (used by K, M, N).
//
// Replaces botorch's LogExpectedImprovement / ProbabilityOfImprovement forward + torch autograd
// backward as driven by optimize_acqf (/root/reference/Algorithms/BayesianOptimization/
// PCA_BO.py:199-203, 607-614).  Per query point x (reduced space, k dims), with the state left by
// the conditioning kernels (ZnT, R = L^-1, alpha, Standardize stats):
//   xn  = (x - lo)/(hi - lo)                               Normalize
//   ks_j = matern52(|xn - zn_j| / l)                       j = 0..n-1
//   mu_s = ks . alpha ;  v = R ks ;  var_s = 1 - |v|^2     (fast_pred_var: root-inverse cache)
//   mu = m_y + s_y mu_s ;  sigma = sqrt(clamp(s_y^2 var_s))
//   u = +-(mu - best_f)/sigma ;  logEI = h(u) + log sigma
// and the reverse-mode gradient  w = R^T v,  grad = sum_j (c_mu alpha_j + c_s w_j) dks_j/dx.
//
// Work decomposition (latency- and issue-bound: ~1 MFLOP per query at n = 450, 10-15 us per launch):
//   grid (S, q): work-group (s, q) owns SLAB rows of R (half from the top, half mirrored from the bottom -> balanced
//   triangular work) for query q.  It recomputes ks (n*k flops, cheaper than a launch boundary), forms its entries
//   of v, its contribution R_slab^T v_slab to w, and contracts that with dks/dx.  Because the gradient is linear in w,
//   partial gradients of different slabs simply add.
//   k_acq_fast<SLAB,NB>   NP = 64 NB <= 512 and k <= 40 (every headline shape): all loops static, ZnT and the R slab
//                  read once into registers, DPP reductions, scalar row bases (see the comment at the kernel).
//   k_acq_fused<SLAB>     any size: the same phases with run-time loops.
//   The S partial records of a query are combined INSIDE the launch by the last work-group to arrive at a per-query
//   ticket counter (split-K style hand-off: write-through (sc1) stores -> vmcnt(0) -> barrier -> relaxed ticket; the
//   last arriver does one agent-scope acquire, then plain loads).  Its four waves sum the partials in a fixed order
//   (deterministic), apply the scalar log-EI chain rule and write value/gradient to device memory and to pinned host
//   memory, followed per query by a sequence word the host polls.  One launch per L-BFGS-B evaluation, no second
//   kernel, no memcpy.  Query points arrive as kernel arguments when they fit (<= 3.5 KB) so that no work-group has
//   to read host memory over PCIe.  Batches of more than 32 queries use the launch boundary instead of tickets
//   (k_acq_combine) and, on the fast path, give each group 8 queries per load of its register tiles.
#include "pcabo_internal.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>

#define PSTRIDE (2 + 2 * PCABO_MAXD)   // doubles per (query, slab) partial record

// Phase stamps for tools/gpu_acq_phases.py (diagnostic build only: `make timing` -> libpcabo_timing.so).
#ifdef PCABO_ACQ_TIMING
__device__ unsigned long long g_acq_stamps[16];
#define STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0) g_acq_stamps[i] = wall_clock64(); } while (0)
#define STAMP_FIN(i) do { if (blockIdx.y == 0 && threadIdx.x == 0) g_acq_stamps[i] = wall_clock64(); } while (0)
extern "C" int pcabo_debug_acq_stamps(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_acq_stamps), sizeof(g_acq_stamps)) == hipSuccess ? 0 : -3;
}
#else
#define STAMP(i)
#define STAMP_FIN(i)
#endif

// Write-through store (global_store ... sc1): the partial records are handed to another work-group inside
// the launch; written this way they never sit dirty in this XCD's L2, so the hand-off needs no L2 write-back
// (agent-scope release) on the producer side - only the drain + ticket, and the consumer's acquire.
__device__ inline void st_wt(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rows of slab s: SLAB/2 from the top plus SLAB/2 mirrored from the bottom (balanced triangular work)
template <int SLAB>
__device__ inline int slab_row(int s, int m, int NP) {
  return m < SLAB / 2 ? (SLAB / 2) * s + m : NP - (SLAB / 2) * (s + 1) + (m - SLAB / 2);
}

// forward declaration: scalar part, defined below
__device__ void acq_finish_query(const double* base, int S, int k, int q, const double* bounds4, const double* ystats,
                                 const AcqParams& p, double* val, double* grad, double* host_val, double* host_grad,
                                 int lane);

// Sum of the gradient partials of slabs first, first + 3, ... (zero if first >= S) for component c (fixed order; shared by the in-launch
// finish, where waves 1-3 take first = 0, 1, 2, and by the large-batch combine pass -> identical bits on both paths).
__device__ inline void grad_partial_sum(const double* base, int S, int c, int first, double* gs_out, double* gm_out) {
  double gs = 0.0, gm = 0.0;
  int sl = first;
  for (; sl + 21 < S; sl += 24) {                 // eight slabs at a time: 16 loads in flight, one round trip
    double a[8], b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      a[t] = base[(size_t)(sl + 3 * t) * PSTRIDE + 2 + c];
      b[t] = base[(size_t)(sl + 3 * t) * PSTRIDE + 2 + PCABO_MAXD + c];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { gs += a[t]; gm += b[t]; }
  }
  {                                               // the rest (< 8 slabs), still issued together
    double a[8], b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int si = sl + 3 * t;
      const int sc = si < S ? si : 0;
      a[t] = base[(size_t)sc * PSTRIDE + 2 + c];
      b[t] = base[(size_t)sc * PSTRIDE + 2 + PCABO_MAXD + c];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { if (sl + 3 * t < S) { gs += a[t]; gm += b[t]; } }
  }
  *gs_out = gs;
  *gm_out = gm;
}

__device__ void acq_finish_scalar(const double* base, int S, int q, const double* ystats, const AcqParams& p, double* val,
                                  double* host_val, double* coef, int lane);
__device__ void acq_scalar_core(double vv, double mus, int q, double ym, double ysd, const AcqParams& p, double* val,
                                double* host_val, double* coef, int l);
// ---- in-launch combine: the last slab group of a query to arrive finishes it ---------------------------
template <int SLAB>
__device__ inline void acq_tail(int combine, double* s_v, double* partial, unsigned int* counters, int q, int S, int k,
                                const double* bounds4, const double* ystats, const AcqParams& prm, double* val,
                                double* grad, double* host_val, double* host_grad, HostMirror* hm,
                                unsigned long long seq, int tid, int w, int l) {
  if (!combine) return;          // large batches: a follow-up k_acq_combine launch reads the partials instead
  int* s_flag = reinterpret_cast<int*>(s_v + SLAB);
  STAMP(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  STAMP(7);
  if (tid == 0) {
    unsigned int t = __hip_atomic_fetch_add(&counters[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = ((t % (unsigned int)S) == (unsigned int)(S - 1));
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *s_flag = last;
  }
  __syncthreads();
  if (!*s_flag) return;
  STAMP_FIN(8);
  // The partial records come from other XCDs (write-through stores): every read below is a trip to memory.  The four
  // waves of the finishing group split the work so that those trips overlap: wave 0 reduces |v|^2 / mu_s and runs the
  // scalar log-EI chain; waves 1-3 meanwhile sum the gradient partials of every third slab (fixed order).
  const double* base = partial + (size_t)q * S * PSTRIDE;
  double* s_fin = s_v + SLAB + 2;                 // [3][2][PCABO_MAXD] partial sums, then 2 coefficients
  double* s_coef = s_fin + 6 * PCABO_MAXD;
  double rng_c = 1.0;
  if (w == 0) {
    if (prm.want_grad && l < k) rng_c = bounds4[PCABO_MAXD + l] - bounds4[l];     // in flight during the scalar chain
    acq_finish_scalar(base, S, q, ystats, prm, val, host_val, s_coef, l);
  } else if (prm.want_grad) {
    for (int c = l; c < k; c += 64) {
      double gs, gm;
      grad_partial_sum(base, S, c, w - 1, &gs, &gm);
      s_fin[(w - 1) * 2 * PCABO_MAXD + c] = gs;
      s_fin[(w - 1) * 2 * PCABO_MAXD + PCABO_MAXD + c] = gm;
    }
  }
  __syncthreads();
  STAMP_FIN(13);
  if (w == 0) {
    if (prm.want_grad) {
      const double c_mu = s_coef[0], c_sg = s_coef[1];
      for (int c = l; c < k; c += 64) {
        const double gs = (s_fin[c] + s_fin[2 * PCABO_MAXD + c]) + s_fin[4 * PCABO_MAXD + c];
        const double gm = (s_fin[PCABO_MAXD + c] + s_fin[3 * PCABO_MAXD + c]) + s_fin[5 * PCABO_MAXD + c];
        const double g = __fma_rn(c_mu, gm, c_sg * gs) / (c == l ? rng_c : bounds4[PCABO_MAXD + c] - bounds4[c]);
        grad[(size_t)q * k + c] = g;
        if (host_grad) host_grad[(size_t)q * k + c] = g;
      }
    }
    // publish: this query's sequence word follows its results with a system-scope release (one wave, so the
    // release store's drain covers every lane's host writes)
    if (hm && l == 0)
      __hip_atomic_store(const_cast<unsigned long long*>(&hm->qflag[q]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    STAMP_FIN(9);
#ifdef PCABO_ACQ_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP_FIN(10);
#endif
  }
}

// Batched launches: which run does this work-group serve, and (table mode) which of its queries?  Offsets every per-run
// operand; k and best_f of the run come from its own device words.
#define ACQ_BATCH_PROLOGUE()                                                                                     \
  unsigned run_ = blockIdx.z;                                                                                    \
  int qsel_ = -1;                                                                                                \
  if (ab.table) {                                                                                                \
    const unsigned e_ = reinterpret_cast<const unsigned*>(qa.x)[blockIdx.y];                                     \
    run_ = e_ >> 16; qsel_ = (int)(e_ & 0xffffu);                                                                \
  }                                                                                                              \
  if (ab.zs) {                                                                                                   \
    ZnT = zrun(ZnT, ab.zs, run_); R = zrun(R, ab.zs, run_); alpha = zrun(alpha, ab.zs, run_);                    \
    bounds4 = zrun(bounds4, ab.zs, run_); ystats = zrun(ystats, ab.zs, run_); partial = zrun(partial, ab.zs, run_); \
    counters = zrun(counters, ab.zs, run_); val = zrun(val, ab.zs, run_); grad = zrun(grad, ab.zs, run_);        \
    Xq = zrun(Xq, ab.xq_host ? ab.hzs : ab.zs, run_);                                                            \
    host_val = zrun(host_val, ab.hzs, run_); host_grad = zrun(host_grad, ab.hzs, run_); hm = zrun(hm, ab.hzs, run_); \
    if (ab.k_dev) k = *zrun(ab.k_dev, ab.zs, run_);                                                              \
    if (ab.bestf) prm.best_f = *zrun(ab.bestf, ab.zs, run_);                                                     \
  }                                                                                                              \
  (void)qsel_;

template <int SLAB>
__global__ __launch_bounds__(256) void k_acq_fused(
    QueryArgs qa, const double* __restrict__ Xq, int q_total, int n, int k, int NP, int ld,
    const double* __restrict__ ZnT, const double* __restrict__ R, const double* __restrict__ alpha,
    const double* __restrict__ bounds4, const double* __restrict__ ystats, AcqParams prm, double* partial,
    unsigned int* counters, double* __restrict__ val, double* __restrict__ grad,
    double* host_val, double* host_grad, HostMirror* hm, unsigned long long seq, int combine, int /*qb*/,
    MailPair* /*dev_mail*/, MailPair* /*part_pairs*/, AcqBatch ab) {
  ACQ_BATCH_PROLOGUE()
  const double inv_ls = prm.inv_ls;
  const int kernel = prm.kernel, want_grad = prm.want_grad;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* s_ks = s_dyn;              // NP
  double* s_cf = s_dyn + NP;         // NP  coef_j (dks_j/dxn = coef_j (xn - zn_j))
  double* s_tm = s_dyn + 2 * NP;     // NP  alpha_j coef_j restricted to this slab's rows
  double* s_xn = s_dyn + 3 * NP;     // PCABO_MAXD
  double* s_v = s_xn + PCABO_MAXD;   // SLAB
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int s = blockIdx.x, q = ab.table ? qsel_ : (int)blockIdx.y, S = gridDim.x;
  double* out = partial + ((size_t)q * S + s) * PSTRIDE;
  STAMP(0);

  if (tid < k) {
    double lo = bounds4[tid], hi = bounds4[PCABO_MAXD + tid];
    double xv = Xq ? Xq[(size_t)q * k + tid] : qa.x[q * k + tid];
    s_xn[tid] = (xv - lo) / (hi - lo);
  }
  __syncthreads();
  STAMP(1);
  // kernel vector and the radial derivative factor
  const double s5 = 2.23606797749979;
  for (int j = tid; j < NP; j += 256) {
    double ks = 0.0, cf = 0.0;
    if (j < n) {
      double sq = 0.0;
      for (int c = 0; c < k; ++c) { double dlt = s_xn[c] - ZnT[(size_t)c * ld + j]; sq += dlt * dlt; }
      sq *= inv_ls * inv_ls;
      if (kernel == 1) {
        ks = exp(-0.5 * sq);
        cf = -ks * inv_ls * inv_ls;
      } else {
        double dist = sqrt(fmax(sq, 1e-30));
        double e = exp(-s5 * dist);
        ks = ((s5 * dist + 1.0) + (5.0 / 3.0) * (dist * dist)) * e;
        cf = -(5.0 / 3.0) * (1.0 + s5 * dist) * e * inv_ls * inv_ls;
      }
    }
    s_ks[j] = ks;
    s_cf[j] = cf;
  }
  __syncthreads();
  STAMP(2);
  // v_i = R[i][0..i] . ks for the slab's 16 rows: each wave owns 4 rows and streams them together (4
  // independent load streams in flight), then 64-lane shuffle reductions
  {
    constexpr int RW = SLAB / 4;          // rows per wave
    int ri[RW];
    const double* Rr[RW];
    double acc[RW];
    int imax = 0;
#pragma unroll
    for (int u = 0; u < RW; ++u) {
      ri[u] = slab_row<SLAB>(s, w + 4 * u, NP); Rr[u] = R + (size_t)ri[u] * ld; acc[u] = 0.0; imax = max(imax, ri[u]);
    }
    for (int j = l; j <= imax; j += 64) {
      const double kj = s_ks[j];
#pragma unroll
      for (int u = 0; u < RW; ++u) acc[u] += Rr[u][j] * kj;      // R is exactly zero above its diagonal
    }
#pragma unroll
    for (int u = 0; u < RW; ++u) acc[u] = wave_sum(acc[u]);
    if (l == 0) {
#pragma unroll
      for (int u = 0; u < RW; ++u) s_v[w + 4 * u] = acc[u];
    }
  }
  __syncthreads();
  STAMP(3);
  // slab contributions to |v|^2 and to mu_s = alpha . ks: 16 lanes, one round trip
  if (w == 0) {
    double vv = 0.0, mu = 0.0;
    if (l < SLAB) {
      const int i = slab_row<SLAB>(s, l, NP);
      const double vi = s_v[l];
      vv = vi * vi;
      if (i < n) mu = alpha[i] * s_ks[i];
    }
    vv = wave_sum(vv);        // lanes >= SLAB hold zeros
    mu = wave_sum(mu);
    if (l == 0) { st_wt(out + 0, vv); st_wt(out + 1, mu); }
  }
  if (want_grad) {
  __syncthreads();   // thread 0 has finished reading s_ks before it is reused below
  STAMP(4);
  // w_j (slab part) = sum_{i in slab} R[i][j] v_i (R[i][j] = 0 for j > i); fold in coef_j.  The 16 row
  // segments are read column-wise (coalesced over j), all loads independent.
  for (int j = tid; j < NP; j += 256) {
    double wj = 0.0;
#pragma unroll
    for (int m = 0; m < SLAB; ++m) wj += R[(size_t)slab_row<SLAB>(s, m, NP) * ld + j] * s_v[m];
    const bool mine = ((j / (SLAB / 2)) == s) || (((NP - 1 - j) / (SLAB / 2)) == s);
    const double cf = s_cf[j];
    s_tm[j] = (mine && j < n) ? alpha[j] * cf : 0.0;
    s_ks[j] = wj * cf;                    // ks no longer needed: reuse as t_sigma
  }
  __syncthreads();
  STAMP(5);
  // contraction with (xn_c - zn_jc): a wave keeps 8 components in flight (c = c0 + w + 4u)
  for (int c0 = 0; c0 < k; c0 += 32) {
    double gs[8], gm[8], xc[8];
    const double* zrow[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + w + 4 * u;
      const int cc = c < k ? c : 0;
      gs[u] = 0.0; gm[u] = 0.0; xc[u] = s_xn[cc]; zrow[u] = ZnT + (size_t)cc * ld;
    }
    for (int j = l; j < n; j += 64) {
      const double ts = s_ks[j], tm = s_tm[j];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double dlt = xc[u] - zrow[u][j];
        gs[u] += ts * dlt;
        gm[u] += tm * dlt;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { gs[u] = wave_sum(gs[u]); gm[u] = wave_sum(gm[u]); }
    if (l == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + w + 4 * u;
        if (c < k) { st_wt(out + 2 + c, gs[u]); st_wt(out + 2 + PCABO_MAXD + c, gm[u]); }
      }
    }
  }
  }  // want_grad

  acq_tail<SLAB>(combine, s_v, partial, counters, q, S, k, bounds4, ystats, prm, val, grad, host_val, host_grad, hm, seq, tid, w, l);
}

// ---- resident mode: the dedicated finishing group of a query -----------------------------------------------------
// In resident mode the slab groups publish their partial records as (value, tag) pairs and go straight to the next
// round; group (S, q) of the grid does nothing but finish query q.  It polls the records themselves (the data arrive
// with their tags: no ticket, no drain, no acquire), and it starts the scalar log-EI chain as soon as the |v|^2 / mu_s
// pairs are there - about 1.7 us before the gradient pairs - so the chain overlaps with the rest of the round:
//   wave 0     |v|^2, mu_s of every slab (one lane per slab, S <= 64) -> sums -> scalar chain -> value, coefficients
//   wave 1-3   gradient pairs of slabs w-1, w+2, ... (items spread over the 64 lanes, staged in LDS), then lanes over
//              components add them in slab order: the same partial sums as grad_partial_sum
//   wave 0     gradient = (c_mu gm + c_sg gs) / range, results and sequence word to the host.
// Returns false on a timeout (the group then leaves the kernel).
#define FIN_STAGE_PER_WAVE (11 * 2 * 40)      // doubles: ceil(32 / 3) slabs x {gs, gm} x k <= 40
__device__ inline bool acq_server_finisher(const MailPair* rec, int S, int k, int q, unsigned long long seq,
                                           const double* ystats, const AcqParams& prm, double* val, double* grad,
                                           double* host_val, double* host_grad, HostMirror* hm, double rng_c,
                                           double* s_fin, double* s_coef, double* s_stage, int* s_srv,
                                           unsigned long long t0, int w, int l) {
  if (w == 0) {
    const double ym = ystats[0], ysd = ystats[1];
    const void* ptr[8];
    pcabo_u4 o[8];
    const MailPair* mine = rec + (size_t)(l < S ? l : 0) * PSTRIDE;
#pragma unroll
    for (int t = 0; t < 8; ++t) ptr[t] = mine + (t & 1);
    for (;;) {
      ld_pairs_sys8(ptr, o);
      const bool ok = pair_tag(o[0]) == seq && pair_tag(o[1]) == seq;
      if (__all(ok)) break;
      if (__any(wall_clock64() - t0 > PCABO_SERVER_TIMEOUT_TICKS)) { if (l == 0) *s_srv = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    double vv = l < S ? pair_value(o[0]) : 0.0, mus = l < S ? pair_value(o[1]) : 0.0;
    vv = wave_sum(vv);
    mus = wave_sum(mus);
    acq_scalar_core(vv, mus, q, ym, ysd, prm, val, host_val, s_coef, l);
  } else if (prm.want_grad) {
    const int first = w - 1;
    const int ns = first < S ? (S - first + 2) / 3 : 0;       // slabs first, first + 3, ...
    const int total = ns * 2 * k;
    double* stage = s_stage + (size_t)(w - 1) * FIN_STAGE_PER_WAVE;   // [slab][gs|gm][c]
    for (int i0 = 0; i0 < total; i0 += 512) {
      const void* ptr[8];
      int item[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int i = i0 + l + 64 * t;
        item[t] = i < total ? i : -1;
        const int ii = i < total ? i : 0;
        const int st = ii / (2 * k), rem = ii - st * 2 * k, which = rem >= k, c = rem - which * k;
        ptr[t] = rec + (size_t)(first + 3 * st) * PSTRIDE + 2 + which * PCABO_MAXD + c;
      }
      pcabo_u4 o[8];
      for (;;) {
        ld_pairs_sys8(ptr, o);
        bool ok = true;
#pragma unroll
        for (int t = 0; t < 8; ++t) ok = ok && (item[t] < 0 || pair_tag(o[t]) == seq);
        if (__all(ok)) break;
        if (__any(wall_clock64() - t0 > PCABO_SERVER_TIMEOUT_TICKS)) { if (l == 0) *s_srv = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) if (item[t] >= 0) stage[item[t]] = pair_value(o[t]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (l < k) {
      double gs = 0.0, gm = 0.0;
      for (int st = 0; st < ns; ++st) { gs += stage[st * 2 * k + l]; gm += stage[st * 2 * k + k + l]; }
      s_fin[(w - 1) * 2 * PCABO_MAXD + l] = gs;
      s_fin[(w - 1) * 2 * PCABO_MAXD + PCABO_MAXD + l] = gm;
    }
  }
  __syncthreads();
  if (*s_srv) return false;
  if (w == 0) {
    if (prm.want_grad && l < k) {
      const double c_mu = s_coef[0], c_sg = s_coef[1];
      const double gs = (s_fin[l] + s_fin[2 * PCABO_MAXD + l]) + s_fin[4 * PCABO_MAXD + l];
      const double gm = (s_fin[PCABO_MAXD + l] + s_fin[3 * PCABO_MAXD + l]) + s_fin[5 * PCABO_MAXD + l];
      const double g = __fma_rn(c_mu, gm, c_sg * gs) / rng_c;
      grad[(size_t)q * k + l] = g;
      if (host_grad) host_grad[(size_t)q * k + l] = g;
    }
    if (hm && l == 0)
      __hip_atomic_store(const_cast<unsigned long long*>(&hm->qflag[q]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return true;
}

// Resident finishing groups: group (S, q) of the resident grid.  The whole role lives in this function, entered at the
// very top of k_acq_fast before any register tile exists, so the kernel's register allocation is the maximum of the
// two roles, not their sum.  (A separate kernel on a second stream would be simpler, but two streams of one process may
// share a hardware queue, and then the two kernels wait for each other until they time out.)
#define FIN_LDS_DOUBLES (6 * PCABO_MAXD + 2 + 3 * FIN_STAGE_PER_WAVE + 2)   // ... + two flag words in the last double
__device__ __noinline__ void acq_server_finish_main(double* s_mem, MailPair* dev_mail,
                                                    int ctrl_idx, const MailPair* part_pairs, int S, int k, int q,
                                                    const double* __restrict__ bounds4, const double* __restrict__ ystats,
                                                    const AcqParams& prm, double* val, double* grad, double* host_val,
                                                    double* host_grad, HostMirror* hm, unsigned long long seq) {
  double* s_fin = s_mem;
  int* s_srvp = reinterpret_cast<int*>(s_mem + FIN_LDS_DOUBLES - 1);
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  double rng_c = 1.0;
  if (tid < k) rng_c = bounds4[PCABO_MAXD + tid] - bounds4[tid];
  for (unsigned long long cur_seq = seq;; ++cur_seq) {
    const unsigned long long t0 = wall_clock64();
    if (tid == 0) { s_srvp[0] = 0; s_srvp[1] = 0; }
    __syncthreads();
    if (w == 0) {                                               // this query's control pair of the round: 1 evaluate, 0 leave
      for (;;) {
        const pcabo_u4 hd = ld_pair_sys(dev_mail + ctrl_idx);
        if (__all(mail_seq(hd) == cur_seq)) { if (pair_value(hd) == 0.0 && l == 0) *s_srvp = 1; break; }
        if (__any(wall_clock64() - t0 > PCABO_SERVER_TIMEOUT_TICKS)) { if (l == 0) *s_srvp = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (*s_srvp) return;
    if (!acq_server_finisher(part_pairs + (size_t)q * S * PSTRIDE, S, k, q, cur_seq, ystats, prm, val, grad, host_val,
                             host_grad, hm, rng_c, s_fin, s_fin + 6 * PCABO_MAXD, s_fin + 6 * PCABO_MAXD + 2, s_srvp, t0, w, l))
      return;
    __syncthreads();
  }
}

// number of slab groups per query that launch_acq uses for this size
#define ACQ_SLAB32_NP 448          // 16 rows per work-group below this padded size, 32 from it on (see launch_acq)
int acq_slabs(int NP) { return NP / (NP >= ACQ_SLAB32_NP ? 32 : 16); }

// ---- fast path: NP = 64 NB <= 512 and k <= 40, everything static ---------------------------------------------
// The generic kernel above re-reads its operands (ZnT for ks and again for the gradient contraction, the R slab
// row-wise for v and again column-wise for w) and pays an exposed round trip per loop trip.  Here trip counts are
// template constants and the two big operands are read ONCE, at kernel entry, into registers:
//   z[u][b] = ZnT[c = w + 4u][l + 64 b]    (wave w: components w + 4u, u < 10; lanes: columns)   <= 80 doubles
//   r[u][b] = R[slab row w + 4u][l + 64 b] (wave w: rows w + 4u of the slab; lanes: columns)     <= 64 doubles
// so the whole kernel has one exposed memory round trip; everything after it is LDS + VALU:
//   ks        partial squared distances per wave (its components) -> LDS -> thread j adds the four partials in a
//             fixed order and applies the radial function
//   v         acc[u] = sum_b r[u][b] ks[l + 64 b], DPP wave sums
//   w_j       per-wave partial sum_u r[u][b] v[w + 4u] -> LDS -> thread j adds the four partials, folds in coef_j
//   gradient  gs[u] = sum_b t_sigma[l + 64 b] (xn_c - z[u][b]) from the SAME registers, DPP wave sums
// Wave-uniform indices are forced into SGPRs (readfirstlane): row bases are scalar, loads need no vector address
// arithmetic.  Columns j >= n hold zeros in ZnT (k_znorm) and get zero weights.
template <int SLAB, int NB, bool SRV>
__global__ __launch_bounds__(256) void k_acq_fast(
    QueryArgs qa, const double* __restrict__ Xq, int q_total, int n, int k, int NP_rt, int ld,
    const double* __restrict__ ZnT, const double* __restrict__ R, const double* __restrict__ alpha,
    const double* __restrict__ bounds4, const double* __restrict__ ystats, AcqParams prm, double* partial,
    unsigned int* counters, double* __restrict__ val, double* __restrict__ grad,
    double* host_val, double* host_grad, HostMirror* hm, unsigned long long seq, int combine, int qb,
    MailPair* dev_mail, MailPair* part_pairs, AcqBatch ab) {
  if (!SRV) { ACQ_BATCH_PROLOGUE() }
  constexpr int NP = NB * 64;
  constexpr int CU = 10;                 // components per wave
  constexpr int RW = SLAB / 4;           // slab rows per wave
  constexpr int NM = (NB + 3) / 4;       // columns per thread in the thread-per-column steps
  const double inv_ls = prm.inv_ls;
  const int kernel = prm.kernel, want_grad = prm.want_grad;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* s_ks = s_dyn;              // NP
  double* s_cf = s_dyn + NP;         // NP  coef_j (dks_j/dxn = coef_j (xn - zn_j))
  double* s_tm = s_dyn + 2 * NP;     // NP  alpha_j coef_j restricted to this slab's rows
  double* s_xn = s_dyn + 3 * NP;     // PCABO_MAXD
  double* s_v = s_xn + PCABO_MAXD;   // SLAB (+ flag word)
  double* s_p4 = s_v + SLAB + 2;     // 4 NP per-wave partials (squared distances, then w_j)
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool server = SRV;                                  // resident mode (own instantiation: the plain one keeps its registers), see below
  const int s = blockIdx.x, S = server ? (int)gridDim.x - 1 : (int)gridDim.x;
  if (server && s == S) {                                        // the finishing group of query blockIdx.y
    acq_server_finish_main(s_dyn, dev_mail, 1 + q_total * k + (int)blockIdx.y, part_pairs, S, k,
                           blockIdx.y, bounds4, ystats, prm, val, grad, host_val, host_grad, hm, seq);
    return;
  }
  STAMP(0);

  // ---- every global operand, issued before anything waits ---------------------------------------------------
  // Addresses are (scalar row base) + (lane offset) + (immediate 512 b): no vector address arithmetic per load.
  // (Rotating the block order per group to spread L2 lines made no difference and cost address arithmetic.)
  double z[CU][NB], r[RW][NB], al[NM];
  const unsigned lane8 = (unsigned)l * 8u;
#pragma unroll
  for (int u = 0; u < CU; ++u) {
    const int c = w + 4 * u;                                    // wave-uniform
    const char* zr = reinterpret_cast<const char*>(ZnT + (size_t)c * ld);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      z[u][b] = 0.0;
      if (c < k) z[u][b] = *reinterpret_cast<const double*>(zr + lane8 + 512 * b);
    }
  }
#pragma unroll
  for (int u = 0; u < RW; ++u) {
    const int ri = slab_row<SLAB>(s, w + 4 * u, NP);          // wave-uniform
    const char* Rr = reinterpret_cast<const char*>(R + (size_t)ri * ld);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      r[u][b] = 0.0;                                            // blocks above the diagonal are exactly zero
      if (64 * b <= ri) r[u][b] = *reinterpret_cast<const double*>(Rr + lane8 + 512 * b);
    }
  }
#pragma unroll
  for (int m = 0; m < NM; ++m) { const int j = tid + 256 * m; al[m] = j < n ? alpha[j] : 0.0; }
  // ---- per query: large batches (value scoring of the raw samples, many-restart optimisation) give each group qb
  // queries, so the register tiles above are loaded once for all of them; the in-launch combine uses qb = 1
  // ---- resident mode (dev_mail != nullptr): the kernel stays for all the evaluations of one optimize call.  Round r of
  // a query carries sequence number seq + r - 1; its coordinates and its control pair (1 evaluate, 0 leave) arrive
  // through the device mailbox, which the host fills through the PCIe BAR.  Every query counts its own rounds - the host may drive the restart groups
  // independently of each other.  A round costs neither a launch nor a refill of the register tiles.  Every wait is
  // bounded (PCABO_SERVER_TIMEOUT_TICKS): a group that times out simply leaves, the host then times out on the
  // missing result and finishes the call with plain launches.
  int* s_srv = reinterpret_cast<int*>(s_v + SLAB + 1);        // 0 continue, 1 leave (set by wave 0)
  unsigned long long cur_seq = seq;
  double b_lo = 0.0, b_hi = 1.0;
  if (server && l < k) { b_lo = bounds4[l]; b_hi = bounds4[PCABO_MAXD + l]; }
  for (;;) {                                                  // rounds (one pass unless resident)
  int q = (!SRV && ab.table) ? (int)(reinterpret_cast<const unsigned*>(qa.x)[blockIdx.y] & 0xffffu) : (int)blockIdx.y * qb;
  double* out = partial + ((size_t)q * S + s) * PSTRIDE;
  if (server) {
    const unsigned long long t0 = wall_clock64();
    if (tid == 0) { s_srv[0] = 0; s_srv[1] = 0; }
    __syncthreads();
    if (w == 0) {                                             // control pair + this query's coordinates: one round trip per poll
      volatile int* flags = s_srv;
      const void* ptr[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) ptr[t] = (t & 1) && l < k ? (const void*)(dev_mail + 1 + q * k + l) : (const void*)(dev_mail + 1 + q_total * k + q);
      for (;;) {
        pcabo_u4 o[8];
        ld_pairs_sys8(ptr, o);
        const bool ctl = mail_seq(o[0]) == cur_seq;           // the same pair for every lane
        const bool bye = ctl && pair_value(o[0]) == 0.0;      // leaving needs no coordinates
        const bool ok = ctl && (bye || l >= k || mail_seq(o[1]) == cur_seq);
        if (__all(ok)) {
          if (pair_value(o[0]) == 0.0) { if (l == 0) flags[0] = 1; }    // control pair 0: the call is over or this query's restart group has finished
          else if (l < k) s_xn[l] = (pair_value(o[1]) - b_lo) / (b_hi - b_lo);
          break;
        }
        if (__any(wall_clock64() - t0 > PCABO_SERVER_TIMEOUT_TICKS)) { if (l == 0) flags[0] = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (*s_srv) return;
  }
  MailPair* pout = server ? part_pairs + ((size_t)q * S + s) * PSTRIDE : nullptr;
  for (int qi = 0; qi < qb; ++qi, ++q, out += (size_t)S * PSTRIDE) {
  if (q >= q_total) break;
  if (qi > 0) __syncthreads();          // the previous query's readers of s_xn / s_ks / s_tm / s_v are done
  if (!server) {
  if (tid < k) {
    double lo = bounds4[tid], hi = bounds4[PCABO_MAXD + tid];
    double xv = Xq ? Xq[(size_t)q * k + tid] : qa.x[q * k + tid];
    s_xn[tid] = (xv - lo) / (hi - lo);
  }
  __syncthreads();
  }
  STAMP(1);
  // ---- ks ------------------------------------------------------------------------------------------------------
  double xc[CU];
#pragma unroll
  for (int u = 0; u < CU; ++u) xc[u] = s_xn[(w + 4 * u < k) ? w + 4 * u : k - 1];
  {
    double sq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) sq[b] = 0.0;
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      if (w + 4 * u < k) {                  // wave-uniform
#pragma unroll
        for (int b = 0; b < NB; ++b) { const double d = xc[u] - z[u][b]; sq[b] += d * d; }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) s_p4[w * NP + l + 64 * b] = sq[b];
  }
  __syncthreads();
  const double s5 = 2.23606797749979;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int j = tid + 256 * m;
    if (j < NP) {
      double ks = 0.0, cf = 0.0;
      if (j < n) {
        double sq = ((s_p4[j] + s_p4[NP + j]) + s_p4[2 * NP + j]) + s_p4[3 * NP + j];
        sq *= inv_ls * inv_ls;
        if (kernel == 1) {
          ks = exp(-0.5 * sq);
          cf = -ks * inv_ls * inv_ls;
        } else {
          double dist = sqrt(fmax(sq, 1e-30));
          double e = exp(-s5 * dist);
          ks = ((s5 * dist + 1.0) + (5.0 / 3.0) * (dist * dist)) * e;
          cf = -(5.0 / 3.0) * (1.0 + s5 * dist) * e * inv_ls * inv_ls;
        }
      }
      s_ks[j] = ks;
      s_cf[j] = cf;
    }
  }
  __syncthreads();
  STAMP(2);
  // ---- v_i = R[i][:] . ks for the slab's rows -------------------------------------------------------------------
  {
    double acc[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) acc[u] = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double kj = s_ks[l + 64 * b];
#pragma unroll
      for (int u = 0; u < RW; ++u) acc[u] += r[u][b] * kj;
    }
#pragma unroll
    for (int u = 0; u < RW; ++u) acc[u] = wave_sum(acc[u]);
    if (l == 0) {
#pragma unroll
      for (int u = 0; u < RW; ++u) s_v[w + 4 * u] = acc[u];
    }
  }
  __syncthreads();
  STAMP(3);
  // slab contributions to |v|^2 and to mu_s = alpha . ks
  if (w == 0) {
    double vv = 0.0, mu = 0.0;
    if (l < SLAB) {
      const int i = slab_row<SLAB>(s, l, NP);
      const double vi = s_v[l];
      vv = vi * vi;
      if (i < n) mu = alpha[i] * s_ks[i];
    }
    vv = wave_sum(vv);
    mu = wave_sum(mu);
    if (l == 0) {
      if (server) { st_pair_sys(pout + 0, make_pair(vv, cur_seq)); st_pair_sys(pout + 1, make_pair(mu, cur_seq)); }
      else { st_wt(out + 0, vv); st_wt(out + 1, mu); }
    }
  }
  if (want_grad) {
  // ---- w_j (slab part) = sum_{i in slab} R[i][j] v_i: per-wave partials from the same registers -----------------
  {
    double vw[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) vw[u] = s_v[w + 4 * u];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      double pw = 0.0;
#pragma unroll
      for (int u = 0; u < RW; ++u) pw += r[u][b] * vw[u];
      s_p4[w * NP + l + 64 * b] = pw;
    }
  }
  __syncthreads();   // also: wave 0 has finished reading s_ks before it is reused below
  STAMP(4);
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int j = tid + 256 * m;
    if (j < NP) {
      const double wj = ((s_p4[j] + s_p4[NP + j]) + s_p4[2 * NP + j]) + s_p4[3 * NP + j];
      const bool mine = ((j / (SLAB / 2)) == s) || (((NP - 1 - j) / (SLAB / 2)) == s);
      const double cf = s_cf[j];
      s_tm[j] = mine ? al[m] * cf : 0.0;     // al = 0 beyond n
      s_ks[j] = wj * cf;                     // ks no longer needed: reuse as t_sigma
    }
  }
  __syncthreads();
  STAMP(5);
  // ---- contraction with (xn_c - zn_jc) -------------------------------------------------------------------------
  {
    double gs[CU], gm[CU];
#pragma unroll
    for (int u = 0; u < CU; ++u) { gs[u] = 0.0; gm[u] = 0.0; }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double ts = s_ks[l + 64 * b], tm = s_tm[l + 64 * b];
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const double dlt = xc[u] - z[u][b];
        gs[u] += ts * dlt;
        gm[u] += tm * dlt;
      }
    }
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      const int c = w + 4 * u;
      if (c < k) {                          // wave-uniform
        const double a = wave_sum(gs[u]), b2 = wave_sum(gm[u]);
        if (l == 0) {
          if (server) { st_pair_sys(pout + 2 + c, make_pair(a, cur_seq)); st_pair_sys(pout + 2 + PCABO_MAXD + c, make_pair(b2, cur_seq)); }
          else { st_wt(out + 2 + c, a); st_wt(out + 2 + PCABO_MAXD + c, b2); }
        }
      }
    }
  }
  }  // want_grad
  }   // queries of this group
  q = (!SRV && ab.table) ? (int)(reinterpret_cast<const unsigned*>(qa.x)[blockIdx.y] & 0xffffu) : (int)blockIdx.y * qb;   // (combine: qb = 1)
  out = partial + ((size_t)q * S + s) * PSTRIDE;
  if (!server) {
    acq_tail<SLAB>(combine, s_v, partial, counters, q, S, k, bounds4, ystats, prm, val, grad, host_val, host_grad, hm, cur_seq, tid, w, l);
    break;
  }
  ++cur_seq;
  __syncthreads();                      // the finish of this round has let go of the shared arrays
  }   // rounds
}

// ---- scalar log-EI helper, value and derivative (botorch/acquisition/analytic.py::_log_ei_helper)
__device__ inline void log_ei_helper(double u, double* h, double* dh) {
  const double inv_sqrt2 = 0.7071067811865476;
  const double inv_sqrt_2pi = 0.3989422804014327;
  const double log2pi = 1.8378770664093453;
  if (u > -1.0) {
    double phi = inv_sqrt_2pi * exp(-0.5 * u * u);
    double Phi = 0.5 * erfc(-inv_sqrt2 * u);
    double ei = phi + u * Phi;
    *h = log(ei);
    *dh = Phi / ei;
    return;
  }
  double log_phi = -0.5 * (u * u + log2pi);
  if (u > -1e6) {
    // botorch: w = log(erfcx(-u/sqrt2) |u|) + log(sqrt(pi/2)); h = log_phi + log1mexp(w).  With E = e^w formed
    // directly, log1mexp(w) = log1p(-E) and expm1(-w) = (1 - E)/E: the same conditioning (both routes lose
    // eps/(1 - E)), three transcendental calls fewer on the one wave every round waits for.
    const double ex = erfcx(-inv_sqrt2 * u);
    const double E = (ex * fabs(u)) * 1.2533141373155003;       // sqrt(pi/2)
    *h = log_phi + log1p(-E);
    const double dw = (u + 0.7978845608028654 / ex) + 1.0 / u;  // sqrt(2/pi)/erfcx + u + 1/u
    *dh = -u - dw * E / (1.0 - E);
  } else {
    *h = log_phi - 2.0 * log(fabs(u));
    *dh = -u - 2.0 / u;
  }
}

// Scalar chain of one query (one wave): mean, sigma, u -> value and the two coefficients of the gradient's chain
// rule, from the summed |v|^2 and mu_s.  coef receives {c_mu, c_sg}.
__device__ void acq_scalar_core(double vv, double mus, int q, double ym, double ysd, const AcqParams& p, double* val,
                                double* host_val, double* coef, int l) {
  const double mu = ym + ysd * mus;
  double var = (1.0 - vv) * (ysd * ysd);
  bool clamped = false;
  if (!(var >= 1e-10)) { var = 1e-10; clamped = true; }     // gpytorch min_variance (double)
  if (var < 1e-12) { var = 1e-12; clamped = true; }          // botorch _mean_and_sigma(min_var)
  const double sigma = sqrt(var);
  double u = (mu - p.best_f) / sigma;
  const double sgn = p.maximize ? 1.0 : -1.0;
  u *= sgn;
  double value, dv_du, dv_dsig;
  if (p.acq == 0) {
    double h, dh;
    log_ei_helper(u, &h, &dh);
    value = h + log(sigma);
    dv_du = dh;
    dv_dsig = 1.0 / sigma;
  } else {
    value = 0.5 * erfc(-0.7071067811865476 * u);
    dv_du = 0.3989422804014327 * exp(-0.5 * u * u);
    dv_dsig = 0.0;
  }
  if (l == 0) {
    val[q] = value;
    if (host_val) host_val[q] = value;
    // du = sgn dmu/sigma - u dsigma/sigma ; dsigma = -s_y^2 g_sigma / sigma (0 where the variance was clamped)
    coef[0] = dv_du * sgn * ysd / sigma;
    coef[1] = clamped ? 0.0 : (dv_dsig - dv_du * u / sigma) * (-(ysd * ysd) / sigma);
  }
  STAMP_FIN(12);
}

// The same, reading the slab sums from the partial records (lanes over slabs, S <= 128).
__device__ void acq_finish_scalar(const double* base, int S, int q, const double* ystats, const AcqParams& p, double* val,
                                  double* host_val, double* coef, int l) {
  const double ym = ystats[0], ysd = ystats[1];            // issued with the partial loads, not after the reductions
  double vv = 0.0, mus = 0.0;
  for (int s = l; s < S; s += 64) { vv += base[(size_t)s * PSTRIDE]; mus += base[(size_t)s * PSTRIDE + 1]; }
  vv = wave_sum(vv);
  mus = wave_sum(mus);
  STAMP_FIN(11);
  acq_scalar_core(vv, mus, q, ym, ysd, p, val, host_val, coef, l);
}

// One query, one wave (the large-batch combine pass): scalar chain, then the gradient from all slabs in order.
__device__ void acq_finish_query(const double* base, int S, int k, int q, const double* bounds4, const double* ystats,
                                 const AcqParams& p, double* val, double* grad, double* host_val, double* host_grad,
                                 int l) {
  __shared__ double s_cf2[4][2];
  double* coef = s_cf2[(threadIdx.x >> 6) & 3];
  acq_finish_scalar(base, S, q, ystats, p, val, host_val, coef, l);
  if (p.want_grad) {
    __builtin_amdgcn_wave_barrier();
    const double c_mu = coef[0], c_sg = coef[1];
    for (int c = l; c < k; c += 64) {
      double a0, b0, a1, b1, a2, b2;
      grad_partial_sum(base, S, c, 0, &a0, &b0);
      grad_partial_sum(base, S, c, 1, &a1, &b1);
      grad_partial_sum(base, S, c, 2, &a2, &b2);
      const double gs = (a0 + a1) + a2, gm = (b0 + b1) + b2;
      double g = __fma_rn(c_mu, gm, c_sg * gs) / (bounds4[PCABO_MAXD + c] - bounds4[c]);
      grad[(size_t)q * k + c] = g;
      if (host_grad) host_grad[(size_t)q * k + c] = g;
    }
  }
}

// Combine pass for large batches (the launch boundary makes the partial records visible): one wave per query.
__global__ __launch_bounds__(256) void k_acq_combine(const double* __restrict__ partial, int q_total, int S, int k,
                                                     const double* __restrict__ bounds4,
                                                     const double* __restrict__ ystats, AcqParams p,
                                                     double* __restrict__ val, double* __restrict__ grad, AcqBatch ab) {
  if (ab.zs) {
    const unsigned run_ = blockIdx.z;
    partial = zrun(partial, ab.zs, run_); bounds4 = zrun(bounds4, ab.zs, run_); ystats = zrun(ystats, ab.zs, run_);
    val = zrun(val, ab.zs, run_); grad = zrun(grad, ab.zs, run_);
    if (ab.k_dev) k = *zrun(ab.k_dev, ab.zs, run_);
    if (ab.bestf) p.best_f = *zrun(ab.bestf, ab.zs, run_);
  }
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q < q_total)
    acq_finish_query(partial + (size_t)q * S * PSTRIDE, S, k, q, bounds4, ystats, p, val, grad, nullptr, nullptr,
                     threadIdx.x & 63);
}

// =====================================================================================================================
// Throughput variant for batched runs: one work-group per (restart group, 64-row slab of R).
// k_acq_fast / k_acq_fused above are built for the latency of ONE run: a work-group per (16-row slab, query), each
// recomputing the kernel vector - ~290 groups of ~10 us for the 10 queries of a run at n = 449, which is the right
// trade for a single run and 100x the necessary CU-time when thirty runs share the chip.  Here a work-group serves the
// (up to) GQ = 5 queries of one joint L-BFGS-B problem together:
//   ks      thread per column j: one pass over ZnT[:, j] serves all 5 queries (ks and the radial factor stay in LDS / registers)
//   v       v_q[i] = sum_j R[i][j] ks_q[j] for the slab's 64 rows: R tiles (64 x 32) go through LDS and are read back
//           row-wise, so a LANE OWNS A ROW - no cross-lane reduction; the four waves split the columns of a tile
//   w       w_q[j] = sum_{i in slab} R[i][j] v_q[i]: thread per column, coalesced straight from global - no reduction either
//   grad    gs_q[c] = sum_j w_q[j] cf_q[j] (xn_q[c] - zn[c][j]) (+ the alpha part restricted to the slab's rows): wave per
//           component, lanes over j, DPP wave sums
// R is read once per pass for 5 queries (two passes: the second needs all of v), ZnT twice.  Slab partials are combined
// by the last work-group to arrive at the queries' tickets, as above; it finishes its 5 queries side by side (scalar
// chains on different waves, the gradient sums one thread per (query, component)).
// The table in the QueryArgs slot names the groups of the launch: 32-bit entries run << 16 | first query << 8 | count.
// Arithmetic differs from the kernels above in summation order only (~1e-15 relative): it is selected per context /
// batch (PCABO_OPT_GROUP_ACQ), never mixed within a run.
#ifdef PCABO_ACQ_TIMING
#define GSTAMP(i) do { if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && threadIdx.x == 0) g_acq_stamps[i] = wall_clock64(); } while (0)
#else
#define GSTAMP(i)
#endif
#define GQ 5
#define GT_LD 17            // LDS leading dimension of a wave's 64 x 16 tile (odd: row-wise reads conflict-free)
// Hand-over of LDS data between the lanes of ONE wave: the LDS pipeline serves a wave's instructions in order, so all
// that is needed is that the compiler keeps the order and the data have arrived - a wait on the LDS counter.  (A
// wavefront-scope fence also waits for every global load in flight, which defeats the prefetching below.)
#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int NT>           // NT = ceil(NP / 256): columns per thread in the thread-per-column phases
__global__ __launch_bounds__(256) void k_acq_group(
    QueryArgs qa, const double* __restrict__ Xq, int n, int k, int NP, int ld,
    const double* __restrict__ ZnT, const double* __restrict__ R, const double* __restrict__ alpha,
    const double* __restrict__ bounds4, const double* __restrict__ ystats, AcqParams prm, double* partial,
    unsigned int* counters, double* __restrict__ val, double* __restrict__ grad, double* host_val, double* host_grad,
    HostMirror* hm, unsigned long long seq, AcqBatch ab) {
  const unsigned ent = reinterpret_cast<const unsigned*>(qa.x)[blockIdx.y];
  const unsigned run_ = ent >> 16;
  const int q0 = (int)((ent >> 8) & 0xffu), nq = (int)(ent & 0xffu);
  if (ab.zs) {
    ZnT = zrun(ZnT, ab.zs, run_); R = zrun(R, ab.zs, run_); alpha = zrun(alpha, ab.zs, run_);
    bounds4 = zrun(bounds4, ab.zs, run_); ystats = zrun(ystats, ab.zs, run_); partial = zrun(partial, ab.zs, run_);
    counters = zrun(counters, ab.zs, run_); val = zrun(val, ab.zs, run_); grad = zrun(grad, ab.zs, run_);
    Xq = zrun(Xq, ab.xq_host ? ab.hzs : ab.zs, run_);
    host_val = zrun(host_val, ab.hzs, run_); host_grad = zrun(host_grad, ab.hzs, run_); hm = zrun(hm, ab.hzs, run_);
    if (ab.k_dev) k = *zrun(ab.k_dev, ab.zs, run_);
    if (ab.bestf) prm.best_f = *zrun(ab.bestf, ab.zs, run_);
  }
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x, S = gridDim.x;
  const int r0 = 64 * s, ncol = 64 * (s + 1);       // the slab's rows; columns >= ncol hold zeros in these rows
  const int KS = (k + 1) & ~1;                        // stride of a query's coordinates in LDS
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* s_ks = s_dyn;                               // [GQ][NP]  kernel vectors, later t_sigma = w * cf
  double* s_tile = s_ks + GQ * NP;                    // [4 waves][64][GT_LD] wave-private transposing tiles (64 rows x 16 columns)
  double* s_v = s_tile + 4 * 64 * GT_LD;              // [64][8]   v of the slab's rows (queries contiguous)
  double* s_tm = s_v + 64 * 8;                        // [GQ][64]  alpha_j cf_q[j] for the slab's own rows
  double* s_coef = s_tm + GQ * 64;                    // [GQ][2]
  int* s_flag = reinterpret_cast<int*>(s_coef + 2 * GQ);   // [GQ + 1]
  double* s_vvmu = s_coef + 2 * GQ + 4;               // [GQ][2]  this slab's |v|^2 / mu_s parts until the final burst of stores
  double* s_xn = s_vvmu + 2 * GQ;                     // [GQ][KS]
  double* s_part = s_tile;                            // [4][GQ][64] per-wave partial row sums: each wave inside its own tile
  double* s_gout = s_ks;                              // [2 GQ][KS] gradient parts of all components, staged for one burst of stores
  const double inv_ls = prm.inv_ls;
  const int kernel = prm.kernel;

  // ---- normalised query points (queries beyond nq repeat the first: computed, never published) -----------------
  GSTAMP(0);
  for (int idx = tid; idx < GQ * k; idx += 256) {
    const int q = idx / k, c = idx - q * k;
    const int qq = q < nq ? q : 0;
    const double lo = bounds4[c], hi = bounds4[PCABO_MAXD + c];
    s_xn[q * KS + c] = (Xq[(size_t)(q0 + qq) * k + c] - lo) / (hi - lo);
  }
  __syncthreads();
  GSTAMP(1);
  // ---- ks and the radial derivative factor, all queries per pass over ZnT.  Only the columns this slab uses (j < ncol:
  // R's rows of the slab are zero beyond their diagonal block); all columns of a thread advance together so that a
  // coordinate read from LDS serves every one of them; CB components' loads are in flight per trip -----------------
  double cfr[NT][GQ];
  const double s5 = 2.23606797749979;
  {
    const int jlim = n < ncol ? n : ncol;
    double sq[NT][GQ];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < GQ; ++q) sq[t][q] = 0.0;
    constexpr int CB = NT == 1 ? 40 : (NT == 2 ? 20 : 8);
    for (int c0 = 0; c0 < k; c0 += CB) {
      double z[NT][CB];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < CB; ++u) {
          // unconditional loads from clamped addresses: a load behind a branch makes the number of loads in flight unknown
          // to the compiler, and every later wait becomes "wait for all of them"
          const int cc = c0 + u < k ? c0 + u : k - 1, jj = tid + 256 * t < NP ? tid + 256 * t : NP - 1;
          z[t][u] = ZnT[(size_t)cc * ld + jj];
        }
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        if (c0 + u < k) {                             // uniform
#pragma unroll
          for (int q = 0; q < GQ; ++q) {
            const double xq = s_xn[q * KS + c0 + u];
#pragma unroll
            for (int t = 0; t < NT; ++t) { const double d = xq - z[t][u]; sq[t][q] += d * d; }
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = tid + 256 * t;
#pragma unroll
      for (int q = 0; q < GQ; ++q) {
        double ks = 0.0, cf = 0.0;
        if (j < jlim) {
          const double sqq = sq[t][q] * (inv_ls * inv_ls);
          if (kernel == 1) {
            ks = exp(-0.5 * sqq);
            cf = -ks * inv_ls * inv_ls;
          } else {
            const double dist = sqrt(fmax(sqq, 1e-30));
            const double e = exp(-s5 * dist);
            ks = ((s5 * dist + 1.0) + (5.0 / 3.0) * (dist * dist)) * e;
            cf = -(5.0 / 3.0) * (1.0 + s5 * dist) * e * inv_ls * inv_ls;
          }
        }
        if (j < NP) s_ks[q * NP + j] = ks;
        cfr[t][q] = cf;
      }
    }
  }
  __syncthreads();
  GSTAMP(2);
  // ---- v for the slab's rows.  Wave w owns columns 16 w .. 16 w + 15 of every 64-column block: it loads its 64 x 16
  // piece coalesced (lanes along the columns), turns it in its PRIVATE LDS tile and reads it back with a lane per row -
  // no work-group barrier in the loop, two pieces in flight ahead of the one being consumed --------------------------
  {
    double acc[GQ];
#pragma unroll
    for (int q = 0; q < GQ; ++q) acc[q] = 0.0;
    const int nblk = s + 1;
    const double* Rs = R + (size_t)r0 * ld + 16 * w + (l & 15);
    double* tw = s_tile + w * 64 * GT_LD;
    auto fetch = [&](int J, double (&dst)[16]) {
#pragma unroll
      for (int u = 0; u < 16; ++u) dst[u] = Rs[(size_t)(4 * u + (l >> 4)) * ld + 64 * J];
    };
    auto put = [&](const double (&src)[16]) {
#pragma unroll
      for (int u = 0; u < 16; ++u) tw[(4 * u + (l >> 4)) * GT_LD + (l & 15)] = src[u];
      WAVE_LDS_SYNC();
    };
    auto use = [&](int J) {
      const double* trow = tw + l * GT_LD;
      const double* kcol = s_ks + 64 * J + 16 * w;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const double rv = trow[jj];
#pragma unroll
        for (int q = 0; q < GQ; ++q) acc[q] += rv * kcol[q * NP + jj];
      }
      WAVE_LDS_SYNC();
    };
    double bufA[16], bufB[16];
    fetch(0, bufA);
    if (nblk > 1) fetch(1, bufB);
    for (int J = 0; J < nblk; J += 2) {
      put(bufA);
      if (J + 2 < nblk) fetch(J + 2, bufA);
      use(J);
      if (J + 1 < nblk) {
        put(bufB);
        if (J + 3 < nblk) fetch(J + 3, bufB);
        use(J + 1);
      }
    }
#pragma unroll
    for (int q = 0; q < GQ; ++q) s_part[w * 64 * GT_LD + q * 64 + l] = acc[q];
  }
  __syncthreads();
  for (int idx = tid; idx < GQ * 64; idx += 256) {
    const int q = idx >> 6, m = idx & 63;
    const double* pp = s_part + q * 64 + m;
    s_v[m * 8 + q] = ((pp[0] + pp[64 * GT_LD]) + pp[2 * 64 * GT_LD]) + pp[3 * 64 * GT_LD];
  }
  __syncthreads();
  GSTAMP(3);
  // ---- slab contributions to |v|^2 and mu_s = alpha . ks: wave q (wave 0 also the fifth query) -------------------
  for (int q = w; q < nq; q += 4) {
    const int i = r0 + l;
    const double vi = s_v[l * 8 + q];
    double vv = vi * vi;
    double mu = i < n ? alpha[i] * s_ks[q * NP + i] : 0.0;
    vv = wave_sum(vv);
    mu = wave_sum(mu);
    if (l == 0) { s_vvmu[2 * q] = vv; s_vvmu[2 * q + 1] = mu; }     // (stored with the gradient parts at the end: a
    // write-through store in front of a load keeps that load's wait open until the store has reached memory)
  }
  if (prm.want_grad) {
    // ---- w (slab part), thread per column: w_q[j] = sum_m R[r0 + m][j] v_q[m] --------------------------------------
    double wacc[NT][GQ];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < GQ; ++q) wacc[t][q] = 0.0;
    constexpr int RU = NT <= 2 ? 16 : 8;      // rows per trip: RU * NT loads in flight
    for (int m0 = 0; m0 < 64; m0 += RU) {
      double rr[RU][NT];
#pragma unroll
      for (int u = 0; u < RU; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int j = tid + 256 * t;
          const double rv = R[(size_t)(r0 + m0 + u) * ld + (j < ncol ? j : 0)];      // (unconditional, see the ks phase)
          rr[u][t] = j < ncol ? rv : 0.0;
        }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        double vq[GQ];
#pragma unroll
        for (int q = 0; q < GQ; ++q) vq[q] = s_v[(m0 + u) * 8 + q];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < GQ; ++q) wacc[t][q] += rr[u][t] * vq[q];
      }
    }
    GSTAMP(4);
    __syncthreads();                      // every wave has read the ks it needed (mu above): s_ks becomes t_sigma
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = tid + 256 * t;
      if (j < NP) {
        const bool mine = j >= r0 && j < r0 + 64;
        const double aj = (mine && j < n) ? alpha[j] : 0.0;
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
          s_ks[q * NP + j] = wacc[t][q] * cfr[t][q];
          if (mine) s_tm[q * 64 + (j - r0)] = aj * cfr[t][q];
        }
      }
    }
    __syncthreads();
    GSTAMP(5);
    // ---- contraction with (xn_c - zn_jc): wave per component, lanes over points.  The lane's share of t_sigma lives in
    // registers; the ZnT row of the NEXT component is loaded before the wave sums of the current one ------------------
    constexpr int NBL = 4 * NT;                        // 64-column blocks per lane
    const int jmax = n < ncol ? n : ncol;
    double ts[GQ][NBL], tmq[GQ];
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
      const int j = l + 64 * b;
#pragma unroll
      for (int q = 0; q < GQ; ++q) ts[q][b] = j < jmax ? s_ks[q * NP + j] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < GQ; ++q) tmq[q] = (r0 + l < n) ? s_tm[q * 64 + l] : 0.0;
    __syncthreads();                                   // t_sigma is in registers everywhere: its LDS array becomes s_gout
    GSTAMP(10);
    double zc[NBL], zn_[NBL];
    auto load_row = [&](int c, double (&dst)[NBL]) {
      const double* zrow = ZnT + (size_t)c * ld;
#pragma unroll
      for (int b = 0; b < NBL; ++b) dst[b] = zrow[(64 * b < jmax) ? l + 64 * b : l];       // unconditional; blocks beyond jmax meet t_sigma = 0
    };
    if (w < k) load_row(w, zc);
    for (int c = w; c < k; c += 4) {
      if (c + 4 < k) load_row(c + 4, zn_);
      if (c == w) { GSTAMP(11); }
      double gs[GQ], gm[GQ];
      const int bm = r0 >> 6;                          // the 64-column block that holds the slab's own rows (= lane's row r0 + l)
#pragma unroll
      for (int q = 0; q < GQ; ++q) {
        const double xq = s_xn[q * KS + c];
        double a = 0.0, zm = 0.0;
#pragma unroll
        for (int b = 0; b < NBL; ++b) { a += ts[q][b] * (xq - zc[b]); if (b == bm) zm = zc[b]; }
        gs[q] = a;
        gm[q] = tmq[q] * (xq - zm);
      }
      // 2 GQ wave sums per component: the lanes' partials go through the wave's private LDS tile (free since pass 1) and
      // come back transposed - lane 4 v + h adds 16 of value v's 64 partials, two quad steps finish - instead of 2 GQ
      // DPP reduction trees (the largest item of the kernel's timeline when they were)
      {
        if (c == w) { asm volatile("" :: "v"(gs[0]), "v"(gs[4]), "v"(gm[4])); GSTAMP(12); }
        double* red = s_tile + w * 64 * GT_LD;            // [2 GQ][66] <= 64 * GT_LD doubles
#pragma unroll
        for (int q = 0; q < GQ; ++q) { red[q * 66 + l] = gs[q]; red[(GQ + q) * 66 + l] = gm[q]; }    // row stride 66: see below
        WAVE_LDS_SYNC();
        if (c == w) { GSTAMP(13); }
        const int v = l >> 2, h = l & 3;                  // lanes 0 .. 4 * 2 GQ - 1 = 39 are active
        double sum = 0.0;
        if (v < 2 * GQ) {
          // lane (v, h) adds partials h, h + 4, ... of value v.  With rows of 64 doubles and 16 contiguous partials per lane
          // every lane of a read hit one of two LDS banks (a 20-way conflict: 1.2 us per component); rows of 66 doubles and
          // interleaved partials spread the 40 lanes over 32 banks
          const double* src = red + v * 66 + h;
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int i = 0; i < 16; i += 2) { s0 += src[4 * i]; s1 += src[4 * i + 4]; }
          sum = s0 + s1;
        }
        sum += dpp_get<0xB1, 0xf>(sum);                   // quad_perm [1,0,3,2]
        sum += dpp_get<0x4E, 0xf>(sum);                   // quad_perm [2,3,0,1]
        if (h == 0 && v < 2 * GQ) s_gout[v * KS + c] = sum;
        WAVE_LDS_SYNC();
        if (c == w) { GSTAMP(14); }
        if (c == w + 4) { GSTAMP(15); }
      }
#pragma unroll
      for (int b = 0; b < NBL; ++b) zc[b] = zn_[b];
    }
  }
  // ---- this slab's partial records, all stores in one burst ---------------------------------------------------------
  __syncthreads();
  if (tid < 2 * nq) st_wt(partial + ((size_t)(q0 + (tid >> 1)) * S + s) * PSTRIDE + (tid & 1), s_vvmu[tid]);
  if (prm.want_grad)
    for (int idx = tid; idx < 2 * GQ * k; idx += 256) {
      const int v = idx / k, c = idx - v * k, q = v < GQ ? v : v - GQ;
      if (q < nq) st_wt(partial + ((size_t)(q0 + q) * S + s) * PSTRIDE + 2 + (v < GQ ? 0 : PCABO_MAXD) + c, s_gout[v * KS + c]);
    }
  // ---- tickets: the last slab group of a query finishes it ---------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  GSTAMP(6);
  if (tid == 0) {
    int any = 0;
    for (int q = 0; q < nq; ++q) {
      const unsigned int t = __hip_atomic_fetch_add(&counters[q0 + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t % (unsigned int)S) == (unsigned int)(S - 1);
      s_flag[q] = last;
      any |= last;
    }
    if (any) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_flag[GQ] = any;
  }
  __syncthreads();
  if (!s_flag[GQ]) return;
  if (threadIdx.x == 0) { GSTAMP(7); }
#ifdef PCABO_ACQ_TIMING
  if (threadIdx.x == 0 && blockIdx.y == 0) g_acq_stamps[7] = wall_clock64();
#endif
  // The partial records come from other XCDs: every read below is a trip to memory, so all of them are issued first -
  // the gradient sums (one thread per (query, component), slabs in order) and, per query, the |v|^2 / mu_s sums - and
  // the scalar log-EI chains of ALL queries then run side by side in the lanes of wave 0.
  double gsum[2] = {0.0, 0.0}, gmsum[2] = {0.0, 0.0};
  if (prm.want_grad) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {                       // nq * k <= 5 * 128 > 256 threads only beyond k = 51
      const int idx = tid + 256 * rep;
      if (idx < nq * k) {
        const int q = idx / k, c = idx - q * k;
        if (s_flag[q]) {
          const double* base = partial + (size_t)(q0 + q) * S * PSTRIDE;
          double gs = 0.0, gm = 0.0;
          for (int sl = 0; sl < S; ++sl) { gs += base[(size_t)sl * PSTRIDE + 2 + c]; gm += base[(size_t)sl * PSTRIDE + 2 + PCABO_MAXD + c]; }
          gsum[rep] = gs; gmsum[rep] = gm;
        }
      }
    }
  }
  if (w == 0) {
    double vv = 0.0, mus = 0.0;
    const int q = l < nq ? l : 0;
    if (l < nq && s_flag[q]) {
      const double* base = partial + (size_t)(q0 + q) * S * PSTRIDE;
      for (int sl = 0; sl < S; ++sl) { vv += base[(size_t)sl * PSTRIDE]; mus += base[(size_t)sl * PSTRIDE + 1]; }
    }
    if (l < nq && s_flag[q]) acq_scalar_core(vv, mus, q0 + q, ystats[0], ystats[1], prm, val, host_val, s_coef + 2 * q, 0);
  }
  __syncthreads();
  if (prm.want_grad) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int idx = tid + 256 * rep;
      if (idx < nq * k) {
        const int q = idx / k, c = idx - q * k;
        if (s_flag[q]) {
          const double g = __fma_rn(s_coef[2 * q], gmsum[rep], s_coef[2 * q + 1] * gsum[rep]) / (bounds4[PCABO_MAXD + c] - bounds4[c]);
          grad[(size_t)(q0 + q) * k + c] = g;
          if (host_grad) host_grad[(size_t)(q0 + q) * k + c] = g;
        }
      }
    }
    for (int idx = tid + 512; idx < nq * k; idx += 256) {     // k > 102: the rest, straightforwardly
      const int q = idx / k, c = idx - q * k;
      if (!s_flag[q]) continue;
      const double* base = partial + (size_t)(q0 + q) * S * PSTRIDE;
      double gs = 0.0, gm = 0.0;
      for (int sl = 0; sl < S; ++sl) { gs += base[(size_t)sl * PSTRIDE + 2 + c]; gm += base[(size_t)sl * PSTRIDE + 2 + PCABO_MAXD + c]; }
      const double g = __fma_rn(s_coef[2 * q], gm, s_coef[2 * q + 1] * gs) / (bounds4[PCABO_MAXD + c] - bounds4[c]);
      grad[(size_t)(q0 + q) * k + c] = g;
      if (host_grad) host_grad[(size_t)(q0 + q) * k + c] = g;
    }
  }
#ifdef PCABO_ACQ_TIMING
  if (threadIdx.x == 0 && blockIdx.y == 0) g_acq_stamps[8] = wall_clock64();
#endif
  // publish: every thread's host writes are out before the sequence words follow
  __threadfence_system();
  __syncthreads();
  if (hm && tid < nq && s_flag[tid])
    __hip_atomic_store(const_cast<unsigned long long*>(&hm->qflag[q0 + tid]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef PCABO_ACQ_TIMING
  if (threadIdx.x == 0 && blockIdx.y == 0) g_acq_stamps[9] = wall_clock64();
#endif
}

bool acq_group_possible(int NP, int k) { return NP <= 1280 && k <= PCABO_MAXD; }

// Returns 0, or -1 when nothing was launched (the caller must not wait for per-query flags then).
int launch_acq_group(hipStream_t st, const QueryArgs* tab, int entries, const double* Xq, int n, int k, int NP, int ld,
                     const double* ZnT, const double* R, const double* alpha, const double* bounds4, const double* ystats,
                     AcqParams p, double* partial, unsigned int* counters, double* val, double* grad, double* host_val,
                     double* host_grad, HostMirror* hm, unsigned long long seq, AcqBatch ab) {
  const size_t lds = ((size_t)GQ * NP + 4 * 64 * GT_LD + 64 * 8 + GQ * 64 + 2 * GQ + 4 + 2 * GQ + GQ * (((size_t)k + 1) & ~(size_t)1) + 8) * sizeof(double);
  const dim3 grid(NP / 64, entries), block(256);
  // Dynamic LDS beyond the 64 KB default needs the attribute - per DEVICE, and for every instantiation that can ask for
  // more: <2> does from (NP, k) = (512, 83) on (65 808 bytes at k = 89), <5> always.  Launched from the worker threads of
  // a batch, hence the lock; on a failure nothing is launched and the caller is told (as launch_lbfgsb_group does).
  {
    static std::mutex attr_mu;
    static bool attr_done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!attr_done[dev]) {
      if (hipFuncSetAttribute((const void*)k_acq_group<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_acq_group<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
        return -1;
      attr_done[dev] = true;
    }
  }
  if (lds > 96 * 1024) return -1;        // (cannot happen for NP <= 1280, k <= 128: 98 048 bytes at most)
#define GROUP_ARGS *tab, Xq, n, k, NP, ld, ZnT, R, alpha, bounds4, ystats, p, partial, counters, val, grad, host_val, host_grad, \
                   hm, seq, ab
  if (NP <= 256) hipLaunchKernelGGL(k_acq_group<1>, grid, block, lds, st, GROUP_ARGS);
  else if (NP <= 512) hipLaunchKernelGGL(k_acq_group<2>, grid, block, lds, st, GROUP_ARGS);
  else hipLaunchKernelGGL(k_acq_group<5>, grid, block, lds, st, GROUP_ARGS);
#undef GROUP_ARGS
  return 0;
}

// =====================================================================================================================
// Large value-only batches (the 512 raw samples of gen_batch_initial_conditions) as a GEMM:
//   KS[q][j] = k(x_q, z_j)                      k_score_ks:   thread per training point, 16 queries per work-group
//   V = R KS^T  (n x q),  |v_q|^2 = column norms  k_score_gemm: 64 x 64 output tiles on v_mfma_f64_16x16x4, operand
//                                                tiles through LDS (leading dimension 66: conflict-free), the next tiles
//                                                travelling in registers while the MFMAs of the current ones run
// Only the column norms leave the kernel (V itself is never stored): per (query, 64-row block) one partial record, the
// same records k_acq_combine sums for the slab kernels.  mu_s = alpha . ks is formed where ks is, in k_score_ks.
// 0.10 GFLOP per run at n = 450: one work-group per (row block, 64 queries) instead of one per (16-row slab, 8 queries).
#define SC_QB 16
__global__ __launch_bounds__(256) void k_score_ks(const double* __restrict__ Xq, int q_total, int n, int k, int NP, int ld,
                                                  const double* __restrict__ ZnT, const double* __restrict__ alpha,
                                                  const double* __restrict__ bounds4, AcqParams prm,
                                                  double* __restrict__ KS, double* __restrict__ partial, int S, AcqBatch ab) {
  if (ab.zs) {
    const unsigned run_ = blockIdx.z;
    Xq = zrun(Xq, ab.zs, run_); ZnT = zrun(ZnT, ab.zs, run_); alpha = zrun(alpha, ab.zs, run_);
    bounds4 = zrun(bounds4, ab.zs, run_); KS = zrun(KS, ab.zs, run_); partial = zrun(partial, ab.zs, run_);
    if (ab.k_dev) k = *zrun(ab.k_dev, ab.zs, run_);
  }
  __shared__ double s_xn[SC_QB][PCABO_MAXD];
  __shared__ double s_mu[4][SC_QB];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int j = blockIdx.x * 256 + tid, q0 = blockIdx.y * SC_QB;
  for (int idx = tid; idx < SC_QB * k; idx += 256) {
    const int q = idx / k, c = idx - q * k;
    const int qq = q0 + q < q_total ? q0 + q : q_total - 1;
    const double lo = bounds4[c], hi = bounds4[PCABO_MAXD + c];
    s_xn[q][c] = (Xq[(size_t)qq * k + c] - lo) / (hi - lo);
  }
  __syncthreads();
  double sq[SC_QB];
#pragma unroll
  for (int q = 0; q < SC_QB; ++q) sq[q] = 0.0;
  if (j < n) {
    for (int c0 = 0; c0 < k; c0 += 8) {
      double z[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = (c0 + u < k) ? ZnT[(size_t)(c0 + u) * ld + j] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + u < k) {
#pragma unroll
          for (int q = 0; q < SC_QB; ++q) { const double d = s_xn[q][c0 + u] - z[u]; sq[q] += d * d; }
        }
    }
  }
  const double inv_ls = prm.inv_ls, s5 = 2.23606797749979;
  const double aj = j < n ? alpha[j] : 0.0;
#pragma unroll
  for (int q = 0; q < SC_QB; ++q) {
    double ks = 0.0;
    if (j < n) {
      const double sqq = sq[q] * (inv_ls * inv_ls);
      if (prm.kernel == 1) ks = exp(-0.5 * sqq);
      else {
        const double dist = sqrt(fmax(sqq, 1e-30));
        ks = ((s5 * dist + 1.0) + (5.0 / 3.0) * (dist * dist)) * exp(-s5 * dist);
      }
    }
    if (j < NP && q0 + q < q_total) KS[(size_t)(q0 + q) * ld + j] = ks;
    const double m = wave_sum(aj * ks);
    if (l == 0) s_mu[w][q] = m;
  }
  __syncthreads();
  // mu_s part of this block of 256 points -> slot 1 of the query's record number blockIdx.x (the GEMM zeroes the others')
  if (tid < SC_QB && q0 + tid < q_total)
    partial[((size_t)(q0 + tid) * S + blockIdx.x) * PSTRIDE + 1] = ((s_mu[0][tid] + s_mu[1][tid]) + s_mu[2][tid]) + s_mu[3][tid];
}

__global__ __launch_bounds__(256) void k_score_gemm(const double* __restrict__ R, const double* __restrict__ KS,
                                                    int q_total, int NP, int ld, int njb, double* __restrict__ partial,
                                                    size_t zs) {
  ZRUN(R); ZRUN(KS); ZRUN(partial);
  __shared__ __attribute__((aligned(16))) double s_a[64 * PCABO_TLD];
  __shared__ __attribute__((aligned(16))) double s_b[64 * PCABO_TLD];
  __shared__ double s_red[4][64];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int I = blockIdx.x, qb = blockIdx.y, S = gridDim.x;
  const double* Rrow = R + (size_t)(I * 64) * ld;
  const double* Krow = KS + (size_t)(qb * 64) * ld;
  double pa[16], pb[16];
  auto fetch = [&](int J) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
      pa[u] = Rrow[(size_t)r * ld + J * 64 + c];
      pb[u] = (qb * 64 + r < q_total) ? Krow[(size_t)r * ld + J * 64 + c] : 0.0;
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = tid + 256 * u, r = idx >> 6, c = idx & 63;
      s_a[r * PCABO_TLD + c] = pa[u];
      s_b[r * PCABO_TLD + c] = pb[u];
    }
  };
  double4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
  fetch(0);
  for (int J = 0; J <= I; ++J) {              // R is lower triangular: column blocks beyond the diagonal hold zeros
    __syncthreads();                          // the previous step's MFMAs have read the LDS tiles
    put();
    if (J < I) fetch(J + 1);
    __syncthreads();
    for (int kk = 0; kk < 64; kk += 4) {
      const double a = s_a[(16 * w + (l & 15)) * PCABO_TLD + kk + (l >> 4)];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double b = s_b[(16 * t + (l & 15)) * PCABO_TLD + kk + (l >> 4)];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // column norms of this 64-row block: acc[t][r] = V[16 w + (l >> 4) + 4 r][16 t + (l & 15)]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    double v = ((acc[t][0] * acc[t][0] + acc[t][1] * acc[t][1]) + acc[t][2] * acc[t][2]) + acc[t][3] * acc[t][3];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (l < 16) s_red[w][16 * t + l] = v;
  }
  __syncthreads();
  if (tid < 64 && qb * 64 + tid < q_total) {
    double* rec = partial + ((size_t)(qb * 64 + tid) * S + I) * PSTRIDE;
    rec[0] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
    if (I >= njb) rec[1] = 0.0;               // (records 0 .. njb-1 got their mu_s part from k_score_ks)
  }
}

bool score_gemm_possible(int q) { return q >= 64; }

// Value-only scoring of q >= 64 points: KS, then V = R KS^T on MFMA, then the scalar chain per query (k_acq_combine).
void launch_score(hipStream_t st, const double* Xq, int q, int n, int k, int NP, int ld, const double* ZnT, const double* R,
                  const double* alpha, const double* bounds4, const double* ystats, AcqParams p, double* KS, double* partial,
                  double* val, AcqBatch ab, int B) {
  const int S = NP / 64, njb = (NP + 255) / 256;
  hipLaunchKernelGGL(k_score_ks, dim3(njb, (q + SC_QB - 1) / SC_QB, B), dim3(256), 0, st, Xq, q, n, k, NP, ld, ZnT, alpha, bounds4,
                     p, KS, partial, S, ab);
  hipLaunchKernelGGL(k_score_gemm, dim3(S, (q + 63) / 64, B), dim3(256), 0, st, R, KS, q, NP, ld, njb, partial, ab.zs);
  p.want_grad = 0;
  hipLaunchKernelGGL(k_acq_combine, dim3((q + 3) / 4, 1, B), dim3(256), 0, st, partial, q, S, k, bounds4, ystats, p, val,
                     (double*)nullptr, ab);
}

// Resident mode needs every group of the grid on the chip at the same time (groups wait for one another through the
// mailbox and the tickets): one group per CU is always possible for these kernels, so S q <= number of CUs is enough.
bool acq_server_possible(int q, int n, int k, int NP) {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  }
  if (k > 40 || NP > 512 || q < 1 || q > PCABO_INLAUNCH_MAXQ || (size_t)q * k > PCABO_QA_MAX) return false;
  const int S = acq_slabs(NP);
  return (S + 1) * q <= cus && S <= 32;      // slab groups + one finishing group per query (acq_server_finish_main)
}

void launch_acq(hipStream_t st, const QueryArgs* qa, const double* Xq, int q, int n, int k, int NP, int ld,
                const double* ZnT, const double* R, const double* alpha, const double* bounds4, const double* ystats,
                AcqParams p, double* partial, unsigned int* counters, double* val,
                double* grad, double* host_val, double* host_grad, HostMirror* hm, unsigned long long seq,
                MailPair* dev_mail, MailPair* part_pairs, AcqBatch ab, int B,
                int table_entries) {
  // 16 rows per work-group while S*q groups fit the 256 CUs (NP <= 384 at q = 10), 32 rows beyond that: measured
  // on MI355X (q=10, with gradient) 16 rows win at n=120/250 (22.2 vs 23.0, 26.9 vs 27.8 us), 32 rows at n=449 (34.4 vs
  // 37.3 us).
  const int slab = NP >= ACQ_SLAB32_NP ? 32 : 16;
  const int S = NP / slab;
  static const QueryArgs empty = {};
  const int combine = hm != nullptr;      // small batches: finish inside the launch and publish to the host
  const int nb = NP / 64;
  // large batches on the fast path: 8 queries per group share one load of the register tiles (qb); the generic
  // kernel and the in-launch combine take one query per group
  const bool fast = k <= 40 && nb <= 8;      // (the resident mode is only requested when this holds)
  const int qb = (fast && !combine && q >= 64) ? 8 : 1;
  // batched: table mode -> one grid row per active (run, query) entry; otherwise grid.z = run
  const int gy = ab.table ? table_entries : (q + qb - 1) / qb;
  const int gz = ab.table ? 1 : B;
#define ACQ_ARGS qa ? *qa : empty, Xq, q, n, k, NP, ld, ZnT, R, alpha, bounds4, ystats, p, partial, counters, val, grad, \
                 host_val, host_grad, hm, seq, combine, qb, dev_mail, part_pairs, ab
  const int FIN_LDS = 6 * PCABO_MAXD + 2;   // finishing group's LDS beyond s_v
#define ACQ_LDS(SL, NBV) (3 * NBV * 64 + PCABO_MAXD + SL + 2 + (4 * NBV * 64 > FIN_LDS ? 4 * NBV * 64 : FIN_LDS))
#define ACQ_FAST(SL, NBV)                                                                                      \
  case NBV:                                                                                                    \
    if (dev_mail)                                                                                              \
      hipLaunchKernelGGL((k_acq_fast<SL, NBV, true>), dim3(S + 1, gy), dim3(256),                              \
                         (size_t)std::max<int>(ACQ_LDS(SL, NBV), FIN_LDS_DOUBLES) * sizeof(double), st, ACQ_ARGS); \
    else                                                                                                       \
      hipLaunchKernelGGL((k_acq_fast<SL, NBV, false>), dim3(S, gy, gz), dim3(256),                             \
                         (size_t)ACQ_LDS(SL, NBV) * sizeof(double), st, ACQ_ARGS);                             \
    break;
  if (fast) {
    if (slab == 16) {
      switch (nb) { ACQ_FAST(16, 1) ACQ_FAST(16, 2) ACQ_FAST(16, 3) ACQ_FAST(16, 4) ACQ_FAST(16, 5) ACQ_FAST(16, 6)
                    ACQ_FAST(16, 7) ACQ_FAST(16, 8) }
    } else {
      switch (nb) { ACQ_FAST(32, 1) ACQ_FAST(32, 2) ACQ_FAST(32, 3) ACQ_FAST(32, 4) ACQ_FAST(32, 5) ACQ_FAST(32, 6)
                    ACQ_FAST(32, 7) ACQ_FAST(32, 8) }
    }
  } else {
    size_t lds = (size_t)(3 * NP + PCABO_MAXD + 32 + 2 + 6 * PCABO_MAXD + 2) * sizeof(double);
    if (slab == 16)
      hipLaunchKernelGGL(k_acq_fused<16>, dim3(S, ab.table ? gy : q, gz), dim3(256), lds, st, ACQ_ARGS);
    else
      hipLaunchKernelGGL(k_acq_fused<32>, dim3(S, ab.table ? gy : q, gz), dim3(256), lds, st, ACQ_ARGS);
  }
#undef ACQ_FAST
#undef ACQ_LDS
#undef ACQ_ARGS
  if (!combine)
    hipLaunchKernelGGL(k_acq_combine, dim3((q + 3) / 4, 1, gz), dim3(256), 0, st, partial, q, S, k, bounds4, ystats, p, val,
                       grad, ab);
}
