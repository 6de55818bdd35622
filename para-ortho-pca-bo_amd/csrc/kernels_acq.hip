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
// Work decomposition (latency-bound: ~1 MFLOP per query at n = 450):
//   k_acq_partial  grid (S, q): work-group (s, q) owns 16 rows of R (8 from the top, 8 mirrored from
//                  the bottom -> balanced triangular work) for query q.  It recomputes ks (n*k
//                  flops, cheaper than a launch boundary), forms its 16 entries of v by
//                  wave-per-row shuffle reductions, its contribution R_slab^T v_slab to w, and
//                  contracts that with dks/dx.  Because the gradient is linear in w, partial
//                  gradients of different slabs simply add.  No inter-group communication.
//   k_acq_combine  one wave per query: fixed-order sum over the S partials (deterministic), then the
//                  scalar log-EI chain rule; results go to device buffers and, for the L-BFGS-B loop,
//                  straight into pinned host memory followed by a sequence flag.
#include "pcabo_internal.h"

#define SLAB PCABO_SLAB
#define PSTRIDE (2 + 2 * PCABO_MAXD)   // doubles per (query, slab) partial record

__device__ inline int slab_row(int s, int m, int NP) { return m < 8 ? 8 * s + m : NP - 8 * (s + 1) + (m - 8); }

__global__ __launch_bounds__(256) void k_acq_partial(
    const double* __restrict__ Xq, int n, int k, int NP, int ld, const double* __restrict__ ZnT,
    const double* __restrict__ R, const double* __restrict__ alpha, const double* __restrict__ bounds4,
    double inv_ls, int kernel, int want_grad, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* s_ks = s_dyn;              // NP
  double* s_cf = s_dyn + NP;         // NP  coef_j (dks_j/dxn = coef_j (xn - zn_j))
  double* s_tm = s_dyn + 2 * NP;     // NP  alpha_j coef_j restricted to this slab's rows
  double* s_xn = s_dyn + 3 * NP;     // PCABO_MAXD
  double* s_v = s_xn + PCABO_MAXD;   // SLAB
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int s = blockIdx.x, q = blockIdx.y, S = gridDim.x;
  double* out = partial + ((size_t)q * S + s) * PSTRIDE;

  if (tid < k) {
    double lo = bounds4[tid], hi = bounds4[PCABO_MAXD + tid];
    s_xn[tid] = (Xq[(size_t)q * k + tid] - lo) / (hi - lo);
  }
  __syncthreads();
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
  // v_i = R[i][0..i] . ks for the slab's 16 rows: one wave per row, 64-lane shuffle reduction
  for (int m = w; m < SLAB; m += 4) {
    const int i = slab_row(s, m, NP);
    const double* Ri = R + (size_t)i * ld;
    double acc = 0.0;
    for (int j = l; j <= i; j += 64) acc += Ri[j] * s_ks[j];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (l == 0) s_v[m] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    double vv = 0.0, mu = 0.0;
    for (int m = 0; m < SLAB; ++m) {
      vv += s_v[m] * s_v[m];
      int i = slab_row(s, m, NP);
      if (i < n) mu += alpha[i] * s_ks[i];
    }
    out[0] = vv;
    out[1] = mu;
  }
  if (!want_grad) return;
  __syncthreads();   // thread 0 has finished reading s_ks before it is reused below
  // w_j (slab part) = sum_{i in slab, i >= j} R[i][j] v_i ; fold in coef_j.  Rows are re-read
  // column-wise here (coalesced over j).
  for (int j = tid; j < NP; j += 256) {
    double wj = 0.0;
    for (int m = 0; m < SLAB; ++m) {
      const int i = slab_row(s, m, NP);
      if (i >= j) wj += R[(size_t)i * ld + j] * s_v[m];
    }
    const bool mine = ((j >> 3) == s) || (((NP - 1 - j) >> 3) == s);
    const double cf = s_cf[j];
    s_tm[j] = (mine && j < n) ? alpha[j] * cf : 0.0;
    s_ks[j] = wj * cf;                    // ks no longer needed: reuse as t_sigma
  }
  __syncthreads();
  // contraction with (xn_c - zn_jc): wave w handles components c = w, w+4, ...
  for (int c = w; c < k; c += 4) {
    const double xc = s_xn[c];
    const double* zrow = ZnT + (size_t)c * ld;
    double gs = 0.0, gm = 0.0;
    for (int j = l; j < n; j += 64) {
      double dlt = xc - zrow[j];
      gs += s_ks[j] * dlt;
      gm += s_tm[j] * dlt;
    }
    for (int off = 32; off > 0; off >>= 1) { gs += __shfl_xor(gs, off, 64); gm += __shfl_xor(gm, off, 64); }
    if (l == 0) { out[2 + c] = gs; out[2 + PCABO_MAXD + c] = gm; }
  }
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
    double ex = erfcx(-inv_sqrt2 * u);
    double wv = log(ex * fabs(u)) + 0.22579135264472738;       // + log(pi/2)/2
    double l1m = (-0.6931471805599453 < wv) ? log(-expm1(wv)) : log1p(-exp(wv));
    *h = log_phi + l1m;
    double dw = (u + 0.7978845608028654 / ex) + 1.0 / u;       // sqrt(2/pi)/erfcx + u + 1/u
    *dh = -u - dw / expm1(-wv);
  } else {
    *h = log_phi - 2.0 * log(fabs(u));
    *dh = -u - 2.0 / u;
  }
}

// One wave per query; lanes over reduced components (c, c + 64).
__global__ __launch_bounds__(1024) void k_acq_combine(
    const double* __restrict__ partial, int q_total, int S, int k, const double* __restrict__ bounds4,
    const double* __restrict__ ystats, AcqParams p, double* __restrict__ val, double* __restrict__ grad,
    double* host_val, double* host_grad, HostMirror* hm, unsigned long long seq) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int q = blockIdx.x * (blockDim.x >> 6) + w;
  if (q < q_total) {
    const double* base = partial + (size_t)q * S * PSTRIDE;
    double vv = 0.0, mus = 0.0;
    for (int s = 0; s < S; ++s) { vv += base[(size_t)s * PSTRIDE]; mus += base[(size_t)s * PSTRIDE + 1]; }
    const double ym = ystats[0], ysd = ystats[1];
    const double mu = ym + ysd * mus;
    double var = (1.0 - vv) * (ysd * ysd);
    bool clamped = false;
    if (!(var >= 1e-10)) { var = 1e-10; clamped = true; }     // gpytorch min_variance (double)
    if (var < 1e-12) { var = 1e-12; clamped = true; }          // botorch _mean_and_sigma(min_var)
    const double sigma = sqrt(var);
    double u = (mu - p.best_f) / sigma;
    const double sgn = p.maximize ? 1.0 : -1.0;
    u *= sgn;
    double value, dv_du, dv_dsig_over;   // d value/du and explicit d value/d sigma * 1 (log sigma term)
    if (p.acq == 0) {
      double h, dh;
      log_ei_helper(u, &h, &dh);
      value = h + log(sigma);
      dv_du = dh;
      dv_dsig_over = 1.0 / sigma;
    } else {
      value = 0.5 * erfc(-0.7071067811865476 * u);
      dv_du = 0.3989422804014327 * exp(-0.5 * u * u);
      dv_dsig_over = 0.0;
    }
    if (l == 0) { val[q] = value; if (host_val) host_val[q] = value; }
    if (p.want_grad) {
      // du = sgn dmu/sigma - u dsigma/sigma ; dsigma = -s_y^2 g_sigma / sigma (0 if clamped)
      const double c_mu = dv_du * sgn * ysd / sigma;
      const double c_sg = clamped ? 0.0 : (dv_dsig_over - dv_du * u / sigma) * (-(ysd * ysd) / sigma);
      for (int c = l; c < k; c += 64) {
        double gs = 0.0, gm = 0.0;
        for (int s = 0; s < S; ++s) {
          gs += base[(size_t)s * PSTRIDE + 2 + c];
          gm += base[(size_t)s * PSTRIDE + 2 + PCABO_MAXD + c];
        }
        double g = (c_mu * gm + c_sg * gs) / (bounds4[PCABO_MAXD + c] - bounds4[c]);
        grad[(size_t)q * k + c] = g;
        if (host_grad) host_grad[(size_t)q * k + c] = g;
      }
    }
  }
  if (hm && seq) {           // single-block launches only: publish the sequence number last
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(const_cast<unsigned long long*>(&hm->flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

void launch_acq_partial(hipStream_t st, const double* Xq, int q, int n, int k, int NP, int ld, const double* ZnT,
                        const double* R, const double* alpha, const double* bounds4, AcqParams p, double* partial) {
  const int S = NP / SLAB;
  size_t lds = (size_t)(3 * NP + PCABO_MAXD + SLAB) * sizeof(double);
  hipLaunchKernelGGL(k_acq_partial, dim3(S, q), dim3(256), lds, st, Xq, n, k, NP, ld, ZnT, R, alpha, bounds4,
                     p.inv_ls, p.kernel, p.want_grad, partial);
}

void launch_acq_combine(hipStream_t st, int q, int k, int NP, const double* bounds4, const double* ystats, AcqParams p,
                        const double* partial, double* val, double* grad, double* host_val, double* host_grad,
                        HostMirror* hm, unsigned long long seq) {
  const int S = NP / SLAB;
  if (q <= 16) {
    hipLaunchKernelGGL(k_acq_combine, dim3(1), dim3(64 * q), 0, st, partial, q, S, k, bounds4, ystats, p, val, grad,
                       host_val, host_grad, hm, seq);
  } else {
    hipLaunchKernelGGL(k_acq_combine, dim3((q + 3) / 4), dim3(256), 0, st, partial, q, S, k, bounds4, ystats, p, val,
                       grad, host_val, host_grad, (HostMirror*)nullptr, 0ull);
  }
}
