// Entry points of include/pcabo.h that are host code only (no HIP call, no device): the L-BFGS-B driver the tests pin against
// scipy, the switch of its vector kernels and the Sobol helpers (torch.quasirandom.SobolEngine restated; reference call site:
// botorch's gen_batch_initial_conditions behind /root/reference/Algorithms/BayesianOptimization/PCA_BO.py:607-614).
// Built by g++ into libpcabo.so and, with -fsanitize=..., into the host-only checker libraries of the Makefile.
#include "../../include/pcabo.h"
#include "lbfgsb.h"

#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

int pcabo_lbfgsb_minimize(int nvar, double* x, const double* lower, const double* upper, pcabo_fg_callback fg,
                          void* user, int m, double factr, double pgtol, int maxiter, int maxfun, int maxls,
                          double* f_out, int* nit, int* nfev, int* task_out) {
  if (nvar < 1 || !x || !fg || m < 1) return PCABO_ERR_ARG;
  Lbfgsb opt;
  opt.init(nvar, m, lower, upper, factr, pgtol, maxls);
  if (lower && upper)
    for (int i = 0; i < nvar; ++i) x[i] = x[i] < lower[i] ? lower[i] : (x[i] > upper[i] ? upper[i] : x[i]);
  std::vector<double> g(nvar, 0.0), xc, gc;
  double f = 0.0, fc = 0.0;
  int iters = 0, evals = 0;
  while (true) {
    int task = opt.step(x, &f, g.data());
    if (task == LBFGSB_FG) {
      if (!xc.empty() && memcmp(x, xc.data(), nvar * sizeof(double)) == 0) { f = fc; g = gc; continue; }   // scipy's memoisation
      f = fg(x, g.data(), user); ++evals;
      xc.assign(x, x + nvar); gc = g; fc = f;
      continue;
    }
    if (task == LBFGSB_NEW_X) {
      ++iters;
      if (iters >= maxiter) opt.stop(LBFGSB_STOP_ITER);
      else if (evals > maxfun) opt.stop(LBFGSB_STOP_FUN);
      continue;
    }
    break;
  }
  if (f_out) *f_out = f;
  if (nit) *nit = iters;
  if (nfev) *nfev = evals;
  if (task_out) *task_out = opt.task();
  return opt.warnflag();
}

int pcabo_lbfgsb_set_vector_kernels(int enabled) { return lbfgsb_set_vector_kernels(enabled); }
int pcabo_lbfgsb_set_sum_order(int order) { return lbfgsb_set_default_sum_order(order); }

int pcabo_sobol_scramble(int64_t* state, const int64_t* ltm, int k) {
  if (!state || !ltm || k < 1) return PCABO_ERR_ARG;
  const int MAXBIT = 30;
  for (int d = 0; d < k; ++d) {
    // matrix d over GF(2): unit diagonal, entries below it from ltm, nothing above (whatever ltm holds there - the caller
    // need not clear the upper triangle).  col[c] = column c as a bit vector, row p at bit MAXBIT-1-p.
    int64_t col[30];
    for (int c = 0; c < MAXBIT; ++c) col[c] = (int64_t)1 << (MAXBIT - 1 - c);
    for (int p = 1; p < MAXBIT; ++p) {
      const int64_t* row = ltm + ((size_t)d * MAXBIT + p) * MAXBIT;
      const int64_t pbit = (int64_t)1 << (MAXBIT - 1 - p);
      for (int c = 0; c < p; ++c) col[c] |= (row[c] & 1) ? pbit : 0;
    }
    for (int j = 0; j < MAXBIT; ++j) {                  // state <- matrix * state, bit MAXBIT-1-c of the state = component c
      unsigned long long v = (unsigned long long)state[(size_t)d * MAXBIT + j] & ((1ull << MAXBIT) - 1);
      int64_t t2 = 0;
      while (v) {
        const int b = __builtin_ctzll(v);
        t2 ^= col[MAXBIT - 1 - b];
        v &= v - 1;
      }
      state[(size_t)d * MAXBIT + j] = t2;
    }
  }
  return PCABO_OK;
}

// n points of a FRESH scrambled Sobol engine (state after pcabo_sobol_scramble, shift = the engine's shift vector), as
// torch.quasirandom.SobolEngine.draw(n, dtype=float64) returns them for num_generated = 0 (first point = float32(shift) / 2^30, then
// Gray-code steps: the state column of the lowest zero bit of the running index is XORed in), mapped into a box as botorch's
// draw_sobol_samples does: out[i][j] = lo[j] + rng[j] * u[i][j] (multiply, then add: two roundings, as torch's two kernels).
// lo = rng = NULL: u itself.
int pcabo_sobol_draw(const int64_t* state, const int64_t* shift, int k, int n, const double* lo, const double* rng, double* out) {
  if (!state || !shift || !out || k < 1 || n < 1 || (lo == nullptr) != (rng == nullptr)) return PCABO_ERR_ARG;
  const int MAXBIT = 30;
  const double recip = 9.31322574615478515625e-10;      // 2^-30
  std::vector<int64_t> q(shift, shift + k);
  for (int i = 0; i < n; ++i) {
    if (i > 0) {
      unsigned v = (unsigned)(i - 1);
      int l = 0;
      while (v & 1u) { v >>= 1; ++l; }
      if (l >= MAXBIT) return PCABO_ERR_ARG;
      for (int j = 0; j < k; ++j) q[j] ^= state[(size_t)j * MAXBIT + l];
    }
    double* o = out + (size_t)i * k;
    for (int j = 0; j < k; ++j) {
      // (torch keeps the FIRST point as `quasi / 2**30` of an int64 tensor: a float32 tensor - that point has 24 bits)
      const double u = i == 0 ? (double)((float)q[j]) * recip : (double)q[j] * recip;
      if (lo) { const double t = rng[j] * u; o[j] = lo[j] + t; }
      else o[j] = u;
    }
  }
  return PCABO_OK;
}


}  // extern "C"
