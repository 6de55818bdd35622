// Entry points of include/pcabo.h that are host code only (no HIP call, no device): the L-BFGS-B driver the tests pin against
// scipy, the switch of its vector kernels and the Sobol helpers (torch.quasirandom.SobolEngine restated; reference call site:
// botorch's gen_batch_initial_conditions behind /root/reference/Algorithms/BayesianOptimization/PCA_BO.py:607-614).
// Built by g++ into libpcabo.so and, with -fsanitize=..., into the host-only checker libraries of the Makefile.
#include "../../include/pcabo.h"
#include "lbfgsb.h"

#include <algorithm>
#include <cmath>
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


// pcabo_sobol_draw for the runs of a lock-step batch in one call: run r's engine (states[r], shifts[r], ks[r] columns) draws n
// points into outs[r] (n x ks[r], contiguous), mapped into the box whose corners lie in boxes + r * box_stride as [lo(k), hi(k)]
// (the packing pcabo_batch_acq_bounds writes); rng = hi - lo is formed here, the subtraction the caller's numpy made.  A run with
// states[r] == NULL is skipped.  Same numbers as `rows` calls of pcabo_sobol_draw.
int pcabo_sobol_draw_rows(const int64_t* const* states, const int64_t* const* shifts, const int* ks, int rows, int n,
                          const double* boxes, long long box_stride, double* const* outs) {
  if (!states || !shifts || !ks || !boxes || !outs || rows < 1 || n < 1 || n > (1 << 30)) return PCABO_ERR_ARG;
  for (int r = 0; r < rows; ++r)
    if (states[r] && (ks[r] < 1 || !shifts[r] || !outs[r])) return PCABO_ERR_ARG;
  std::vector<double> rng;
  for (int r = 0; r < rows; ++r) {
    if (!states[r]) continue;
    const int k = ks[r];
    const double* lo = boxes + (size_t)r * box_stride;
    rng.resize(k);
    for (int j = 0; j < k; ++j) rng[j] = lo[k + j] - lo[j];
    pcabo_sobol_draw(states[r], shifts[r], k, n, lo, rng.data(), outs[r]);       // (arguments checked above: cannot fail)
  }
  return PCABO_OK;
}

// ---- torch's CPU generator, restated (the host's pacing thread draws the reference's random numbers without torch calls) ----
// The reference's restart heuristic draws from torch's global CPU generator: the Sobol scramble bits (torch.randint(2, ...)) and
// botorch's initialize_q_batch -> torch.multinomial(weights, n, replacement=False) (behind PCA_BO.py:607-614).  With hundreds of
// runs per host thread those two calls were 40 % of the thread's time.  torch's generator is a 32-bit Mersenne Twister
// (at::mt19937: the published MT19937 with `left` / `next` bookkeeping); `blob` is its state exactly as
// torch.Generator.get_state() exports it (CPUGeneratorImplStateLegacy: uint64 seed, int left, int seeded, uint64 next, uint64
// state[624], ...), read and ADVANCED in place, so set_state(blob) puts torch's own generator where torch's own calls would
// have left it.  pcabo/hostrng.py checks both functions against torch when it is imported and falls back to torch on a mismatch.
namespace {
struct TorchMt {
  uint64_t seed; int32_t left; int32_t seeded; uint64_t next; uint64_t state[624];
};
inline uint32_t mt_twist(uint64_t u, uint64_t v) {
  return (uint32_t)((((uint32_t)u & 0x80000000u) | ((uint32_t)v & 0x7fffffffu)) >> 1) ^ (((uint32_t)v & 1u) ? 0x9908b0dfu : 0u);
}
// (index form, blocks shorter than the recurrence's reach of N - M = 227 words: no iteration of a block reads what the block writes
// except st[i + 1] of its last element, read before it is written - the compiler may vectorise)
__attribute__((optimize("O3"))) inline void mt_next_state(TorchMt* s) {
  constexpr int N = 624, M = 397;
  uint64_t* __restrict__ st = s->state;
  for (int i = 0; i < N - M; ++i) st[i] = (uint32_t)st[i + M] ^ mt_twist(st[i], st[i + 1]);
  for (int i = N - M; i < N - 1; ++i) st[i] = (uint32_t)st[i + M - N] ^ mt_twist(st[i], st[i + 1]);
  st[N - 1] = (uint32_t)st[M - 1] ^ mt_twist(st[N - 1], st[0]);
  s->left = N;
  s->next = 0;
}
inline uint32_t mt_draw(TorchMt* s) {
  if (--s->left == 0) mt_next_state(s);
  uint32_t y = (uint32_t)s->state[s->next++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
// the low bit of the tempered words w[0 .. m)
__attribute__((optimize("O3"))) inline void mt_temper_bits(const uint64_t* w, int64_t m, int64_t* out) {
  for (int64_t i = 0; i < m; ++i) {
    uint32_t y = (uint32_t)w[i];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    out[i] = (int64_t)(y & 1u);
  }
}
}  // namespace

// torch.randint(2, (count,), generator=g): one 32-bit draw per element, value = draw % 2 (int64 out).
int pcabo_torch_randint2(void* blob, int64_t count, int64_t* out) {
  if (!blob || !out || count < 0) return PCABO_ERR_ARG;
  TorchMt* s = static_cast<TorchMt*>(blob);
  if (!s->seeded || s->left < 1 || s->left > 624 || s->next > 624) return PCABO_ERR_ARG;
  // whole runs of state words at a time (the tempering of a run is a loop without dependencies: the compiler vectorises it);
  // the bookkeeping is mt_draw's: a fresh state serves N draws, `left` stays N after the first of them
  while (count > 0) {
    int64_t take;
    if (s->left == 1) {
      mt_next_state(s);
      take = count < 624 ? count : 624;
      mt_temper_bits(s->state + s->next, take, out);
      s->next += (uint64_t)take;
      s->left = 624 - (int32_t)(take - 1);
    } else {
      take = count < s->left - 1 ? count : s->left - 1;
      mt_temper_bits(s->state + s->next, take, out);
      s->next += (uint64_t)take;
      s->left -= (int32_t)take;
    }
    out += take;
    count -= take;
  }
  return PCABO_OK;
}

namespace {
// the n_pick largest of w_i / q_i, largest first (torch's topk on the ratios), q_i an Exp(1) variate drawn as torch's exponential_ does
void multinomial_row(TorchMt* s, const double* w, int n, int n_pick, double* ratio, int* best, int64_t* out) {
  for (int i = 0; i < n; ++i) {
    const uint64_t hi = mt_draw(s), lo = mt_draw(s);
    const uint64_t r64 = (hi << 32) | lo;
    const double u = (double)(r64 & ((1ull << 53) - 1)) * 1.1102230246251565e-16;      // 2^-53
    const double q = -std::log1p(-u);
    ratio[i] = w[i] / q;
  }
  int have = 0;                        // (n_pick is 10 of 512: selection by insertion)
  for (int i = 0; i < n; ++i) {
    const double v = ratio[i];
    if (have == n_pick && !(v > ratio[best[have - 1]])) continue;
    int pos = have < n_pick ? have : n_pick - 1;
    while (pos > 0 && v > ratio[best[pos - 1]]) { if (pos < n_pick) best[pos] = best[pos - 1]; --pos; }
    best[pos] = i;
    if (have < n_pick) ++have;
  }
  for (int j = 0; j < n_pick; ++j) out[j] = best[j];
}
}  // namespace

// botorch's initialize_q_batch for `rows` runs at once (behind PCA_BO.py:607-614): per row z = (v - mean) / std (unbiased), weights
// exp(eta z) (halved exponents while one overflows), torch.multinomial(weights, n_pick) on the row's generator, the arg-max forced
// into the last place if it was not drawn.  flags[r]: 0 picked; 1 all values equal (std == 0: the caller takes botorch's random
// permutation path on torch's generator - nothing was drawn here); 2 row skipped (blobs[r] == NULL).  The statistics are formed
// in this file's order (Welford), not torch's: the weights can differ from a torch-formed row in the last bit, the picks - an
// ordering of weights over independent exponential variates - do not (tests/test_abi_and_host.py compares 10 000 rows).
namespace {
// rows [r0, r1) of pcabo_boltzmann_pick_rows
void boltzmann_rows(void* const* blobs, const double* vals, int r0, int r1, int n, int n_pick, double eta, int64_t* out, int* flags) {
  std::vector<double> w((size_t)n), ratio((size_t)n);
  std::vector<int> best((size_t)n_pick);
  for (int r = r0; r < r1; ++r) {
    TorchMt* s = static_cast<TorchMt*>(blobs[r]);
    if (!s) { flags[r] = 2; continue; }
    const double* v = vals + (size_t)r * n;
    double mean = 0.0, m2 = 0.0;
    int arg = 0;
    for (int i = 0; i < n; ++i) {
      const double d = v[i] - mean;
      mean += d / (double)(i + 1);
      m2 += d * (v[i] - mean);
      if (v[i] > v[arg]) arg = i;
    }
    const double sd = std::sqrt(m2 / (double)(n - 1));
    if (!(sd > 0.0)) { flags[r] = 1; continue; }
    double scale = eta;
    for (;;) {
      bool inf = false;
      for (int i = 0; i < n; ++i) { w[i] = std::exp(scale * ((v[i] - mean) / sd)); inf = inf || std::isinf(w[i]); }
      if (!inf) break;
      scale *= 0.5;
    }
    int64_t* o = out + (size_t)r * n_pick;
    multinomial_row(s, w.data(), n, n_pick, ratio.data(), best.data(), o);
    bool has = false;
    for (int j = 0; j < n_pick; ++j) has = has || o[j] == arg;
    if (!has) o[n_pick - 1] = arg;
    flags[r] = 0;
  }
}
}  // namespace

int pcabo_boltzmann_pick_rows(void* const* blobs, const double* vals, int rows, int n, int n_pick, double eta, int64_t* out, int* flags) {
  if (!blobs || !vals || !out || !flags || rows < 1 || n < 2 || n_pick < 1 || n_pick > n) return PCABO_ERR_ARG;
  for (int r = 0; r < rows; ++r) {
    const TorchMt* s = static_cast<const TorchMt*>(blobs[r]);
    if (s && (!s->seeded || s->left < 1 || s->left > 624 || s->next > 624)) return PCABO_ERR_ARG;
  }
  boltzmann_rows(blobs, vals, 0, rows, n, n_pick, eta, out, flags);
  return PCABO_OK;
}

// torch.multinomial(weights[r], n_pick, replacement=False, generator=g_r) for `rows` rows with a generator each (blobs[r]; NULL =
// skip the row): q_i = -log1p(-u_i) with u_i = ((hi << 32 | lo) & (2^53 - 1)) 2^-53 from two draws per element (an Exp(1) variate:
// torch's exponential_), picks = the n_pick largest weights[i] / q_i, largest first (torch's topk).  out[rows][n_pick].
int pcabo_torch_multinomial_rows(void* const* blobs, const double* weights, int rows, int n, int n_pick, int64_t* out) {
  if (!blobs || !weights || !out || rows < 1 || n < 1 || n_pick < 1 || n_pick > n) return PCABO_ERR_ARG;
  std::vector<double> ratio((size_t)n);
  std::vector<int> best((size_t)n_pick);
  for (int r = 0; r < rows; ++r) {
    TorchMt* s = static_cast<TorchMt*>(blobs[r]);
    if (!s) continue;
    if (!s->seeded || s->left < 1 || s->left > 624 || s->next > 624) return PCABO_ERR_ARG;
    multinomial_row(s, weights + (size_t)r * n, n, n_pick, ratio.data(), best.data(), out + (size_t)r * n_pick);
  }
  return PCABO_OK;
}

}  // extern "C"
