// L-BFGS-B 3.0 restated as a reverse-communication C++ class.  See lbfgsb.h.
// Array layouts follow the published Fortran description (column-major, circular column storage
// for S and Y) so that every inner product is accumulated in the same order.
#include "lbfgsb.h"

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#define WS(i, p) ws_[(size_t)(p) * n_ + (i)]
#define WY(i, p) wy_[(size_t)(p) * n_ + (i)]
#define SY(i, j) sy_[(size_t)(j) * m_ + (i)]
#define SS(i, j) ss_[(size_t)(j) * m_ + (i)]
#define WT(i, j) wt_[(size_t)(j) * m_ + (i)]
#define WN(i, j) wn_[(size_t)(j) * 2 * m_ + (i)]
#define WN1(i, j) snd_[(size_t)(j) * 2 * m_ + (i)]

namespace {

// circular column index of the limited-memory matrices (avoids an integer division per use)
inline int nxt(int p, int m) { return p + 1 == m ? 0 : p + 1; }

// ---- vector kernels -----------------------------------------------------------------------------------------
// The O(m n) loops of L-BFGS-B come in two shapes.  Both are vectorised WITHOUT changing what any single number
// goes through (same operands, same order, separate multiply and add - no fma), so the iterates stay scipy's:
//  (a) "reduce over variables": 2m accumulators  acc_j += coef_k * W(k, j), k in increasing order.  SIMD lanes =
//      different j, which needs the j's of one variable side by side -> a row-major mirror `wr_` of WY|WS
//      (row stride RS = 2 MP, MP = m rounded up to 4; physical circular-buffer columns, mapped to logical order
//      afterwards).
//  (b) "chain per variable": out_k = out_k + f(W(k, j)) for j = 0..col-1 in order.  SIMD lanes = different k, on the
//      usual column-major WY / WS, for ALL variables (then gathered through the free-variable index).
typedef void (*accum_fn)(const double* wr, int rs, int mp, const int* rows, const double* coef, int count, double* acc);
typedef void (*chain_fn)(double* out, const double* wy, const double* ws, double a1, double a2, double th, int n);

void accum_scalar(const double* wr, int rs, int mp, const int* rows, const double* coef, int count, double* acc) {
  const int w = 2 * mp;
  for (int t = 0; t < count; ++t) {
    const double* row = wr + (size_t)(rows ? rows[t] : t) * rs;
    const double c = coef[t];
    for (int j = 0; j < w; ++j) acc[j] += c * row[j];
  }
}
// cmprlb:  out = out + (wy a1 + ws a2)
void chain_r_scalar(double* out, const double* wy, const double* ws, double a1, double a2, double, int n) {
  for (int k = 0; k < n; ++k) out[k] += wy[k] * a1 + ws[k] * a2;
}
// subsm:   out = (out + wy a1 / theta) + ws a2
void chain_d_scalar(double* out, const double* wy, const double* ws, double a1, double a2, double th, int n) {
  for (int k = 0; k < n; ++k) out[k] = out[k] + wy[k] * a1 / th + ws[k] * a2;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void accum_avx2(const double* wr, int rs, int mp, const int* rows, const double* coef,
                                                int count, double* acc) {
  if (mp == 12) {                      // m = 9..12 (scipy's default m = 10): six accumulators stay in registers
    __m256d a0 = _mm256_loadu_pd(acc), a1 = _mm256_loadu_pd(acc + 4), a2 = _mm256_loadu_pd(acc + 8);
    __m256d a3 = _mm256_loadu_pd(acc + 12), a4 = _mm256_loadu_pd(acc + 16), a5 = _mm256_loadu_pd(acc + 20);
    for (int t = 0; t < count; ++t) {
      const double* row = wr + (size_t)(rows ? rows[t] : t) * rs;
      const __m256d c = _mm256_set1_pd(coef[t]);
      a0 = _mm256_add_pd(a0, _mm256_mul_pd(c, _mm256_loadu_pd(row)));
      a1 = _mm256_add_pd(a1, _mm256_mul_pd(c, _mm256_loadu_pd(row + 4)));
      a2 = _mm256_add_pd(a2, _mm256_mul_pd(c, _mm256_loadu_pd(row + 8)));
      a3 = _mm256_add_pd(a3, _mm256_mul_pd(c, _mm256_loadu_pd(row + 12)));
      a4 = _mm256_add_pd(a4, _mm256_mul_pd(c, _mm256_loadu_pd(row + 16)));
      a5 = _mm256_add_pd(a5, _mm256_mul_pd(c, _mm256_loadu_pd(row + 20)));
    }
    _mm256_storeu_pd(acc, a0); _mm256_storeu_pd(acc + 4, a1); _mm256_storeu_pd(acc + 8, a2);
    _mm256_storeu_pd(acc + 12, a3); _mm256_storeu_pd(acc + 16, a4); _mm256_storeu_pd(acc + 20, a5);
    return;
  }
  const int w = 2 * mp;
  for (int t = 0; t < count; ++t) {
    const double* row = wr + (size_t)(rows ? rows[t] : t) * rs;
    const __m256d c = _mm256_set1_pd(coef[t]);
    for (int j = 0; j < w; j += 4)
      _mm256_storeu_pd(acc + j, _mm256_add_pd(_mm256_loadu_pd(acc + j), _mm256_mul_pd(c, _mm256_loadu_pd(row + j))));
  }
}
__attribute__((target("avx2"))) void chain_r_avx2(double* out, const double* wy, const double* ws, double a1, double a2,
                                                  double, int n) {
  const __m256d v1 = _mm256_set1_pd(a1), v2 = _mm256_set1_pd(a2);
  int k = 0;
  for (; k + 4 <= n; k += 4) {
    const __m256d t = _mm256_add_pd(_mm256_mul_pd(_mm256_loadu_pd(wy + k), v1), _mm256_mul_pd(_mm256_loadu_pd(ws + k), v2));
    _mm256_storeu_pd(out + k, _mm256_add_pd(_mm256_loadu_pd(out + k), t));
  }
  for (; k < n; ++k) out[k] += wy[k] * a1 + ws[k] * a2;
}
__attribute__((target("avx2"))) void chain_d_avx2(double* out, const double* wy, const double* ws, double a1, double a2,
                                                  double th, int n) {
  const __m256d v1 = _mm256_set1_pd(a1), v2 = _mm256_set1_pd(a2), vt = _mm256_set1_pd(th);
  int k = 0;
  for (; k + 4 <= n; k += 4) {
    __m256d t = _mm256_div_pd(_mm256_mul_pd(_mm256_loadu_pd(wy + k), v1), vt);
    t = _mm256_add_pd(_mm256_loadu_pd(out + k), t);
    _mm256_storeu_pd(out + k, _mm256_add_pd(t, _mm256_mul_pd(_mm256_loadu_pd(ws + k), v2)));
  }
  for (; k < n; ++k) out[k] = out[k] + wy[k] * a1 / th + ws[k] * a2;
}
#endif

struct VecKernels { accum_fn accum; chain_fn chain_r, chain_d; };
const VecKernels g_scalar_kernels{accum_scalar, chain_r_scalar, chain_d_scalar};
std::atomic<const VecKernels*> g_kernels{nullptr};
const VecKernels* default_kernels() {
#if defined(__x86_64__)
  static const VecKernels avx2{accum_avx2, chain_r_avx2, chain_d_avx2};
  if (__builtin_cpu_supports("avx2")) return &avx2;
#endif
  return &g_scalar_kernels;
}
inline const VecKernels& vec_kernels() {
  const VecKernels* k = g_kernels.load(std::memory_order_acquire);
  if (!k) { k = default_kernels(); g_kernels.store(k, std::memory_order_release); }
  return *k;
}

inline double ddot(int n, const double* x, const double* y) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}

// ---- the second summation order ("tree", Lbfgsb::set_sum_order(1)) ---------------------------------------------------
// The sums over the VARIABLES (the inner products d'd, g'd, r'r, the 2m accumulations W'd of cauchy / formk / subsm / matupd,
// f1 of the Cauchy search, the descent test of subsm) in the order a 64-lane wavefront forms them without a serial chain:
// lane l adds the terms l, l + 64, l + 128, ... in that order, then the 64 lane sums meet in a balanced tree of adjacent
// pairs (1 + 1, 2 + 2, 4 + 4 ... 32 + 32).  csrc/kernels_lbfgsb.hip steps in this order on the device (DPP row operations form
// exactly this tree); with the switch set the host class is its twin, bit for bit.  The short sums over the <= 2m history
// columns (bmv, dtrsl, dpofa, the small ddots) keep the published order in both modes.  A variable that the published
// algorithm skips carries the coefficient 0.0 here (full-length vectors): x + 0.0 w leaves every partial sum as it is.
inline double tree64(double* t) {
  for (int w = 1; w < 64; w <<= 1)
    for (int i = 0; i < 64; i += 2 * w) t[i] += t[i + w];
  return t[0];
}
inline double tree_sum(int n, const double* p) {
  double t[64];
  const int n0 = n < 64 ? n : 64;
  for (int i = 0; i < n0; ++i) t[i] = p[i];
  for (int i = n0; i < 64; ++i) t[i] = 0.0;
  for (int i = 64; i < n; ++i) t[i & 63] += p[i];
  return tree64(t);
}
inline double tree_dot(int n, const double* x, const double* y) {
  double t[64];
  const int n0 = n < 64 ? n : 64;
  for (int i = 0; i < n0; ++i) t[i] = x[i] * y[i];
  for (int i = n0; i < 64; ++i) t[i] = 0.0;
  for (int i = 64; i < n; ++i) t[i & 63] += x[i] * y[i];
  return tree64(t);
}
// acc[j] = tree sum over ALL variables i of coef[i] * W(i, j), j over the 2 mp physical columns of the mirror
void tree_accum(const double* wr, int rs, int mp, const double* coef, int n, double* acc) {
  const int w = 2 * mp;
  static thread_local double t[2 * LBFGSB_MAXM + 8][64];
  for (int j = 0; j < w; ++j) for (int l = 0; l < 64; ++l) t[j][l] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* row = wr + (size_t)i * rs;
    const double c = coef[i];
    const int l = i & 63;
    if (i < 64) for (int j = 0; j < w; ++j) t[j][l] = c * row[j];
    else for (int j = 0; j < w; ++j) t[j][l] += c * row[j];
  }
  for (int j = 0; j < w; ++j) acc[j] = tree64(t[j]);
}

// LINPACK dpofa: upper Cholesky factor of the leading nn x nn block (leading dimension lda).
int dpofa(double* a, int lda, int nn) {
  for (int j = 0; j < nn; ++j) {
    double s = 0.0;
    for (int k = 0; k < j; ++k) {
      double t = a[(size_t)j * lda + k] - ddot(k, a + (size_t)k * lda, a + (size_t)j * lda);
      t = t / a[(size_t)k * lda + k];
      a[(size_t)j * lda + k] = t;
      s += t * t;
    }
    s = a[(size_t)j * lda + j] - s;
    if (s <= 0.0) return j + 1;
    a[(size_t)j * lda + j] = std::sqrt(s);
  }
  return 0;
}

// LINPACK dtrsl for an upper-triangular t: job 1 -> t x = b, job 11 -> t' x = b.
int dtrsl(const double* t, int ldt, int nn, double* b, int job) {
  for (int i = 0; i < nn; ++i)
    if (t[(size_t)i * ldt + i] == 0.0) return i + 1;
  if (job == 1) {
    b[nn - 1] = b[nn - 1] / t[(size_t)(nn - 1) * ldt + nn - 1];
    for (int j = nn - 2; j >= 0; --j) {
      double temp = -b[j + 1];
      const double* col = t + (size_t)(j + 1) * ldt;
      for (int i = 0; i <= j; ++i) b[i] += temp * col[i];
      b[j] = b[j] / t[(size_t)j * ldt + j];
    }
  } else {
    b[0] = b[0] / t[0];
    for (int j = 1; j < nn; ++j) {
      b[j] = b[j] - ddot(j, t + (size_t)j * ldt, b);
      b[j] = b[j] / t[(size_t)j * ldt + j];
    }
  }
  return 0;
}

// dtrsl with the pivots' reciprocals formed first (r_j = 1 / t_jj, one division each) and multiplied in: the form of the tree /
// wave order (set_sum_order(1)) - on a wavefront the pivots' reciprocals are formed side by side, and the solve's chain of nn
// dependent divisions becomes a chain of multiplications.  One more rounding per pivot than the published division.
int dtrsl_recip(const double* t, int ldt, int nn, double* b, int job) {
  double r[2 * LBFGSB_MAXM];
  for (int i = 0; i < nn; ++i) {
    if (t[(size_t)i * ldt + i] == 0.0) return i + 1;
    r[i] = 1.0 / t[(size_t)i * ldt + i];
  }
  if (job == 1) {
    b[nn - 1] = b[nn - 1] * r[nn - 1];
    for (int j = nn - 2; j >= 0; --j) {
      double temp = -b[j + 1];
      const double* col = t + (size_t)(j + 1) * ldt;
      for (int i = 0; i <= j; ++i) b[i] += temp * col[i];
      b[j] = b[j] * r[j];
    }
  } else {
    b[0] = b[0] * r[0];
    for (int j = 1; j < nn; ++j) {
      b[j] = b[j] - ddot(j, t + (size_t)j * ldt, b);
      b[j] = b[j] * r[j];
    }
  }
  return 0;
}

}  // namespace

namespace { std::atomic<int> g_default_sum_order{0}; }
int lbfgsb_set_default_sum_order(int order) { return g_default_sum_order.exchange(order ? 1 : 0); }

int lbfgsb_set_vector_kernels(int enabled) {
  const bool was = &vec_kernels() != &g_scalar_kernels;
  g_kernels.store(enabled ? default_kernels() : &g_scalar_kernels, std::memory_order_release);
  return was ? 1 : 0;
}

void Lbfgsb::init(int n, int m, const double* lower, const double* upper, double factr, double pgtol, int maxls) {
  if (m > LBFGSB_MAXM) m = LBFGSB_MAXM;
  n_ = n; m_ = m; factr_ = factr; pgtol_ = pgtol; maxls_ = maxls;
  sum_order_ = g_default_sum_order.load();
  l_.assign(n, 0.0); u_.assign(n, 0.0); nbd_.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    double lo = lower ? lower[i] : -INFINITY, hi = upper ? upper[i] : INFINITY;
    bool hl = std::isfinite(lo), hu = std::isfinite(hi);
    l_[i] = hl ? lo : 0.0; u_[i] = hu ? hi : 0.0;
    nbd_[i] = hl ? (hu ? 2 : 1) : (hu ? 3 : 0);
  }
  ws_.assign((size_t)n * m, 0.0); wy_.assign((size_t)n * m, 0.0);
  mp_ = (m + 3) & ~3; rs_ = 2 * mp_;
  wr_.assign((size_t)n * rs_, 0.0); sc_coef_.assign(n, 0.0); sc_full_.assign(n, 0.0); sc_rows_.assign(n, 0);
  sy_.assign((size_t)m * m, 0.0); ss_.assign((size_t)m * m, 0.0); wt_.assign((size_t)m * m, 0.0);
  wn_.assign((size_t)4 * m * m, 0.0); snd_.assign((size_t)4 * m * m, 0.0);
  z_.assign(n, 0.0); r_.assign(n, 0.0); d_.assign(n, 0.0); t_.assign(n, 0.0); xp_.assign(n, 0.0);
  wa_.assign((size_t)8 * m, 0.0);
  index_.assign(n, 0); iwhere_.assign(n, 0); indx2_.assign(n, 0);
  task_ = LBFGSB_START; phase_ = 0;
}

void Lbfgsb::reset_memory() {
  info_ = 0; col_ = 0; head_ = 0; theta_ = 1.0; iupdat_ = 0; updatd_ = false;
}

void Lbfgsb::projgr(const double* x, const double* g) {
  sbgnrm_ = 0.0;
  if (boxed_) {
    // every variable has both bounds: branch-free (the sign of g_i is unpredictable), four running maxima
    // (max is exact and associative, the result is the same number)
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
    // (plain comparisons instead of fmax/fmin: those are libm calls without -ffinite-math; no NaN reaches this point)
    auto proj = [&](int i) {
      const double gi = g[i], hi = x[i] - u_[i], lo = x[i] - l_[i];
      const double a = hi > gi ? hi : gi, b = lo < gi ? lo : gi;
      return std::fabs(gi < 0.0 ? a : b);
    };
    auto mx = [](double p, double q) { return p > q ? p : q; };
    int i = 0;
    for (; i + 4 <= n_; i += 4) { m0 = mx(m0, proj(i)); m1 = mx(m1, proj(i + 1)); m2 = mx(m2, proj(i + 2)); m3 = mx(m3, proj(i + 3)); }
    for (; i < n_; ++i) m0 = mx(m0, proj(i));
    sbgnrm_ = mx(mx(m0, m1), mx(m2, m3));
    return;
  }
  for (int i = 0; i < n_; ++i) {
    double gi = g[i];
    if (nbd_[i] != 0) {
      if (gi < 0.0) { if (nbd_[i] >= 2) gi = std::fmax(x[i] - u_[i], gi); }
      else { if (nbd_[i] <= 2) gi = std::fmin(x[i] - l_[i], gi); }
    }
    sbgnrm_ = std::fmax(sbgnrm_, std::fabs(gi));
  }
}

bool Lbfgsb::active_init(double* x) {
  prjctd_ = false; cnstnd_ = false; boxed_ = true;
  for (int i = 0; i < n_; ++i) {
    if (nbd_[i] > 0) {
      if (nbd_[i] <= 2 && x[i] <= l_[i]) { if (x[i] < l_[i]) { prjctd_ = true; x[i] = l_[i]; } }
      else if (nbd_[i] >= 2 && x[i] >= u_[i]) { if (x[i] > u_[i]) { prjctd_ = true; x[i] = u_[i]; } }
    }
  }
  for (int i = 0; i < n_; ++i) {
    if (nbd_[i] != 2) boxed_ = false;
    if (nbd_[i] == 0) iwhere_[i] = -1;
    else {
      cnstnd_ = true;
      iwhere_[i] = (nbd_[i] == 2 && u_[i] - l_[i] <= 0.0) ? 3 : 0;
    }
  }
  return true;
}

void Lbfgsb::hpsolb(int n, double* t, int* iorder, int iheap) {
  // 1-based heap arithmetic on 0-based storage
  if (iheap == 0) {
    for (int k = 2; k <= n; ++k) {
      double ddum = t[k - 1];
      int indxin = iorder[k - 1];
      int i = k;
      while (i > 1) {
        int j = i / 2;
        if (ddum < t[j - 1]) { t[i - 1] = t[j - 1]; iorder[i - 1] = iorder[j - 1]; i = j; }
        else break;
      }
      t[i - 1] = ddum; iorder[i - 1] = indxin;
    }
  }
  if (n > 1) {
    int i = 1;
    double out = t[0];
    int indxou = iorder[0];
    double ddum = t[n - 1];
    int indxin = iorder[n - 1];
    while (true) {
      int j = i + i;
      if (j <= n - 1) {
        if (t[j] < t[j - 1]) j = j + 1;
        if (t[j - 1] < ddum) { t[i - 1] = t[j - 1]; iorder[i - 1] = iorder[j - 1]; i = j; }
        else break;
      } else break;
    }
    t[i - 1] = ddum; iorder[i - 1] = indxin;
    t[n - 1] = out; iorder[n - 1] = indxou;
  }
}

// p = M v with the 2col x 2col middle matrix of the compact representation
void Lbfgsb::bmv(const double* v, double* p) {
  const int col = col_;
  if (col == 0) return;
  if (sum_order_ == 1) {
    // the same products with the reciprocals rs_k = 1 / SY(k, k), rq_i = 1 / sqrt(SY(i, i)) and the pivots' reciprocals multiplied in
    double rs[LBFGSB_MAXM], rq[LBFGSB_MAXM];
    for (int i = 0; i < col; ++i) { rs[i] = 1.0 / SY(i, i); rq[i] = 1.0 / std::sqrt(SY(i, i)); }
    p[col] = v[col];
    for (int i = 1; i < col; ++i) {
      double sum = 0.0;
      for (int k = 0; k < i; ++k) sum += SY(i, k) * v[k] * rs[k];
      p[col + i] = v[col + i] + sum;
    }
    info_ = dtrsl_recip(wt_.data(), m_, col, p + col, 11);
    if (info_ != 0) return;
    for (int i = 0; i < col; ++i) p[i] = v[i] * rq[i];
    info_ = dtrsl_recip(wt_.data(), m_, col, p + col, 1);
    if (info_ != 0) return;
    for (int i = 0; i < col; ++i) p[i] = -p[i] * rq[i];
    for (int i = 0; i < col; ++i) {
      double sum = 0.0;
      for (int k = i + 1; k < col; ++k) sum += SY(k, i) * p[col + k] * rs[i];
      p[i] += sum;
    }
    return;
  }
  p[col] = v[col];
  for (int i = 1; i < col; ++i) {
    double sum = 0.0;
    for (int k = 0; k < i; ++k) sum += SY(i, k) * v[k] / SY(k, k);
    p[col + i] = v[col + i] + sum;
  }
  info_ = dtrsl(wt_.data(), m_, col, p + col, 11);
  if (info_ != 0) return;
  for (int i = 0; i < col; ++i) p[i] = v[i] / std::sqrt(SY(i, i));
  info_ = dtrsl(wt_.data(), m_, col, p + col, 1);
  if (info_ != 0) return;
  for (int i = 0; i < col; ++i) p[i] = -p[i] / std::sqrt(SY(i, i));
  for (int i = 0; i < col; ++i) {
    double sum = 0.0;
    for (int k = i + 1; k < col; ++k) sum += SY(k, i) * p[col + k] / SY(i, i);
    p[i] += sum;
  }
}

// Generalised Cauchy point along the projected steepest-descent path
void Lbfgsb::cauchy(const double* x, const double* g) {
  const int n = n_, m = m_, col = col_;
  double* p = wa_.data();
  double* c = wa_.data() + 2 * m;
  double* wbp = wa_.data() + 4 * m;
  double* v = wa_.data() + 6 * m;
  double* t = t_.data();
  double* d = d_.data();
  double* xcp = z_.data();
  int* iorder = indx2_.data();
  if (sbgnrm_ <= 0.0) { std::memcpy(xcp, x, sizeof(double) * n); return; }
  bool bnded = true;
  int nfree = n, nbreak = 0, ibkmin = 0, nmove = 0;
  double bkmin = 0.0;
  const int col2 = 2 * col;
  double f1 = 0.0;
  for (int i = 0; i < col2; ++i) p[i] = 0.0;
  if (boxed_) {
    // every variable has both bounds (the acquisition's search box): the same classification with the sign of the gradient
    // handled by selects instead of branches (it is unpredictable: half of the ~25 cycles per variable were mispredictions) -
    // tl / (-neggi) for neggi < 0 and tu / neggi for neggi > 0 are both "distance to the bound ahead / |neggi|"
    for (int i = 0; i < n; ++i) {
      const double neggi = -g[i];
      const double tl = x[i] - l_[i], tu = u_[i] - x[i];
      int iw = iwhere_[i];
      if (iw != 3) {
        iw = 0;
        if (tl <= 0.0) { if (neggi <= 0.0) iw = 1; }
        else if (tu <= 0.0) { if (neggi >= 0.0) iw = 2; }
        else if (std::fabs(neggi) <= 0.0) iw = -3;
        iwhere_[i] = iw;
      }
      if (iw != 0) { d[i] = 0.0; continue; }
      d[i] = neggi;
      f1 -= neggi * neggi;
      sc_rows_[nmove] = i;
      sc_coef_[nmove++] = neggi;
      if (neggi != 0.0) {                      // (NaN never gets here; neggi == 0 with iw == 0 cannot happen either)
        const double ahead = neggi < 0.0 ? tl : tu;
        const double tb = ahead / std::fabs(neggi);
        iorder[nbreak] = i;
        t[nbreak] = tb;
        if (nbreak == 0 || tb < bkmin) { bkmin = tb; ibkmin = nbreak; }
        ++nbreak;
      } else {
        --nfree;
        iorder[nfree] = i;
      }
    }
  } else
  for (int i = 0; i < n; ++i) {
    const double neggi = -g[i];
    double tl = 0.0, tu = 0.0;
    if (iwhere_[i] != 3 && iwhere_[i] != -1) {
      if (nbd_[i] <= 2) tl = x[i] - l_[i];
      if (nbd_[i] >= 2) tu = u_[i] - x[i];
      const bool xlower = nbd_[i] <= 2 && tl <= 0.0;
      const bool xupper = nbd_[i] >= 2 && tu <= 0.0;
      iwhere_[i] = 0;
      if (xlower) { if (neggi <= 0.0) iwhere_[i] = 1; }
      else if (xupper) { if (neggi >= 0.0) iwhere_[i] = 2; }
      else { if (std::fabs(neggi) <= 0.0) iwhere_[i] = -3; }
    }
    if (iwhere_[i] != 0 && iwhere_[i] != -1) {
      d[i] = 0.0;
    } else {
      d[i] = neggi;
      f1 -= neggi * neggi;
      sc_rows_[nmove] = i;                    // p = W' d is accumulated after the loop, in this (increasing) order
      sc_coef_[nmove++] = neggi;
      if (nbd_[i] <= 2 && nbd_[i] != 0 && neggi < 0.0) {
        iorder[nbreak] = i;
        t[nbreak] = tl / (-neggi);
        if (nbreak == 0 || t[nbreak] < bkmin) { bkmin = t[nbreak]; ibkmin = nbreak; }
        ++nbreak;
      } else if (nbd_[i] >= 2 && neggi > 0.0) {
        iorder[nbreak] = i;
        t[nbreak] = tu / neggi;
        if (nbreak == 0 || t[nbreak] < bkmin) { bkmin = t[nbreak]; ibkmin = nbreak; }
        ++nbreak;
      } else {
        --nfree;
        iorder[nfree] = i;
        if (std::fabs(neggi) > 0.0) bnded = false;
      }
    }
  }
  if (sum_order_ == 1) {
    // f1 = -(sum of neggi^2 over the moving variables), d itself is the full-length coefficient vector (0.0 elsewhere)
    for (int i = 0; i < n; ++i) sc_full_[i] = d[i] * d[i];
    f1 = -tree_sum(n, sc_full_.data());
  }
  if (col > 0) {
    double acc[2 * LBFGSB_MAXM] = {0.0};
    if (sum_order_ == 1) tree_accum(wr_.data(), rs_, mp_, d, n, acc);
    else vec_kernels().accum(wr_.data(), rs_, mp_, sc_rows_.data(), sc_coef_.data(), nmove, acc);
    int pointr = head_;
    for (int j = 0; j < col; ++j) { p[j] = acc[pointr]; p[col + j] = acc[mp_ + pointr]; pointr = nxt(pointr, m); }
#ifdef LBFGSB_CHECK
    { int pr = head_; for (int j = 0; j < col; ++j) { double a = 0, b = 0; for (int i = 0; i < n; ++i) if (d[i] != 0.0 || true) { if (iwhere_[i]==0||iwhere_[i]==-1) { a += WY(i, pr) * d[i]; b += WS(i, pr) * d[i]; } }
        if (a != p[j] || b != p[col + j]) fprintf(stderr, "cauchy mismatch j=%d %.17g %.17g | %.17g %.17g\n", j, a, p[j], b, p[col+j]); pr = nxt(pr, m); } }
#endif
  }
  if (theta_ != 1.0)
    for (int j = 0; j < col; ++j) p[col + j] *= theta_;
  std::memcpy(xcp, x, sizeof(double) * n);
  if (nbreak == 0 && nfree == n) return;
  for (int j = 0; j < col2; ++j) c[j] = 0.0;
  double f2 = -theta_ * f1;
  const double f2_org = f2;
  if (col > 0) {
    bmv(p, v);
    if (info_ != 0) return;
    f2 -= ddot(col2, v, p);
  }
  double dtm = -f1 / f2;
  double tsum = 0.0;
  nseg_ = 1;
  bool skip_to_999 = false;
  if (nbreak > 0) {
    int nleft = nbreak, iter = 1;
    double tj = 0.0;
    while (true) {
      const double tj0 = tj;
      int ibp;
      if (iter == 1) {
        tj = bkmin;
        ibp = iorder[ibkmin];
      } else {
        if (iter == 2) {
          if (ibkmin != nbreak - 1) { t[ibkmin] = t[nbreak - 1]; iorder[ibkmin] = iorder[nbreak - 1]; }
        }
        hpsolb(nleft, t, iorder, iter - 2);
        tj = t[nleft - 1];
        ibp = iorder[nleft - 1];
      }
      const double dt = tj - tj0;
      if (dtm < dt) break;                     // minimiser inside this segment
      tsum += dt;
      --nleft;
      ++iter;
      const double dibp = d[ibp];
      d[ibp] = 0.0;
      double zibp;
      if (dibp > 0.0) { zibp = u_[ibp] - x[ibp]; xcp[ibp] = u_[ibp]; iwhere_[ibp] = 2; }
      else { zibp = l_[ibp] - x[ibp]; xcp[ibp] = l_[ibp]; iwhere_[ibp] = 1; }
      if (nleft == 0 && nbreak == n) { dtm = dt; skip_to_999 = true; break; }
      ++nseg_;
      const double dibp2 = dibp * dibp;
      f1 = f1 + dt * f2 + dibp2 - theta_ * dibp * zibp;
      f2 = f2 - theta_ * dibp2;
      if (col > 0) {
        for (int j = 0; j < col2; ++j) c[j] += dt * p[j];
        int pointr = head_;
        for (int j = 0; j < col; ++j) {
          wbp[j] = WY(ibp, pointr);
          wbp[col + j] = theta_ * WS(ibp, pointr);
          pointr = nxt(pointr, m);
        }
        bmv(wbp, v);
        if (info_ != 0) return;
        const double wmc = ddot(col2, c, v);
        const double wmp = ddot(col2, p, v);
        const double wmw = ddot(col2, wbp, v);
        for (int j = 0; j < col2; ++j) p[j] -= dibp * wbp[j];
        f1 += dibp * wmc;
        f2 = f2 + 2.0 * dibp * wmp - dibp2 * wmw;
      }
      f2 = std::fmax(epsmch_ * f2_org, f2);
      if (nleft > 0) {
        dtm = -f1 / f2;
        continue;
      } else if (bnded) {
        f1 = 0.0; f2 = 0.0; dtm = 0.0;
      } else {
        dtm = -f1 / f2;
      }
      break;
    }
  }
  if (!skip_to_999) {
    if (dtm <= 0.0) dtm = 0.0;
    tsum += dtm;
    for (int i = 0; i < n; ++i) xcp[i] += tsum * d[i];
  }
  if (col > 0)
    for (int j = 0; j < col2; ++j) c[j] += dtm * p[j];
}

void Lbfgsb::freev() {
  const int n = n_;
  nenter_ = 0;
  ileave_ = n;
  if (iter_ > 0 && cnstnd_) {
    for (int i = 0; i < nfree_; ++i) {
      int k = index_[i];
      if (iwhere_[k] > 0) { --ileave_; indx2_[ileave_] = k; }
    }
    for (int i = nfree_; i < n; ++i) {
      int k = index_[i];
      if (iwhere_[k] <= 0) { indx2_[nenter_] = k; ++nenter_; }
    }
  }
  wrk_ = (ileave_ < n) || (nenter_ > 0) || updatd_;
  nfree_ = 0;
  int iact = n;
  for (int i = 0; i < n; ++i) {
    if (iwhere_[i] <= 0) { index_[nfree_] = i; ++nfree_; }
    else { --iact; index_[iact] = i; }
  }
}

// LEL^T factorisation of the indefinite subspace matrix
void Lbfgsb::formk() {
  const int n = n_, m = m_, col = col_, head = head_, nsub = nfree_;
  const int* ind = index_.data();
  const int* indx2 = indx2_.data();
  int upcl;
  if (updatd_) {
    if (iupdat_ > m) {
      for (int jy = 0; jy < m - 1; ++jy) {
        const int js = m + jy;
        for (int t = 0; t < m - 1 - jy; ++t) {
          WN1(jy + t, jy) = WN1(jy + 1 + t, jy + 1);
          WN1(js + t, js) = WN1(js + 1 + t, js + 1);
        }
        for (int t = 0; t < m - 1; ++t) WN1(m + t, jy) = WN1(m + 1 + t, jy + 1);
      }
    }
    int ipntr = (head + col - 1) % m;
    const int iy = col - 1, is = m + col - 1;
    // Elements jy = 0..col-1 of the new rows: col independent dot products per block, accumulated side by
    // side (k outer) - every accumulator still sums k in increasing order, exactly like one-at-a-time ddots.
    {
      // free variables: row col of Y'ZZ'Y (t1) and column col of R_z (t4) share the coefficient WY(k, ipntr);
      // active variables: row col of S'AA'S (t2) and of L_a (t3) share WS(k, ipntr).  One pass over the mirror each.
      double accf[2 * LBFGSB_MAXM] = {0.0}, acca[2 * LBFGSB_MAXM] = {0.0};
      double* coef = sc_coef_.data();
      for (int k = 0; k < nsub; ++k) coef[k] = WY(ind[k], ipntr);
      for (int k = nsub; k < n; ++k) coef[k] = WS(ind[k], ipntr);
      if (sum_order_ == 1) {
        // full-length coefficient vectors: WY(k, ipntr) on the free variables / WS(k, ipntr) on the active ones, 0.0 elsewhere
        double* cb = sc_full_.data();
        for (int k = 0; k < n; ++k) { const bool fr = iwhere_[k] <= 0; coef[k] = fr ? WY(k, ipntr) : 0.0; cb[k] = fr ? 0.0 : WS(k, ipntr); }
        tree_accum(wr_.data(), rs_, mp_, coef, n, accf);
        tree_accum(wr_.data(), rs_, mp_, cb, n, acca);
      } else {
      vec_kernels().accum(wr_.data(), rs_, mp_, ind, coef, nsub, accf);
      vec_kernels().accum(wr_.data(), rs_, mp_, ind + nsub, coef + nsub, n - nsub, acca);
      }
      int jp = head;
      const int jyc = col - 1;
      for (int jy = 0; jy < col; ++jy) {
        WN1(iy, jy) = accf[jp];            // t1
        WN1(is, m + jy) = acca[mp_ + jp];  // t2
        WN1(is, jy) = acca[jp];            // t3
        jp = nxt(jp, m);
      }
      jp = head;                           // t4 last: element (m+col-1, col-1) is written by both t3 and t4, t4 stays
      for (int jy = 0; jy < col; ++jy) { WN1(m + jy, jyc) = accf[mp_ + jp]; jp = nxt(jp, m); }
    }
    upcl = col - 1;
  } else {
    upcl = col;
  }
  int ipntr = head;
  for (int iy = 0; iy < upcl; ++iy) {
    const int is = m + iy;
    int jpntr = head;
    for (int jy = 0; jy <= iy; ++jy) {
      const int js = m + jy;
      double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0, temp4 = 0.0;
      for (int k = 0; k < nenter_; ++k) {
        int k1 = indx2[k];
        temp1 += WY(k1, ipntr) * WY(k1, jpntr);
        temp2 += WS(k1, ipntr) * WS(k1, jpntr);
      }
      for (int k = ileave_; k < n; ++k) {
        int k1 = indx2[k];
        temp3 += WY(k1, ipntr) * WY(k1, jpntr);
        temp4 += WS(k1, ipntr) * WS(k1, jpntr);
      }
      WN1(iy, jy) = WN1(iy, jy) + temp1 - temp3;
      WN1(is, js) = WN1(is, js) - temp2 + temp4;
      jpntr = nxt(jpntr, m);
    }
    ipntr = nxt(ipntr, m);
  }
  ipntr = head;
  for (int is = m; is < m + upcl; ++is) {
    int jpntr = head;
    for (int jy = 0; jy < upcl; ++jy) {
      double temp1 = 0.0, temp3 = 0.0;
      for (int k = 0; k < nenter_; ++k) { int k1 = indx2[k]; temp1 += WS(k1, ipntr) * WY(k1, jpntr); }
      for (int k = ileave_; k < n; ++k) { int k1 = indx2[k]; temp3 += WS(k1, ipntr) * WY(k1, jpntr); }
      if (is <= jy + m) WN1(is, jy) = WN1(is, jy) + temp1 - temp3;
      else WN1(is, jy) = WN1(is, jy) - temp1 + temp3;
      jpntr = nxt(jpntr, m);
    }
    ipntr = nxt(ipntr, m);
  }
  const int m2 = 2 * m;
  for (int iy = 0; iy < col; ++iy) {
    const int is = col + iy, is1 = m + iy;
    for (int jy = 0; jy <= iy; ++jy) {
      const int js = col + jy, js1 = m + jy;
      WN(jy, iy) = WN1(iy, jy) / theta_;
      WN(js, is) = WN1(is1, js1) * theta_;
    }
    for (int jy = 0; jy < iy; ++jy) WN(jy, is) = -WN1(is1, jy);
    for (int jy = iy; jy < col; ++jy) WN(jy, is) = WN1(is1, jy);
    WN(iy, iy) += SY(iy, iy);
  }
  int info = dpofa(wn_.data(), m2, col);
  if (info != 0) { info_ = -1; return; }
  const int col2 = 2 * col;
  for (int js = col; js < col2; ++js) (sum_order_ == 1 ? dtrsl_recip : dtrsl)(wn_.data(), m2, col, &WN(0, js), 11);
  for (int is = col; is < col2; ++is)
    for (int js = is; js < col2; ++js) WN(is, js) += ddot(col, &WN(0, is), &WN(0, js));
  info = dpofa(&WN(col, col), m2, col);
  if (info != 0) info_ = -2;
}

void Lbfgsb::cmprlb(const double* x, const double* g) {
  const int n = n_, m = m_, col = col_;
  if (!cnstnd_ && col > 0) {
    for (int i = 0; i < n; ++i) r_[i] = -g[i];
    return;
  }
  const bool all_free = (nfree_ == n);
  double* full = all_free ? r_.data() : sc_full_.data();
  for (int k = 0; k < n; ++k) full[k] = -theta_ * (z_[k] - x[k]) - g[k];
  bmv(wa_.data() + 2 * m, wa_.data());
  if (info_ != 0) { info_ = -8; return; }
  int pointr = head_;
  for (int j = 0; j < col; ++j) {                    // a chain per variable, all n side by side (see vec_kernels)
    vec_kernels().chain_r(full, &WY(0, pointr), &WS(0, pointr), wa_[j], theta_ * wa_[col + j], 0.0, n);
    pointr = nxt(pointr, m);
  }
  if (!all_free) for (int i = 0; i < nfree_; ++i) r_[i] = full[index_[i]];
#ifdef LBFGSB_CHECK
  { std::vector<double> rr(nfree_); for (int i = 0; i < nfree_; ++i) { int k = index_[i]; rr[i] = -theta_ * (z_[k] - x[k]) - g[k]; }
    int pr = head_; for (int j = 0; j < col; ++j) { const double a1 = wa_[j], a2 = theta_ * wa_[col + j]; for (int i = 0; i < nfree_; ++i) { int k = index_[i]; rr[i] += WY(k, pr) * a1 + WS(k, pr) * a2; } pr = nxt(pr, m); }
    for (int i = 0; i < nfree_; ++i) if (rr[i] != r_[i]) { fprintf(stderr, "cmprlb mismatch i=%d %.17g %.17g allfree=%d\n", i, rr[i], r_[i], (int)all_free); break; } }
#endif
}

void Lbfgsb::subsm(const double* xx, const double* gg) {
  const int n = n_, m = m_, col = col_, nsub = nfree_;
  const int* ind = index_.data();
  double* x = z_.data();      // on entry the Cauchy point, on exit the subspace minimiser
  double* d = r_.data();      // reduced gradient -> Newton direction
  double* wv = wa_.data();
  if (nsub <= 0) return;
  int pointr = head_;
  {
    // wv = W' Z d over the free variables (their order), all 2 col inner products in one pass over the mirror
    double acc[2 * LBFGSB_MAXM] = {0.0};
    if (sum_order_ == 1) {
      double* full = sc_coef_.data();                // d scattered to the variables' own places, 0.0 elsewhere
      for (int k = 0; k < n; ++k) full[k] = 0.0;
      for (int i = 0; i < nsub; ++i) full[ind[i]] = d[i];
      tree_accum(wr_.data(), rs_, mp_, full, n, acc);
    } else
    vec_kernels().accum(wr_.data(), rs_, mp_, ind, d, nsub, acc);
    for (int i = 0; i < col; ++i) { wv[i] = acc[pointr]; wv[col + i] = theta_ * acc[mp_ + pointr]; pointr = nxt(pointr, m); }
#ifdef LBFGSB_CHECK
    { int pr = head_; for (int i = 0; i < col; ++i) { double a = 0, b = 0; for (int j = 0; j < nsub; ++j) { a += WY(ind[j], pr) * d[j]; b += WS(ind[j], pr) * d[j]; }
        if (a != wv[i] || theta_ * b != wv[col + i]) fprintf(stderr, "subsm1 mismatch i=%d\n", i); pr = nxt(pr, m); } }
    for (int i = 0; i < nsub; ++i) sc_coef_[i] = d[i];
#endif
  }
  const int m2 = 2 * m, col2 = 2 * col;
  info_ = (sum_order_ == 1 ? dtrsl_recip : dtrsl)(wn_.data(), m2, col2, wv, 11);
  if (info_ != 0) return;
  for (int i = 0; i < col; ++i) wv[i] = -wv[i];
  info_ = (sum_order_ == 1 ? dtrsl_recip : dtrsl)(wn_.data(), m2, col2, wv, 1);
  if (info_ != 0) return;
  pointr = head_;
  {
    // d_k = d_k + WY(k, j) wv_j / theta + WS(k, j) wv_{col+j}, j in order: a chain per variable, run for all n variables
    // side by side on the column-major matrices, then read back through the free-variable index
    const bool all_free = (nsub == n);               // freev lists the free variables in increasing order
    double* full = all_free ? d : sc_full_.data();
    if (!all_free) { for (int k = 0; k < n; ++k) full[k] = 0.0; for (int i = 0; i < nsub; ++i) full[ind[i]] = d[i]; }
    for (int jy = 0; jy < col; ++jy) {
      if (sum_order_ == 1) {                         // (out + wy a1 (1 / theta)) + ws a2: the reciprocal of theta formed once
        const double it = 1.0 / theta_, a1 = wv[jy], a2 = wv[col + jy];
        const double* wy = &WY(0, pointr); const double* ws = &WS(0, pointr);
        for (int k = 0; k < n; ++k) full[k] = full[k] + wy[k] * a1 * it + ws[k] * a2;
      } else
      vec_kernels().chain_d(full, &WY(0, pointr), &WS(0, pointr), wv[jy], wv[col + jy], theta_, n);
      pointr = nxt(pointr, m);
    }
    if (!all_free) for (int i = 0; i < nsub; ++i) d[i] = full[ind[i]];
#ifdef LBFGSB_CHECK
    { int pr = head_; std::vector<double> dd(sc_coef_.begin(), sc_coef_.begin() + nsub);
      for (int jy = 0; jy < col; ++jy) { for (int i = 0; i < nsub; ++i) dd[i] = dd[i] + WY(ind[i], pr) * wv[jy] / theta_ + WS(ind[i], pr) * wv[col + jy]; pr = nxt(pr, m); }
      for (int i = 0; i < nsub; ++i) if (dd[i] != d[i]) { fprintf(stderr, "subsm2 mismatch i=%d %.17g %.17g\n", i, dd[i], d[i]); break; } }
#endif
  }
  const double inv_theta = 1.0 / theta_;
  for (int i = 0; i < nsub; ++i) d[i] *= inv_theta;
  // projected Newton step (3.0), falling back to a truncated step when it is not a descent direction
  iword_ = 0;
  std::memcpy(xp_.data(), x, sizeof(double) * n);
  for (int i = 0; i < nsub; ++i) {
    const int k = ind[i];
    const double dk = d[i];
    double xk = x[k];
    // (plain comparisons instead of fmax / fmin: those are libm calls without -ffinite-math, 360 of them per step at 180
    // variables; no NaN reaches this point and on a tie both operands are the same number)
    if (nbd_[k] != 0) {
      const double v = xk + dk, lk = l_[k], uk = u_[k];
      if (nbd_[k] == 1) {
        x[k] = lk > v ? lk : v;
        if (x[k] == lk) iword_ = 1;
      } else if (nbd_[k] == 2) {
        xk = lk > v ? lk : v;
        const double w = uk < xk ? uk : xk;
        x[k] = w;
        if (w == lk || w == uk) iword_ = 1;
      } else if (nbd_[k] == 3) {
        x[k] = uk < v ? uk : v;
        if (x[k] == uk) iword_ = 1;
      }
    } else {
      x[k] = xk + dk;
    }
  }
  if (iword_ == 0) return;
  double dd_p = 0.0;
  if (sum_order_ == 1) {
    for (int i = 0; i < n; ++i) sc_coef_[i] = (x[i] - xx[i]) * gg[i];
    dd_p = tree_sum(n, sc_coef_.data());
  } else
  for (int i = 0; i < n; ++i) dd_p += (x[i] - xx[i]) * gg[i];
  if (dd_p > 0.0) {
    std::memcpy(x, xp_.data(), sizeof(double) * n);
    double alpha = 1.0, temp1 = alpha;
    int ibd = 0;
    for (int i = 0; i < nsub; ++i) {
      const int k = ind[i];
      const double dk = d[i];
      if (nbd_[k] != 0) {
        if (dk < 0.0 && nbd_[k] <= 2) {
          double temp2 = l_[k] - x[k];
          if (temp2 >= 0.0) temp1 = 0.0;
          else if (dk * alpha < temp2) temp1 = temp2 / dk;
        } else if (dk > 0.0 && nbd_[k] >= 2) {
          double temp2 = u_[k] - x[k];
          if (temp2 <= 0.0) temp1 = 0.0;
          else if (dk * alpha > temp2) temp1 = temp2 / dk;
        }
        if (temp1 < alpha) { alpha = temp1; ibd = i; }
      }
    }
    if (alpha < 1.0) {
      const double dk = d[ibd];
      const int k = ind[ibd];
      if (dk > 0.0) { x[k] = u_[k]; d[ibd] = 0.0; }
      else if (dk < 0.0) { x[k] = l_[k]; d[ibd] = 0.0; }
    }
    for (int i = 0; i < nsub; ++i) { int k = ind[i]; x[k] += alpha * d[i]; }
  }
}

void Lbfgsb::dcstep(double* stx, double* fx, double* dx, double* sty, double* fy, double* dy, double* stp, double fp,
                    double dp, bool* brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (*dx / std::fabs(*dx));
  double stpf, stpc, stpq, theta, s, gamma, p, q, r;
  if (fp > *fx) {
    theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = std::fmax(std::fabs(theta), std::fmax(std::fabs(*dx), std::fabs(dp)));
    gamma = s * std::sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    if (*stp < *stx) gamma = -gamma;
    p = (gamma - *dx) + theta;
    q = ((gamma - *dx) + gamma) + dp;
    r = p / q;
    stpc = *stx + r * (*stp - *stx);
    stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
    if (std::fabs(stpc - *stx) < std::fabs(stpq - *stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    *brackt = true;
  } else if (sgnd < 0.0) {
    theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = std::fmax(std::fabs(theta), std::fmax(std::fabs(*dx), std::fabs(dp)));
    gamma = s * std::sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + *dx;
    r = p / q;
    stpc = *stp + r * (*stx - *stp);
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (std::fabs(stpc - *stp) > std::fabs(stpq - *stp)) stpf = stpc;
    else stpf = stpq;
    *brackt = true;
  } else if (std::fabs(dp) < std::fabs(*dx)) {
    theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
    s = std::fmax(std::fabs(theta), std::fmax(std::fabs(*dx), std::fabs(dp)));
    gamma = s * std::sqrt(std::fmax(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
    if (*stp > *stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (*dx - dp)) + gamma;
    r = p / q;
    if (r < 0.0 && gamma != 0.0) stpc = *stp + r * (*stx - *stp);
    else if (*stp > *stx) stpc = stpmax;
    else stpc = stpmin;
    stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
    if (*brackt) {
      if (std::fabs(stpc - *stp) < std::fabs(stpq - *stp)) stpf = stpc;
      else stpf = stpq;
      if (*stp > *stx) stpf = std::fmin(*stp + 0.66 * (*sty - *stp), stpf);
      else stpf = std::fmax(*stp + 0.66 * (*sty - *stp), stpf);
    } else {
      if (std::fabs(stpc - *stp) > std::fabs(stpq - *stp)) stpf = stpc;
      else stpf = stpq;
      stpf = std::fmin(stpmax, stpf);
      stpf = std::fmax(stpmin, stpf);
    }
  } else {
    if (*brackt) {
      theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
      s = std::fmax(std::fabs(theta), std::fmax(std::fabs(*dy), std::fabs(dp)));
      gamma = s * std::sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
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

void Lbfgsb::dcsrch(double f, double g, double* stp, double ftol, double gtol, double xtol, double stpmin,
                    double stpmax, Dcsrch& s) {
  const double xtrapl = 1.1, xtrapu = 4.0, p5 = 0.5, p66 = 0.66;
  if (s.task == 0) {
    if (*stp < stpmin || *stp > stpmax || g >= 0.0) { s.task = 4; return; }
    s.brackt = false; s.stage = 1; s.finit = f; s.ginit = g; s.gtest = ftol * s.ginit;
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
  if (f <= ftest && std::fabs(g) <= gtol * (-s.ginit)) task = 2;
  if (task == 2 || task == 3) { s.task = task; return; }
  if (s.stage == 1 && f <= s.fx && f > ftest) {
    double fm = f - *stp * s.gtest, fxm = s.fx - s.stx * s.gtest, fym = s.fy - s.sty * s.gtest;
    double gm = g - s.gtest, gxm = s.gx - s.gtest, gym = s.gy - s.gtest;
    dcstep(&s.stx, &fxm, &gxm, &s.sty, &fym, &gym, stp, fm, gm, &s.brackt, s.stmin, s.stmax);
    s.fx = fxm + s.stx * s.gtest; s.fy = fym + s.sty * s.gtest; s.gx = gxm + s.gtest; s.gy = gym + s.gtest;
  } else {
    dcstep(&s.stx, &s.fx, &s.gx, &s.sty, &s.fy, &s.gy, stp, f, g, &s.brackt, s.stmin, s.stmax);
  }
  if (s.brackt) {
    if (std::fabs(s.sty - s.stx) >= p66 * s.width1) *stp = s.stx + p5 * (s.sty - s.stx);
    s.width1 = s.width;
    s.width = std::fabs(s.sty - s.stx);
  }
  if (s.brackt) { s.stmin = std::fmin(s.stx, s.sty); s.stmax = std::fmax(s.stx, s.sty); }
  else { s.stmin = *stp + xtrapl * (*stp - s.stx); s.stmax = *stp + xtrapu * (*stp - s.stx); }
  *stp = std::fmax(*stp, stpmin);
  *stp = std::fmin(*stp, stpmax);
  if ((s.brackt && (*stp <= s.stmin || *stp >= s.stmax)) || (s.brackt && s.stmax - s.stmin <= xtol * s.stmax))
    *stp = s.stx;
  s.task = 1;
}

// One call of the line-search driver.  Sets task_ to LBFGSB_FG (x holds the trial point) or
// LBFGSB_NEW_X (line search finished); info_ != 0 on failure.
void Lbfgsb::lnsrlb(double* x, double f, const double* g) {
  const int n = n_;
  const double big = 1e10, ftol = 1e-3, gtol = 0.9, xtol = 0.1;
  if (phase_ != 2) {
    dtd_ = sum_order_ == 1 ? tree_dot(n, d_.data(), d_.data()) : ddot(n, d_.data(), d_.data());
    dnorm_ = std::sqrt(dtd_);
    stpmx_ = big;
    if (cnstnd_) {
      if (iter_ == 0) stpmx_ = 1.0;
      else {
        for (int i = 0; i < n; ++i) {
          const double a1 = d_[i];
          if (nbd_[i] != 0) {
            if (a1 < 0.0 && nbd_[i] <= 2) {
              double a2 = l_[i] - x[i];
              if (a2 >= 0.0) stpmx_ = 0.0;
              else if (a1 * stpmx_ < a2) stpmx_ = a2 / a1;
            } else if (a1 > 0.0 && nbd_[i] >= 2) {
              double a2 = u_[i] - x[i];
              if (a2 <= 0.0) stpmx_ = 0.0;
              else if (a1 * stpmx_ > a2) stpmx_ = a2 / a1;
            }
          }
        }
      }
    }
    if (iter_ == 0 && !boxed_) stp_ = std::fmin(1.0 / dnorm_, stpmx_);
    else stp_ = 1.0;
    std::memcpy(t_.data(), x, sizeof(double) * n);
    std::memcpy(r_.data(), g, sizeof(double) * n);
    fold_ = f;
    ifun_ = 0;
    iback_ = 0;
    ls_ = Dcsrch();
  }
  gd_ = sum_order_ == 1 ? tree_dot(n, g, d_.data()) : ddot(n, g, d_.data());
  if (ifun_ == 0) {
    gdold_ = gd_;
    if (gd_ >= 0.0) { info_ = -4; return; }     // ascent direction: line search impossible
  }
  dcsrch(f, gd_, &stp_, ftol, gtol, xtol, 0.0, stpmx_, ls_);
  xstep_ = stp_ * dnorm_;
  if (ls_.task != 2 && ls_.task != 3) {
    task_ = LBFGSB_FG;
    ++ifun_;
    ++nfgv_;
    iback_ = ifun_ - 1;
    if (stp_ == 1.0) std::memcpy(x, z_.data(), sizeof(double) * n);
    else
      for (int i = 0; i < n; ++i) x[i] = stp_ * d_[i] + t_[i];
  } else {
    task_ = LBFGSB_NEW_X;
  }
}

void Lbfgsb::matupd(double rr, double dr) {
  const int n = n_, m = m_;
  if (iupdat_ <= m) {
    col_ = iupdat_;
    itail_ = (head_ + iupdat_ - 1) % m;
  } else {
    itail_ = nxt(itail_, m);
    head_ = nxt(head_, m);
  }
  std::memcpy(&WS(0, itail_), d_.data(), sizeof(double) * n);
  std::memcpy(&WY(0, itail_), r_.data(), sizeof(double) * n);
  for (int i = 0; i < n; ++i) {                      // row-major mirror (see vec_kernels)
    wr_[(size_t)i * rs_ + itail_] = r_[i];
    wr_[(size_t)i * rs_ + mp_ + itail_] = d_[i];
  }
  theta_ = rr / dr;
  const int col = col_;
  if (iupdat_ > m) {
    for (int j = 0; j < col - 1; ++j) {
      for (int t = 0; t <= j; ++t) SS(t, j) = SS(t + 1, j + 1);
      for (int t = 0; t < col - 1 - j; ++t) SY(j + t, j) = SY(j + 1 + t, j + 1);
    }
  }
  {
    // SY(col-1, j) = d . WY(:, j), SS(j, col-1) = WS(:, j) . d for the older columns j: one pass over the mirror
    double acc[2 * LBFGSB_MAXM] = {0.0};
    if (sum_order_ == 1) tree_accum(wr_.data(), rs_, mp_, d_.data(), n, acc);
    else vec_kernels().accum(wr_.data(), rs_, mp_, nullptr, d_.data(), n, acc);
    int pointr = head_;
    for (int j = 0; j < col - 1; ++j) { SY(col - 1, j) = acc[pointr]; SS(j, col - 1) = acc[mp_ + pointr]; pointr = nxt(pointr, m); }
#ifdef LBFGSB_CHECK
    { int pr = head_; for (int j = 0; j < col - 1; ++j) { double a = 0, b = 0; for (int k = 0; k < n; ++k) { a += d_[k] * WY(k, pr); b += WS(k, pr) * d_[k]; }
        if (a != SY(col - 1, j) || b != SS(j, col - 1)) fprintf(stderr, "matupd mismatch j=%d\n", j); pr = nxt(pr, m); } }
#endif
  }
  if (stp_ == 1.0) SS(col - 1, col - 1) = dtd_;
  else SS(col - 1, col - 1) = stp_ * stp_ * dtd_;
  SY(col - 1, col - 1) = dr;
}

void Lbfgsb::formt() {
  const int col = col_;
  for (int j = 0; j < col; ++j) WT(0, j) = theta_ * SS(0, j);
  for (int i = 1; i < col; ++i) {
    for (int j = i; j < col; ++j) {
      const int k1 = (i < j ? i : j);
      double ddum = 0.0;
      if (sum_order_ == 1) for (int k = 0; k < k1; ++k) ddum += SY(i, k) * SY(j, k) * (1.0 / SY(k, k));
      else
      for (int k = 0; k < k1; ++k) ddum += SY(i, k) * SY(j, k) / SY(k, k);
      WT(i, j) = ddum + theta_ * SS(i, j);
    }
  }
  int info = dpofa(wt_.data(), m_, col);
  if (info != 0) info_ = -3;
}

int Lbfgsb::step(double* x, double* fp, double* g) {
  const int n = n_;
  double f = *fp;
  if (task_ >= LBFGSB_CONV_PG) return task_;         // finished (or stopped by the caller)

  bool need_iteration_start = false;   // jump target 222
  bool resume_linesearch = false;      // jump target 666 with task FG_LN

  if (phase_ == 0) {
    epsmch_ = DBL_EPSILON;
    col_ = 0; head_ = 0; theta_ = 1.0; iupdat_ = 0; updatd_ = false;
    iback_ = 0; itail_ = 0; iword_ = 0; nact_ = 0; ileave_ = 0; nenter_ = 0;
    fold_ = 0; dnorm_ = 0; gd_ = 0; stpmx_ = 0; sbgnrm_ = 0; stp_ = 0; gdold_ = 0; dtd_ = 0;
    iter_ = 0; nfgv_ = 0; nseg_ = 0; nintol_ = 0; nskip_ = 0; nfree_ = n; ifun_ = 0;
    tol_ = factr_ * epsmch_;
    info_ = 0;
    for (int i = 0; i < n; ++i)
      if (nbd_[i] == 2 && l_[i] > u_[i]) { task_ = LBFGSB_ERROR; return task_; }
    if (n <= 0 || m_ <= 0 || factr_ < 0.0) { task_ = LBFGSB_ERROR; return task_; }
    active_init(x);
    phase_ = 1;
    task_ = LBFGSB_FG;
    return task_;
  }
  if (phase_ == 1) {
    nfgv_ = 1;
    projgr(x, g);
    if (sbgnrm_ <= pgtol_) { task_ = LBFGSB_CONV_PG; return task_; }
    need_iteration_start = true;
  } else if (phase_ == 2) {
    resume_linesearch = true;
  } else {  // phase_ == 3: back from NEW_X
    if (sbgnrm_ <= pgtol_) { task_ = LBFGSB_CONV_PG; return task_; }
    const double ddum = std::fmax(std::fmax(std::fabs(fold_), std::fabs(f)), 1.0);
    if ((fold_ - f) <= tol_ * ddum) {
      task_ = LBFGSB_CONV_F;
      if (iback_ >= 10) info_ = -5;
      return task_;
    }
    for (int i = 0; i < n; ++i) r_[i] = g[i] - r_[i];
    const double rr = sum_order_ == 1 ? tree_dot(n, r_.data(), r_.data()) : ddot(n, r_.data(), r_.data());
    double dr, ddum2;
    if (stp_ == 1.0) { dr = gd_ - gdold_; ddum2 = -gdold_; }
    else {
      dr = (gd_ - gdold_) * stp_;
      for (int i = 0; i < n; ++i) d_[i] *= stp_;
      ddum2 = -gdold_ * stp_;
    }
    if (dr <= epsmch_ * ddum2) {
      ++nskip_;
      updatd_ = false;
    } else {
      updatd_ = true;
      ++iupdat_;
      matupd(rr, dr);
      formt();
      if (info_ != 0) reset_memory();
    }
    need_iteration_start = true;
  }

  while (true) {
    if (need_iteration_start) {
      need_iteration_start = false;
      // ----- 222: start of an iteration -----
      iword_ = -1;
      bool skip_cauchy = (!cnstnd_ && col_ > 0);
      if (skip_cauchy) {
        std::memcpy(z_.data(), x, sizeof(double) * n);
        wrk_ = updatd_;
        nseg_ = 0;
      } else {
        cauchy(x, g);
        if (info_ != 0) { reset_memory(); need_iteration_start = true; continue; }
        nintol_ += nseg_;
        freev();
        nact_ = n - nfree_;
      }
      // ----- 333: subspace minimisation -----
      if (nfree_ != 0 && col_ != 0) {
        if (wrk_) formk();
        if (info_ != 0) { reset_memory(); need_iteration_start = true; continue; }
        cmprlb(x, g);
        if (info_ == 0) subsm(x, g);
        if (info_ != 0) { reset_memory(); need_iteration_start = true; continue; }
      }
      // ----- 555: line search direction -----
      for (int i = 0; i < n; ++i) d_[i] = z_[i] - x[i];
      phase_ = 0;   // lnsrlb: fresh line search (anything != 2)
    }
    // ----- 666 -----
    if (resume_linesearch) { resume_linesearch = false; phase_ = 2; }
    lnsrlb(x, f, g);
    if (info_ != 0 || iback_ >= maxls_) {
      std::memcpy(x, t_.data(), sizeof(double) * n);
      std::memcpy(g, r_.data(), sizeof(double) * n);
      f = fold_;
      *fp = f;
      if (col_ == 0) {
        if (info_ == 0) { info_ = -9; --nfgv_; --ifun_; --iback_; }
        task_ = LBFGSB_ABNORMAL;
        ++iter_;
        return task_;
      }
      if (info_ == 0) --nfgv_;
      reset_memory();
      need_iteration_start = true;
      continue;
    }
    if (task_ == LBFGSB_FG) { phase_ = 2; return task_; }
    // NEW_X
    ++iter_;
    projgr(x, g);
    phase_ = 3;
    task_ = LBFGSB_NEW_X;
    return task_;
  }
}
