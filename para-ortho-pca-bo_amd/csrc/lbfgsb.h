// Host-side limited-memory BFGS with box bounds (L-BFGS-B 3.0: Byrd, Lu, Nocedal, Zhu 1995;
// Morales & Nocedal 2011), reverse-communication form.
//
// The reference reaches this algorithm through botorch.optimize_acqf -> gen_candidates_scipy ->
// scipy.optimize.minimize(method="L-BFGS-B") (/root/reference/Algorithms/BayesianOptimization/
// PCA_BO.py:607-614).  This is a from-scratch restatement of the published algorithm (generalised
// Cauchy point, subspace minimisation with the 3.0 projection step, More-Thuente line search
// dcsrch/dcstep, compact L-BFGS matrices) written so that, fed the same f/g values, it takes the
// same iterates as scipy's implementation; tests/test_lbfgsb_vs_scipy.py pins it against scipy.
#pragma once
#include <vector>

#define LBFGSB_MAXM 32   // largest history length supported (scipy's default is 10)

enum {
  LBFGSB_START = 0,
  LBFGSB_NEW_X = 1,        // an iteration finished; caller may stop() or call step() again
  LBFGSB_FG = 3,           // caller must evaluate f and g at x, then call step() again
  LBFGSB_CONV_PG = 40,     // CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL
  LBFGSB_CONV_F = 41,      // CONVERGENCE: REL_REDUCTION_OF_F <= FACTR*EPSMCH
  LBFGSB_STOP_ITER = 50,   // STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT   (set by the caller)
  LBFGSB_STOP_FUN = 51,    // STOP: TOTAL NO. OF F,G EVALUATIONS EXCEEDS LIMIT (set by the caller)
  LBFGSB_ABNORMAL = 70,    // ABNORMAL TERMINATION IN LNSRCH
  LBFGSB_ERROR = 90        // invalid input (l > u, ...)
};

// 0: scalar O(m n) loops, 1: the default (AVX2 where available); same iterates either way.  Returns the previous setting.
int lbfgsb_set_vector_kernels(int enabled);
// Summation order of the sums over the variables for optimisers initialised from now on: 0 the published order (scipy's iterates),
// 1 the 64-lane tree order the device-resident optimiser steps in (see lbfgsb.cpp).  Returns the previous default.
int lbfgsb_set_default_sum_order(int order);

class Lbfgsb {
 public:
  // lower/upper may be null (unbounded); +-inf entries mean "no bound on that side".
  void init(int n, int m, const double* lower, const double* upper, double factr, double pgtol, int maxls);
  int step(double* x, double* f, double* g);
  void stop(int code) { task_ = code; }
  int task() const { return task_; }
  // scipy's warnflag: 0 converged, 1 iteration/evaluation limit, 2 anything else
  int warnflag() const {
    if (task_ == LBFGSB_CONV_PG || task_ == LBFGSB_CONV_F) return 0;
    if (task_ == LBFGSB_STOP_ITER || task_ == LBFGSB_STOP_FUN) return 1;
    return 2;
  }
  int iterations() const { return iter_; }
  void set_sum_order(int order) { sum_order_ = order ? 1 : 0; }      // (after init, before the first step)
  int sum_order() const { return sum_order_; }

 private:
  // problem
  int n_ = 0, m_ = 0, maxls_ = 20;
  int sum_order_ = 0;        // 0: the published order, 1: the 64-lane tree order (the device optimiser's twin)
  double factr_ = 1e7, pgtol_ = 1e-5;
  std::vector<double> l_, u_;
  std::vector<int> nbd_;
  // L-BFGS matrices (column-major, Fortran layout)
  std::vector<double> ws_, wy_, sy_, ss_, wt_, wn_, snd_;
  std::vector<double> wr_, sc_coef_, sc_full_;   // row-major mirror of WY|WS (stride rs_) and scratch, see lbfgsb.cpp
  std::vector<int> sc_rows_;
  int mp_ = 0, rs_ = 0;
  std::vector<double> z_, r_, d_, t_, xp_, wa_;
  std::vector<int> index_, iwhere_, indx2_;
  // scalars of mainlb
  bool prjctd_ = false, cnstnd_ = false, boxed_ = false, updatd_ = false, wrk_ = false;
  int nintol_ = 0, iback_ = 0, nskip_ = 0, head_ = 0, col_ = 0, itail_ = 0, iter_ = 0, iupdat_ = 0;
  int nseg_ = 0, nfgv_ = 0, info_ = 0, ifun_ = 0, iword_ = 0, nfree_ = 0, nact_ = 0, ileave_ = 0, nenter_ = 0;
  double theta_ = 1, fold_ = 0, tol_ = 0, dnorm_ = 0, epsmch_ = 0, gd_ = 0, stpmx_ = 0, sbgnrm_ = 0, stp_ = 0;
  double gdold_ = 0, dtd_ = 0, xstep_ = 0;
  int task_ = LBFGSB_START;
  int phase_ = 0;            // 0 start, 1 waiting for f,g at x0, 2 inside the line search, 3 after NEW_X
  // dcsrch state
  struct Dcsrch {
    int task = 0;            // 0 START, 1 FG, 2 CONVERGENCE, 3 WARNING, 4 ERROR
    bool brackt = false;
    int stage = 1;
    double ginit = 0, gtest = 0, gx = 0, gy = 0, finit = 0, fx = 0, fy = 0, stx = 0, sty = 0, stmin = 0, stmax = 0;
    double width = 0, width1 = 0;
  } ls_;

  void reset_memory();
  void projgr(const double* x, const double* g);
  bool active_init(double* x);
  void cauchy(const double* x, const double* g);
  void hpsolb(int n, double* t, int* iorder, int iheap);
  void bmv(const double* v, double* p);
  void freev();
  void formk();
  void cmprlb(const double* x, const double* g);
  void subsm(const double* x, const double* g);
  void lnsrlb(double* x, double f, const double* g);
  void matupd(double rr, double dr);
  void formt();
  static void dcsrch(double f, double g, double* stp, double ftol, double gtol, double xtol, double stpmin,
                     double stpmax, Dcsrch& s);
  static void dcstep(double* stx, double* fx, double* dx, double* sty, double* fy, double* dy, double* stp, double fp,
                     double dp, bool* brackt, double stpmin, double stpmax);
};
