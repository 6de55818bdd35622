// Sanitizer driver for the host side of libpcabo (Makefile targets asan / ubsan / tsan; tests/test_host_sanitizers.py runs them).
// Everything here is CPU code of the product compiled by g++ with -fsanitize=...: csrc/lbfgsb.cpp, csrc/host_entry.cpp (the
// L-BFGS-B driver and the Sobol helpers), csrc/host_side.h (RestartGroup, GangPool) and csrc/lb_plan.h (the work plan of the device
// optimiser's passes).  The launcher is a stub: where the product launches an acquisition kernel and polls its flags, the
// workers here evaluate a bounded test objective on the CPU - same table packing, same per-thread tables, same pool protocol.
// Exit code 0 and "host selftest ok" = every check passed; a sanitizer report aborts the process (halt_on_error).
#include "../../include/pcabo.h"
#include "host_side.h"
#include "lb_plan.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

int g_fail = 0;
#define CHECK(c, ...) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); ++g_fail; } } while (0)

struct Lcg {                                 // deterministic inputs (no libc rand state shared between threads)
  uint64_t s;
  explicit Lcg(uint64_t seed) : s(seed * 6364136223846793005ull + 1442695040888963407ull) {}
  double uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) * (1.0 / 9007199254740992.0); }
};

// a smooth multi-modal objective with a box: f = sum (x_i - c_i)^2 (1 + 0.3 sin(3 x_i)) + 0.1 sum x_i x_{i+1}
struct Objective { std::vector<double> c; };
double fg_objective(const double* x, double* g, void* user) {
  const Objective* o = static_cast<const Objective*>(user);
  const int n = (int)o->c.size();
  double f = 0.0;
  for (int i = 0; i < n; ++i) {
    const double d = x[i] - o->c[i], s = 1.0 + 0.3 * std::sin(3.0 * x[i]);
    f += d * d * s;
    g[i] = 2.0 * d * s + d * d * 0.9 * std::cos(3.0 * x[i]);
  }
  for (int i = 0; i + 1 < n; ++i) { f += 0.1 * x[i] * x[i + 1]; g[i] += 0.1 * x[i + 1]; g[i + 1] += 0.1 * x[i]; }
  return f;
}

void test_minimize() {
  for (int nvar : {1, 2, 7, 33, 64, 65, 200}) {
    for (int variant = 0; variant < 3; ++variant) {       // scalar loops, vector kernels, the device optimiser's tree order
      pcabo_lbfgsb_set_vector_kernels(variant == 1);
      pcabo_lbfgsb_set_sum_order(variant == 2);
      Lcg r(17 + nvar);
      Objective o; o.c.resize(nvar);
      std::vector<double> x(nvar), lo(nvar), hi(nvar);
      for (int i = 0; i < nvar; ++i) { o.c[i] = 4.0 * r.uni() - 2.0; lo[i] = -1.0 - r.uni(); hi[i] = 1.0 + r.uni(); x[i] = 6.0 * r.uni() - 3.0; }
      double f = 0.0; int nit = 0, nfev = 0, task = 0;
      const int wf = pcabo_lbfgsb_minimize(nvar, x.data(), lo.data(), hi.data(), fg_objective, &o, 10, 1e7, 1e-5, 200, 15000, 20, &f, &nit, &nfev, &task);
      CHECK(wf >= 0 && wf <= 2, "warnflag %d", wf);
      CHECK(std::isfinite(f) && nfev >= 1, "f %g nfev %d", f, nfev);
      for (int i = 0; i < nvar; ++i) CHECK(x[i] >= lo[i] && x[i] <= hi[i], "x[%d] = %g outside the box", i, x[i]);
    }
  }
  pcabo_lbfgsb_set_vector_kernels(1);
  pcabo_lbfgsb_set_sum_order(0);
  // unbounded and half-bounded sides, a history longer than the default
  {
    const int nvar = 12;
    Objective o; o.c.assign(nvar, 0.5);
    std::vector<double> x(nvar, 2.0), lo(nvar, -INFINITY), hi(nvar, INFINITY);
    for (int i = 0; i < nvar; i += 3) lo[i] = 0.75;
    for (int i = 1; i < nvar; i += 3) hi[i] = 0.25;
    double f = 0.0; int nit = 0, nfev = 0, task = 0;
    const int wf = pcabo_lbfgsb_minimize(nvar, x.data(), lo.data(), hi.data(), fg_objective, &o, 17, 1e7, 1e-5, 200, 15000, 20, &f, &nit, &nfev, &task);
    CHECK(wf == 0, "half-bounded problem: warnflag %d task %d", wf, task);
  }
  CHECK(pcabo_lbfgsb_minimize(0, nullptr, nullptr, nullptr, fg_objective, nullptr, 10, 1e7, 1e-5, 1, 1, 20, nullptr, nullptr, nullptr, nullptr) == PCABO_ERR_ARG, "bad arguments accepted");
}

void test_sobol() {
  for (int k : {1, 3, 36, 89}) {
    Lcg r(5 + k);
    std::vector<int64_t> state((size_t)k * 30), ltm((size_t)k * 30 * 30), shift(k);
    for (auto& v : state) v = (int64_t)(r.uni() * 1073741824.0);
    for (auto& v : ltm) v = r.uni() < 0.5 ? 0 : 1;
    for (auto& v : shift) v = (int64_t)(r.uni() * 1073741824.0);
    CHECK(pcabo_sobol_scramble(state.data(), ltm.data(), k) == PCABO_OK, "scramble");
    for (int n : {1, 2, 512, 513}) {
      std::vector<double> lo(k, -1.5), rng(k, 3.0), out((size_t)n * k, -7.0);
      CHECK(pcabo_sobol_draw(state.data(), shift.data(), k, n, lo.data(), rng.data(), out.data()) == PCABO_OK, "draw");
      for (double v : out) CHECK(v >= -1.5 && v <= 1.5, "sample %g outside the box", v);
      CHECK(pcabo_sobol_draw(state.data(), shift.data(), k, n, nullptr, nullptr, out.data()) == PCABO_OK, "draw (unit cube)");
      for (double v : out) CHECK(v >= 0.0 && v < 1.0, "unit sample %g", v);
    }
  }
}

// torch's CPU generator restated (csrc/host_entry.cpp): state blobs of the published layout, the bit draws, the multinomial rows
// and the Boltzmann pick; all rows in one call = one call per row.
void test_torch_rng() {
  struct Blob { uint64_t seed; int32_t left, seeded; uint64_t next; uint64_t state[624]; unsigned char tail[32]; };
  const int rows = 70, n = 512, n_pick = 10;
  std::vector<Blob> blobs(rows), again;
  std::vector<double> vals((size_t)rows * n);
  for (int r = 0; r < rows; ++r) {
    Lcg g(900 + r);
    blobs[r].seed = 900 + r; blobs[r].left = 1 + (r * 37) % 624; blobs[r].seeded = 1; blobs[r].next = 624 - blobs[r].left;
    for (auto& w : blobs[r].state) w = (uint64_t)(g.uni() * 4294967296.0);
    for (auto& c : blobs[r].tail) c = 0;
    for (int i = 0; i < n; ++i) vals[(size_t)r * n + i] = r == 5 ? 1.25 : 40.0 * g.uni() - 20.0;        // row 5: all equal
  }
  again = blobs;
  std::vector<void*> ptr(rows);
  for (int r = 0; r < rows; ++r) ptr[r] = &blobs[r];
  ptr[7] = nullptr;                                                                                       // row 7: skipped
  std::vector<int64_t> out((size_t)rows * n_pick, -1), out1((size_t)rows * n_pick, -1);
  std::vector<int> flags(rows, -1), flags1(rows, -1);
  CHECK(pcabo_boltzmann_pick_rows(ptr.data(), vals.data(), rows, n, n_pick, 1.0, out.data(), flags.data()) == PCABO_OK, "pick rows");
  for (int r = 0; r < rows; ++r) {                            // one row per call
    void* p1 = r == 7 ? nullptr : (void*)&again[r];
    CHECK(pcabo_boltzmann_pick_rows(&p1, vals.data() + (size_t)r * n, 1, n, n_pick, 1.0, out1.data() + (size_t)r * n_pick, flags1.data() + r) == PCABO_OK, "pick row %d", r);
    CHECK(flags[r] == flags1[r] && flags[r] == (r == 7 ? 2 : r == 5 ? 1 : 0), "flag of row %d: %d / %d", r, flags[r], flags1[r]);
    if (flags[r] != 0) continue;
    for (int j = 0; j < n_pick; ++j) {
      CHECK(out[(size_t)r * n_pick + j] == out1[(size_t)r * n_pick + j], "row %d pick %d differs between the all-rows call and the one-row call", r, j);
      CHECK(out[(size_t)r * n_pick + j] >= 0 && out[(size_t)r * n_pick + j] < n, "row %d pick %d out of range", r, j);
    }
    CHECK(blobs[r].left == again[r].left && blobs[r].next == again[r].next && blobs[r].state[0] == again[r].state[0], "generator of row %d", r);
  }
  std::vector<int64_t> bits(2000);
  CHECK(pcabo_torch_randint2(&blobs[0], 2000, bits.data()) == PCABO_OK, "randint2");
  for (int64_t b : bits) CHECK(b == 0 || b == 1, "bit %lld", (long long)b);
  std::vector<double> w((size_t)3 * n);
  Lcg g(77);
  for (auto& v : w) v = 0.01 + g.uni();
  void* three[3] = {&blobs[1], nullptr, &blobs[2]};
  std::vector<int64_t> idx(3 * n_pick, -1);
  CHECK(pcabo_torch_multinomial_rows(three, w.data(), 3, n, n_pick, idx.data()) == PCABO_OK, "multinomial rows");
  for (int j = 0; j < n_pick; ++j) CHECK(idx[j] >= 0 && idx[j] < n && idx[n_pick + j] == -1 && idx[2 * n_pick + j] >= 0, "multinomial pick %d", j);
  blobs[3].left = 900;
  CHECK(pcabo_torch_randint2(&blobs[3], 4, bits.data()) == PCABO_ERR_ARG, "a broken blob accepted");
}

// pcabo_sobol_draw_rows: ragged k, a skipped run, the boxes in pcabo_batch_acq_bounds' packing - equal to the per-run calls
void test_sobol_rows() {
  const int rows = 37, n = 130, kmax = 36;          // (more rows than a small batch)
  int ks[rows];
  for (int r = 0; r < rows; ++r) ks[r] = r == 0 ? 3 : r == 2 ? 1 : 1 + (r * 7) % 36;
  std::vector<std::vector<int64_t>> st(rows), sh(rows);
  std::vector<double> boxes((size_t)rows * 2 * kmax, NAN), out((size_t)rows * n * kmax, -7.0), one((size_t)n * kmax);
  std::vector<const int64_t*> sp(rows), hp(rows);
  std::vector<double*> op(rows);
  for (int r = 0; r < rows; ++r) {
    Lcg g(40 + r);
    st[r].resize((size_t)ks[r] * 30); sh[r].resize(ks[r]);
    for (auto& v : st[r]) v = (int64_t)(g.uni() * 1073741824.0);
    for (auto& v : sh[r]) v = (int64_t)(g.uni() * 1073741824.0);
    for (int j = 0; j < ks[r]; ++j) { boxes[(size_t)r * 2 * kmax + j] = -1.0 - j; boxes[(size_t)r * 2 * kmax + ks[r] + j] = 2.0 + r; }
    sp[r] = r == 2 ? nullptr : st[r].data(); hp[r] = sh[r].data(); op[r] = out.data() + (size_t)r * n * kmax;
  }
  CHECK(pcabo_sobol_draw_rows(sp.data(), hp.data(), ks, rows, n, boxes.data(), 2 * kmax, op.data()) == PCABO_OK, "draw rows");
  for (int r = 0; r < rows; ++r) {
    if (r == 2) { CHECK(op[r][0] == -7.0, "a skipped run was written"); continue; }
    std::vector<double> rng(ks[r]);
    for (int j = 0; j < ks[r]; ++j) rng[j] = boxes[(size_t)r * 2 * kmax + ks[r] + j] - boxes[(size_t)r * 2 * kmax + j];
    CHECK(pcabo_sobol_draw(st[r].data(), sh[r].data(), ks[r], n, boxes.data() + (size_t)r * 2 * kmax, rng.data(), one.data()) == PCABO_OK, "draw");
    for (int i = 0; i < n * ks[r]; ++i) CHECK(one[i] == op[r][i], "run %d element %d", r, i);
    if (ks[r] < kmax) CHECK(op[r][(size_t)n * ks[r]] == -7.0, "write behind the points of run %d", r);
  }
}

void test_plan() {
  std::vector<int> plan(LB_PLAN_INTS + 64, 0x5a5a5a5a);        // guard words behind the plan: they must stay
  for (int NP = 64; NP <= LB_MAXNP; NP += 64) {
    const int S = NP / 64;
    for (int n = NP - 63; n <= NP; ++n) {
      lb_build_plan_t(plan.data(), n, S);
      for (int i = LB_PLAN_INTS; i < (int)plan.size(); ++i) CHECK(plan[i] == 0x5a5a5a5a, "write behind the plan (n %d)", n);
      for (int pass = 0; pass < 2; ++pass) {
        const int* pw = plan.data() + pass * LB_PLAN_PASS;
        std::vector<int> cover(S * 512, 0);
        int max_dest = -1;
        for (int w = 0; w < LB_WAVES; ++w) {
          const int* e = pw + w * LB_PLAN_WAVE;
          CHECK(e[0] >= 0 && e[0] <= 2, "segments of a wave: %d", e[0]);
          for (int g = 0; g < e[0]; ++g) {
            const int u = e[1 + 4 * g], a = e[2 + 4 * g], b = e[3 + 4 * g], dest = e[4 + 4 * g];
            CHECK(u >= 0 && u < S && a >= 0 && b <= NP && a < b, "segment (%d, %d, %d)", u, a, b);
            CHECK(dest >= -1 && dest < lb_max_slots(NP), "slot %d of %d (n %d NP %d)", dest, lb_max_slots(NP), n, NP);
            if (dest > max_dest) max_dest = dest;
            for (int t = a; t < b && u >= 0 && u < S && t >= 0 && t < 512; ++t) cover[u * 512 + t] += 1;
          }
        }
        for (int u = 0; u < S; ++u) {
          const int lo = pass == 0 ? 0 : 64 * u, hi = pass == 0 ? (n < 64 * (u + 1) ? n : 64 * (u + 1)) : (n > 64 * u ? n : 64 * u);
          for (int t = 0; t < 512; ++t) CHECK(cover[u * 512 + t] == ((t >= lo && t < hi) ? 1 : 0), "pass %d unit %d index %d covered %d times (n %d)", pass, u, t, cover[u * 512 + t], n);
        }
      }
    }
  }
}

// RestartGroups of several "runs" stepped by the pool's workers, each worker with its own launch table on its stack (the
// product's shape: run << 16 | first query << 8 | count), values and gradients written into per-run blocks and absorbed.
void test_gang_pool(int workers, int runs) {
  const int k = 7, nq = 5, ngroups = 2, restarts = nq * ngroups;
  std::vector<std::vector<RestartGroup>> groups(runs);
  std::vector<Objective> obj(runs);
  std::vector<std::vector<double>> hXq(runs), hVal(runs), hGrad(runs);
  for (int b = 0; b < runs; ++b) {
    Lcg r(100 + b);
    obj[b].c.resize(k);
    for (auto& v : obj[b].c) v = 2.0 * r.uni() - 1.0;
    std::vector<double> ics((size_t)restarts * k), bounds(2 * k);
    for (auto& v : ics) v = 4.0 * r.uni() - 2.0;
    for (int c = 0; c < k; ++c) { bounds[c] = -1.25; bounds[k + c] = 1.5; }
    groups[b].resize(ngroups);
    for (int gi = 0; gi < ngroups; ++gi) {
      groups[b][gi].init(ics.data(), bounds.data(), gi * nq, nq, k, 200);
      groups[b][gi].opt.set_sum_order(b & 1);          // every second run steps in the device optimiser's order (the twin's setting)
    }
    hXq[b].assign((size_t)restarts * k, 0.0); hVal[b].assign(restarts, 0.0); hGrad[b].assign((size_t)restarts * k, 0.0);
  }
  GangPool pool;
  pool.start(workers);
  std::atomic<int> rounds{0}, bad{0};
  for (int call = 0; call < 3; ++call) {              // the pool is reused across calls, as a batch does per BO iteration
    if (call > 0) for (int b = 0; b < runs; ++b) for (auto& rg : groups[b]) { rg.active = true; rg.niter = 0; rg.have_cache = false; rg.opt.init(nq * k, 10, rg.lo.data(), rg.hi.data(), 1e7, 1e-5, 20); }
    pool.run([&](int t) {
      unsigned table[64];                              // the worker's own launch table
      struct Pending { int b, gi; };
      std::vector<Pending> pend;
      for (;;) {
        pend.clear();
        int nent = 0;
        for (int b = t; b < runs; b += workers)
          for (int gi = 0; gi < ngroups; ++gi) {
            RestartGroup& rg = groups[b][gi];
            if (!rg.active) continue;
            rg.advance();
            if (!rg.active) continue;
            std::memcpy(hXq[b].data() + (size_t)rg.q0 * k, rg.x.data(), (size_t)rg.nq * k * sizeof(double));
            if (nent < 64) table[nent++] = ((unsigned)b << 16) | ((unsigned)rg.q0 << 8) | (unsigned)rg.nq;
            pend.push_back({b, gi});
          }
        if (nent == 0) break;
        for (int e = 0; e < nent; ++e) {               // the stub launcher: what the kernel would do with the table entry
          const int b = (int)(table[e] >> 16), q0 = (int)((table[e] >> 8) & 0xffu), cnt = (int)(table[e] & 0xffu);
          for (int j = 0; j < cnt; ++j) {
            std::vector<double> g(k);
            const double f = fg_objective(hXq[b].data() + (size_t)(q0 + j) * k, g.data(), &obj[b]);
            hVal[b][q0 + j] = -f;                      // (the acquisition is maximised: RestartGroup::absorb negates)
            for (int c = 0; c < k; ++c) hGrad[b][(size_t)(q0 + j) * k + c] = -g[c];
          }
        }
        for (const Pending& pe : pend) if (!groups[pe.b][pe.gi].absorb(hVal[pe.b].data(), hGrad[pe.b].data())) bad.fetch_add(1);
        rounds.fetch_add(1);
      }
    });
    for (int b = 0; b < runs; ++b)
      for (auto& rg : groups[b]) {
        CHECK(!rg.active, "a group is still active after the call");
        CHECK(rg.opt.warnflag() >= 0 && rg.opt.warnflag() <= 2, "warnflag");
        for (size_t i = 0; i < rg.x.size(); ++i) CHECK(rg.x[i] >= rg.lo[i] && rg.x[i] <= rg.hi[i], "end point outside the box");
      }
  }
  pool.shutdown();
  CHECK(bad.load() == 0, "NaN gradients: %d", bad.load());
  CHECK(rounds.load() > 0, "no rounds");
}

}  // namespace

int main(int argc, char** argv) {
  const int workers = argc > 1 ? std::atoi(argv[1]) : 4;
  test_minimize();
  test_sobol();
  test_sobol_rows();
  test_torch_rng();
  test_plan();
  test_gang_pool(1, 3);
  test_gang_pool(workers, 11);
  if (g_fail) { std::fprintf(stderr, "host selftest: %d check(s) failed\n", g_fail); return 1; }
  std::printf("host selftest ok\n");
  return 0;
}
