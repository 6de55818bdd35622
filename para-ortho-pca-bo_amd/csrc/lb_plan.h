// Work plan of the two triangular passes of the device-resident optimiser's evaluation (kernels_lbfgsb.hip: lb_eval), as a
// template over the pointer type so that the kernel builds it in LDS and the host-side checks (pcabo_debug_lbfgsb_plan,
// host_selftest.cpp under the sanitizers) run the same code on an ordinary array.
#pragma once
#define LB_THREADS 1024
#define LB_MAXNP 512
#define LB_WAVES (LB_THREADS / 64)
// per pass and wave [count, 2 x (unit, lo, hi, dest)], then per unit [first extra slot, extra slots]
#define LB_PLAN_WAVE 9
#define LB_PLAN_PASS (LB_WAVES * LB_PLAN_WAVE + 2 * (LB_MAXNP / 64))
#define LB_PLAN_INTS (2 * LB_PLAN_PASS)
#if defined(__HIPCC__)
#define LB_PLAN_FN __host__ __device__ inline
#else
#define LB_PLAN_FN inline
#endif
// the passes' partial slots: at most (waves + two-slab pairs - slabs) segments do not start their slab
static inline int lb_max_slots(int NP) { const int S = NP / 64; return 16 + S / 2 - S; }

// The triangular passes, balanced.  Unit u of pass 1 is the 64-row slab u of RT' (columns 0 .. min(n, 64 (u + 1)) - 1), unit u of
// pass 2 the 64-column block u of R (rows 64 u .. n - 1): work 1 : 2 : ... : S.  A slab per wave (split in equal parts) leaves the
// pass waiting for the longest wave - a chain of load round trips that the CU's load rate does not explain (16.4 us against
// 12.3).  Here the units are folded into pairs (largest with smallest), the 16 waves are dealt out to the pairs in proportion
// to their work, and the waves of a pair cut its columns (rows) into equal ranges: a wave gets one or two segments
// (unit, lo, hi).  The segment that starts a unit writes the unit's sums where the next phase reads them (dest -1), the others
// go to numbered partial slots that the next phase adds in ascending order - a fixed order for a given (n, NP).
// One thread, once per kernel (n and NP are the launch's).
template <typename IP>
LB_PLAN_FN void lb_build_plan_t(IP plan, int n, int S) {
  for (int i = 0; i < LB_PLAN_INTS; ++i) plan[i] = 0;
  for (int pass = 0; pass < 2; ++pass) {
    IP pw = plan + pass * LB_PLAN_PASS;
    IP pu = pw + LB_WAVES * LB_PLAN_WAVE;
    int lo[8], hi[8], order[8];
    for (int u = 0; u < S; ++u) {
      if (pass == 0) { lo[u] = 0; hi[u] = n < 64 * (u + 1) ? n : 64 * (u + 1); }
      else { lo[u] = 64 * u; hi[u] = n > 64 * u ? n : 64 * u; }
    }
    for (int t = 0; t < S; ++t) {                     // pass 1: S-1, 0, S-2, 1 ...; pass 2 (largest unit first): 0, S-1, 1, S-2 ...
      const int a = t / 2, big = pass == 0 ? S - 1 - a : a, small = pass == 0 ? a : S - 1 - a;
      order[t] = (t & 1) ? small : big;
    }
    const int P = (S + 1) / 2;
    long long total = 0;
    for (int u = 0; u < S; ++u) total += hi[u] - lo[u];
    int waves_left = LB_WAVES, slot = 0, wave = 0;
    long long work_left = total;
    for (int p = 0; p < P; ++p) {
      const int ua = order[2 * p], ub = 2 * p + 1 < S ? order[2 * p + 1] : -1;
      const int wa = hi[ua] - lo[ua], wb = ub >= 0 ? hi[ub] - lo[ub] : 0, work = wa + wb;
      int wp;
      if (p == P - 1) wp = waves_left;
      else {
        wp = work_left > 0 ? (int)(((long long)work * waves_left + work_left / 2) / work_left) : 1;
        const int keep = P - 1 - p;                   // a wave at least for every pair still to come
        if (wp > waves_left - keep) wp = waves_left - keep;
        if (wp < 1) wp = 1;
      }
      waves_left -= wp; work_left -= work;
      int chunk = (work + wp - 1) / wp;
      chunk = (chunk + 1) & ~1;                         // (pairs of columns / rows: both half-waves busy)
      if (chunk < 2) chunk = 2;
      for (int t = 0; t < wp; ++t, ++wave) {
        const int r0 = t * chunk < work ? t * chunk : work, r1 = (t + 1) * chunk < work ? (t + 1) * chunk : work;
        IP e = pw + wave * LB_PLAN_WAVE;
        int cnt = 0;
        for (int g = 0; g < 2; ++g) {                   // the range's part in unit a ([0, wa) of the pair), then in unit b
          const int u = g == 0 ? ua : ub, base = g == 0 ? 0 : wa, len = g == 0 ? wa : wb;
          if (u < 0) continue;
          const int a0 = r0 > base ? r0 - base : 0, a1 = (r1 - base) < len ? r1 - base : len;
          if (a1 <= a0) continue;
          int dest = -1;
          if (a0 > 0) { dest = slot++; if (pu[2 * u + 1] == 0) pu[2 * u] = dest; pu[2 * u + 1] += 1; }
          e[1 + 4 * cnt] = u; e[2 + 4 * cnt] = lo[u] + a0; e[3 + 4 * cnt] = lo[u] + a1; e[4 + 4 * cnt] = dest;
          ++cnt;
        }
        e[0] = cnt;
      }
    }
  }
}

