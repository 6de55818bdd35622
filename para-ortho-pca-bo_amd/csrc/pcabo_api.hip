// C-ABI of libpcabo.so (declared in include/pcabo.h): context management, host<->device staging and
// the launch sequences of the wPCA / GP-conditioning / acquisition phases.
#include "../../include/pcabo.h"
#include "pcabo_internal.h"
#include "lbfgsb.h"
#include "host_side.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>
#include <dirent.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <cerrno>
#include <unistd.h>
#include <emmintrin.h>

#define PCABO_ABI_VERSION 1
#define PROF_GROUPS 6
#define PROF_POOL 4096

struct ProfPair { hipEvent_t a, b; int group; double bytes, flops; };

// One helper thread per context for the odd-numbered restart groups of pcabo_optimize_acqf.  It lives as long as the
// context (creating and joining a thread per call cost ~0.1 ms per BO iteration), sleeps on a condition variable
// between calls and spin-waits on `go` only while a call is active.  `go`/`done` are monotonic round counters.
struct OptHelper {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<unsigned> go{0}, done{0};
  std::atomic<bool> quit{false}, active{false};
  std::function<void()> fn;          // written by the caller only while the helper is idle (done == go)
  void run() {
    unsigned seen = 0;
    while (true) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return active.load(std::memory_order_acquire) || quit.load(std::memory_order_acquire); });
      }
      if (quit.load(std::memory_order_acquire)) return;
      while (active.load(std::memory_order_acquire)) {
        const unsigned cur = go.load(std::memory_order_acquire);
        if (cur == seen) { __builtin_ia32_pause(); continue; }
        seen = cur;
        fn();
        done.store(cur, std::memory_order_release);
      }
    }
  }
  void start() { if (!th.joinable()) th = std::thread([this] { run(); }); }
  void begin(std::function<void()> f) {
    start();
    fn = std::move(f);
    { std::lock_guard<std::mutex> lk(mu); active.store(true, std::memory_order_release); }
    cv.notify_one();
  }
  void end() { active.store(false, std::memory_order_release); }
  void shutdown() {
    if (!th.joinable()) return;
    { std::lock_guard<std::mutex> lk(mu); quit.store(true, std::memory_order_release); active.store(false, std::memory_order_release); }
    cv.notify_one();
    th.join();
  }
};

struct pcabo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t evBounds = nullptr;         // recorded after k_zstats: the search box is on the host before the Cholesky ends
  hipEvent_t evPca = nullptr;            // recorded after the wPCA results left for the host (conditioning may follow it)
  int max_n = 0, max_d = 0, max_q = 0;
  int NPcap = 0, ld = 0, DPcap = 0, KPcap = 0, Scap = 0;
  int ptr_mode = PCABO_PTR_HOST;
  // problem state
  int n = 0, d = 0, k = 0, NP = 0, KP = 0;
  bool have_wpca = false, have_gp = false, gp_pending = false;
  int wpca_n = 0, wpca_d = 0; bool wpca_uncollected = false;   // a wPCA whose results have not been waited for yet
  double lengthscale = 0.0, noise = 0.0;
  int kernel = 0;
  // device buffers
  double *dX = nullptr, *dNoise = nullptr, *dF = nullptr, *dWeights = nullptr, *dWc = nullptr;
  long long* dRanks = nullptr;
  double *dDataMean = nullptr, *dPcaMean = nullptr, *dC = nullptr, *dG = nullptr, *dLam = nullptr;
  double* dGbuf[2] = {nullptr, nullptr};   // eigenvector ping-pong: the previous result warm-starts the next Jacobi
  int gcur = 0, vprev_d = 0;
  double *dComps = nullptr, *dEvr = nullptr, *dZ = nullptr;
  double* dPcaOut = nullptr;             // [data_mean | pca_mean | evr | comps]: one allocation, one copy to the host
  double *dIn = nullptr, *hIn = nullptr; // host-pointer mode: X, noise, ranks (or f) and y travel packed in ONE copy
  int *dK = nullptr, *dSweeps = nullptr, *dInfo = nullptr;
  double *dY = nullptr, *dYs = nullptr, *dYstats = nullptr, *dBounds4 = nullptr, *dZnMean = nullptr, *dUserNB = nullptr;
  double *dZnT = nullptr, *dAT = nullptr, *dNrm = nullptr, *dGram = nullptr, *dL = nullptr, *dR = nullptr;
  double *dTmp = nullptr, *dAlpha = nullptr, *dDiag = nullptr;   // dDiag: 64x64 hand-over tile of the Cholesky panels
  double *dXq = nullptr, *dPartial = nullptr, *dVal = nullptr, *dGrad = nullptr, *dZq = nullptr, *dXout = nullptr;
  unsigned int* dCounters = nullptr;     // per-query tickets of the in-launch combine
  int cnt_S = 0; bool cnt_dirty = true;  // slab-group count the tickets are consistent with / a launch may have died
  double* dKS = nullptr;                 // q x ld kernel vectors of the GEMM scoring path
  double* dBestF = nullptr;              // best_f of this run for the batched acquisition launches (set by the batch)
  char *dRegion = nullptr, *hRegion = nullptr;   // the two allocations everything above / below is carved from
  size_t region_bytes = 0, hregion_bytes = 0;
  bool in_batch = false; int batch_index = 0;
  bool rt_stale = false;                 // pcabo_get_gram has overwritten the transposed root inverse kept in dGram (device-resident L-BFGS-B)
  // pinned host
  HostMirror* hm = nullptr;
  double *hXq = nullptr, *hVal = nullptr, *hGrad = nullptr, *hSmall = nullptr, *hBestF = nullptr;
  MailPair* dMail = nullptr;             // mailbox of the resident acquisition kernel, in device memory
  bool mail_bar = false;                 // the host can write dMail itself through the PCIe BAR (the resident mode needs it)
  MailPair* dPairs = nullptr;            // its partial records as (value, tag) pairs: 32 queries x 32 slabs
  bool opt_resident = true;             // PCABO_OPT_RESIDENT: may pcabo_optimize_acqf use the resident acquisition kernel
  bool opt_group_acq = false;           // PCABO_OPT_GROUP_ACQ: gradient evaluations through k_acq_group (throughput variant)
  bool bestf_f32 = true;                // PCABO_OPT_BESTF_F32: best_f rounded like torch.as_tensor(python float)
  int srv_penalty = 0;                   // > 0: the next calls use plain launches (the GPU looked shared, see pcabo_optimize_acqf)
  unsigned long long seq = 0;
  OptHelper helper;
  std::vector<std::unique_ptr<OptHelper>> more_helpers;   // many restart groups (>= 12): more threads step the state machines
  bool registered = false;
  bool alone = true; int alone_age = 0;   // cached answer of presence_alone(), refreshed every few calls
  char err[512] = {0};
  // profiling
  bool prof = false;
  std::vector<ProfPair> pairs;
  size_t pairs_used = 0;
  double prof_ms[PROF_GROUPS] = {0};
  double prof_pair_ms = 0.0;             // reading of two events recorded back to back (calibrated when profiling is enabled)
  double prof_empty_ms = 0.0;            // reading of an event pair around an empty kernel
  int64_t prof_launches[PROF_GROUPS] = {0};
  double prof_bytes[PROF_GROUPS] = {0}, prof_flops[PROF_GROUPS] = {0};
};

// ---- who else drives this GPU?  ---------------------------------------------------------------------------------
// The resident acquisition kernel holds most of the chip for a whole optimize call.  That is the fastest way to run ONE
// process per GPU (the contract of this package), but when several processes share a device their resident kernels can
// only run one after the other, whole calls at a time (measured: 4 processes 140 instead of 336 it/s in aggregate).
// Every process therefore leaves a marker /dev/shm/pcabo.<device>.<pid> while it has a context on a device, and the
// resident mode is used only by a process that finds itself alone there (markers of dead processes are removed).
static std::mutex g_presence_mu;
static int g_presence_refs[64] = {0};
static void presence_path(char* buf, size_t n, int device, long pid) { snprintf(buf, n, "/dev/shm/pcabo.%d.%ld", device, pid); }
static void presence_register(int device) {
  if (device < 0 || device >= 64) return;
  std::lock_guard<std::mutex> lk(g_presence_mu);
  if (g_presence_refs[device]++ == 0) {
    char path[128]; presence_path(path, sizeof(path), device, (long)getpid());
    int fd = open(path, O_CREAT | O_WRONLY, 0644);
    if (fd >= 0) close(fd);
  }
}
static void presence_unregister(int device) {
  if (device < 0 || device >= 64) return;
  std::lock_guard<std::mutex> lk(g_presence_mu);
  if (g_presence_refs[device] > 0 && --g_presence_refs[device] == 0) {
    char path[128]; presence_path(path, sizeof(path), device, (long)getpid());
    unlink(path);
  }
}
// contexts of THIS process on the device (stand-alone ones and the members of batches alike)
static int presence_local_contexts(int device) {
  if (device < 0 || device >= 64) return 1;
  std::lock_guard<std::mutex> lk(g_presence_mu);
  return g_presence_refs[device];
}
static bool presence_alone(int device) {
  DIR* d = opendir("/dev/shm");
  if (!d) return true;
  char prefix[64]; snprintf(prefix, sizeof(prefix), "pcabo.%d.", device);
  const size_t pl = strlen(prefix);
  bool alone = true;
  while (struct dirent* e = readdir(d)) {
    if (strncmp(e->d_name, prefix, pl) != 0) continue;
    const long pid = atol(e->d_name + pl);
    if (pid <= 0 || pid == (long)getpid()) continue;
    if (kill((pid_t)pid, 0) == 0 || errno == EPERM) { alone = false; break; }
    char path[300]; snprintf(path, sizeof(path), "/dev/shm/%s", e->d_name);
    unlink(path);                                   // left behind by a process that is gone
  }
  closedir(d);
  return alone;
}

static int set_err(pcabo_ctx* c, int code, const char* fmt, const char* a = "", int v = 0) {
  if (c) snprintf(c->err, sizeof(c->err), fmt, a, v);
  if (c && (code == PCABO_ERR_TIMEOUT || code == PCABO_ERR_HIP)) c->cnt_dirty = true;   // a launch may have died half-way
  return code;
}
#define HIPCHK(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) return set_err(ctx, PCABO_ERR_HIP, "HIP error: %s (line %d)", hipGetErrorString(e_), __LINE__); \
  } while (0)

// ---- waiting for the device -----------------------------------------------------------------------------------
// hipStreamSynchronize / hipEventSynchronize put the calling thread to sleep once the wait lasts longer than the
// runtime's short active-wait window; it then comes back late and cold (measured: the first ~0.15 ms of host work after a
// 0.2 ms wait ran several times slower).  The waits of a BO iteration are 0.1-0.4 ms, so they poll instead, and only a
// wait that lasts longer than 2 ms falls back to the blocking call.
static hipError_t wait_stream(hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e == hipSuccess ? hipStreamSynchronize(s) : e;
    if ((spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    __builtin_ia32_pause();
  }
  return hipStreamSynchronize(s);
}
static hipError_t wait_event(hipEvent_t ev) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if ((spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    __builtin_ia32_pause();
  }
  return hipEventSynchronize(ev);
}

// ---- the resident kernel's mailbox, written by the host straight into device memory --------------------------------
// With a large PCIe BAR the CPU can address device memory.  The host then stores the round's (value, tag) pairs into the
// DEVICE mailbox itself (16-byte stores, write-combined, one sfence) and the relay work-group that used to fetch them
// from pinned host memory over PCIe has nothing to do: 5.5 -> 4.6 us per round trip in profiles/tools/bar_mailbox_rtt.hip.
// Used when the device reports a large BAR, the process maps the mailbox read-write (/proc/self/maps - no store is tried
// otherwise, no fault is caught) AND a probe store is read back from the device.  Without it there is no resident mode
// (round 1's relay work-group, which fetched the pairs from pinned host memory, is gone): one launch per evaluation.
static inline void put_mail_pair(MailPair* dst, double v, unsigned long long seq) {
  unsigned long long vb; memcpy(&vb, &v, 8);
  const __m128i x = _mm_set_epi64x((long long)(seq ^ mail_mix(vb)), (long long)vb);
  _mm_store_si128(reinterpret_cast<__m128i*>(dst), x);        // value and tag leave in one store
}
// Can the CPU store to [p, p+len)?  Answered from /proc/self/maps: on a large-BAR system the runtime maps device
// allocations into the process read-write; without host access the range is reserved without write permission.  No
// store is attempted unless the mapping says it is allowed, so nothing can fault and no signal handler is touched.
static bool host_mapping_writable(const void* p, size_t len) {
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return false;
  const unsigned long long a = (unsigned long long)(uintptr_t)p, b = a + len;
  char line[512];
  bool ok = false;
  while (fgets(line, sizeof(line), f)) {
    unsigned long long lo = 0, hi = 0;
    char perms[8] = {0};
    if (sscanf(line, "%llx-%llx %7s", &lo, &hi, perms) != 3) continue;
    if (a >= lo && b <= hi) { ok = perms[0] == 'r' && perms[1] == 'w'; break; }
  }
  fclose(f);
  return ok;
}
static bool mail_bar_usable(pcabo_ctx* ctx) {
  int large = 0;
  if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, ctx->device) != hipSuccess) { (void)hipGetLastError(); large = 0; }
  if (!large) return false;
  if (!host_mapping_writable(ctx->dMail, PCABO_MAIL_PAIRS * sizeof(MailPair))) return false;
  // the mapping is writable: verify that a store through it is what the device sees (read back with a copy)
  MailPair* probe = ctx->dMail + (PCABO_MAIL_PAIRS - 1);
  put_mail_pair(probe, 42.5, 0x1234ull);
  _mm_sfence();
  MailPair back{0.0, 0};
  if (hipMemcpy(&back, probe, sizeof(back), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
  unsigned long long vb; const double v = 42.5; memcpy(&vb, &v, 8);
  const bool seen = back.v == 42.5 && back.tag == (0x1234ull ^ mail_mix(vb));
  put_mail_pair(probe, 0.0, 0); _mm_sfence();
  return seen;
}

// ---- profiling helpers ------------------------------------------------------------------------
__global__ void k_nop() {}

static void prof_resolve(pcabo_ctx* c) {
  if (c->pairs_used == 0) return;
  (void)hipStreamSynchronize(c->stream);
  for (size_t i = 0; i < c->pairs_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->pairs[i].a, c->pairs[i].b) == hipSuccess) {
      // two events recorded back to back already read c->prof_pair_ms; the rest is attributed to the launch
      ms = ms > (float)c->prof_pair_ms ? ms - (float)c->prof_pair_ms : 0.f;
      c->prof_ms[c->pairs[i].group] += ms;
      c->prof_launches[c->pairs[i].group] += 1;
      c->prof_bytes[c->pairs[i].group] += c->pairs[i].bytes;
      c->prof_flops[c->pairs[i].group] += c->pairs[i].flops;
    }
  }
  c->pairs_used = 0;
}
struct ProfScope {
  pcabo_ctx* c; ProfPair* p = nullptr;
  // bytes / flops: ALGORITHMIC work of the bracketed launches (formulas in DESIGN.md section 4)
  ProfScope(pcabo_ctx* ctx, int group, double bytes = 0.0, double flops = 0.0) : c(ctx) {
    if (!c->prof) return;
    if (c->pairs_used == c->pairs.size()) prof_resolve(c);
    p = &c->pairs[c->pairs_used++];
    p->group = group; p->bytes = bytes; p->flops = flops;
    (void)hipEventRecord(p->a, c->stream);
  }
  ~ProfScope() { if (p) (void)hipEventRecord(p->b, c->stream); }
};

// ---- algorithmic work models (per launch), see DESIGN.md section 4 ----------------------------
static double acq_bytes(int n, int k, int q, int grad) {
  return 4.0 * n * (n + 1.0) + 8.0 * n * k + 8.0 * n + 8.0 * q * k + 8.0 * q * (grad ? 1 + k : 1);
}
static double acq_flops(int n, int k, int q, int grad) {
  double v = 3.0 * n * k + 20.0 * n + (double)n * n;          // ks, mu, v = R ks (triangular)
  if (grad) v += (double)n * n + 4.0 * n * k;                   // w = R^T v, contraction with dks/dx
  return q * v;
}

template <typename T>
static hipError_t dalloc(T** p, size_t count) { return hipMalloc((void**)p, count * sizeof(T)); }

extern "C" {

int pcabo_abi_version(void) { return PCABO_ABI_VERSION; }

int pcabo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C" (helpers with C++ linkage follow)

// ---- memory of a context --------------------------------------------------------------------------------------------
// All device buffers of a context are carved out of ONE allocation (its "region"), all pinned host buffers out of another.
// A stand-alone context owns both.  The contexts of a batch (pcabo_batch_create) live side by side in one device slab and
// one pinned slab with the SAME layout, so that buffer X of run b sits at (X of run 0) + b * region_bytes: the batched
// launches address a run's operands as `pointer + blockIdx.z * stride` with a single stride for every buffer.
struct Carver {
  char* base; size_t off = 0;
  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};
static size_t carve_device(pcabo_ctx* ctx, char* base) {
  Carver c{base};
  const size_t N = ctx->NPcap, D = ctx->DPcap, n = ctx->max_n, d = ctx->max_d, Q = ctx->max_q;
  ctx->dX = c.take<double>(n * d);        ctx->dNoise = c.take<double>(n * d);
  ctx->dF = c.take<double>(n);            ctx->dWeights = c.take<double>(n);
  ctx->dWc = c.take<double>((n + 4) * D); ctx->dRanks = c.take<long long>(n);
  ctx->dPcaOut = c.take<double>(3 * d + d * d);
  ctx->dDataMean = ctx->dPcaOut; ctx->dPcaMean = ctx->dPcaOut + d; ctx->dEvr = ctx->dPcaOut + 2 * d;
  ctx->dComps = ctx->dPcaOut + 3 * d;
  ctx->dIn = c.take<double>(n * (2 * d + 2));
  ctx->dC = c.take<double>(D * D);        ctx->dGbuf[0] = c.take<double>(d * d); ctx->dGbuf[1] = c.take<double>(d * d);
  ctx->dG = ctx->dGbuf[0];
  ctx->dLam = c.take<double>(d);          ctx->dZ = c.take<double>(n * d);
  ctx->dK = c.take<int>(1);               ctx->dSweeps = c.take<int>(1);         ctx->dInfo = c.take<int>(1);
  ctx->dY = c.take<double>(n);            ctx->dYs = c.take<double>(N);
  ctx->dYstats = c.take<double>(2);       ctx->dBounds4 = c.take<double>(4 * PCABO_MAXD);
  ctx->dZnMean = c.take<double>(PCABO_MAXD); ctx->dUserNB = c.take<double>(2 * PCABO_MAXD);
  ctx->dZnT = c.take<double>((size_t)ctx->KPcap * N); ctx->dAT = c.take<double>((size_t)ctx->KPcap * N);
  ctx->dNrm = c.take<double>(N);
  ctx->dGram = c.take<double>(N * N);     ctx->dL = c.take<double>(N * N);       ctx->dR = c.take<double>(N * N);
  ctx->dTmp = c.take<double>(N);          ctx->dAlpha = c.take<double>(N);
  ctx->dDiag = c.take<double>((size_t)2 * PCABO_BS * PCABO_BS);      // two hand-over tiles (k_chol_step)
  ctx->dXq = c.take<double>(Q * d + 8);   // (+ the run's best_f behind the query block in batched scoring)
  ctx->dPartial = c.take<double>(Q * (size_t)ctx->Scap * (2 + 2 * PCABO_MAXD));
  ctx->dVal = c.take<double>(Q);          ctx->dGrad = c.take<double>(Q * d);
  ctx->dZq = c.take<double>(d);           ctx->dXout = c.take<double>(d);
  ctx->dBestF = c.take<double>(2);
  ctx->dKS = c.take<double>(Q * N);        // kernel vectors of a large value-only batch (GEMM scoring)
  ctx->dCounters = c.take<unsigned int>(PCABO_CNT_DONE + 1);
  return (c.off + 4095) & ~(size_t)4095;
}
static size_t carve_host(pcabo_ctx* ctx, char* base) {
  Carver c{base};
  const size_t n = ctx->max_n, d = ctx->max_d, Q = ctx->max_q;
  ctx->hm = c.take<HostMirror>(1);
  // (hXq: at least one QueryArgs block - small batches are handed to the kernel by value from here)
  ctx->hXq = c.take<double>(std::max<size_t>(Q * d, PCABO_QA_MAX) + 8);
  ctx->hVal = c.take<double>(Q);
  ctx->hGrad = c.take<double>(Q * d);
  ctx->hSmall = c.take<double>(d * d + 8 * d + 64);
  ctx->hIn = c.take<double>(n * (2 * d + 2));
  ctx->hBestF = c.take<double>(2);
  return (c.off + 4095) & ~(size_t)4095;
}

// Fill in a context over (optionally) somebody else's memory and stream: `dreg` / `hreg` / `stream` non-null = a member
// of a batch (no resident-mode buffers, nothing owned).
static int ctx_setup(pcabo_ctx* ctx, int device, int max_n, int max_d, int max_q, char* dreg, char* hreg,
                     hipStream_t stream) {
  ctx->device = device;
  ctx->max_n = max_n; ctx->max_d = max_d; ctx->max_q = max_q;
  ctx->NPcap = round_up(max_n, PCABO_BS);
  ctx->ld = ctx->NPcap;
  ctx->DPcap = round_up(max_d, 16);
  ctx->KPcap = round_up(max_d, 4);
  ctx->Scap = ctx->NPcap / PCABO_SLAB;
  ctx->in_batch = dreg != nullptr;
  HIPCHK(hipSetDevice(device));
  if (stream) ctx->stream = stream;
  else HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  presence_register(ctx->device); ctx->registered = true;
  HIPCHK(hipEventCreateWithFlags(&ctx->evBounds, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ctx->evPca, hipEventDisableTiming));
  ctx->region_bytes = carve_device(ctx, nullptr);
  ctx->hregion_bytes = carve_host(ctx, nullptr);
  if (dreg) {
    ctx->dRegion = dreg; ctx->hRegion = hreg;        // zeroed by the batch
  } else {
    HIPCHK(hipMalloc((void**)&ctx->dRegion, ctx->region_bytes));
    HIPCHK(hipHostMalloc((void**)&ctx->hRegion, ctx->hregion_bytes, hipHostMallocDefault));
    memset(ctx->hRegion, 0, ctx->hregion_bytes);
  }
  carve_device(ctx, ctx->dRegion);
  carve_host(ctx, ctx->hRegion);
  if (!dreg) {
    // zero: the per-query tickets, the padded tail of y_s, and R (its blocks above the diagonal stay zero for good)
    HIPCHK(hipMemsetAsync(ctx->dRegion, 0, ctx->region_bytes, ctx->stream));
    if (hipExtMallocWithFlags((void**)&ctx->dMail, PCABO_MAIL_PAIRS * sizeof(MailPair), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      HIPCHK(dalloc(&ctx->dMail, PCABO_MAIL_PAIRS));
    }
    HIPCHK(hipMemsetAsync(ctx->dMail, 0, PCABO_MAIL_PAIRS * sizeof(MailPair), ctx->stream));
    const size_t np = (size_t)PCABO_INLAUNCH_MAXQ * 32 * (2 + 2 * PCABO_MAXD);
    HIPCHK(dalloc(&ctx->dPairs, np));
    HIPCHK(hipMemsetAsync(ctx->dPairs, 0, np * sizeof(MailPair), ctx->stream));   // on OUR stream: a memset on the
    // null stream is not ordered with the non-blocking streams the resident kernels run on
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->mail_bar = mail_bar_usable(ctx);
  } else {
    ctx->opt_resident = false;           // the resident kernel wants most of the chip for one run: not inside a batch
  }
  // Sequence numbers double as mailbox tags.  Memory handed out by the allocator may come from a context that was
  // destroyed earlier in this process, so every context numbers from its own base: a stale tag can never match.
  {
    static std::atomic<unsigned long long> ctx_counter{0};
    ctx->seq = ((unsigned long long)(getpid() & 0xffff) << 44) + ((ctx_counter.fetch_add(1) + 1) << 30);
  }
  return PCABO_OK;
}

static void ctx_teardown(pcabo_ctx* ctx) {
  (void)hipSetDevice(ctx->device);        // tear-down: nothing useful to do with an error from here on
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->pairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  if (!ctx->in_batch) {
    if (ctx->dRegion) (void)hipFree(ctx->dRegion);
    if (ctx->hRegion) (void)hipHostFree(ctx->hRegion);
    if (ctx->dMail) (void)hipFree(ctx->dMail);
    if (ctx->dPairs) (void)hipFree(ctx->dPairs);
  }
  if (ctx->evBounds) (void)hipEventDestroy(ctx->evBounds);
  if (ctx->evPca) (void)hipEventDestroy(ctx->evPca);
  if (ctx->stream && !ctx->in_batch) (void)hipStreamDestroy(ctx->stream);
  ctx->helper.shutdown();
  for (auto& h : ctx->more_helpers) h->shutdown();
  ctx->more_helpers.clear();
  if (ctx->registered) presence_unregister(ctx->device);
}

static bool ctx_sizes_ok(int max_n, int max_d, int max_q) {
  return max_n >= 2 && max_d >= 1 && max_d <= PCABO_MAXD && max_q >= 1 && max_q < PCABO_CNT_DONE;
}

extern "C" {

int pcabo_ctx_create(int device, int max_n, int max_d, int max_q, pcabo_ctx** out) {
  if (!out) return PCABO_ERR_ARG;
  *out = nullptr;
  if (!ctx_sizes_ok(max_n, max_d, max_q)) return PCABO_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PCABO_ERR_HIP;
  pcabo_ctx* ctx = new (std::nothrow) pcabo_ctx();
  if (!ctx) return PCABO_ERR_HIP;
  *out = ctx;                       // handed back even on failure so the caller can read the message
  return ctx_setup(ctx, device, max_n, max_d, max_q, nullptr, nullptr, nullptr);
}

int pcabo_ctx_destroy(pcabo_ctx* ctx) {
  if (!ctx) return PCABO_ERR_ARG;
  if (ctx->in_batch) return set_err(ctx, PCABO_ERR_ARG, "pcabo_ctx_destroy: the context belongs to a batch (pcabo_batch_destroy frees it)%s", "");
  ctx_teardown(ctx);
  delete ctx;
  return PCABO_OK;
}

int pcabo_set_pointer_mode(pcabo_ctx* ctx, int mode) {
  if (!ctx || (mode != PCABO_PTR_HOST && mode != PCABO_PTR_DEVICE)) return PCABO_ERR_ARG;
  ctx->ptr_mode = mode;
  return PCABO_OK;
}

int pcabo_set_option(pcabo_ctx* ctx, int option, int value) {
  if (!ctx) return PCABO_ERR_ARG;
  switch (option) {
    case PCABO_OPT_RESIDENT: ctx->opt_resident = value != 0 && !ctx->in_batch; return PCABO_OK;   // (never inside a batch)
    case PCABO_OPT_BESTF_F32: ctx->bestf_f32 = value != 0; return PCABO_OK;
    case PCABO_OPT_GROUP_ACQ: ctx->opt_group_acq = value != 0; return PCABO_OK;
    default: return set_err(ctx, PCABO_ERR_ARG, "pcabo_set_option: unknown option %s%d", "", option);
  }
}

int pcabo_last_error(pcabo_ctx* ctx, char* buf, int buflen) {
  if (!ctx || !buf || buflen <= 0) return PCABO_ERR_ARG;
  snprintf(buf, buflen, "%s", ctx->err);
  return PCABO_OK;
}

// bulk input: copy host data to the staging buffer, or use the caller's device pointer as is
#define STAGE_IN(dst, src, count, T)                                                                        \
  do {                                                                                                      \
    if (ctx->ptr_mode == PCABO_PTR_HOST) {                                                                  \
      HIPCHK(hipMemcpyAsync((void*)(dst), (const void*)(src), (size_t)(count) * sizeof(T), hipMemcpyHostToDevice, ctx->stream)); \
    } else {                                                                                                \
      HIPCHK(hipMemcpyAsync((void*)(dst), (const void*)(src), (size_t)(count) * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream)); \
    }                                                                                                       \
  } while (0)
#define STAGE_OUT(dst, src, count, T)                                                                       \
  do {                                                                                                      \
    HIPCHK(hipMemcpyAsync((void*)(dst), (const void*)(src), (size_t)(count) * sizeof(T),                    \
                          ctx->ptr_mode == PCABO_PTR_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream)); \
  } while (0)
#define HOST_OUT(dst, src, count, T) \
  HIPCHK(hipMemcpyAsync((void*)(dst), (const void*)(src), (size_t)(count) * sizeof(T), hipMemcpyDeviceToHost, ctx->stream))

// ---- rows A-C (+ D-H behind them) ---------------------------------------------------------------------------------
// Inputs of one BO iteration.  Host-pointer mode: X, the noise block, the ranks (or f) and y are packed into one pinned
// buffer and cross in ONE copy (four pageable copies cost 60-170 us of host time per iteration); device-pointer mode:
// device-to-device copies into the context's own buffers as before.
struct WpcaInputs { const double *X, *noise, *y; const long long* ranks; };
static int stage_inputs(pcabo_ctx* ctx, const double* X, const double* f, const int64_t* ranks, const double* noise,
                        const double* y, int n, int d, int maximize, WpcaInputs* in) {
  hipStream_t s = ctx->stream;
  const size_t nd = (size_t)n * d;
  if (ctx->ptr_mode == PCABO_PTR_HOST) {
    double* h = ctx->hIn;
    size_t off = 0;
    memcpy(h, X, nd * sizeof(double));                               in->X = ctx->dIn; off += nd;
    if (noise) { memcpy(h + off, noise, nd * sizeof(double));        in->noise = ctx->dIn + off; off += nd; }
    const size_t off_r = off;
    memcpy(h + off, ranks ? (const void*)ranks : (const void*)f, (size_t)n * 8); off += n;
    if (y) { memcpy(h + off, y, (size_t)n * sizeof(double));         in->y = ctx->dIn + off; off += n; }
    // (segments are only 8-byte aligned: none of the consumers uses wider loads on them)
    HIPCHK(hipMemcpyAsync(ctx->dIn, h, off * sizeof(double), hipMemcpyHostToDevice, s));
    if (ranks) in->ranks = (const long long*)(ctx->dIn + off_r);
    else { launch_rank(s, ctx->dIn + off_r, n, maximize, ctx->dRanks); in->ranks = ctx->dRanks; }
    return PCABO_OK;
  }
  STAGE_IN(ctx->dX, X, nd, double);                                  in->X = ctx->dX;
  if (ranks) { STAGE_IN(ctx->dRanks, ranks, n, long long); }
  else { STAGE_IN(ctx->dF, f, n, double); launch_rank(s, ctx->dF, n, maximize, ctx->dRanks); }
  in->ranks = ctx->dRanks;
  if (noise) { STAGE_IN(ctx->dNoise, noise, nd, double);             in->noise = ctx->dNoise; }
  if (y) { STAGE_IN(ctx->dY, y, n, double);                          in->y = ctx->dY; }
  return PCABO_OK;
}

// wPCA launches, then ONE copy of [data_mean | pca_mean | evr | comps] to pinned memory and the event that says it
// arrived.  k arrives through the HostMirror (written by the selection step at the end of k_jacobi).
static int enqueue_wpca(pcabo_ctx* ctx, const WpcaInputs& in, int n, int d, double var_threshold, int n_components) {
  hipStream_t s = ctx->stream;
  const int DP = round_up(d, 16);
  {
    ProfScope ps(ctx, 0, 8.0 * n * d * 2 + 8.0 * n * d, 2.0 * n * d * d + 9.0 * d * d * d + 2.0 * n * d * d);
    launch_wpca_prep(s, in.X, in.ranks, in.noise, n, d, DP, ctx->dWeights, ctx->dDataMean, ctx->dPcaMean, ctx->dWc);
    launch_cov(s, ctx->dWc, n, DP, ctx->dC);
    const double* v0 = (ctx->vprev_d == d) ? ctx->dGbuf[ctx->gcur] : nullptr;
    ctx->gcur ^= 1;
    ctx->dG = ctx->dGbuf[ctx->gcur];
    launch_jacobi(s, ctx->dC, d, DP, v0, ctx->dG, ctx->dLam, ctx->dSweeps, n, var_threshold, n_components, ctx->dComps,
                  ctx->dEvr, ctx->dK, ctx->hm);
    ctx->vprev_d = d;
    launch_project(s, in.X, ctx->dDataMean, ctx->dPcaMean, ctx->dComps, ctx->dK, n, d, ctx->dZ);
  }
  const int rcount = n < d ? n : d;
  HIPCHK(hipMemcpyAsync(ctx->hSmall, ctx->dPcaOut, ((size_t)3 * ctx->max_d + (size_t)rcount * d) * sizeof(double),
                        hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(ctx->evPca, s));
  return PCABO_OK;
}

// wait for the wPCA results only (whatever was enqueued behind them keeps running) and hand them out
static int collect_wpca(pcabo_ctx* ctx, int n, int d, double* data_mean, double* pca_mean, double* comps, double* evr,
                        int* k) {
  HIPCHK(wait_event(ctx->evPca));
  HIPCHK(hipGetLastError());
  const int rcount = n < d ? n : d;
  const double* h = ctx->hSmall;
  const size_t D = ctx->max_d;
  if (data_mean) memcpy(data_mean, h, (size_t)d * sizeof(double));
  if (pca_mean) memcpy(pca_mean, h + D, (size_t)d * sizeof(double));
  if (evr) memcpy(evr, h + 2 * D, (size_t)rcount * sizeof(double));
  if (comps) memcpy(comps, h + 3 * D, (size_t)rcount * d * sizeof(double));
  ctx->d = d; ctx->k = ctx->hm->k;
  ctx->have_wpca = true;
  if (k) *k = ctx->k;
  return PCABO_OK;
}

int pcabo_wpca(pcabo_ctx* ctx, const double* X, const double* f, const int64_t* ranks, int n, int d, int maximize,
               double var_threshold, int n_components, const double* noise, double* data_mean, double* pca_mean,
               double* comps, double* evr, int* k, double* Z) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!X || (!f && !ranks) || n < 2 || n > ctx->max_n || d < 1 || d > ctx->max_d)
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_wpca: bad argument or size beyond context capacity%s", "");
  if (ctx->gp_pending) return set_err(ctx, PCABO_ERR_ARG, "pcabo_wpca: a conditioning is in flight, call pcabo_gp_condition_end first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  WpcaInputs in{nullptr, nullptr, nullptr, nullptr};
  int rc = stage_inputs(ctx, X, f, ranks, noise, nullptr, n, d, maximize, &in);
  if (rc != PCABO_OK) return rc;
  rc = enqueue_wpca(ctx, in, n, d, var_threshold, n_components);
  if (rc != PCABO_OK) return rc;
  ctx->have_gp = false;
  rc = collect_wpca(ctx, n, d, data_mean, pca_mean, comps, evr, k);
  if (rc != PCABO_OK) return rc;
  ctx->n = n;
  if (Z) {
    STAGE_OUT(Z, ctx->dZ, (size_t)n * ctx->k, double);
    HIPCHK(hipStreamSynchronize(s));
  }
  return PCABO_OK;
}

static int launch_factorisation(pcabo_ctx* ctx, double jitter) {
  hipStream_t s = ctx->stream;
  if (jitter > 0.0) {      // a retry: K is built again (same kernel, same bits, flag cleared), then the jitter goes on its diagonal
    launch_gram(s, ctx->dAT, ctx->dNrm, ctx->n, ctx->NP, round_up(ctx->k, 4), ctx->ld, ctx->noise, ctx->kernel, nullptr, nullptr,
                ctx->dL, ctx->dInfo);
    launch_add_jitter(s, ctx->dL, ctx->n, ctx->ld, jitter);
  }
  { ProfScope ps(ctx, 2, 16.0 * ctx->n * ctx->n, (double)ctx->n * ctx->n * ctx->n / 3.0); launch_cholesky(s, ctx->dL, ctx->NP, ctx->ld, ctx->dInfo, ctx->dDiag); }
  {
    ProfScope ps(ctx, 3, 16.0 * ctx->n * ctx->n, (double)ctx->n * ctx->n * ctx->n / 3.0 + 2.0 * ctx->n * ctx->n);
    launch_trinv(s, ctx->dL, ctx->NP, ctx->ld, ctx->dR);
    launch_alpha(s, ctx->dR, ctx->dYs, ctx->n, ctx->NP, ctx->ld, ctx->dTmp, ctx->dAlpha);
  }
  HIPCHK(hipMemcpyAsync((void*)&ctx->hm->chol_info, ctx->dInfo, sizeof(int), hipMemcpyDeviceToHost, s));
  return PCABO_OK;
}

// Rows D-H on the stream, inputs on the device.  k < 0: the reduced dimension is still on its way (the launches sit
// right behind the wPCA) - the three kernels that need it read it from ctx->dK.
static int enqueue_condition(pcabo_ctx* ctx, const double* y_dev, int n, int k, const double* unb, double lengthscale,
                             double noise, int kernel) {
  hipStream_t s = ctx->stream;
  const int* k_dev = k < 0 ? ctx->dK : nullptr;
  const int kk = k < 0 ? ctx->max_d : k;              // work model only
  ctx->n = n;
  ctx->NP = round_up(n, PCABO_BS);
  if (k >= 0) { ctx->k = k; ctx->KP = round_up(k, 4); }
  ctx->lengthscale = lengthscale; ctx->noise = noise; ctx->kernel = kernel;
  ctx->have_gp = false;
  ctx->gp_pending = true;
  // the per-query tickets count modulo the number of slab groups: they only need a reset when that number changes
  // (every 64th iteration) or after a launch that did not complete
  if (acq_slabs(ctx->NP) != ctx->cnt_S || ctx->cnt_dirty) {
    HIPCHK(hipMemsetAsync(ctx->dCounters, 0, (PCABO_CNT_DONE + 1) * sizeof(unsigned int), s));
    ctx->cnt_S = acq_slabs(ctx->NP); ctx->cnt_dirty = false;
  }
  {
    ProfScope ps(ctx, 1, 8.0 * n * kk + 4.0 * n * (n + 1.0), 2.0 * n * n * kk + 12.0 * n * n);
    launch_zstats(s, ctx->dZ, y_dev, n, k, unb, ctx->dBounds4, ctx->dZnMean, ctx->dYstats, ctx->dYs, ctx->hm, k_dev);
    HIPCHK(hipEventRecord(ctx->evBounds, s));
    launch_znorm(s, ctx->dZ, n, k, ctx->NP, ctx->KP, ctx->ld, ctx->dBounds4, ctx->dZnMean, 1.0 / lengthscale, ctx->dZnT,
                 ctx->dAT, ctx->dNrm, k_dev);
    launch_gram(s, ctx->dAT, ctx->dNrm, n, ctx->NP, ctx->KP, ctx->ld, noise, kernel, nullptr, k_dev, ctx->dL, ctx->dInfo);
  }
  return launch_factorisation(ctx, 0.0);       // asynchronous: pcabo_gp_condition_end() waits and checks
}

static bool gp_args_ok(const pcabo_ctx* ctx, int n, double lengthscale, double noise, int kernel) {
  return n >= 2 && n <= ctx->max_n && lengthscale > 0.0 && noise >= 0.0 &&
         (kernel == PCABO_KERNEL_MATERN52 || kernel == PCABO_KERNEL_RBF);
}

int pcabo_gp_condition_begin(pcabo_ctx* ctx, const double* Z, const double* y, int n, int k, const double* norm_bounds,
                             double lengthscale, double noise, int kernel) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!y || k < 1 || k > ctx->max_d || !gp_args_ok(ctx, n, lengthscale, noise, kernel))
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_gp_condition: bad argument or size beyond context capacity%s", "");
  if (!Z && (!ctx->have_wpca || ctx->n != n || ctx->k != k))
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_gp_condition: Z == NULL needs a matching pcabo_wpca call first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (Z) STAGE_IN(ctx->dZ, Z, (size_t)n * k, double);
  STAGE_IN(ctx->dY, y, n, double);
  const double* unb = nullptr;
  if (norm_bounds) {
    HIPCHK(hipMemcpyAsync(ctx->dUserNB, norm_bounds, (size_t)2 * k * sizeof(double), hipMemcpyHostToDevice, s));
    unb = ctx->dUserNB;
  }
  return enqueue_condition(ctx, ctx->dY, n, k, unb, lengthscale, noise, kernel);
}

// pcabo_wpca followed by pcabo_gp_condition_begin(Z = NULL, norm_bounds = NULL) as ONE enqueue: the conditioning
// launches go onto the stream right behind the projection, before the host has seen k, so the device does not idle
// while the wPCA results travel and the caller prepares the second call.  Returns as soon as the wPCA results are on
// the host; finish with pcabo_gp_condition_end().
int pcabo_wpca_gp_condition_begin(pcabo_ctx* ctx, const double* X, const double* f, const int64_t* ranks, int n, int d,
                                  int maximize, double var_threshold, int n_components, const double* noise,
                                  const double* y, double lengthscale, double gp_noise, int kernel, double* data_mean,
                                  double* pca_mean, double* comps, double* evr, int* k) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!X || !y || (!f && !ranks) || d < 1 || d > ctx->max_d || !gp_args_ok(ctx, n, lengthscale, gp_noise, kernel))
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_wpca_gp_condition_begin: bad argument or size beyond context capacity%s", "");
  if (ctx->gp_pending) return set_err(ctx, PCABO_ERR_ARG, "pcabo_wpca_gp_condition_begin: a conditioning is in flight, call pcabo_gp_condition_end first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  WpcaInputs in{nullptr, nullptr, nullptr, nullptr};
  int rc = stage_inputs(ctx, X, f, ranks, noise, y, n, d, maximize, &in);
  if (rc != PCABO_OK) return rc;
  rc = enqueue_wpca(ctx, in, n, d, var_threshold, n_components);
  if (rc != PCABO_OK) return rc;
  ctx->wpca_n = n; ctx->wpca_d = d;
  if (ctx->prof) {                       // profiled runs keep the two phases apart (the work model of group 1 needs k)
    rc = collect_wpca(ctx, n, d, data_mean, pca_mean, comps, evr, k);
    if (rc != PCABO_OK) return rc;
    return enqueue_condition(ctx, in.y, n, ctx->k, nullptr, lengthscale, gp_noise, kernel);
  }
  rc = enqueue_condition(ctx, in.y, n, -1, nullptr, lengthscale, gp_noise, kernel);
  if (rc != PCABO_OK) return rc;
  ctx->wpca_n = n; ctx->wpca_d = d; ctx->wpca_uncollected = true;
  if (!data_mean && !pca_mean && !comps && !evr && !k) return PCABO_OK;      // enqueue only: pcabo_wpca_results() waits
  return pcabo_wpca_results(ctx, data_mean, pca_mean, comps, evr, k);
}

// Second half of pcabo_wpca_gp_condition_begin when that was called without output pointers: waits for the wPCA
// results (the conditioning keeps running) and hands them out.  Lets the host do work that only needs a GUESS of k (the
// scrambled Sobol engine of the initial-condition draw, with last iteration's k) while the eigen-decomposition runs.
int pcabo_wpca_results(pcabo_ctx* ctx, double* data_mean, double* pca_mean, double* comps, double* evr, int* k) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!ctx->wpca_uncollected && !ctx->have_wpca)
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_wpca_results: no weighted PCA enqueued%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = collect_wpca(ctx, ctx->wpca_n, ctx->wpca_d, data_mean, pca_mean, comps, evr, k);
  if (rc != PCABO_OK) return rc;
  ctx->wpca_uncollected = false;
  ctx->KP = round_up(ctx->k, 4);
  return PCABO_OK;
}

int pcabo_gp_condition_end(pcabo_ctx* ctx) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!ctx->gp_pending) return set_err(ctx, PCABO_ERR_ARG, "pcabo_gp_condition_end: no conditioning in flight%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->wpca_uncollected) { int rc0 = pcabo_wpca_results(ctx, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc0 != PCABO_OK) return rc0; }
  ctx->gp_pending = false;
  double jitter = 0.0;
  for (int attempt = 0; attempt < 4; ++attempt) {       // psd_safe_cholesky: 0, 1e-8, 1e-7, 1e-6
    HIPCHK(wait_stream(ctx->stream));
    HIPCHK(hipGetLastError());
    if (ctx->hm->chol_info == 0) { ctx->have_gp = true; return PCABO_OK; }
    if (attempt == 3) break;
    jitter = (attempt == 0) ? 1e-8 : jitter * 10.0;
    int rc = launch_factorisation(ctx, jitter);
    if (rc != PCABO_OK) return rc;
  }
  return set_err(ctx, PCABO_ERR_NOT_PD, "K + s2 I not positive definite after jitter retries (pivot %s%d)", "",
                 ctx->hm->chol_info);
}

int pcabo_gp_condition(pcabo_ctx* ctx, const double* Z, const double* y, int n, int k, const double* norm_bounds,
                       double lengthscale, double noise, int kernel) {
  int rc = pcabo_gp_condition_begin(ctx, Z, y, n, k, norm_bounds, lengthscale, noise, kernel);
  if (rc != PCABO_OK) return rc;
  return pcabo_gp_condition_end(ctx);
}

int pcabo_acq_bounds(pcabo_ctx* ctx, double* bounds) {
  if (!ctx || !bounds) return PCABO_ERR_ARG;
  if (ctx->wpca_uncollected) { int rc0 = pcabo_wpca_results(ctx, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc0 != PCABO_OK) return rc0; }
  if (ctx->gp_pending) {                   // conditioning in flight: the box only needs k_zstats, wait for that alone
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(wait_event(ctx->evBounds));
  } else if (!ctx->have_gp) {
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_acq_bounds: call pcabo_gp_condition first%s", "");
  }
  for (int c = 0; c < ctx->k; ++c) { bounds[c] = ctx->hm->acq_lo[c]; bounds[ctx->k + c] = ctx->hm->acq_hi[c]; }
  return PCABO_OK;
}

static AcqParams make_params(pcabo_ctx* ctx, double best_f, int maximize, int acq, int want_grad) {
  AcqParams p;
  // torch.as_tensor(python float) gives a float32 tensor; a numpy float64 scalar keeps its 64 bits (PCABO_OPT_BESTF_F32)
  p.best_f = ctx->bestf_f32 ? (double)(float)best_f : best_f;
  p.y_mean = 0.0; p.y_std = 1.0;
  p.inv_ls = 1.0 / ctx->lengthscale;
  p.maximize = maximize ? 1 : 0;
  p.acq = acq;
  p.kernel = ctx->kernel;
  p.want_grad = want_grad;
  return p;
}

// One evaluation of the acquisition at the nq points staged in ctx->hXq (pinned): a single fused launch
// whose results land in ctx->hVal / ctx->hGrad; the host spins on the sequence flag.
static int eval_staged(pcabo_ctx* ctx, int nq, AcqParams& p, bool allow_gemm = true) {
  hipStream_t s = ctx->stream;
  const int k = ctx->k;
  const unsigned long long seq = ++ctx->seq;
  const QueryArgs* qa = nullptr;
  const double* xdev = nullptr;
  if ((size_t)nq * k <= PCABO_QA_MAX) {
    qa = reinterpret_cast<const QueryArgs*>(ctx->hXq);        // copied by value into the kernel arguments
  } else {
    HIPCHK(hipMemcpyAsync(ctx->dXq, ctx->hXq, (size_t)nq * k * sizeof(double), hipMemcpyHostToDevice, s));
    xdev = ctx->dXq;
  }
  const bool small = nq <= PCABO_INLAUNCH_MAXQ;   // in-launch combine + host flag; larger batches: two launches + copy
  if (!small && !p.want_grad && allow_gemm && score_gemm_possible(nq)) {
    // value-only scoring of a large batch (the raw samples): V = R KS^T on MFMA
    if (!xdev) {
      HIPCHK(hipMemcpyAsync(ctx->dXq, ctx->hXq, (size_t)nq * k * sizeof(double), hipMemcpyHostToDevice, s));
      xdev = ctx->dXq;
    }
    ProfScope ps(ctx, 5, acq_bytes(ctx->n, k, nq, 0), acq_flops(ctx->n, k, nq, 0));
    launch_score(s, xdev, nq, ctx->n, k, ctx->NP, ctx->ld, ctx->dZnT, ctx->dR, ctx->dAlpha, ctx->dBounds4, ctx->dYstats, p,
                 ctx->dKS, ctx->dPartial, ctx->dVal);
  } else {
    ProfScope ps(ctx, small ? 4 : 5, acq_bytes(ctx->n, k, nq, p.want_grad), acq_flops(ctx->n, k, nq, p.want_grad));
    launch_acq(s, qa, xdev, nq, ctx->n, k, ctx->NP, ctx->ld, ctx->dZnT, ctx->dR, ctx->dAlpha, ctx->dBounds4, ctx->dYstats,
               p, ctx->dPartial, ctx->dCounters, ctx->dVal, ctx->dGrad, small ? ctx->hVal : nullptr,
               small ? ctx->hGrad : nullptr, small ? ctx->hm : nullptr, seq);
  }
  if (!small) {
    HIPCHK(hipMemcpyAsync(ctx->hVal, ctx->dVal, (size_t)nq * sizeof(double), hipMemcpyDeviceToHost, s));
    if (p.want_grad) HIPCHK(hipMemcpyAsync(ctx->hGrad, ctx->dGrad, (size_t)nq * k * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(wait_stream(s));
    HIPCHK(hipGetLastError());
    return PCABO_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long spins = 0;
  for (int qi = 0; qi < nq; ++qi) {
    while (__atomic_load_n(&ctx->hm->qflag[qi], __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xFFFF) == 0) {
        if (hipStreamQuery(s) == hipSuccess) {              // kernel done: the flag must be there
          if (__atomic_load_n(&ctx->hm->qflag[qi], __ATOMIC_ACQUIRE) == seq) break;
          HIPCHK(hipGetLastError());
          return set_err(ctx, PCABO_ERR_TIMEOUT, "acquisition kernel finished without publishing its results%s", "");
        }
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > 20.0) return set_err(ctx, PCABO_ERR_TIMEOUT, "device did not publish acquisition results%s", "");
      }
    }
  }
  return PCABO_OK;
}

// The same for restart groups through the throughput kernel (PCABO_OPT_GROUP_ACQ): group g = the gn[g] <= 5 points staged
// at query slots g0[g].. of ctx->hXq, which the kernel reads in place (pinned memory); one work-group per (group, 64-row slab).
static int eval_staged_groups(pcabo_ctx* ctx, const int* g0, const int* gn, int ng, AcqParams& p) {
  hipStream_t s = ctx->stream;
  const unsigned long long seq = ++ctx->seq;
  QueryArgs tab;
  unsigned* ent = reinterpret_cast<unsigned*>(tab.x);
  for (int g = 0; g < ng; ++g) ent[g] = ((unsigned)g0[g] << 8) | (unsigned)gn[g];
  {
    int nq = 0;
    for (int g = 0; g < ng; ++g) nq += gn[g];
    ProfScope ps(ctx, 4, acq_bytes(ctx->n, ctx->k, nq, p.want_grad), acq_flops(ctx->n, ctx->k, nq, p.want_grad));
    if (launch_acq_group(s, &tab, ng, ctx->hXq, ctx->n, ctx->k, ctx->NP, ctx->ld, ctx->dZnT, ctx->dR, ctx->dAlpha, ctx->dBounds4,
                         ctx->dYstats, p, ctx->dPartial, ctx->dCounters + PCABO_GROUP_CNT_OFFSET, ctx->dVal, ctx->dGrad, ctx->hVal,
                         ctx->hGrad, ctx->hm, seq, AcqBatch()) != 0)
      return set_err(ctx, PCABO_ERR_HIP, "the restart-group acquisition kernel could not be launched%s", "");
  }
  HIPCHK(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long spins = 0;
  for (int g = 0; g < ng; ++g)
    for (int j = 0; j < gn[g]; ++j) {
      while (__atomic_load_n(&ctx->hm->qflag[g0[g] + j], __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xFFFF) == 0) {
          if (hipStreamQuery(s) == hipSuccess) {
            if (__atomic_load_n(&ctx->hm->qflag[g0[g] + j], __ATOMIC_ACQUIRE) == seq) break;
            HIPCHK(hipGetLastError());
            return set_err(ctx, PCABO_ERR_TIMEOUT, "acquisition kernel finished without publishing its results%s", "");
          }
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0)
            return set_err(ctx, PCABO_ERR_TIMEOUT, "device did not publish acquisition results%s", "");
        }
      }
    }
  return PCABO_OK;
}
static bool group_mode(const pcabo_ctx* ctx, int q, int want_grad) {
  return ctx->opt_group_acq && want_grad && q <= PCABO_INLAUNCH_MAXQ && acq_group_possible(ctx->NP, ctx->k);
}

int pcabo_acq_eval(pcabo_ctx* ctx, const double* Xq, int q, double best_f, int maximize, int acq, double* val,
                   double* grad) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!Xq || !val || q < 1 || q > ctx->max_q || (acq != PCABO_ACQ_LOG_EI && acq != PCABO_ACQ_PI))
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_acq_eval: bad argument or q beyond context capacity%s", "");
  if (!ctx->have_gp) return set_err(ctx, PCABO_ERR_ARG, "pcabo_acq_eval: call pcabo_gp_condition first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int k = ctx->k;
  const size_t nx = (size_t)q * k;
  if (ctx->ptr_mode == PCABO_PTR_HOST) memcpy(ctx->hXq, Xq, nx * sizeof(double));
  else {
    HIPCHK(hipMemcpyAsync(ctx->hXq, Xq, nx * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  AcqParams p = make_params(ctx, best_f, maximize, acq, grad ? 1 : 0);
  int rc;
  if (group_mode(ctx, q, grad != nullptr) && ctx->ptr_mode == PCABO_PTR_HOST) {
    int g0[(PCABO_INLAUNCH_MAXQ + PCABO_GROUP_Q - 1) / PCABO_GROUP_Q], gn[(PCABO_INLAUNCH_MAXQ + PCABO_GROUP_Q - 1) / PCABO_GROUP_Q], ng = 0;
    for (int a = 0; a < q; a += PCABO_GROUP_Q) { g0[ng] = a; gn[ng] = std::min(PCABO_GROUP_Q, q - a); ++ng; }
    rc = eval_staged_groups(ctx, g0, gn, ng, p);
  } else {
    rc = eval_staged(ctx, q, p);
  }
  if (rc != PCABO_OK) return rc;
  if (ctx->ptr_mode == PCABO_PTR_HOST) {
    memcpy(val, ctx->hVal, (size_t)q * sizeof(double));
    if (grad) memcpy(grad, ctx->hGrad, nx * sizeof(double));
  } else {
    HIPCHK(hipMemcpyAsync(val, ctx->dVal, (size_t)q * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (grad) HIPCHK(hipMemcpyAsync(grad, ctx->dGrad, nx * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  return PCABO_OK;
}

// pcabo_gp_condition_end + pcabo_acq_eval (values only) with the evaluation ENQUEUED BEHIND the conditioning: the raw
// samples of the initial-condition draw are known before the factorisation has finished, so their copy and launches
// need not wait for the host to see it end.  If the factorisation turns out to have needed jitter, the values are
// computed again after the retries.
int pcabo_gp_condition_end_eval(pcabo_ctx* ctx, const double* Xq, int q, double best_f, int maximize, int acq,
                                double* val) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!Xq || !val || q < 1 || q > ctx->max_q || (acq != PCABO_ACQ_LOG_EI && acq != PCABO_ACQ_PI))
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_gp_condition_end_eval: bad argument or q beyond context capacity%s", "");
  if (!ctx->gp_pending) return set_err(ctx, PCABO_ERR_ARG, "pcabo_gp_condition_end_eval: no conditioning in flight%s", "");
  if (ctx->wpca_uncollected) { int rc0 = pcabo_wpca_results(ctx, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc0 != PCABO_OK) return rc0; }
  if (q <= PCABO_INLAUNCH_MAXQ || ctx->ptr_mode != PCABO_PTR_HOST) {      // nothing to gain: the two calls in a row
    int rc = pcabo_gp_condition_end(ctx);
    if (rc != PCABO_OK) return rc;
    return pcabo_acq_eval(ctx, Xq, q, best_f, maximize, acq, val, nullptr);
  }
  HIPCHK(hipSetDevice(ctx->device));
  memcpy(ctx->hXq, Xq, (size_t)q * ctx->k * sizeof(double));
  AcqParams p = make_params(ctx, best_f, maximize, acq, 0);
  int rc = eval_staged(ctx, q, p);                  // ends with a stream synchronisation: the conditioning is over too
  if (rc != PCABO_OK) { ctx->gp_pending = false; return rc; }
  if (ctx->hm->chol_info != 0) {                    // rare: jitter retries, then the evaluation again
    rc = pcabo_gp_condition_end(ctx);
    if (rc != PCABO_OK) return rc;
    rc = eval_staged(ctx, q, p);
    if (rc != PCABO_OK) return rc;
  } else {
    ctx->gp_pending = false;
    ctx->have_gp = true;
  }
  memcpy(val, ctx->hVal, (size_t)q * sizeof(double));
  return PCABO_OK;
}

int pcabo_logei(pcabo_ctx* ctx, const double* Xq, int q, double best_f, int maximize, double* val, double* grad) {
  return pcabo_acq_eval(ctx, Xq, q, best_f, maximize, PCABO_ACQ_LOG_EI, val, grad);
}

// ---- resident acquisition kernel: host side of the mailbox ---------------------------------------------------------
// One round: all cap*k coordinate pairs get the round's tag (coordinates of queries that are no longer active are
// whatever is left in hXq), then one control pair per query (1 = evaluate, 0 = leave for good).  A pair leaves in one 16-byte store and
// its tag is mixed with the value's bits (mail_mix), so a reader never takes a value that does not belong to its tag.
static void server_post(pcabo_ctx* ctx, int cap, int nq, int k, unsigned long long tag) {
  MailPair* m = ctx->dMail;                                    // device memory, through the PCIe BAR
  const int np = cap * k;
  for (int i = 0; i < np; ++i) put_mail_pair(m + 1 + i, ctx->hXq[i], tag);
  for (int q = 0; q < cap; ++q) put_mail_pair(m + 1 + np + q, q < nq ? 1.0 : 0.0, tag);     // control pairs: evaluate / leave
  _mm_sfence();                                                // flush the write-combining buffers now
}
static int server_wait(pcabo_ctx* ctx, int nq, unsigned long long tag) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long spins = 0;
  for (int qi = 0; qi < nq; ++qi) {
    while (__atomic_load_n(&ctx->hm->qflag[qi], __ATOMIC_ACQUIRE) != tag) {
      if ((++spins & 0xFFFF) == 0) {
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > 3.0) return set_err(ctx, PCABO_ERR_TIMEOUT, "resident acquisition kernel did not answer%s", "");
      }
    }
  }
  return PCABO_OK;
}

// ---- multi-start L-BFGS-B over the device acquisition ------------------------------------------
// All restart groups advance in lock-step: per round every still-active group asks for one joint
// value+gradient evaluation; the points of all groups go to the device in ONE launch whose results
// land in pinned host memory (eval_staged above).
int pcabo_optimize_acqf(pcabo_ctx* ctx, const double* ics, int num_restarts, int batch_limit, const double* bounds,
                        int maxiter, double best_f, int maximize, int acq, double* cand, double* vals, int* info,
                        int* failed) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!ics || !bounds || !cand || !vals || num_restarts < 1 || batch_limit < 1 || num_restarts > ctx->max_q ||
      (acq != PCABO_ACQ_LOG_EI && acq != PCABO_ACQ_PI))
    return set_err(ctx, PCABO_ERR_ARG, "pcabo_optimize_acqf: bad argument%s", "");
  if (!ctx->have_gp) return set_err(ctx, PCABO_ERR_ARG, "pcabo_optimize_acqf: call pcabo_gp_condition first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  const int k = ctx->k;
  const int ngroups = (num_restarts + batch_limit - 1) / batch_limit;
  std::vector<Lbfgsb> opt(ngroups);
  std::vector<int> gstart(ngroups), gsize(ngroups), niter(ngroups, 0), nfev(ngroups, 0);
  std::vector<char> active(ngroups, 1);     // char, not vector<bool>: groups are touched from two threads
  std::vector<std::vector<double>> x(ngroups), g(ngroups), lo(ngroups), hi(ngroups);
  std::vector<double> fval(ngroups, 0.0), fc(ngroups, 0.0);
  std::vector<std::vector<double>> xc(ngroups), gc(ngroups), vc(ngroups);   // vc: per-restart values of the cached evaluation
  std::vector<char> have_cache(ngroups, 0);
  for (int gi = 0; gi < ngroups; ++gi) {
    gstart[gi] = gi * batch_limit;
    gsize[gi] = std::min(batch_limit, num_restarts - gstart[gi]);
    const int nv = gsize[gi] * k;
    x[gi].resize(nv); g[gi].assign(nv, 0.0); lo[gi].resize(nv); hi[gi].resize(nv);
    for (int j = 0; j < gsize[gi]; ++j)
      for (int c = 0; c < k; ++c) {
        double l = bounds[c], h = bounds[k + c];
        double v = ics[(size_t)(gstart[gi] + j) * k + c];
        lo[gi][j * k + c] = l; hi[gi][j * k + c] = h;
        x[gi][j * k + c] = v < l ? l : (v > h ? h : v);          // columnwise_clamp / np.clip
      }
    opt[gi].init(nv, 10, lo[gi].data(), hi[gi].data(), 1e7, 1e-5, 20);
  }
  AcqParams p = make_params(ctx, best_f, maximize, acq, 1);
  const int maxfun = 15000;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  int any_failed = 0;
  // advance one group's state machine until it needs f,g at x[gi] (or stops)
  // scipy wraps the objective in a ScalarFunction that memoises the last evaluated point: when a shrinking
  // line-search step underflows and the trial point repeats, the function is neither called nor counted.
  // Same here (xc/fc/gc = last evaluated point of the group and its value/gradient).
  std::vector<char> pending(ngroups, 0);    // x[gi] waits for its evaluation (left so by the free-running mode below)
  auto advance = [&](int gi) {
    if (!active[gi]) return;
    if (pending[gi]) { pending[gi] = 0; return; }
    while (true) {
      int task = opt[gi].step(x[gi].data(), &fval[gi], g[gi].data());
      if (task == LBFGSB_FG) {
        if (have_cache[gi] && memcmp(x[gi].data(), xc[gi].data(), x[gi].size() * sizeof(double)) == 0) {
          fval[gi] = fc[gi];
          g[gi] = gc[gi];
          continue;
        }
        return;
      }
      if (task == LBFGSB_NEW_X) {
        niter[gi] += 1;
        if (niter[gi] >= maxiter) opt[gi].stop(LBFGSB_STOP_ITER);
        else if (nfev[gi] > maxfun) opt[gi].stop(LBFGSB_STOP_FUN);
        continue;
      }
      active[gi] = false;
      return;
    }
  };
  // The groups are independent between evaluations: odd-numbered groups advance on the context's helper thread
  // while the calling thread advances the even ones (hand-off through two monotonic round counters, spin-waiting).
  OptHelper& hp = ctx->helper;
  // With many groups (the stress configuration: 256 restarts = 52 joint problems of up to 5 x 78 variables) two threads
  // spend longer on the state machines than the device on the evaluations (90-150 us against 100-180 us per round): then up
  // to five more helpers share the groups (stride T over the group index; the calling thread takes every T-th).
  const int T = ngroups >= 12 ? std::min(7, 1 + ngroups / 8) : 2;          // threads that step groups, the caller included
  if (T > 2) while ((int)ctx->more_helpers.size() < T - 2) ctx->more_helpers.emplace_back(new OptHelper());
  std::vector<OptHelper*> hs;                                               // hs[i] steps the groups gi = i + 1 (mod T)
  if (ngroups > 1) hs.push_back(&hp);
  for (int i = 0; i < T - 2; ++i) hs.push_back(ctx->more_helpers[i].get());
  for (size_t i = 0; i < hs.size(); ++i) {
    const int first = (int)i + 1;
    hs[i]->begin([&, first] { for (int gi = first; gi < ngroups; gi += T) advance(gi); });
  }
  struct Ender { std::vector<OptHelper*>& h; ~Ender() { for (OptHelper* x : h) x->end(); } } ender{hs};
  unsigned round_no = 0;
  std::vector<int> qoff(ngroups, -1);
  int srv_cap = 0;                       // > 0 while a resident kernel of that many queries is in flight
  struct ServerStop {                    // whatever way the loop is left: tell the kernel to go, wait until it has gone
    pcabo_ctx* c; int& cap; int k;
    void stop() {
      if (cap <= 0) return;
      server_post(c, cap, 0, k, ++c->seq);
      (void)hipStreamSynchronize(c->stream);
      cap = 0;
    }
    ~ServerStop() { stop(); }
  } server_stop{ctx, srv_cap, k};
  while (true) {
    ++round_no;
    unsigned tickets[8] = {0};
    for (size_t i = 0; i < hs.size(); ++i) {
      tickets[i] = hs[i]->go.load(std::memory_order_relaxed) + 1;
      hs[i]->go.store(tickets[i], std::memory_order_release);
    }
    for (int gi = 0; gi < ngroups; gi += T) advance(gi);
    for (size_t i = 0; i < hs.size(); ++i)
      while (hs[i]->done.load(std::memory_order_acquire) != tickets[i]) __builtin_ia32_pause();
    int nq = 0;
    for (int gi = 0; gi < ngroups; ++gi) {
      qoff[gi] = -1;
      if (!active[gi]) continue;
      qoff[gi] = nq;
      memcpy(ctx->hXq + (size_t)nq * k, x[gi].data(), (size_t)gsize[gi] * k * sizeof(double));
      nq += gsize[gi];
    }
    if (nq == 0) break;
    int rc;
    if (round_no == 1 && (ctx->alone_age++ & 7) == 0) ctx->alone = presence_alone(ctx->device);
    // several contexts in ONE process (one run per host thread, or a batch next to a single run): their resident kernels
    // would each want most of the chip at the same time - one launch per evaluation then, which interleaves well
    const bool only_context_here = presence_local_contexts(ctx->device) == 1;
    if (round_no == 1 && ctx->srv_penalty > 0) --ctx->srv_penalty;
    else if (round_no == 1 && ctx->opt_resident && ctx->mail_bar && !ctx->opt_group_acq && ctx->alone && only_context_here && !ctx->prof && acq_server_possible(nq, ctx->n, k, ctx->NP)) {
      // the evaluations of this call go to ONE resident launch (see k_acq_fast): no launch and no operand refill per round
      srv_cap = nq;
      launch_acq(ctx->stream, nullptr, nullptr, srv_cap, ctx->n, k, ctx->NP, ctx->ld, ctx->dZnT, ctx->dR, ctx->dAlpha,
                 ctx->dBounds4, ctx->dYstats, p, ctx->dPartial, ctx->dCounters, ctx->dVal, ctx->dGrad, ctx->hVal, ctx->hGrad,
                 ctx->hm, ctx->seq + 1, ctx->dMail, ctx->dPairs);
      HIPCHK(hipGetLastError());
      if (ngroups == 2 && active[0] && active[1] && nq == num_restarts) {
        // ---- the two restart groups free of each other ------------------------------------------------------------
        // Each group has its own slots, control pairs and round counter in the mailbox, so each host thread drives its
        // own group (post - wait - L-BFGS-B step) without meeting the other: a round no longer waits for the slower of
        // the two steps and there is no hand-off between the threads.  Same evaluations, same order per group.
        const unsigned long long seq0 = ctx->seq;
        MailPair* m = ctx->dMail;
        const int cap = srv_cap;
        std::atomic<int> abort_flag{0};
        int timed_out[2] = {0, 0}, got_nan[2] = {0, 0}, left[2] = {0, 0}, slow_first[2] = {0, 0};
        unsigned long long used[2] = {0, 0};
        auto free_loop = [&](int gi) {
          const int q0 = gstart[gi], nqg = gsize[gi];
          unsigned long long r = 0;
          for (;;) {
            if (abort_flag.load(std::memory_order_relaxed)) { pending[gi] = 1; break; }
            const unsigned long long tag = seq0 + (++r);
            for (int t = 0; t < nqg * k; ++t) put_mail_pair(m + 1 + q0 * k + t, x[gi][t], tag);
            for (int j = 0; j < nqg; ++j) put_mail_pair(m + 1 + cap * k + q0 + j, 1.0, tag);
            _mm_sfence();
            const auto t0 = std::chrono::steady_clock::now();
            unsigned long spins = 0;
            bool ok = true;
            for (int j = 0; j < nqg && ok; ++j) {
              while (__atomic_load_n(&ctx->hm->qflag[q0 + j], __ATOMIC_ACQUIRE) != tag) {
                if ((++spins & 0xFFF) == 0) {
                  if (abort_flag.load(std::memory_order_relaxed)) { ok = false; break; }
                  if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) {
                    timed_out[gi] = 1; abort_flag.store(1, std::memory_order_relaxed); ok = false; break;
                  }
                }
              }
            }
            if (!ok) { pending[gi] = 1; break; }
            if (r == 1 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) slow_first[gi] = 1;
            double fs = 0.0;
            bool nan = false;
            for (int j = 0; j < nqg; ++j) fs += ctx->hVal[q0 + j];
            for (int t = 0; t < nqg * k; ++t) {
              const double gv = -ctx->hGrad[(size_t)q0 * k + t];
              if (gv != gv) nan = true;
              g[gi][t] = gv;
            }
            if (nan) { got_nan[gi] = 1; abort_flag.store(1, std::memory_order_relaxed); break; }
            fval[gi] = -fs;
            nfev[gi] += 1;
            xc[gi] = x[gi]; gc[gi] = g[gi]; fc[gi] = fval[gi]; have_cache[gi] = 1;
            vc[gi].assign(ctx->hVal + q0, ctx->hVal + q0 + nqg);
            advance(gi);
            if (!active[gi]) break;
          }
          // this group's slab and finishing groups may go (they wait for round r + 1 of their own count)
          const unsigned long long bye = seq0 + r + 1;
          for (int j = 0; j < nqg; ++j) put_mail_pair(m + 1 + cap * k + q0 + j, 0.0, bye);
          _mm_sfence();
          left[gi] = 1;
          used[gi] = r + 1;
        };
        hp.fn = [&] { free_loop(1); };                 // the helper is idle here (its ticket of this round is done)
        const unsigned fticket = hp.go.load(std::memory_order_relaxed) + 1;
        hp.go.store(fticket, std::memory_order_release);
        free_loop(0);
        while (hp.done.load(std::memory_order_acquire) != fticket) __builtin_ia32_pause();
        hp.fn = [&] { for (int gi = 1; gi < ngroups; gi += T) advance(gi); };
        HIPCHK(hipStreamSynchronize(ctx->stream));      // every group of the grid has been told to leave
        ctx->seq = seq0 + std::max(used[0], used[1]) + 1;
        srv_cap = 0;
        if (got_nan[0] || got_nan[1]) return set_err(ctx, PCABO_ERR_NAN, "NaN in acquisition gradient%s", "");
        if (timed_out[0] || timed_out[1]) ctx->srv_penalty = 1000;      // plain launches from here on (see below)
        else if (slow_first[0] || slow_first[1]) ctx->srv_penalty = 40;
        if (!active[0] && !active[1]) break;            // the normal end: both groups ran to their stop
        continue;                                       // a wait failed: the lock-step loop below finishes the call
      }
    }
    if (srv_cap > 0) {
      const unsigned long long tag = ++ctx->seq;
      const double tp = now();
      server_post(ctx, srv_cap, nq, k, tag);
      rc = server_wait(ctx, nq, tag);
      if (rc == PCABO_ERR_TIMEOUT) {       // e.g. this thread lost the CPU for longer than the kernel waits: plain launches from here on
        server_stop.stop();
        ctx->srv_penalty = 1000;
        rc = eval_staged(ctx, nq, p);
      } else if (round_no == 1 && now() - tp > 1e-3) {
        // A first answer after more than a millisecond (normal: ~20 us) means the grid could not become resident at once:
        // another process holds the GPU with its own resident kernel.  Resident kernels of different processes then run
        // one after the other, whole calls at a time; plain launches interleave much better, so use them for a while.
        ctx->srv_penalty = 40;
      }
    } else if (group_mode(ctx, nq, 1) && batch_limit <= PCABO_GROUP_Q && ngroups <= 8) {
      int g0[8], gn[8], ng = 0;
      for (int gi = 0; gi < ngroups; ++gi) if (qoff[gi] >= 0) { g0[ng] = qoff[gi]; gn[ng] = gsize[gi]; ++ng; }
      rc = eval_staged_groups(ctx, g0, gn, ng, p);
    } else {
      rc = eval_staged(ctx, nq, p);
    }
    if (rc != PCABO_OK) return rc;
    for (int gi = 0; gi < ngroups; ++gi) {
      if (qoff[gi] < 0) continue;
      double fs = 0.0;
      bool nan = false;
      for (int j = 0; j < gsize[gi]; ++j) fs += ctx->hVal[qoff[gi] + j];
      for (int t = 0; t < gsize[gi] * k; ++t) {
        double gv = -ctx->hGrad[(size_t)qoff[gi] * k + t];
        if (gv != gv) nan = true;
        g[gi][t] = gv;
      }
      if (nan) return set_err(ctx, PCABO_ERR_NAN, "NaN in acquisition gradient%s", "");
      fval[gi] = -fs;
      nfev[gi] += 1;
      xc[gi] = x[gi]; gc[gi] = g[gi]; fc[gi] = fval[gi]; have_cache[gi] = 1;
      vc[gi].assign(ctx->hVal + qoff[gi], ctx->hVal + qoff[gi] + gsize[gi]);
    }
  }
  server_stop.stop();
  // final clamp and acquisition values at the candidates (no gradient)
  for (int gi = 0; gi < ngroups; ++gi) {
    for (int t = 0; t < gsize[gi] * k; ++t) {
      double v = x[gi][t];
      v = v < lo[gi][t] ? lo[gi][t] : (v > hi[gi][t] ? hi[gi][t] : v);
      cand[(size_t)gstart[gi] * k + t] = v;
      ctx->hXq[(size_t)gstart[gi] * k + t] = v;
    }
    int wf = opt[gi].warnflag();
    if (info) { info[4 * gi] = niter[gi]; info[4 * gi + 1] = nfev[gi]; info[4 * gi + 2] = wf; info[4 * gi + 3] = opt[gi].task(); }
    if (wf == 2) any_failed = 1;
  }
  // botorch evaluates the acquisition once more at the clamped end points.  An L-BFGS-B run normally ends ON the last point
  // it had evaluated (the accepted trial of its last line search), whose per-restart values are still here - the same
  // kernel arithmetic, so the same bits; only a run that ended elsewhere (abnormal line search) needs the launch.
  bool reuse = true;
  for (int gi = 0; gi < ngroups && reuse; ++gi)
    reuse = have_cache[gi] && (int)vc[gi].size() == gsize[gi] &&
            memcmp(cand + (size_t)gstart[gi] * k, xc[gi].data(), (size_t)gsize[gi] * k * sizeof(double)) == 0;
  if (reuse) {
    for (int gi = 0; gi < ngroups; ++gi)
      for (int j = 0; j < gsize[gi]; ++j) vals[gstart[gi] + j] = vc[gi][j];
  } else {
    AcqParams pv = make_params(ctx, best_f, maximize, acq, 0);
    // (through the slab kernels whatever the number of restarts: the values of a restart must not depend on how many
    // restarts share the call)
    int rc = eval_staged(ctx, num_restarts, pv, false);
    if (rc != PCABO_OK) return rc;
    for (int j = 0; j < num_restarts; ++j) vals[j] = ctx->hVal[j];
  }
  if (failed) *failed = any_failed;
  return PCABO_OK;
}

int pcabo_inverse_map(pcabo_ctx* ctx, const double* z, double* x) {
  if (!ctx || !z || !x) return PCABO_ERR_ARG;
  if (!ctx->have_wpca) return set_err(ctx, PCABO_ERR_ARG, "pcabo_inverse_map: call pcabo_wpca first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  // z travels through the pinned query block (the kernel reads it in place over PCIe: 36 doubles), x comes back the way the
  // acquisition results do - stores to pinned host memory followed by a sequence word the host polls: no copy, no stream
  // synchronisation (40 -> 15 us per BO iteration on the host clock)
  memcpy(ctx->hXq, z, (size_t)ctx->k * sizeof(double));
  const unsigned long long seq = ++ctx->seq;
  double* hx = ctx->hSmall + (size_t)ctx->max_d * ctx->max_d + 3 * (size_t)ctx->max_d;      // behind the wPCA results' block
  launch_inverse_map(s, ctx->hXq, ctx->dComps, ctx->dDataMean, ctx->dPcaMean, ctx->k, ctx->d, ctx->dXout, nullptr, ZB(),
                     hx, ctx->hm, seq);
  HIPCHK(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long spins = 1; __atomic_load_n(&ctx->hm->qflag[0], __ATOMIC_ACQUIRE) != seq; ++spins) {
    if ((spins & 0xFFFF) == 0) {
      if (hipStreamQuery(s) == hipSuccess) {
        if (__atomic_load_n(&ctx->hm->qflag[0], __ATOMIC_ACQUIRE) == seq) break;
        HIPCHK(hipGetLastError());
        return set_err(ctx, PCABO_ERR_TIMEOUT, "inverse-map kernel finished without publishing its result%s", "");
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0)
        return set_err(ctx, PCABO_ERR_TIMEOUT, "device did not publish the inverse map%s", "");
    }
  }
  memcpy(x, hx, (size_t)ctx->d * sizeof(double));
  return PCABO_OK;
}

int pcabo_get_gp_state(pcabo_ctx* ctx, double* K_chol, double* Rinv, double* alpha, double* y_mean_std,
                       double* norm_bounds) {
  if (!ctx) return PCABO_ERR_ARG;
  if (!ctx->have_gp) return set_err(ctx, PCABO_ERR_ARG, "pcabo_get_gp_state: call pcabo_gp_condition first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t n = ctx->n, w = n * sizeof(double), pitch = (size_t)ctx->ld * sizeof(double);
  if (K_chol) HIPCHK(hipMemcpy2DAsync(K_chol, w, ctx->dL, pitch, w, n, hipMemcpyDeviceToHost, s));
  if (Rinv) HIPCHK(hipMemcpy2DAsync(Rinv, w, ctx->dR, pitch, w, n, hipMemcpyDeviceToHost, s));
  if (alpha) HOST_OUT(alpha, ctx->dAlpha, n, double);
  HIPCHK(hipStreamSynchronize(s));
  if (y_mean_std) { y_mean_std[0] = ctx->hm->y_mean; y_mean_std[1] = ctx->hm->y_std; }
  if (norm_bounds)
    for (int c = 0; c < ctx->k; ++c) { norm_bounds[c] = ctx->hm->norm_lo[c]; norm_bounds[ctx->k + c] = ctx->hm->norm_hi[c]; }
  return PCABO_OK;
}

int pcabo_get_gram(pcabo_ctx* ctx, double* K) {
  if (!ctx || !K) return PCABO_ERR_ARG;
  if (!ctx->have_gp) return set_err(ctx, PCABO_ERR_ARG, "pcabo_get_gram: call pcabo_gp_condition first%s", "");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = ctx->n, w = n * sizeof(double), pitch = (size_t)ctx->ld * sizeof(double);
  // built on demand: the conditioning itself writes only the copy that the factorisation then overwrites
  ctx->rt_stale = true;
  launch_gram(ctx->stream, ctx->dAT, ctx->dNrm, ctx->n, ctx->NP, round_up(ctx->k, 4), ctx->ld, ctx->noise, ctx->kernel, ctx->dGram,
              nullptr, nullptr, nullptr);
  HIPCHK(hipMemcpy2DAsync(K, w, ctx->dGram, pitch, w, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; ++i)                  // only the lower tiles are built on the device
    for (size_t j = i + 1; j < n; ++j) K[i * n + j] = K[j * n + i];
  return PCABO_OK;
}

// (pcabo_lbfgsb_minimize, pcabo_lbfgsb_set_vector_kernels, pcabo_sobol_scramble, pcabo_sobol_draw: host_entry.cpp - host code only,
// built by g++ and, for the sanitizer targets of the Makefile, without the HIP runtime)

int pcabo_set_profiling(pcabo_ctx* ctx, int enabled) {
  if (!ctx) return PCABO_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (enabled && ctx->pairs.empty()) {
    ctx->pairs.resize(PROF_POOL);
    for (auto& p : ctx->pairs) { HIPCHK(hipEventCreate(&p.a)); HIPCHK(hipEventCreate(&p.b)); p.group = 0; }
  }
  if (!enabled) prof_resolve(ctx);
  if (enabled && !ctx->prof) {
    // calibrate the event-pair overhead on this stream: median reading of 64 back-to-back pairs (nothing in between;
    // subtracted from every pair) - the reading around an EMPTY kernel is reported with PCABO_TRACE_OPT for comparison
    prof_resolve(ctx);
    double med[2] = {0.0, 0.0};
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<float> r;
      for (int i = 0; i < 64; ++i) {
        HIPCHK(hipEventRecord(ctx->pairs[i].a, ctx->stream));
        if (mode == 1) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, ctx->stream);
        HIPCHK(hipEventRecord(ctx->pairs[i].b, ctx->stream));
      }
      HIPCHK(hipStreamSynchronize(ctx->stream));
      for (int i = 0; i < 64; ++i) { float ms = 0.f; if (hipEventElapsedTime(&ms, ctx->pairs[i].a, ctx->pairs[i].b) == hipSuccess) r.push_back(ms); }
      if (!r.empty()) { std::sort(r.begin(), r.end()); med[mode] = r[r.size() / 2]; }
    }
    ctx->prof_pair_ms = med[0];
    ctx->prof_empty_ms = med[1];
  }
  ctx->prof = enabled != 0;
  return PCABO_OK;
}

int pcabo_get_profile_calibration(pcabo_ctx* ctx, double* pair_ms, double* empty_kernel_ms) {
  if (!ctx) return PCABO_ERR_ARG;
  if (pair_ms) *pair_ms = ctx->prof_pair_ms;
  if (empty_kernel_ms) *empty_kernel_ms = ctx->prof_empty_ms;
  return PCABO_OK;
}

int pcabo_get_profile(pcabo_ctx* ctx, int which, double* ms, int64_t* launches, double* bytes, double* flops) {
  if (!ctx || which < 0 || which >= PROF_GROUPS) return PCABO_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  prof_resolve(ctx);
  if (ms) *ms = ctx->prof_ms[which];
  if (launches) *launches = ctx->prof_launches[which];
  if (bytes) *bytes = ctx->prof_bytes[which];
  if (flops) *flops = ctx->prof_flops[which];
  return PCABO_OK;
}

int pcabo_reset_profile(pcabo_ctx* ctx) {
  if (!ctx) return PCABO_ERR_ARG;
  prof_resolve(ctx);
  for (int i = 0; i < PROF_GROUPS; ++i) { ctx->prof_ms[i] = 0.0; ctx->prof_launches[i] = 0; ctx->prof_bytes[i] = 0.0; ctx->prof_flops[i] = 0.0; }
  return PCABO_OK;
}


// =====================================================================================================================
// Batched contexts (declared in include/pcabo.h): B runs advancing in lock-step.
// =====================================================================================================================
}  // extern "C"

// RestartGroup (one joint L-BFGS-B problem of a run) and GangPool (the batch's worker threads): host_side.h

struct pcabo_batch {
  int device = 0, B = 0, max_n = 0, max_d = 0, max_q = 0;
  hipStream_t stream = nullptr;
  hipEvent_t evPca = nullptr, evBounds = nullptr;
  char *dSlab = nullptr, *hSlab = nullptr;
  size_t zs = 0, hzs = 0;
  std::vector<pcabo_ctx*> ctx;
  std::vector<char> active;              // runs that still take part in pcabo_batch_optimize_acqf (pcabo_batch_set_active)
  int n = 0, d = 0, NP = 0;
  bool wpca_uncollected = false, gp_pending = false, have_gp = false;
  double lengthscale = 0.0, noise = 0.0;
  int kernel = 0;
  int gcur = 0, vprev_d = 0;             // eigenvector ping-pong of ALL runs (they advance together)
  int cnt_S = 0; bool cnt_dirty = true;
  bool prof = false; hipEvent_t pev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // phase marks of the last conditioning
  bool group_acq = true;                 // L-BFGS-B rounds through k_acq_group (pcabo_batch_set_option(PCABO_OPT_GROUP_ACQ, 0): the per-query kernels)
  size_t in_stride_x = 0, in_stride_noise = 0, in_stride_y = 0;   // pcabo_batch_set_input_strides (doubles between two runs' blocks; 0: dense)
  bool score_enqueued = false;           // pcabo_batch_gp_condition_end_eval_begin without its _end yet
  bool imap_enqueued = false;            // pcabo_batch_inverse_map_begin without its _end yet
  bool opt_enqueued = false; int opt_restarts = 0, opt_limit = 0; std::vector<int> opt_act;   // pcabo_batch_optimize_acqf_begin without its _end yet
  int dev_lbfgsb = 0;                    // PCABO_OPT_DEVICE_LBFGSB: 1 device-resident L-BFGS-B, 2 its host-stepped twin
  unsigned *dOptTab = nullptr, *hOptTab = nullptr;   // launch table of the device-resident optimiser (B * 32 entries)
  int opt_cus = 0;                       // PCABO_OPT_LBFGSB_CUS: > 0 = the optimiser's launches run on a stream confined to that many CUs
  hipStream_t optStream = nullptr; int optStream_cus = 0;
  hipEvent_t evOptIn = nullptr, evOptOut = nullptr;
  int G = 0;                             // gangs = worker threads of the L-BFGS-B phase
  std::vector<hipStream_t> gstream;
  GangPool pool;
  std::atomic<unsigned long long> seq{0};
  char err[512] = {0};
};

static int bset_err(pcabo_batch* b, int code, const char* fmt, const char* a = "", int v = 0) {
  if (b) snprintf(b->err, sizeof(b->err), fmt, a, v);
  return code;
}
#define BHIPCHK(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) return bset_err(batch, PCABO_ERR_HIP, "HIP error: %s (line %d)", hipGetErrorString(e_), __LINE__); \
  } while (0)

static ZB batch_zb(const pcabo_batch* b) { ZB z; z.B = b->B; z.zs = b->zs; z.hzs = b->hzs; return z; }

static void batch_free(pcabo_batch* batch) {
  (void)hipSetDevice(batch->device);
  batch->pool.shutdown();
  if (batch->stream) (void)hipStreamSynchronize(batch->stream);
  for (pcabo_ctx* c : batch->ctx) if (c) { ctx_teardown(c); delete c; }
  for (hipStream_t s : batch->gstream) if (s) (void)hipStreamDestroy(s);
  if (batch->evPca) (void)hipEventDestroy(batch->evPca);
  if (batch->evBounds) (void)hipEventDestroy(batch->evBounds);
  for (hipEvent_t e : batch->pev) if (e) (void)hipEventDestroy(e);
  if (batch->dSlab) (void)hipFree(batch->dSlab);
  if (batch->hSlab) (void)hipHostFree(batch->hSlab);
  if (batch->optStream) { (void)hipStreamSynchronize(batch->optStream); (void)hipStreamDestroy(batch->optStream); }
  if (batch->evOptIn) (void)hipEventDestroy(batch->evOptIn);
  if (batch->evOptOut) (void)hipEventDestroy(batch->evOptOut);
  if (batch->dOptTab) (void)hipFree(batch->dOptTab);
  if (batch->hOptTab) (void)hipHostFree(batch->hOptTab);
  if (batch->stream) (void)hipStreamDestroy(batch->stream);
  delete batch;
}

extern "C" {

int pcabo_batch_create(int device, int B, int max_n, int max_d, int max_q, pcabo_batch** out) {
  if (!out) return PCABO_ERR_ARG;
  *out = nullptr;
  if (B < 1 || B > 4096 || !ctx_sizes_ok(max_n, max_d, max_q)) return PCABO_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PCABO_ERR_HIP;
  pcabo_batch* batch = new (std::nothrow) pcabo_batch();
  if (!batch) return PCABO_ERR_HIP;
  *out = batch;                      // handed back even on failure so the caller can read the message (and must destroy it)
  batch->device = device; batch->B = B; batch->max_n = max_n; batch->max_d = max_d; batch->max_q = max_q;
  BHIPCHK(hipSetDevice(device));
  BHIPCHK(hipStreamCreateWithFlags(&batch->stream, hipStreamNonBlocking));
  BHIPCHK(hipEventCreateWithFlags(&batch->evPca, hipEventDisableTiming));
  BHIPCHK(hipEventCreateWithFlags(&batch->evBounds, hipEventDisableTiming));
  {
    pcabo_ctx probe;                 // layout only
    probe.max_n = max_n; probe.max_d = max_d; probe.max_q = max_q;
    probe.NPcap = round_up(max_n, PCABO_BS); probe.DPcap = round_up(max_d, 16); probe.KPcap = round_up(max_d, 4);
    probe.Scap = probe.NPcap / PCABO_SLAB;
    batch->zs = carve_device(&probe, nullptr);
    batch->hzs = carve_host(&probe, nullptr);
  }
  BHIPCHK(hipMalloc((void**)&batch->dSlab, batch->zs * (size_t)B));
  BHIPCHK(hipHostMalloc((void**)&batch->hSlab, batch->hzs * (size_t)B, hipHostMallocDefault));
  memset(batch->hSlab, 0, batch->hzs * (size_t)B);
  BHIPCHK(hipMalloc((void**)&batch->dOptTab, (size_t)B * PCABO_INLAUNCH_MAXQ * sizeof(unsigned)));
  BHIPCHK(hipHostMalloc((void**)&batch->hOptTab, (size_t)B * PCABO_INLAUNCH_MAXQ * sizeof(unsigned), hipHostMallocDefault));
  BHIPCHK(hipMemsetAsync(batch->dSlab, 0, batch->zs * (size_t)B, batch->stream));
  BHIPCHK(hipStreamSynchronize(batch->stream));
  batch->ctx.assign(B, nullptr);
  batch->active.assign(B, 1);
  for (int b = 0; b < B; ++b) {
    pcabo_ctx* c = new (std::nothrow) pcabo_ctx();
    if (!c) return bset_err(batch, PCABO_ERR_HIP, "out of memory%s", "");
    batch->ctx[b] = c;
    int rc = ctx_setup(c, device, max_n, max_d, max_q, batch->dSlab + batch->zs * (size_t)b,
                       batch->hSlab + batch->hzs * (size_t)b, batch->stream);
    c->batch_index = b;
    if (rc != PCABO_OK) return bset_err(batch, rc, "context %s%d of the batch could not be set up", "", b);
    if (c->region_bytes != batch->zs || c->hregion_bytes != batch->hzs)
      return bset_err(batch, PCABO_ERR_ARG, "internal: layout mismatch%s", "");
  }
  // worker threads of the L-BFGS-B phase: one gang of runs per thread, a HIP stream per gang
  // (hardware_concurrency reports the host, not this process's share of it: a GPU of a shared node comes with ~16 cores,
  // and every worker spins while it waits - 8 by default)
  int T = std::min(8, (int)std::thread::hardware_concurrency() - 2);        // (pcabo_batch_set_workers changes it)
  T = std::max(1, std::min(T, std::min(B, 32)));
  batch->G = T;
  batch->gstream.assign((size_t)T, nullptr);
  for (size_t g = 0; g < batch->gstream.size(); ++g) BHIPCHK(hipStreamCreateWithFlags(&batch->gstream[g], hipStreamNonBlocking));
  batch->pool.start(T);
  for (pcabo_ctx* c : batch->ctx) c->opt_group_acq = batch->group_acq;      // a run's single-context calls match its batch
  batch->seq.store(((unsigned long long)(getpid() & 0xffff) << 44) + (1ull << 43));
  return PCABO_OK;
}

int pcabo_batch_destroy(pcabo_batch* batch) {
  if (!batch) return PCABO_ERR_ARG;
  batch_free(batch);
  return PCABO_OK;
}

// Worker threads (= gangs) of the L-BFGS-B phase.  Every worker spins while its launch is in flight, so the workers of all
// batches of a process should fit the cores the process owns: two batches advancing side by side on two host threads take
// 4 each on a 16-core share.  Not during a call on this batch.  The runs' results do not depend on it.
int pcabo_batch_set_workers(pcabo_batch* batch, int workers) {
  if (!batch || workers < 1) return PCABO_ERR_ARG;
  BHIPCHK(hipSetDevice(batch->device));
  const int T = std::max(1, std::min(workers, std::min(batch->B, 32)));
  if (T == batch->G) return PCABO_OK;
  batch->pool.shutdown();
  BHIPCHK(hipStreamSynchronize(batch->stream));
  for (hipStream_t& s : batch->gstream) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); s = nullptr; }
  batch->gstream.assign((size_t)T, nullptr);
  for (size_t g = 0; g < batch->gstream.size(); ++g) BHIPCHK(hipStreamCreateWithFlags(&batch->gstream[g], hipStreamNonBlocking));
  batch->G = T;
  batch->pool.start(T);
  return PCABO_OK;
}

// PCABO_OPT_GROUP_ACQ (default 1): the L-BFGS-B rounds of the batch go through the throughput kernel k_acq_group; 0: through
// the per-query kernels a single context uses by default (same formulas, another summation order) - with it a run of the batch
// is bit-identical to the same run in a stand-alone context with default options.  Not during a call on this batch.
int pcabo_batch_set_option(pcabo_batch* batch, int option, int value) {
  if (!batch) return PCABO_ERR_ARG;
  if (option == PCABO_OPT_LBFGSB_CUS) {
    if (value < 0 || value > 1024) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_set_option: PCABO_OPT_LBFGSB_CUS takes 0 .. 1024 (%s%d)", "", value);
    batch->opt_cus = value;
    return PCABO_OK;
  }
  if (option == PCABO_OPT_DEVICE_LBFGSB) {
    if (value < 0 || value > 2) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_set_option: PCABO_OPT_DEVICE_LBFGSB takes 0, 1 or 2 (%s%d)", "", value);
    // switched on after a conditioning: dGram holds K (or an older RT), not this GP's transposed root inverse - the next
    // launch of the device optimiser rebuilds it (k_rt_build) instead of reading whatever is there
    if (value != 0 && batch->dev_lbfgsb == 0) for (pcabo_ctx* c : batch->ctx) if (c->have_gp) c->rt_stale = true;
    batch->dev_lbfgsb = value;
    return PCABO_OK;
  }
  if (option != PCABO_OPT_GROUP_ACQ) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_set_option: unknown option %s%d", "", option);
  batch->group_acq = value != 0;
  for (pcabo_ctx* c : batch->ctx) c->opt_group_acq = batch->group_acq;
  return PCABO_OK;
}

// Device time of the phases of the LAST pcabo_batch_wpca_gp_condition_begin (HIP events on the batch's stream; call after
// the conditioning has been waited for): ms[0] rows A-C (wPCA), ms[1] Normalize + Gram, ms[2] Cholesky, ms[3] root inverse + alpha.
int pcabo_batch_set_profiling(pcabo_batch* batch, int enabled) {
  if (!batch) return PCABO_ERR_ARG;
  BHIPCHK(hipSetDevice(batch->device));
  if (enabled && !batch->pev[0]) for (auto& e : batch->pev) BHIPCHK(hipEventCreate(&e));
  batch->prof = enabled != 0;
  return PCABO_OK;
}
int pcabo_batch_get_profile(pcabo_batch* batch, double* ms) {
  if (!batch || !ms || !batch->prof) return PCABO_ERR_ARG;
  BHIPCHK(hipSetDevice(batch->device));
  BHIPCHK(hipEventSynchronize(batch->pev[4]));
  for (int i = 0; i < 4; ++i) { float t = 0.f; BHIPCHK(hipEventElapsedTime(&t, batch->pev[i], batch->pev[i + 1])); ms[i] = t; }
  return PCABO_OK;
}

// A run whose optimisation cannot go on (botorch would raise there: NaN in the acquisition gradient once the reference's
// unclipped candidates have blown the search box up) is parked by the caller: it stays in the lock-step launches of rows
// A-K (on whatever finite data the caller keeps feeding it) but no longer takes part in pcabo_batch_optimize_acqf.
int pcabo_batch_set_active(pcabo_batch* batch, const int* active) {
  if (!batch || !active) return PCABO_ERR_ARG;
  for (int b = 0; b < batch->B; ++b) batch->active[b] = active[b] != 0;
  return PCABO_OK;
}

int pcabo_batch_last_error(pcabo_batch* batch, char* buf, int buflen) {
  if (!batch || !buf || buflen <= 0) return PCABO_ERR_ARG;
  snprintf(buf, buflen, "%s", batch->err);
  return PCABO_OK;
}

pcabo_ctx* pcabo_batch_ctx(pcabo_batch* batch, int b) {
  if (!batch || b < 0 || b >= batch->B) return nullptr;
  return batch->ctx[b];
}

// X, noise and y of pcabo_batch_wpca_gp_condition_begin as they lie in the caller's own arrays: doubles between the blocks of two
// runs (0 = dense, the default).  A driver that keeps X as [B][budget][d] and y as [B][budget] hands the first n rows of every run
// over without building a dense copy per iteration.
int pcabo_batch_set_input_strides(pcabo_batch* batch, size_t x_stride, size_t noise_stride, size_t y_stride) {
  if (!batch) return PCABO_ERR_ARG;
  batch->in_stride_x = x_stride; batch->in_stride_noise = noise_stride; batch->in_stride_y = y_stride;
  return PCABO_OK;
}

int pcabo_batch_wpca_gp_condition_begin(pcabo_batch* batch, const double* X, const int64_t* ranks, const double* noise,
                                        const double* y, int n, int d, int maximize, double var_threshold,
                                        int n_components, double lengthscale, double gp_noise, int kernel) {
  (void)maximize;                    // ranks are given: nothing on the device depends on the direction
  if (!batch) return PCABO_ERR_ARG;
  pcabo_ctx* c0 = batch->ctx[0];
  if (!X || !ranks || !y || n < 2 || n > batch->max_n || d < 1 || d > batch->max_d || !gp_args_ok(c0, n, lengthscale, gp_noise, kernel))
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_wpca_gp_condition_begin: bad argument or size beyond the batch's capacity%s", "");
  if (batch->gp_pending) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_wpca_gp_condition_begin: a conditioning is in flight%s", "");
  BHIPCHK(hipSetDevice(batch->device));
  hipStream_t s = batch->stream;
  const int B = batch->B;
  const size_t nd = (size_t)n * d;
  // pack [X | noise | ranks | y] of every run into its pinned block; ONE strided copy carries all runs
  const size_t off_noise = nd, off_r = noise ? 2 * nd : nd, off_y = off_r + n, total = off_y + n;
  const size_t sx = batch->in_stride_x ? batch->in_stride_x : nd, sn = batch->in_stride_noise ? batch->in_stride_noise : nd;
  const size_t sy = batch->in_stride_y ? batch->in_stride_y : (size_t)n;
  if (sx < nd || sn < nd || sy < (size_t)n)
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_wpca_gp_condition_begin: an input stride is smaller than a run's block%s", "");
  for (int b = 0; b < B; ++b) {
    double* h = batch->ctx[b]->hIn;
    memcpy(h, X + (size_t)b * sx, nd * sizeof(double));
    if (noise) memcpy(h + off_noise, noise + (size_t)b * sn, nd * sizeof(double));
    memcpy(h + off_r, ranks + (size_t)b * n, (size_t)n * 8);
    memcpy(h + off_y, y + (size_t)b * sy, (size_t)n * sizeof(double));
  }
  BHIPCHK(hipMemcpy2DAsync(c0->dIn, batch->zs, c0->hIn, batch->hzs, total * sizeof(double), B, hipMemcpyHostToDevice, s));
  const double* inX = c0->dIn;
  const double* inNoise = noise ? c0->dIn + off_noise : nullptr;
  const long long* inRanks = reinterpret_cast<const long long*>(c0->dIn + off_r);
  const double* inY = c0->dIn + off_y;
  const ZB zb = batch_zb(batch);
  const int DP = round_up(d, 16);
  auto mark = [&](int i) { if (batch->prof) (void)hipEventRecord(batch->pev[i], s); };
  mark(0);
  // rows A-C
  launch_wpca_prep(s, inX, inRanks, inNoise, n, d, DP, c0->dWeights, c0->dDataMean, c0->dPcaMean, c0->dWc, zb);
  launch_cov(s, c0->dWc, n, DP, c0->dC, zb);
  const double* v0 = (batch->vprev_d == d) ? c0->dGbuf[batch->gcur] : nullptr;
  batch->gcur ^= 1;
  launch_jacobi(s, c0->dC, d, DP, v0, c0->dGbuf[batch->gcur], c0->dLam, c0->dSweeps, n, var_threshold, n_components,
                c0->dComps, c0->dEvr, c0->dK, c0->hm, zb);
  batch->vprev_d = d;
  launch_project(s, inX, c0->dDataMean, c0->dPcaMean, c0->dComps, c0->dK, n, d, c0->dZ, zb);
  const int rcount = n < d ? n : d;
  BHIPCHK(hipMemcpy2DAsync(c0->hSmall, batch->hzs, c0->dPcaOut, batch->zs,
                           ((size_t)3 * batch->max_d + (size_t)rcount * d) * sizeof(double), B, hipMemcpyDeviceToHost, s));
  BHIPCHK(hipEventRecord(batch->evPca, s));
  // rows D-H, queued behind the projection (k is read on the device)
  const int NP = round_up(n, PCABO_BS);
  if (acq_slabs(NP) != batch->cnt_S || batch->cnt_dirty) {
    BHIPCHK(hipMemset2DAsync(c0->dCounters, batch->zs, 0, (PCABO_CNT_DONE + 1) * sizeof(unsigned int), B, s));
    batch->cnt_S = acq_slabs(NP); batch->cnt_dirty = false;
  }
  mark(1);
  launch_zstats(s, c0->dZ, inY, n, -1, nullptr, c0->dBounds4, c0->dZnMean, c0->dYstats, c0->dYs, c0->hm, c0->dK, zb);
  BHIPCHK(hipEventRecord(batch->evBounds, s));
  launch_znorm(s, c0->dZ, n, -1, NP, 0, c0->ld, c0->dBounds4, c0->dZnMean, 1.0 / lengthscale, c0->dZnT, c0->dAT, c0->dNrm,
               c0->dK, zb);
  launch_gram(s, c0->dAT, c0->dNrm, n, NP, 0, c0->ld, gp_noise, kernel, nullptr, c0->dK, c0->dL, c0->dInfo, zb);
  mark(2);
  launch_cholesky(s, c0->dL, NP, c0->ld, c0->dInfo, c0->dDiag, zb);
  mark(3);
  launch_trinv(s, c0->dL, NP, c0->ld, c0->dR, zb);
  launch_alpha(s, c0->dR, c0->dYs, n, NP, c0->ld, c0->dTmp, c0->dAlpha, zb);
  mark(4);
  // the device-resident optimiser reads the root inverse transposed as well (the Gram buffer is free: K is only kept on demand)
  if (batch->dev_lbfgsb) launch_rt_build(s, c0->dR, n, NP, c0->ld, c0->dGram, zb);
  BHIPCHK(hipMemcpy2DAsync((void*)&c0->hm->chol_info, batch->hzs, c0->dInfo, batch->zs, sizeof(int), B, hipMemcpyDeviceToHost, s));
  BHIPCHK(hipGetLastError());
  batch->n = n; batch->d = d; batch->NP = NP;
  batch->lengthscale = lengthscale; batch->noise = gp_noise; batch->kernel = kernel;
  batch->wpca_uncollected = true; batch->gp_pending = true; batch->have_gp = false;
  for (int b = 0; b < B; ++b) {          // the per-run contexts see the same state (single-context calls keep working)
    pcabo_ctx* c = batch->ctx[b];
    c->n = n; c->d = d; c->NP = NP; c->lengthscale = lengthscale; c->noise = gp_noise; c->kernel = kernel;
    c->have_gp = false; c->have_wpca = false; c->gp_pending = false; c->wpca_uncollected = false;
    c->gcur = batch->gcur; c->dG = c->dGbuf[c->gcur]; c->vprev_d = d;
    c->cnt_S = batch->cnt_S; c->cnt_dirty = false;
  }
  return PCABO_OK;
}

// Rows D-H for every run WITHOUT the weighted PCA: the GPs live on the given points (pcabo_gp_condition_begin for B runs; the
// reference's Vanilla_BO, Vanilla_BO.py:166-196, conditions on the raw d-dimensional points with Normalize switched off - the
// caller passes identity bounds).  Z[B][n*k] and y[B][n] (strides: pcabo_batch_set_input_strides), norm_bounds[2*k] for all runs
// (lo[k], hi[k]) or NULL (bounds from the data as for PCA_BO).  Finish with pcabo_batch_gp_condition_end_eval.
int pcabo_batch_gp_condition_begin(pcabo_batch* batch, const double* Z, const double* y, int n, int k, const double* norm_bounds,
                                   double lengthscale, double gp_noise, int kernel) {
  if (!batch) return PCABO_ERR_ARG;
  pcabo_ctx* c0 = batch->ctx[0];
  if (!Z || !y || n < 2 || n > batch->max_n || k < 1 || k > batch->max_d || !gp_args_ok(c0, n, lengthscale, gp_noise, kernel))
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_gp_condition_begin: bad argument or size beyond the batch's capacity%s", "");
  if (batch->gp_pending) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_gp_condition_begin: a conditioning is in flight%s", "");
  BHIPCHK(hipSetDevice(batch->device));
  hipStream_t s = batch->stream;
  const int B = batch->B;
  const size_t nk = (size_t)n * k, off_y = nk, total = nk + n;
  const size_t sx = batch->in_stride_x ? batch->in_stride_x : nk, sy = batch->in_stride_y ? batch->in_stride_y : (size_t)n;
  if (sx < nk || sy < (size_t)n)
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_gp_condition_begin: an input stride is smaller than a run's block%s", "");
  for (int b = 0; b < B; ++b) {
    pcabo_ctx* c = batch->ctx[b];
    memcpy(c->hIn, Z + (size_t)b * sx, nk * sizeof(double));
    memcpy(c->hIn + off_y, y + (size_t)b * sy, (size_t)n * sizeof(double));
    c->hm->k = k;
    if (norm_bounds) memcpy(c->hSmall, norm_bounds, (size_t)2 * k * sizeof(double));
  }
  BHIPCHK(hipMemcpy2DAsync(c0->dIn, batch->zs, c0->hIn, batch->hzs, total * sizeof(double), B, hipMemcpyHostToDevice, s));
  BHIPCHK(hipMemcpy2DAsync(c0->dK, batch->zs, &c0->hm->k, batch->hzs, sizeof(int), B, hipMemcpyHostToDevice, s));
  if (norm_bounds)
    BHIPCHK(hipMemcpy2DAsync(c0->dUserNB, batch->zs, c0->hSmall, batch->hzs, (size_t)2 * k * sizeof(double), B, hipMemcpyHostToDevice, s));
  const double* inZ = c0->dIn;
  const double* inY = c0->dIn + off_y;
  const ZB zb = batch_zb(batch);
  auto mark = [&](int i) { if (batch->prof) (void)hipEventRecord(batch->pev[i], s); };
  mark(0);
  const int NP = round_up(n, PCABO_BS);
  if (acq_slabs(NP) != batch->cnt_S || batch->cnt_dirty) {
    BHIPCHK(hipMemset2DAsync(c0->dCounters, batch->zs, 0, (PCABO_CNT_DONE + 1) * sizeof(unsigned int), B, s));
    batch->cnt_S = acq_slabs(NP); batch->cnt_dirty = false;
  }
  mark(1);
  launch_zstats(s, inZ, inY, n, -1, norm_bounds ? c0->dUserNB : nullptr, c0->dBounds4, c0->dZnMean, c0->dYstats, c0->dYs, c0->hm, c0->dK, zb);
  BHIPCHK(hipEventRecord(batch->evBounds, s));
  BHIPCHK(hipEventRecord(batch->evPca, s));
  launch_znorm(s, inZ, n, -1, NP, 0, c0->ld, c0->dBounds4, c0->dZnMean, 1.0 / lengthscale, c0->dZnT, c0->dAT, c0->dNrm, c0->dK, zb);
  launch_gram(s, c0->dAT, c0->dNrm, n, NP, 0, c0->ld, gp_noise, kernel, nullptr, c0->dK, c0->dL, c0->dInfo, zb);
  mark(2);
  launch_cholesky(s, c0->dL, NP, c0->ld, c0->dInfo, c0->dDiag, zb);
  mark(3);
  launch_trinv(s, c0->dL, NP, c0->ld, c0->dR, zb);
  launch_alpha(s, c0->dR, c0->dYs, n, NP, c0->ld, c0->dTmp, c0->dAlpha, zb);
  mark(4);
  if (batch->dev_lbfgsb) launch_rt_build(s, c0->dR, n, NP, c0->ld, c0->dGram, zb);
  BHIPCHK(hipMemcpy2DAsync((void*)&c0->hm->chol_info, batch->hzs, c0->dInfo, batch->zs, sizeof(int), B, hipMemcpyDeviceToHost, s));
  BHIPCHK(hipGetLastError());
  batch->n = n; batch->d = k; batch->NP = NP;
  batch->lengthscale = lengthscale; batch->noise = gp_noise; batch->kernel = kernel;
  batch->wpca_uncollected = false; batch->gp_pending = true; batch->have_gp = false;
  for (int b = 0; b < B; ++b) {
    pcabo_ctx* c = batch->ctx[b];
    c->n = n; c->d = k; c->k = k; c->KP = round_up(k, 4); c->NP = NP; c->lengthscale = lengthscale; c->noise = gp_noise; c->kernel = kernel;
    c->have_gp = false; c->have_wpca = false; c->gp_pending = false; c->wpca_uncollected = false;
    c->cnt_S = batch->cnt_S; c->cnt_dirty = false;
  }
  return PCABO_OK;
}

int pcabo_batch_wpca_results(pcabo_batch* batch, double* data_mean, double* pca_mean, double* comps, double* evr, int* k) {
  if (!batch) return PCABO_ERR_ARG;
  if (!batch->wpca_uncollected && batch->n == 0) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_wpca_results: nothing enqueued%s", "");
  BHIPCHK(hipSetDevice(batch->device));
  BHIPCHK(wait_event(batch->evPca));
  BHIPCHK(hipGetLastError());
  const int n = batch->n, d = batch->d, rcount = n < d ? n : d;
  const size_t D = batch->max_d;
  for (int b = 0; b < batch->B; ++b) {
    pcabo_ctx* c = batch->ctx[b];
    const double* h = c->hSmall;
    if (data_mean) memcpy(data_mean + (size_t)b * d, h, (size_t)d * sizeof(double));
    if (pca_mean) memcpy(pca_mean + (size_t)b * d, h + D, (size_t)d * sizeof(double));
    if (evr) memcpy(evr + (size_t)b * d, h + 2 * D, (size_t)rcount * sizeof(double));
    if (comps) memcpy(comps + (size_t)b * d * d, h + 3 * D, (size_t)rcount * d * sizeof(double));
    c->k = c->hm->k; c->KP = round_up(c->k, 4); c->have_wpca = true;
    if (k) k[b] = c->k;
  }
  batch->wpca_uncollected = false;
  return PCABO_OK;
}

int pcabo_batch_acq_bounds(pcabo_batch* batch, double* bounds) {
  if (!batch || !bounds) return PCABO_ERR_ARG;
  if (batch->wpca_uncollected) { int rc = pcabo_batch_wpca_results(batch, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc != PCABO_OK) return rc; }
  if (batch->gp_pending) {
    BHIPCHK(hipSetDevice(batch->device));
    BHIPCHK(wait_event(batch->evBounds));
  } else if (!batch->have_gp) {
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_acq_bounds: no conditioning enqueued%s", "");
  }
  for (int b = 0; b < batch->B; ++b) {
    const pcabo_ctx* c = batch->ctx[b];
    double* o = bounds + (size_t)b * 2 * batch->max_d;
    for (int j = 0; j < c->k; ++j) { o[j] = c->hm->acq_lo[j]; o[c->k + j] = c->hm->acq_hi[j]; }
  }
  return PCABO_OK;
}

static int batch_max_k(const pcabo_batch* batch) {
  int km = 1;
  for (const pcabo_ctx* c : batch->ctx) km = std::max(km, c->k);
  return km;
}

static AcqBatch batch_ab(const pcabo_batch* batch, int table, int xq_host) {
  AcqBatch ab;
  ab.zs = batch->zs; ab.hzs = batch->hzs; ab.k_dev = batch->ctx[0]->dK; ab.bestf = batch->ctx[0]->dBestF;
  ab.table = table; ab.xq_host = xq_host;
  return ab;
}

// best_f of every run -> its device word (rounded per run like the single-context calls do)
static int batch_put_best_f(pcabo_batch* batch, const double* best_f) {
  pcabo_ctx* c0 = batch->ctx[0];
  for (int b = 0; b < batch->B; ++b) {
    pcabo_ctx* c = batch->ctx[b];
    c->hBestF[0] = c->bestf_f32 ? (double)(float)best_f[b] : best_f[b];
  }
  BHIPCHK(hipMemcpy2DAsync(c0->dBestF, batch->zs, c0->hBestF, batch->hzs, sizeof(double), batch->B, hipMemcpyHostToDevice, batch->stream));
  return PCABO_OK;
}

// The scoring of the raw samples in two halves: _begin packs the points, enqueues copy - scoring launch - copy back and
// returns; _end waits for the stream and finishes (jitter retries of single runs included).  pcabo_batch_busy tells a caller
// that drives several batches from one thread when _end (or any other waiting call) would return at once.
int pcabo_batch_busy(pcabo_batch* batch) {
  if (!batch) return PCABO_ERR_ARG;
  if (hipSetDevice(batch->device) != hipSuccess) return PCABO_ERR_HIP;
  const hipError_t e = hipStreamQuery(batch->stream);
  return e == hipErrorNotReady ? 1 : (e == hipSuccess ? 0 : PCABO_ERR_HIP);
}

static int batch_score_impl(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize,
                            int acq, double* val, int* status, int phase /* 0 both, 1 begin, 2 end */) {
  if (!batch) return PCABO_ERR_ARG;
  if (!Xq || !best_f || (!val && phase != 1) || q < 1 || q > batch->max_q || (acq != PCABO_ACQ_LOG_EI && acq != PCABO_ACQ_PI))
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_gp_condition_end_eval: bad argument%s", "");
  if (!batch->gp_pending) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_gp_condition_end_eval: no conditioning in flight%s", "");
  if (phase == 2 && !batch->score_enqueued) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_gp_condition_end_eval_end: no _begin before it%s", "");
  if (phase != 2 && batch->score_enqueued) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_gp_condition_end_eval: a scoring is already enqueued%s", "");
  if (batch->wpca_uncollected) { int rc = pcabo_batch_wpca_results(batch, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc != PCABO_OK) return rc; }
  BHIPCHK(hipSetDevice(batch->device));
  hipStream_t s = batch->stream;
  const int B = batch->B, kmax = batch_max_k(batch);
  pcabo_ctx* c0 = batch->ctx[0];
  if (phase != 2) {
  for (int b = 0; b < B; ++b)
    memcpy(batch->ctx[b]->hXq, Xq + (size_t)b * q * batch->max_d, (size_t)q * batch->ctx[b]->k * sizeof(double));
  BHIPCHK(hipMemcpy2DAsync(c0->dXq, batch->zs, c0->hXq, batch->hzs, (size_t)q * kmax * sizeof(double), B, hipMemcpyHostToDevice, s));
  int rc = batch_put_best_f(batch, best_f);
  if (rc != PCABO_OK) return rc;
  AcqParams p = make_params(c0, 0.0, maximize, acq, 0);
  p.inv_ls = 1.0 / batch->lengthscale; p.kernel = batch->kernel;
  if (score_gemm_possible(q))
    launch_score(s, c0->dXq, q, batch->n, kmax, batch->NP, c0->ld, c0->dZnT, c0->dR, c0->dAlpha, c0->dBounds4, c0->dYstats, p,
                 c0->dKS, c0->dPartial, c0->dVal, batch_ab(batch, 0, 0), B);
  else
    launch_acq(s, nullptr, c0->dXq, q, batch->n, kmax, batch->NP, c0->ld, c0->dZnT, c0->dR, c0->dAlpha, c0->dBounds4, c0->dYstats,
               p, c0->dPartial, c0->dCounters, c0->dVal, c0->dGrad, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
               batch_ab(batch, 0, 0), B, 0);
  BHIPCHK(hipMemcpy2DAsync(c0->hVal, batch->hzs, c0->dVal, batch->zs, (size_t)q * sizeof(double), B, hipMemcpyDeviceToHost, s));
  BHIPCHK(hipGetLastError());
  batch->score_enqueued = true;
  }
  if (phase == 1) return PCABO_OK;
  BHIPCHK(wait_stream(s));
  BHIPCHK(hipGetLastError());
  batch->score_enqueued = false;
  batch->gp_pending = false; batch->have_gp = true;
  int worst = PCABO_OK;
  for (int b = 0; b < B; ++b) {
    pcabo_ctx* c = batch->ctx[b];
    int st = PCABO_OK;
    if (c->hm->chol_info != 0) {
      // rare: this run's K needed jitter (psd_safe_cholesky: 1e-8, 1e-7, 1e-6) - redo it alone, then score again
      double jitter = 0.0;
      st = PCABO_ERR_NOT_PD;
      for (int attempt = 0; attempt < 3; ++attempt) {
        jitter = attempt == 0 ? 1e-8 : jitter * 10.0;
        int r2 = launch_factorisation(c, jitter);
        if (r2 != PCABO_OK) { st = r2; break; }
        if (wait_stream(c->stream) != hipSuccess) { st = PCABO_ERR_HIP; break; }
        if (c->hm->chol_info == 0) { st = PCABO_OK; break; }
      }
      if (st == PCABO_OK) {
        c->have_gp = true;
        if (batch->dev_lbfgsb) launch_rt_build(c->stream, c->dR, c->n, c->NP, c->ld, c->dGram);
        st = pcabo_acq_eval(c, Xq + (size_t)b * q * batch->max_d, q, best_f[b], maximize, acq, c->hVal, nullptr);
      } else {
        set_err(c, st, "K + s2 I not positive definite after jitter retries (pivot %s%d)", "", c->hm->chol_info);
      }
    }
    c->have_gp = st == PCABO_OK;
    if (st == PCABO_OK) memcpy(val + (size_t)b * q, c->hVal, (size_t)q * sizeof(double));
    if (status) status[b] = st;
    if (st != PCABO_OK) worst = st;
  }
  if (worst != PCABO_OK && !status) return bset_err(batch, worst, "a run of the batch failed in the conditioning (status array not given)%s", "");
  return PCABO_OK;
}

int pcabo_batch_gp_condition_end_eval(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize,
                                      int acq, double* val, int* status) {
  return batch_score_impl(batch, Xq, q, best_f, maximize, acq, val, status, 0);
}
int pcabo_batch_gp_condition_end_eval_begin(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize, int acq) {
  return batch_score_impl(batch, Xq, q, best_f, maximize, acq, nullptr, nullptr, 1);
}
int pcabo_batch_gp_condition_end_eval_end(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize,
                                          int acq, double* val, int* status) {
  return batch_score_impl(batch, Xq, q, best_f, maximize, acq, val, status, 2);
}

// pcabo_batch_optimize_acqf with PCABO_OPT_DEVICE_LBFGSB: every restart group's L-BFGS-B inside one launch of k_lbfgsb_group
// (value 1), or the host's L-BFGS-B over the same kernel's evaluation-only mode, one launch per round (value 2: the twin the
// device stepping is compared with).  Returns PCABO_OK / an error, or 1 when the call is not eligible (the caller then takes
// the host-paced path).
static int batch_optimize_device(pcabo_batch* batch, const double* ics, int num_restarts, int batch_limit, const double* bounds,
                                 int maxiter, int maximize, int acq, double* cand, double* vals, int* info, int* failed,
                                 int* status, int phase = 0 /* 0 whole call, 1 enqueue only, 2 wait + collect */) {
  const int B = batch->B, MD = batch->max_d, kmax = batch_max_k(batch);
  const int ngroups = (num_restarts + batch_limit - 1) / batch_limit;
  pcabo_ctx* c0 = batch->ctx[0];
  if (phase == 2) {
    if (!batch->opt_enqueued || batch->opt_restarts != num_restarts || batch->opt_limit != batch_limit)
      return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf_end: no matching _begin before it%s", "");
  } else if (batch->opt_enqueued) {
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf: an optimisation of this batch is already enqueued%s", "");
  } else if (!lbfgsb_device_possible(batch->NP, kmax, batch_limit) || num_restarts > PCABO_INLAUNCH_MAXQ || maxiter < 1 ||
      batch->max_q < 64 + 8 * ngroups || (size_t)(num_restarts + 2) * kmax > (size_t)batch->max_q * MD)
    return 1;
  std::vector<int> act;
  if (phase == 2) act = batch->opt_act;
  else
  for (int b = 0; b < B; ++b) {
    const pcabo_ctx* c = batch->ctx[b];
    if (!batch->active[b] || !c->have_gp) continue;
    const double* bd = bounds + (size_t)b * 2 * MD;
    for (int j = 0; j < c->k; ++j)
      if (!std::isfinite(bd[j]) || !std::isfinite(bd[c->k + j]) || bd[j] > bd[c->k + j]) return 1;
    act.push_back(b);
  }
  hipStream_t s = batch->stream;
  if (phase != 2)
  for (int b : act) {
    pcabo_ctx* c = batch->ctx[b];
    if (c->rt_stale) { launch_rt_build(s, c->dR, c->n, c->NP, c->ld, c->dGram); c->rt_stale = false; }
  }
  auto fill_status = [&](const std::vector<int>& run_status) {
    for (int b = 0; b < B; ++b) {
      const pcabo_ctx* c = batch->ctx[b];
      int st = run_status[b];
      if (!batch->active[b]) st = PCABO_ERR_ARG; else if (!c->have_gp) st = PCABO_ERR_NOT_PD;
      if (status) status[b] = st;
    }
  };
  const size_t xq_doubles = (size_t)(num_restarts + 2) * kmax;
  const double inv_ls = 1.0 / batch->lengthscale;
  // PCABO_OPT_LBFGSB_CUS: the optimiser's work-groups take a whole CU each for milliseconds; confined to a part of the chip they
  // leave the rest to the short kernels of the other batches of the process (conditioning, scoring)
  hipStream_t ks = s;
  if (batch->opt_cus > 0) {
    if (batch->optStream && batch->optStream_cus != batch->opt_cus) {
      (void)hipStreamSynchronize(batch->optStream); (void)hipStreamDestroy(batch->optStream); batch->optStream = nullptr;
    }
    if (!batch->optStream) {
      hipDeviceProp_t prop;
      BHIPCHK(hipGetDeviceProperties(&prop, batch->device));
      const int total = prop.multiProcessorCount, words = (total + 31) / 32;
      std::vector<uint32_t> mask((size_t)words, 0u);
      for (int i = 0; i < std::min(batch->opt_cus, total); ++i) mask[i >> 5] |= 1u << (i & 31);
      if (hipExtStreamCreateWithCUMask(&batch->optStream, (uint32_t)words, mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        batch->optStream = nullptr;
        return bset_err(batch, PCABO_ERR_HIP, "PCABO_OPT_LBFGSB_CUS: a stream with a CU mask could not be created (%s%d CUs)", "", batch->opt_cus);
      }
      batch->optStream_cus = batch->opt_cus;
      if (!batch->evOptIn) { BHIPCHK(hipEventCreateWithFlags(&batch->evOptIn, hipEventDisableTiming)); BHIPCHK(hipEventCreateWithFlags(&batch->evOptOut, hipEventDisableTiming)); }
    }
    ks = batch->optStream;
  }
  auto enqueue = [&](int nent, int mode) -> int {
    BHIPCHK(hipMemcpyAsync(batch->dOptTab, batch->hOptTab, (size_t)nent * sizeof(unsigned), hipMemcpyHostToDevice, s));
    BHIPCHK(hipMemcpy2DAsync(c0->dXq, batch->zs, c0->hXq, batch->hzs, xq_doubles * sizeof(double), B, hipMemcpyHostToDevice, s));
    if (ks != s) { BHIPCHK(hipEventRecord(batch->evOptIn, s)); BHIPCHK(hipStreamWaitEvent(ks, batch->evOptIn, 0)); }
    if (launch_lbfgsb_group(ks, batch->dOptTab, nent, mode, num_restarts, maxiter, batch->n, batch->NP, c0->ld, c0->dXq, c0->dZnT,
                            c0->dR, c0->dGram, c0->dAlpha, c0->dBounds4, c0->dYstats, c0->dBestF, c0->dK, inv_ls, maximize ? 1 : 0,
                            acq, batch->kernel, c0->dGrad, c0->dVal, batch->zs) != 0)
      return bset_err(batch, PCABO_ERR_HIP, "the device-resident optimiser could not be launched%s", "");
    if (ks != s) { BHIPCHK(hipEventRecord(batch->evOptOut, ks)); BHIPCHK(hipStreamWaitEvent(s, batch->evOptOut, 0)); }
    BHIPCHK(hipMemcpy2DAsync(c0->hVal, batch->hzs, c0->dVal, batch->zs, (size_t)(64 + 8 * ngroups) * sizeof(double), B, hipMemcpyDeviceToHost, s));
    BHIPCHK(hipMemcpy2DAsync(c0->hGrad, batch->hzs, c0->dGrad, batch->zs, (size_t)num_restarts * kmax * sizeof(double), B, hipMemcpyDeviceToHost, s));
    BHIPCHK(hipGetLastError());
    return PCABO_OK;
  };
  auto launch = [&](int nent, int mode) -> int {
    const int rc = enqueue(nent, mode);
    if (rc != PCABO_OK) return rc;
    BHIPCHK(wait_stream(s));
    BHIPCHK(hipGetLastError());
    return PCABO_OK;
  };
  std::vector<int> run_status(B, PCABO_OK);
  if (batch->dev_lbfgsb == 1) {
    // ---- everything on the device
    int nent = 0;
    if (phase != 2) {
    for (size_t blk = 0; blk < act.size(); blk += 8)          // both groups of a run on one XCD (work-groups go round the 8 XCDs)
      for (int gi = 0; gi < ngroups; ++gi)
        for (size_t r = blk; r < std::min(act.size(), blk + 8); ++r) {
          const int q0 = gi * batch_limit, nq = std::min(batch_limit, num_restarts - q0);
          batch->hOptTab[nent++] = ((unsigned)act[r] << 16) | ((unsigned)q0 << 8) | (unsigned)nq;
        }
    for (int b : act) {
      pcabo_ctx* c = batch->ctx[b];
      const int k = c->k;
      memcpy(c->hXq, ics + (size_t)b * num_restarts * MD, (size_t)num_restarts * k * sizeof(double));
      memcpy(c->hXq + (size_t)num_restarts * k, bounds + (size_t)b * 2 * MD, (size_t)2 * k * sizeof(double));
    }
    if (nent > 0) { const int rc = enqueue(nent, 1); if (rc != PCABO_OK) return rc; }
    batch->opt_act = act; batch->opt_restarts = num_restarts; batch->opt_limit = batch_limit; batch->opt_enqueued = true;
    }
    if (phase == 1) return PCABO_OK;
    batch->opt_enqueued = false;
    BHIPCHK(wait_stream(s));
    BHIPCHK(hipGetLastError());
    for (int b : act) {
      const pcabo_ctx* c = batch->ctx[b];
      const int k = c->k;
      int any_failed = 0;
      for (int gi = 0; gi < ngroups; ++gi) {
        const double* o = c->hVal + 64 + 8 * gi;
        if (info) { int* io = info + ((size_t)b * ngroups + gi) * 4; io[0] = (int)o[0]; io[1] = (int)o[1]; io[2] = (int)o[2]; io[3] = (int)o[3]; }
        if ((int)o[2] == 2) any_failed = 1;
        if ((int)o[4] != 0) run_status[b] = (int)o[4];
        if ((int)o[7] != 0) run_status[b] = PCABO_ERR_TIMEOUT;       // evaluation cap (cannot happen within maxiter / maxls)
      }
      if (run_status[b] == PCABO_ERR_NAN) set_err(batch->ctx[b], PCABO_ERR_NAN, "NaN in acquisition gradient%s", "");
      memcpy(cand + (size_t)b * num_restarts * MD, c->hGrad, (size_t)num_restarts * k * sizeof(double));
      memcpy(vals + (size_t)b * num_restarts, c->hVal, (size_t)num_restarts * sizeof(double));
      if (failed) failed[b] = any_failed;
    }
    for (int b = 0; b < B; ++b) if (failed && (!batch->active[b] || !batch->ctx[b]->have_gp)) failed[b] = 0;
    fill_status(run_status);
    return PCABO_OK;
  }
  if (phase != 0) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf_begin / _end need PCABO_OPT_DEVICE_LBFGSB = 1%s", "");
  // ---- the twin: host L-BFGS-B (csrc/lbfgsb.cpp), evaluations through mode 0 of the same kernel, one launch per round
  std::vector<std::vector<RestartGroup>> groups(B);
  for (int b : act) {
    groups[b].resize(ngroups);
    for (int gi = 0; gi < ngroups; ++gi) {
      const int q0 = gi * batch_limit, nq = std::min(batch_limit, num_restarts - q0);
      groups[b][gi].init(ics + (size_t)b * num_restarts * MD, bounds + (size_t)b * 2 * MD, q0, nq, batch->ctx[b]->k, maxiter);
      groups[b][gi].opt.set_sum_order(1);              // the device steps in the 64-lane tree order (lbfgsb.cpp): so does its twin
    }
  }
  struct Pending { int b, gi; };
  std::vector<Pending> pend;
  for (;;) {
    pend.clear();
    int nent = 0;
    for (int b : act) {
      if (run_status[b] != PCABO_OK) continue;
      pcabo_ctx* c = batch->ctx[b];
      for (int gi = 0; gi < ngroups; ++gi) {
        RestartGroup& rg = groups[b][gi];
        if (!rg.active) continue;
        rg.advance();
        if (!rg.active) continue;
        memcpy(c->hXq + (size_t)rg.q0 * c->k, rg.x.data(), (size_t)rg.nq * c->k * sizeof(double));
        batch->hOptTab[nent++] = ((unsigned)b << 16) | ((unsigned)rg.q0 << 8) | (unsigned)rg.nq;
        pend.push_back({b, gi});
      }
    }
    if (nent == 0) break;
    const int rc = launch(nent, 0);
    if (rc != PCABO_OK) return rc;
    for (const Pending& pe : pend) {
      pcabo_ctx* c = batch->ctx[pe.b];
      if (!groups[pe.b][pe.gi].absorb(c->hVal, c->hGrad)) {
        run_status[pe.b] = PCABO_ERR_NAN;
        set_err(c, PCABO_ERR_NAN, "NaN in acquisition gradient%s", "");
      }
    }
  }
  {
    int nent = 0;
    std::vector<Pending> redo;
    for (int b : act) {
      if (run_status[b] != PCABO_OK) continue;
      pcabo_ctx* c = batch->ctx[b];
      const int k = c->k;
      double* cb = cand + (size_t)b * num_restarts * MD;
      for (int gi = 0; gi < ngroups; ++gi) {
        RestartGroup& rg = groups[b][gi];
        for (int t = 0; t < rg.nq * k; ++t) {
          double v = rg.x[t];
          v = v < rg.lo[t] ? rg.lo[t] : (v > rg.hi[t] ? rg.hi[t] : v);
          cb[(size_t)rg.q0 * k + t] = v;
        }
        const bool reuse = rg.have_cache && (int)rg.vc.size() == rg.nq &&
                           memcmp(cb + (size_t)rg.q0 * k, rg.xc.data(), (size_t)rg.nq * k * sizeof(double)) == 0;
        if (reuse) { for (int j = 0; j < rg.nq; ++j) vals[(size_t)b * num_restarts + rg.q0 + j] = rg.vc[j]; }
        else {
          memcpy(c->hXq + (size_t)rg.q0 * k, cb + (size_t)rg.q0 * k, (size_t)rg.nq * k * sizeof(double));
          batch->hOptTab[nent++] = ((unsigned)b << 16) | ((unsigned)rg.q0 << 8) | (unsigned)rg.nq;
          redo.push_back({b, gi});
        }
      }
    }
    if (nent > 0) {
      const int rc = launch(nent, 0);
      if (rc != PCABO_OK) return rc;
      for (const Pending& pe : redo) {
        const RestartGroup& rg = groups[pe.b][pe.gi];
        for (int j = 0; j < rg.nq; ++j) vals[(size_t)pe.b * num_restarts + rg.q0 + j] = batch->ctx[pe.b]->hVal[rg.q0 + j];
      }
    }
  }
  for (int b = 0; b < B; ++b) {
    int any_failed = 0;
    for (int gi = 0; gi < (int)groups[b].size(); ++gi) {
      const RestartGroup& rg = groups[b][gi];
      const int wf = rg.opt.warnflag();
      if (info) { int* o = info + ((size_t)b * ngroups + gi) * 4; o[0] = rg.niter; o[1] = rg.nfev; o[2] = wf; o[3] = rg.opt.task(); }
      if (wf == 2) any_failed = 1;
    }
    if (failed) failed[b] = any_failed;
  }
  fill_status(run_status);
  return PCABO_OK;
}

int pcabo_batch_optimize_acqf(pcabo_batch* batch, const double* ics, int num_restarts, int batch_limit,
                              const double* bounds, int maxiter, const double* best_f, int maximize, int acq,
                              double* cand, double* vals, int* info, int* failed, int* status) {
  if (!batch) return PCABO_ERR_ARG;
  if (!ics || !bounds || !best_f || !cand || !vals || num_restarts < 1 || batch_limit < 1 ||
      num_restarts > PCABO_INLAUNCH_MAXQ || (acq != PCABO_ACQ_LOG_EI && acq != PCABO_ACQ_PI))
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf: bad argument (num_restarts <= 32)%s", "");
  if (!batch->have_gp) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf: no conditioned GP%s", "");
  BHIPCHK(hipSetDevice(batch->device));
  const int B = batch->B, MD = batch->max_d, kmax = batch_max_k(batch), G = batch->G;
  const int ngroups = (num_restarts + batch_limit - 1) / batch_limit;
  int rc = batch_put_best_f(batch, best_f);
  if (rc != PCABO_OK) return rc;
  if (batch->dev_lbfgsb) {
    const int drc = batch_optimize_device(batch, ics, num_restarts, batch_limit, bounds, maxiter, maximize, acq, cand, vals, info,
                                          failed, status);
    if (drc != 1) return drc;                       // 1: not eligible (size, bounds) - the host-paced path below
  }
  BHIPCHK(hipStreamSynchronize(batch->stream));
  pcabo_ctx* c0 = batch->ctx[0];
  AcqParams pg = make_params(c0, 0.0, maximize, acq, 1);
  pg.inv_ls = 1.0 / batch->lengthscale; pg.kernel = batch->kernel;
  AcqParams pv = pg; pv.want_grad = 0;
  std::vector<std::vector<RestartGroup>> groups(B);
  std::vector<int> run_status(B, PCABO_OK);
  for (int b = 0; b < B; ++b) {
    const pcabo_ctx* c = batch->ctx[b];
    if (!batch->active[b]) { groups[b].clear(); run_status[b] = PCABO_ERR_ARG; continue; }      // parked by the caller
    groups[b].resize(c->have_gp ? ngroups : 0);
    if (!c->have_gp) { run_status[b] = PCABO_ERR_NOT_PD; continue; }
    for (int gi = 0; gi < ngroups; ++gi) {
      const int q0 = gi * batch_limit, nq = std::min(batch_limit, num_restarts - q0);
      groups[b][gi].init(ics + (size_t)b * num_restarts * MD, bounds + (size_t)b * 2 * MD, q0, nq, c->k, maxiter);
    }
  }
  std::atomic<int> hip_failed{0};
  const int table_cap = PCABO_QA_MAX * 2;          // 32-bit entries in the QueryArgs slot
  if (((B + G - 1) / G) * num_restarts > table_cap)
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf: too many runs per worker thread for one launch table (%s%d entries)", "", table_cap);
  // One gang = the runs b with b % G == g, driven by worker g on its own stream: advance the gang's restart groups,
  // hand every pending (run, query) to ONE launch, wait for the per-query sequence words, feed the results back.
  auto gang = [&](int g) {
    if (hipSetDevice(batch->device) != hipSuccess) { hip_failed.store(1); return; }
    std::vector<int> mine;
    for (int b = g; b < B; b += G) if (run_status[b] == PCABO_OK) mine.push_back(b);
    struct Pending { int b, gi; };
    const bool use_group = batch->group_acq && batch_limit <= PCABO_GROUP_Q && acq_group_possible(batch->NP, kmax);
    hipStream_t st = batch->gstream[g];
    QueryArgs tab;                                   // the round's launch table (travels in the kernel arguments)
    unsigned* ent = reinterpret_cast<unsigned*>(tab.x);
    std::vector<Pending> pend;
    auto flags_ready = [&](int nent, unsigned long long seq) -> bool {
      for (int e = 0; e < nent; ++e) {
        const pcabo_ctx* c = batch->ctx[ent[e] >> 16];
        if (use_group) {
          const int q0 = (int)((ent[e] >> 8) & 0xffu), nqe = (int)(ent[e] & 0xffu);
          for (int j = 0; j < nqe; ++j) if (__atomic_load_n(&c->hm->qflag[q0 + j], __ATOMIC_ACQUIRE) != seq) return false;
        } else if (__atomic_load_n(&c->hm->qflag[ent[e] & 0xffffu], __ATOMIC_ACQUIRE) != seq) return false;
      }
      return true;
    };
    for (;;) {
      // step every active restart group of the gang to its next evaluation point
      pend.clear();
      int nent = 0;
      for (int b : mine) {
        if (run_status[b] != PCABO_OK) continue;
        pcabo_ctx* c = batch->ctx[b];
        for (int gi = 0; gi < ngroups; ++gi) {
          RestartGroup& rg = groups[b][gi];
          if (!rg.active) continue;
          rg.advance();
          if (!rg.active) continue;
          memcpy(c->hXq + (size_t)rg.q0 * c->k, rg.x.data(), (size_t)rg.nq * c->k * sizeof(double));
          if (use_group) ent[nent++] = ((unsigned)b << 16) | ((unsigned)rg.q0 << 8) | (unsigned)rg.nq;
          else for (int j = 0; j < rg.nq; ++j) ent[nent++] = ((unsigned)b << 16) | (unsigned)(rg.q0 + j);
          pend.push_back({b, gi});
        }
      }
      if (nent == 0) break;
      const unsigned long long seq = batch->seq.fetch_add(1) + 1;
      if (use_group) {
        if (launch_acq_group(st, &tab, nent, c0->hXq, batch->n, kmax, batch->NP, c0->ld, c0->dZnT, c0->dR, c0->dAlpha,
                             c0->dBounds4, c0->dYstats, pg, c0->dPartial, c0->dCounters + PCABO_GROUP_CNT_OFFSET, c0->dVal,
                             c0->dGrad, c0->hVal, c0->hGrad, c0->hm, seq, batch_ab(batch, 1, 1)) != 0) {
          hip_failed.store(1); return;                       // nothing was launched: no flags to wait for
        }
      } else {
        launch_acq(st, &tab, c0->hXq, PCABO_INLAUNCH_MAXQ, batch->n, kmax, batch->NP, c0->ld, c0->dZnT, c0->dR,
                   c0->dAlpha, c0->dBounds4, c0->dYstats, pg, c0->dPartial, c0->dCounters, c0->dVal, c0->dGrad, c0->hVal,
                   c0->hGrad, c0->hm, seq, nullptr, nullptr, batch_ab(batch, 1, 1), B, nent);
      }
      if (hipGetLastError() != hipSuccess) { hip_failed.store(1); return; }
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned long spins = 1; !flags_ready(nent, seq); ++spins) {
        if ((spins & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
          hip_failed.store(1); return;                       // the launch did not publish its results
        }
      }
      for (const Pending& pe : pend) {
        pcabo_ctx* c = batch->ctx[pe.b];
        if (!groups[pe.b][pe.gi].absorb(c->hVal, c->hGrad)) {
          run_status[pe.b] = PCABO_ERR_NAN;
          set_err(c, PCABO_ERR_NAN, "NaN in acquisition gradient%s", "");
        }
      }
    }
    auto launch_and_wait = [&](int nent, const AcqParams& p, unsigned long long seq) -> bool {
      launch_acq(st, &tab, c0->hXq, PCABO_INLAUNCH_MAXQ, batch->n, kmax, batch->NP, c0->ld, c0->dZnT, c0->dR, c0->dAlpha,
                 c0->dBounds4, c0->dYstats, p, c0->dPartial, c0->dCounters, c0->dVal, c0->dGrad, c0->hVal, c0->hGrad, c0->hm,
                 seq, nullptr, nullptr, batch_ab(batch, 1, 1), B, nent);
      if (hipGetLastError() != hipSuccess) return false;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned long spins = 0;
      for (int e = 0; e < nent; ++e) {
        const pcabo_ctx* c = batch->ctx[ent[e] >> 16];
        const int q = (int)(ent[e] & 0xffffu);
        while (__atomic_load_n(&c->hm->qflag[q], __ATOMIC_ACQUIRE) != seq) {
          if ((++spins & 0xFFFF) == 0 &&
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) return false;
        }
      }
      return true;
    };
    // end points, values there (botorch evaluates once more at the clamped end points; normally that is the last point
    // the run evaluated - same kernel arithmetic, same bits - otherwise one value-only launch for the gang's leftovers)
    int nent = 0;
    std::vector<int> redo;
    for (int b : mine) {
      if (run_status[b] != PCABO_OK) continue;
      pcabo_ctx* c = batch->ctx[b];
      const int k = c->k;
      double* cb = cand + (size_t)b * num_restarts * MD;
      bool reuse = true;
      for (int gi = 0; gi < ngroups; ++gi) {
        RestartGroup& rg = groups[b][gi];
        for (int t = 0; t < rg.nq * k; ++t) {
          double v = rg.x[t];
          v = v < rg.lo[t] ? rg.lo[t] : (v > rg.hi[t] ? rg.hi[t] : v);
          cb[(size_t)rg.q0 * k + t] = v;
        }
        reuse = reuse && rg.have_cache && (int)rg.vc.size() == rg.nq &&
                memcmp(cb + (size_t)rg.q0 * k, rg.xc.data(), (size_t)rg.nq * k * sizeof(double)) == 0;
      }
      if (reuse) {
        for (int gi = 0; gi < ngroups; ++gi)
          for (int j = 0; j < groups[b][gi].nq; ++j) vals[(size_t)b * num_restarts + groups[b][gi].q0 + j] = groups[b][gi].vc[j];
      } else {
        memcpy(c->hXq, cb, (size_t)num_restarts * k * sizeof(double));
        for (int j = 0; j < num_restarts; ++j) ent[nent++] = ((unsigned)b << 16) | (unsigned)j;
        redo.push_back(b);
      }
    }
    if (nent > 0) {
      const unsigned long long seq = batch->seq.fetch_add(1) + 1;
      if (!launch_and_wait(nent, pv, seq)) { hip_failed.store(1); return; }
      for (int b : redo)
        for (int j = 0; j < num_restarts; ++j) vals[(size_t)b * num_restarts + j] = batch->ctx[b]->hVal[j];
    }
  };
  batch->pool.run(gang);
  if (hip_failed.load()) {
    batch->cnt_dirty = true;
    (void)hipDeviceSynchronize();
    return bset_err(batch, PCABO_ERR_HIP, "an acquisition launch of the batch failed or did not answer%s", "");
  }
  for (int b = 0; b < B; ++b) {
    int any_failed = 0;
    for (int gi = 0; gi < (int)groups[b].size(); ++gi) {
      const RestartGroup& rg = groups[b][gi];
      const int wf = rg.opt.warnflag();
      if (info) {
        int* o = info + ((size_t)b * ngroups + gi) * 4;
        o[0] = rg.niter; o[1] = rg.nfev; o[2] = wf; o[3] = rg.opt.task();
      }
      if (wf == 2) any_failed = 1;
    }
    if (failed) failed[b] = any_failed;
    if (status) status[b] = run_status[b];
  }
  return PCABO_OK;
}

// pcabo_batch_optimize_acqf in two halves for callers that drive several batches from one thread (PCABO_OPT_DEVICE_LBFGSB = 1
// only: the whole optimisation is one launch, so _begin returns as soon as it is enqueued).  _begin returns PCABO_OK, an error,
// or 1 when the call does not qualify for the device-resident optimiser - the caller then uses pcabo_batch_optimize_acqf.
int pcabo_batch_optimize_acqf_begin(pcabo_batch* batch, const double* ics, int num_restarts, int batch_limit,
                                    const double* bounds, int maxiter, const double* best_f, int maximize, int acq) {
  if (!batch) return PCABO_ERR_ARG;
  if (!ics || !bounds || !best_f || num_restarts < 1 || batch_limit < 1 || num_restarts > PCABO_INLAUNCH_MAXQ ||
      (acq != PCABO_ACQ_LOG_EI && acq != PCABO_ACQ_PI))
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf_begin: bad argument (num_restarts <= 32)%s", "");
  if (!batch->have_gp) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_optimize_acqf_begin: no conditioned GP%s", "");
  if (batch->dev_lbfgsb != 1) return 1;
  BHIPCHK(hipSetDevice(batch->device));
  const int rc = batch_put_best_f(batch, best_f);
  if (rc != PCABO_OK) return rc;
  return batch_optimize_device(batch, ics, num_restarts, batch_limit, bounds, maxiter, maximize, acq, nullptr, nullptr, nullptr,
                               nullptr, nullptr, 1);
}
int pcabo_batch_optimize_acqf_end(pcabo_batch* batch, int num_restarts, int batch_limit, double* cand, double* vals, int* info,
                                  int* failed, int* status) {
  if (!batch || !cand || !vals) return PCABO_ERR_ARG;
  BHIPCHK(hipSetDevice(batch->device));
  return batch_optimize_device(batch, nullptr, num_restarts, batch_limit, nullptr, 1, 0, PCABO_ACQ_LOG_EI, cand, vals, info, failed,
                               status, 2);
}

// Value and gradient of the acquisition at q <= 32 points per run through the evaluation of the device-resident optimiser
// (mode 0 of k_lbfgsb_group; PCABO_OPT_DEVICE_LBFGSB must be on so that the transposed root inverse exists).  Xq[B][q * max_d]
// (run b's points packed with stride k_b), val[B][q], grad[B][q * max_d] (same packing).  Diagnostics / parity tests.
int pcabo_batch_device_acq_eval(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize, int acq,
                                double* val, double* grad) {
  if (!batch) return PCABO_ERR_ARG;
  if (!Xq || !best_f || !val || !grad || q < 1 || q > PCABO_INLAUNCH_MAXQ || (acq != PCABO_ACQ_LOG_EI && acq != PCABO_ACQ_PI))
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_device_acq_eval: bad argument (q <= 32)%s", "");
  if (!batch->have_gp || !batch->dev_lbfgsb)
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_device_acq_eval: needs a conditioned GP and PCABO_OPT_DEVICE_LBFGSB%s", "");
  // the pinned table / point / value / gradient buffers below belong to an enqueued call until its _end has collected them
  if (batch->opt_enqueued || batch->score_enqueued || batch->imap_enqueued)
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_device_acq_eval: a _begin call of this batch has not been ended%s", "");
  BHIPCHK(hipSetDevice(batch->device));
  const int B = batch->B, MD = batch->max_d, kmax = batch_max_k(batch);
  if (!lbfgsb_device_possible(batch->NP, kmax, PCABO_GROUP_Q) || batch->max_q < 64)
    return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_device_acq_eval: shape beyond the device optimiser (n <= 512, k <= 40)%s", "");
  int rc = batch_put_best_f(batch, best_f);
  if (rc != PCABO_OK) return rc;
  pcabo_ctx* c0 = batch->ctx[0];
  hipStream_t s = batch->stream;
  int nent = 0;
  for (int b = 0; b < B; ++b) {
    pcabo_ctx* c = batch->ctx[b];
    if (!c->have_gp) continue;
    if (c->rt_stale) { launch_rt_build(s, c->dR, c->n, c->NP, c->ld, c->dGram); c->rt_stale = false; }
    memcpy(c->hXq, Xq + (size_t)b * q * MD, (size_t)q * c->k * sizeof(double));
    for (int q0 = 0; q0 < q; q0 += PCABO_GROUP_Q)
      batch->hOptTab[nent++] = ((unsigned)b << 16) | ((unsigned)q0 << 8) | (unsigned)std::min(PCABO_GROUP_Q, q - q0);
  }
  if (nent == 0) return PCABO_OK;
  BHIPCHK(hipMemcpyAsync(batch->dOptTab, batch->hOptTab, (size_t)nent * sizeof(unsigned), hipMemcpyHostToDevice, s));
  BHIPCHK(hipMemcpy2DAsync(c0->dXq, batch->zs, c0->hXq, batch->hzs, (size_t)(q + 2) * kmax * sizeof(double), B, hipMemcpyHostToDevice, s));
  if (launch_lbfgsb_group(s, batch->dOptTab, nent, 0, q, 1, batch->n, batch->NP, c0->ld, c0->dXq, c0->dZnT, c0->dR, c0->dGram,
                          c0->dAlpha, c0->dBounds4, c0->dYstats, c0->dBestF, c0->dK, 1.0 / batch->lengthscale, maximize ? 1 : 0, acq,
                          batch->kernel, c0->dGrad, c0->dVal, batch->zs) != 0)
    return bset_err(batch, PCABO_ERR_HIP, "the device-resident optimiser could not be launched%s", "");
  BHIPCHK(hipMemcpy2DAsync(c0->hVal, batch->hzs, c0->dVal, batch->zs, (size_t)q * sizeof(double), B, hipMemcpyDeviceToHost, s));
  BHIPCHK(hipMemcpy2DAsync(c0->hGrad, batch->hzs, c0->dGrad, batch->zs, (size_t)q * kmax * sizeof(double), B, hipMemcpyDeviceToHost, s));
  BHIPCHK(wait_stream(s));
  BHIPCHK(hipGetLastError());
  for (int b = 0; b < B; ++b) {
    const pcabo_ctx* c = batch->ctx[b];
    if (!c->have_gp) continue;
    memcpy(val + (size_t)b * q, c->hVal, (size_t)q * sizeof(double));
    memcpy(grad + (size_t)b * q * MD, c->hGrad, (size_t)q * c->k * sizeof(double));
  }
  return PCABO_OK;
}

static int batch_inverse_map_impl(pcabo_batch* batch, const double* z, double* x, int phase /* 0 whole call, 1 enqueue, 2 wait + collect */) {
  if (!batch || (!z && phase != 2) || (!x && phase != 1)) return PCABO_ERR_ARG;
  if (batch->n == 0 || batch->wpca_uncollected) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_inverse_map: no weighted PCA collected%s", "");
  if (phase == 2 && !batch->imap_enqueued) return bset_err(batch, PCABO_ERR_ARG, "pcabo_batch_inverse_map_end: no _begin before it%s", "");
  BHIPCHK(hipSetDevice(batch->device));
  hipStream_t s = batch->stream;
  pcabo_ctx* c0 = batch->ctx[0];
  const int B = batch->B, MD = batch->max_d, d = batch->d;
  if (phase != 2) {
    for (int b = 0; b < B; ++b) memcpy(batch->ctx[b]->hXq, z + (size_t)b * MD, (size_t)batch->ctx[b]->k * sizeof(double));
    BHIPCHK(hipMemcpy2DAsync(c0->dZq, batch->zs, c0->hXq, batch->hzs, (size_t)MD * sizeof(double), B, hipMemcpyHostToDevice, s));
    launch_inverse_map(s, c0->dZq, c0->dComps, c0->dDataMean, c0->dPcaMean, 0, d, c0->dXout, c0->dK, batch_zb(batch));
    BHIPCHK(hipMemcpy2DAsync(c0->hSmall, batch->hzs, c0->dXout, batch->zs, (size_t)d * sizeof(double), B, hipMemcpyDeviceToHost, s));
    BHIPCHK(hipGetLastError());
    batch->imap_enqueued = true;
  }
  if (phase == 1) return PCABO_OK;
  batch->imap_enqueued = false;
  BHIPCHK(wait_stream(s));
  BHIPCHK(hipGetLastError());
  for (int b = 0; b < B; ++b) memcpy(x + (size_t)b * d, batch->ctx[b]->hSmall, (size_t)d * sizeof(double));
  return PCABO_OK;
}
int pcabo_batch_inverse_map(pcabo_batch* batch, const double* z, double* x) { return batch_inverse_map_impl(batch, z, x, 0); }
int pcabo_batch_inverse_map_begin(pcabo_batch* batch, const double* z) { return batch_inverse_map_impl(batch, z, nullptr, 1); }
int pcabo_batch_inverse_map_end(pcabo_batch* batch, double* x) { return batch_inverse_map_impl(batch, nullptr, x, 2); }

}  // extern "C"

// =====================================================================================================================
// Final gather of best-so-far values over RCCL (declared in include/pcabo.h).
// The data path has no collective: runs are independent and every rank drives its own GPU.  The ONE exchange of the design -
// the all-gather of a few doubles after the runs - is available here for callers without torch.distributed.  librccl.so is
// opened on first use (no link-time dependency: a process that never gathers never loads it).
// =====================================================================================================================
#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } rccl_unique_id;            // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(void**, int, rccl_unique_id, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (!a.lib) return a;
    a.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(a.lib, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void**, int, rccl_unique_id, int))dlsym(a.lib, "ncclCommInitRank");
    a.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(a.lib, "ncclAllGather");
    a.CommDestroy = (int (*)(void*))dlsym(a.lib, "ncclCommDestroy");
    a.GetErrorString = (const char* (*)(int))dlsym(a.lib, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy;
    return a;
  }();
  return api;
}
}  // namespace

struct pcabo_comm {
  void* comm = nullptr;
  int world = 0, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  double* dbuf = nullptr;
  size_t cap = 0;                       // doubles per rank the device buffer holds
  char err[256] = {0};
};
static int cset_err(pcabo_comm* c, int code, const char* what, int rc) {
  if (c) {
    const RcclApi& a = rccl();
    snprintf(c->err, sizeof(c->err), "%s (%d%s%s)", what, rc, a.GetErrorString ? ": " : "", a.GetErrorString ? a.GetErrorString(rc) : "");
  }
  return code;
}

extern "C" {

int pcabo_comm_unique_id(char* id128) {
  if (!id128) return PCABO_ERR_ARG;
  RcclApi& a = rccl();
  if (!a.ok) return PCABO_ERR_HIP;
  rccl_unique_id id;
  if (a.GetUniqueId(&id) != 0) return PCABO_ERR_HIP;
  memcpy(id128, id.internal, 128);
  return PCABO_OK;
}

int pcabo_comm_create(const char* id128, int world, int rank, int device, pcabo_comm** out) {
  if (!out) return PCABO_ERR_ARG;
  *out = nullptr;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return PCABO_ERR_ARG;
  RcclApi& a = rccl();
  if (!a.ok) return PCABO_ERR_HIP;
  pcabo_comm* c = new (std::nothrow) pcabo_comm();
  if (!c) return PCABO_ERR_HIP;
  *out = c;                             // handed back even on failure (message in pcabo_comm_last_error)
  c->world = world; c->rank = rank; c->device = device;
  if (hipSetDevice(device) != hipSuccess) return cset_err(c, PCABO_ERR_HIP, "hipSetDevice failed", (int)hipGetLastError());
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return cset_err(c, PCABO_ERR_HIP, "stream", (int)hipGetLastError());
  rccl_unique_id id;
  memcpy(id.internal, id128, 128);
  const int rc = a.CommInitRank(&c->comm, world, id, rank);
  if (rc != 0) { c->comm = nullptr; return cset_err(c, PCABO_ERR_HIP, "ncclCommInitRank failed", rc); }
  return PCABO_OK;
}

int pcabo_gather_best(pcabo_comm* c, const double* local, int n_local, double* all) {
  if (!c || !c->comm || !local || !all || n_local < 1) return PCABO_ERR_ARG;
  RcclApi& a = rccl();
  if (hipSetDevice(c->device) != hipSuccess) return cset_err(c, PCABO_ERR_HIP, "hipSetDevice failed", (int)hipGetLastError());
  if ((size_t)n_local > c->cap) {
    if (c->dbuf) (void)hipFree(c->dbuf);
    c->dbuf = nullptr; c->cap = 0;
    if (hipMalloc((void**)&c->dbuf, (size_t)n_local * (c->world + 1) * sizeof(double)) != hipSuccess)
      return cset_err(c, PCABO_ERR_HIP, "hipMalloc failed", (int)hipGetLastError());
    c->cap = (size_t)n_local;
  }
  double* send = c->dbuf;
  double* recv = c->dbuf + c->cap;
  if (hipMemcpyAsync(send, local, (size_t)n_local * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
    return cset_err(c, PCABO_ERR_HIP, "copy to the device failed", (int)hipGetLastError());
  const int rc = a.AllGather(send, recv, (size_t)n_local, /* ncclFloat64 */ 8, c->comm, c->stream);
  if (rc != 0) return cset_err(c, PCABO_ERR_HIP, "ncclAllGather failed", rc);
  if (hipMemcpyAsync(all, recv, (size_t)n_local * c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess)
    return cset_err(c, PCABO_ERR_HIP, "copy from the device failed", (int)hipGetLastError());
  return PCABO_OK;
}

int pcabo_comm_last_error(pcabo_comm* c, char* buf, int buflen) {
  if (!c || !buf || buflen <= 0) return PCABO_ERR_ARG;
  snprintf(buf, buflen, "%s", c->err);
  return PCABO_OK;
}

int pcabo_comm_destroy(pcabo_comm* c) {
  if (!c) return PCABO_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->dbuf) (void)hipFree(c->dbuf);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return PCABO_OK;
}

}  // extern "C"
