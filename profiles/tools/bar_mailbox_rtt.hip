// Can the HOST write the resident kernel's mailbox straight into device memory (through the PCIe BAR) instead of into
// pinned host memory that a relay work-group has to fetch over PCIe?  Variant of server_rtt.hip without the relay: every
// group polls the device mailbox, which the host fills with 16-byte stores + sfence.
//   mode 0: mailbox in pinned host memory, relay group (reference: server_rtt.hip)   [not built here]
//   mode 1: hipExtMallocWithFlags(hipDeviceMallocFinegrained), host writes it directly
//   mode 2: hipExtMallocWithFlags(hipDeviceMallocUncached),    host writes it directly
//   mode 3: plain hipMalloc, host writes it directly (coarse-grained: the L2 may serve stale lines)
// A SIGSEGV/SIGBUS handler reports "host cannot address this memory" instead of dying silently.
// Build: hipcc --offload-arch=gfx950 -O2 -o bar_mailbox_rtt bar_mailbox_rtt.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <emmintrin.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
struct Pair { double v; unsigned long long tag; };
__device__ inline u4 ld_sys(const void* p) {
  u4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ inline unsigned long long tag_of(u4 v) { return ((unsigned long long)v.w << 32) | v.z; }
#define TIMEOUT_TICKS 2000000ull       // 20 ms at 100 MHz
__global__ __launch_bounds__(256) void k_server(const Pair* box, int npairs, int per_group, unsigned long long base,
                                                unsigned int* counter, volatile unsigned long long* done, double* sink) {
  __shared__ int s_state;
  const int tid = threadIdx.x, g = blockIdx.x;
  double acc = 0.0;
  for (unsigned long long round = 1;; ++round) {
    const unsigned long long seq = base + round;
    const unsigned long long t0 = wall_clock64();
    if (tid == 0) s_state = 0;
    __syncthreads();
    if (tid < 64) {
      bool got = false;
      while (!got) {
        bool ok = true; u4 h = ld_sys(box); ok = tag_of(h) == seq;
        int p = 1 + (g * per_group + tid) % (npairs - 1);
        u4 x = {0, 0, 0, 0};
        if (tid < per_group) { x = ld_sys(box + p); ok = ok && tag_of(x) == seq; }
        if (__all(ok)) { got = true; if (__builtin_bit_cast(double, ((unsigned long long)h.y << 32) | h.x) == 0.0) { if (tid == 0) s_state = 1; }
                         acc += __builtin_bit_cast(double, ((unsigned long long)x.y << 32) | x.x); }
        else if (__any(wall_clock64() - t0 > TIMEOUT_TICKS)) { if (tid == 0) s_state = 1; got = true; }
        else __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_state) break;
    if (tid == 0) {
      unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((t % gridDim.x) == gridDim.x - 1)
        __hip_atomic_store((unsigned long long*)done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (sink && acc == 123.456) sink[0] = acc;
}
static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }
static inline void put_pair(Pair* dst, double v, unsigned long long tag) {
  __m128i x = _mm_set_epi64x((long long)tag, (long long)__builtin_bit_cast(unsigned long long, v));
  _mm_store_si128((__m128i*)dst, x);                 // one 16-byte store: value and tag travel together
}
static int run(int mode) {
  const int npairs = 1 + 170, per_group = 17, groups = 160, rounds = 5000;
  Pair* box = nullptr; unsigned long long* h_done; unsigned int* counter;
  hipError_t e;
  if (mode == 1) e = hipExtMallocWithFlags((void**)&box, sizeof(Pair) * 512, hipDeviceMallocFinegrained);
  else if (mode == 2) e = hipExtMallocWithFlags((void**)&box, sizeof(Pair) * 512, hipDeviceMallocUncached);
  else e = hipMalloc((void**)&box, sizeof(Pair) * 512);
  if (e != hipSuccess) { printf("mode %d: allocation failed: %s\n", mode, hipGetErrorString(e)); return 1; }
  hipMemset(box, 0, sizeof(Pair) * 512);
  hipHostMalloc(&h_done, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipMalloc(&counter, 4); hipMemset(counter, 0, 4);
  *h_done = 0;
  hipDeviceSynchronize();
  signal(SIGSEGV, on_fault); signal(SIGBUS, on_fault);
  if (sigsetjmp(g_jmp, 1)) { printf("mode %d: the host cannot address this memory (fault on the first store)\n", mode); return 1; }
  put_pair(box + 511, 1.0, 7); _mm_sfence();           // probe
  signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const unsigned long long base = 1000;
  hipLaunchKernelGGL(k_server, dim3(groups), dim3(256), 0, s, box, npairs, per_group, base, counter, h_done, (double*)nullptr);
  double acc = 0, wr = 0; int bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    double a = now();
    while (now() - a < 6e-6) {}
    const unsigned long long seq = base + r;
    double b = now();
    for (int p = 1; p < npairs; ++p) put_pair(box + p, r * 0.5 + p, seq);
    put_pair(box, 1.0, seq);
    _mm_sfence();
    wr += now() - b;
    while (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) != seq) { if (now() - b > 0.05) { ++bad; break; } }
    acc += now() - b;
    if (bad) break;
  }
  for (int p = 1; p < npairs; ++p) put_pair(box + p, 0.0, base + (bad ? 0 : rounds) + 1);
  put_pair(box, 0.0, base + (bad ? 0 : rounds) + 1); _mm_sfence();
  e = hipStreamSynchronize(s);
  printf("mode %d: host writes the device mailbox directly, %d groups, %d pairs: round trip %.2f us (host stores %.2f us) per round "
         "(%d rounds, gave up %d, sync: %s)\n", mode, groups, npairs, 1e6 * acc / (bad ? 1 : rounds), 1e6 * wr / (bad ? 1 : rounds),
         rounds, bad, hipGetErrorString(e));
  return 0;
}
int main(int argc, char** argv) {
  int large_bar = -1;
  hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
  printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
  int mode = argc > 1 ? atoi(argv[1]) : 1;
  return run(mode);
}
