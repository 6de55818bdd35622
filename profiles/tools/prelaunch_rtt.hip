// Handshake latency of a PRE-LAUNCHED kernel that waits for its input: the host launches the kernel, does other
// work (here: spins 8 us), then publishes "go"; every work-group polls it (thread 0, bounded), the last group
// to finish publishes "done" to pinned host memory.  Measured: go-written -> done-seen on the host.
//   mode 0: go in pinned host memory (every poll crosses PCIe)
//   mode 1: go in fine-grained device memory written by the host through the BAR (if the allocation is
//           host-accessible on this system), groups poll device memory
// Build: hipcc --offload-arch=gfx950 -O2 -o prelaunch_rtt prelaunch_rtt.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void k_wait(const volatile unsigned long long* go, unsigned long long seq, unsigned int* counter,
                       volatile unsigned long long* done, const double* xin, double* sink) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    int good = 0;
    while (true) {
      if (__hip_atomic_load((const unsigned long long*)go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == seq) { good = 1; break; }
      if (wall_clock64() - t0 > 100000000ull / 20) break;          // 50 ms at 100 MHz: give up
      __builtin_amdgcn_s_sleep(2);
    }
    ok = good;
  }
  __syncthreads();
  double v = 0.0;
  if (ok && threadIdx.x < 40) v = xin[threadIdx.x];                  // the query point, read after the flag
  if (sink && v == 123.456) sink[0] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((t % gridDim.x) == gridDim.x - 1)
      __hip_atomic_store((unsigned long long*)done, ok ? seq : ~0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main() {
  unsigned long long *h_go, *h_done; double* h_x;
  hipHostMalloc(&h_go, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostMalloc(&h_done, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostMalloc(&h_x, 4096, hipHostMallocMapped | hipHostMallocCoherent);
  unsigned long long* d_go = nullptr; double* d_x = nullptr;
  hipError_t e1 = hipExtMallocWithFlags((void**)&d_go, 4096, hipDeviceMallocFinegrained);
  hipError_t e2 = hipExtMallocWithFlags((void**)&d_x, 4096, hipDeviceMallocFinegrained);
  printf("fine-grained device allocation: %s %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  unsigned int* counter; hipMalloc(&counter, 4); hipMemset(counter, 0, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  *h_go = 0; *h_done = 0;
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 1) {
      if (e1 != hipSuccess) break;
      hipPointerAttribute_t at; hipPointerGetAttributes(&at, d_go);
      printf("mode 1: host pointer of the device allocation = %p\n", at.hostPointer);
      hipMemset(d_go, 0, 4096); hipDeviceSynchronize();
    }
    for (int groups : {1, 160, 290}) {
      const int N = 3000; double acc = 0, accl = 0; unsigned long long seq = 1000000ull * (mode + 1) + groups * 10000ull;
      int bad = 0;
      for (int i = 0; i < N; ++i) {
        ++seq;
        const unsigned long long* go = mode == 0 ? h_go : d_go;
        const double* x = mode == 0 ? h_x : d_x;
        double a = now();
        hipLaunchKernelGGL(k_wait, dim3(groups), dim3(256), 0, s, go, seq, counter, h_done, x, (double*)nullptr);
        accl += now() - a;
        while (now() - a < 8e-6) {}                      // the host's own work (L-BFGS-B step)
        double b = now();
        if (mode == 0) { h_x[0] = (double)i; __atomic_store_n(h_go, seq, __ATOMIC_RELEASE); }
        else {
          volatile double* xd = (volatile double*)d_x; xd[0] = (double)i;       // host stores through the BAR
          __atomic_store_n((unsigned long long*)d_go, seq, __ATOMIC_RELEASE);
        }
        unsigned long long dv;
        while ((dv = __atomic_load_n(h_done, __ATOMIC_ACQUIRE)) != seq) { if (dv == ~0ull) { ++bad; break; } if (now() - b > 0.2) { ++bad; break; } }
        acc += now() - b;
        hipStreamSynchronize(s);
      }
      printf("mode %d, %3d groups: go->done %.2f us (launch call %.2f us, gave up %d times)\n", mode, groups, 1e6 * acc / N, 1e6 * accl / N, bad);
      if (bad > N / 2) break;
    }
  }
  return 0;
}
