// Round-trip latency of "launch a kernel, kernel writes a flag to pinned host memory, host spins on it" on MI355X:
// the floor under every L-BFGS-B round of the acquisition optimiser (the kernel itself adds its own time).
// Build: hipcc --offload-arch=gfx950 -O2 -o launch_rtt launch_rtt.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { double x[384]; };
__global__ void k_small(volatile unsigned long long* flag, unsigned long long seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((unsigned long long*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_big(Big b, volatile unsigned long long* flag, unsigned long long seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((unsigned long long*)flag, seq + (b.x[5] > 1e300), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
  unsigned long long* flag;
  hipHostMalloc(&flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
  *flag = 0;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  Big b = {};
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int mode = 0; mode < 4; ++mode) {
    const int N = 20000; unsigned long long seq = *flag; double tl = 0;
    double t0 = now();
    for (int i = 0; i < N; ++i) {
      ++seq;
      double a = now();
      if (mode == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, flag, seq);
      if (mode == 1) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b, flag, seq);
      if (mode == 2) hipLaunchKernelGGL(k_big, dim3(160), dim3(256), 40000, s, b, flag, seq);
      if (mode == 3) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, flag, seq); }
      tl += now() - a;
      if (mode == 3) hipStreamSynchronize(s);
      else while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
    }
    double dt = now() - t0;
    const char* names[] = {"1 group, 16 B args, host spins on flag", "1 group, 3 KB args, host spins on flag",
                           "160 groups x 256 thr, 40 KB LDS, 3 KB args, spin", "1 group, hipStreamSynchronize"};
    printf("%-52s round trip %.2f us (launch call itself %.2f us)\n", names[mode], 1e6 * dt / N, 1e6 * tl / N);
  }
  return 0;
}
