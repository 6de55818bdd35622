// Stand-alone model of the launch pattern behind the abort recorded in gpurun_out/bclock30_prof.log (round 2): worker
// threads, each with its own stream, launching a kernel whose EXPLICIT arguments are 3840 bytes (a 3584-byte block by value
// + 256 bytes of pointers and scalars, as k_acq_fast had; with the 256 hidden bytes of code object v5 the kernarg segment is
// exactly HIP's 4 KB limit), while rocprofv3 --kernel-trace intercepts the dispatches.  Nothing of libpcabo is linked.
//   hipcc --offload-arch=gfx950 -O2 -pthread -o kernarg_threads kernarg_threads.hip
//   ./kernarg_threads [threads=8] [launches per thread=20000] [block doubles=448]
//   rocprofv3 --kernel-trace --stats -d out -- ./kernarg_threads
// Prints one line per block size; a crash under the profiler with 448 doubles and none with 64 points at the argument size.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

template <int N> struct Block { double x[N]; };
struct Rest { const void* p[22]; long long s[8]; };      // 240 bytes

template <int N>
__global__ void k_args(Block<N> b, Rest r, unsigned long long* out, unsigned long long seq) {
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockDim.x == 64) {      // (grid / block size: the hidden arguments are in use)
    double s = 0.0;
    for (int i = 0; i < N; i += 61) s += b.x[i];
    out[0] = seq + (unsigned long long)(s > 1e300) + (unsigned long long)(r.s[3] == 77);
  }
}

template <int N>
static int run(int T, int L) {
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      if (hipSetDevice(0) != hipSuccess) { bad++; return; }
      hipStream_t s;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { bad++; return; }
      unsigned long long* out = nullptr;
      if (hipMalloc((void**)&out, 64) != hipSuccess) { bad++; return; }
      // the block lives on the heap of the worker thread (as the launch table of a gang did) and is passed by value
      std::vector<Block<N>> heap(3);
      Rest r = {};
      for (int i = 0; i < L; ++i) {
        heap[i % 3].x[i % N] = (double)i;
        hipLaunchKernelGGL(k_args<N>, dim3(2), dim3(64), 0, s, heap[i % 3], r, out, (unsigned long long)i);
        if ((i & 1023) == 1023 && hipStreamSynchronize(s) != hipSuccess) { bad++; return; }
      }
      if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) bad++;
      (void)hipFree(out);
      (void)hipStreamDestroy(s);
    });
  for (auto& x : th) x.join();
  printf("block of %4d doubles (%4zu explicit argument bytes): %d threads x %d launches, failures %d\n", N,
         sizeof(Block<N>) + sizeof(Rest) + 16, T, L, bad.load());
  fflush(stdout);
  return bad.load();
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8, L = argc > 2 ? atoi(argv[2]) : 20000;
  const int N = argc > 3 ? atoi(argv[3]) : 0;
  int bad = 0;
  if (N == 0 || N == 64) bad += run<64>(T, L);
  if (N == 0 || N == 256) bad += run<256>(T, L);
  if (N == 0 || N == 400) bad += run<400>(T, L);
  if (N == 0 || N == 448) bad += run<448>(T, L);
  return bad ? 1 : 0;
}
