// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 with THIS library's access pattern:
// coalesced 8-byte-per-lane loads (doubles).  Reads (and in a second kernel writes) a known byte count that is
// larger than the 256 MiB Infinity Cache, so the counters can be compared with the truth.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read8(const double* __restrict__ p, size_t n, double* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  double s = 0.0;
  for (; i < n; i += stride) s += p[i];
  if (s == 123.456) *out = s;
}
__global__ void k_write8(double* __restrict__ p, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (double)i;
}
int main() {
  const size_t n = (size_t)1 << 27;   // 2^27 doubles = 1 GiB
  double *p, *o;
  hipMalloc(&p, n * sizeof(double)); hipMalloc(&o, 8);
  hipMemset(p, 0, n * sizeof(double));
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_write8, dim3(2048), dim3(256), 0, 0, p, n);
  hipLaunchKernelGGL(k_read8, dim3(2048), dim3(256), 0, 0, p, n, o);
  hipDeviceSynchronize();
  printf("calibration: each kernel moved %zu bytes\n", n * sizeof(double));
  return 0;
}
