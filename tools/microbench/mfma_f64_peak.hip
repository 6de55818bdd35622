// What the f64 matrix cores of one MI355X deliver in practice (diagnostic, not part of the product):
//  mode 0: back-to-back v_mfma_f64_16x16x4 on register operands, 4 independent accumulators per wave (no memory at all);
//  mode 1: the look-back's inner loop - operands from LDS (ds_read_b64, leading dimension 34), 4 accumulators, no barriers, no
//          global loads;
//  mode 2: mode 1 with the look-back's two barriers per 32 MFMAs.
// usage: mfma_f64_peak <work-groups per CU> <iterations>
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define TLH 34
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  __shared__ double s_a[64 * TLH], s_b[64 * TLH];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  for (int i = tid; i < 64 * TLH; i += 256) { s_a[i] = 1e-3 * (i % 7); s_b[i] = 1e-3 * (i % 5); }
  __syncthreads();
  double4_t acc[4];
  for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = 1e-3 * l, b0 = 1e-3 * w, b1 = 2e-3, b2 = 3e-3, b3 = 4e-3;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) __syncthreads();
    if (MODE == 2) __syncthreads();
    for (int kk = 0; kk < 32; kk += 4) {
      if (MODE >= 1) {
        a = s_a[(16 * w + (l & 15)) * TLH + kk + (l >> 4)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double bb = s_b[(16 * q + (l & 15)) * TLH + kk + (l >> 4)];
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[q], 0, 0, 0);
        }
      } else {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[3], 0, 0, 0);
      }
    }
  }
  double s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) s += acc[q][r];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}
template <int MODE>
static void run(int wg_per_cu, int iters, double* d) {
  const int groups = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(groups), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(groups), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)groups * 4 /*waves*/ * iters * 32.0 * 2048.0;
  std::printf("mode %d  %d work-groups per CU  %.3f ms  %.1f TFLOP/s  (%.1f %% of 78.6)\n", MODE, wg_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
}
int main(int argc, char** argv) {
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  double* d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
  for (int wg : {1, 2, 3}) {
    if (argc > 1 && atoi(argv[1]) != wg) continue;
    run<0>(wg, iters, d); run<1>(wg, iters, d); run<2>(wg, iters, d);
  }
  hipFree(d);
  return 0;
}
