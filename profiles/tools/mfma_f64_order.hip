// Is v_mfma_f64_16x16x4 a chain of four fused multiply-adds in ascending k on top of the accumulator?
// (If yes, a VALU loop acc = fma(a[k], b[k], acc) and the matrix-core instruction give the same bits, and a kernel may
// move its rank-k updates to the matrix cores without changing its results.)   hipcc --offload-arch=gfx950 -O2 mfma_f64_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D_mfma) {
  const int l = threadIdx.x;
  // A[16][4] (m, k), B[4][16] (k, n), C/D[16][16]: a = A[l & 15][l >> 4], b = B[l >> 4][l & 15], acc[r] = C[(l >> 4) + 4 r][l & 15]
  double4_t acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D_mfma[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
int main() {
  double hA[64], hB[64], hC[256], hD[256];
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
  long same_asc = 0, same_desc = 0, same_pair = 0, same_exact = 0, total = 0;
  srand(1);
  for (int trial = 0; trial < 2000; ++trial) {
    const double scale = trial % 3 == 0 ? 1.0 : (trial % 3 == 1 ? 1e-3 : 1e3);
    for (auto& v : hA) v = (rand() / (double)RAND_MAX - 0.5) * scale;
    for (auto& v : hB) v = (rand() / (double)RAND_MAX - 0.5);
    for (auto& v : hC) v = (rand() / (double)RAND_MAX - 0.5) * (trial % 2 ? 1.0 : 1e-2);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        double asc = hC[m * 16 + n], desc = hC[m * 16 + n];
        for (int kk = 0; kk < 4; ++kk) asc = fma(hA[m * 4 + kk], hB[kk * 16 + n], asc);
        for (int kk = 3; kk >= 0; --kk) desc = fma(hA[m * 4 + kk], hB[kk * 16 + n], desc);
        const double pair = fma(hA[m * 4 + 3], hB[48 + n], fma(hA[m * 4 + 2], hB[32 + n], 0.0)) +
                            fma(hA[m * 4 + 1], hB[16 + n], fma(hA[m * 4 + 0], hB[n], hC[m * 16 + n]));
        long double ex = hC[m * 16 + n];
        for (int kk = 0; kk < 4; ++kk) ex += (long double)hA[m * 4 + kk] * hB[kk * 16 + n];
        const double got = hD[m * 16 + n];
        same_asc += memcmp(&got, &asc, 8) == 0; same_desc += memcmp(&got, &desc, 8) == 0; same_pair += memcmp(&got, &pair, 8) == 0;
        const double exd = (double)ex; same_exact += memcmp(&got, &exd, 8) == 0;
        ++total;
      }
  }
  printf("elements %ld: equal to ascending-k fma chain %ld, descending %ld, pairwise %ld, rounded 80-bit sum %ld\n", total, same_asc, same_desc,
         same_pair, same_exact);
  return 0;
}
