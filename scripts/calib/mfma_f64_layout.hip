// Output / operand layout of v_mfma_f64_16x16x4_f64, found by experiment: A[i][k] = (i+1) for k == kk, B[k][j] = 100 (j+1)
// for k == kk  =>  D[i][j] = 100 (i+1)(j+1); prints which (i, j) every (lane, register) holds, for kk = 0 and kk = 2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* out, int kk) {
  const int lane = threadIdx.x;
  const double a = (lane / 16 == kk) ? (double)(lane % 16 + 1) : 0.0;          // assumed: lane = 16 k + i
  const double b = (lane / 16 == kk) ? 100.0 * (lane % 16 + 1) : 0.0;          // assumed: lane = 16 k + j
  double4_t c = { 0, 0, 0, 0 };
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[lane * 4 + r] = c[r];
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  double h[256];
  for (int kk = 0; kk < 4; kk += 2) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, kk);
    hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
    printf("kk=%d\n", kk);
    const int lanes[6] = { 0, 1, 15, 16, 17, 63 };
    for (int q = 0; q < 6; q++) {
      const int l = lanes[q];
      printf("  lane %2d:", l);
      for (int r = 0; r < 4; r++) {
        const int v = (int)(h[l * 4 + r] / 100.0 + 0.5);          // (i+1)(j+1)
        int fi = -1, fj = -1;
        // decode assuming j = lane % 16
        const int j = l % 16;
        if (v % (j + 1) == 0) { fi = v / (j + 1) - 1; fj = j; }
        printf("  r%d: val %6.0f -> i=%d j=%d", r, h[l * 4 + r], fi, fj);
      }
      printf("\n");
    }
  }
  return 0;
}
