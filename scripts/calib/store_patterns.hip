// What do the K stores of the tile kernels cost?  Write throughput of the access patterns they produce, on a 4 GiB buffer:
//   coalesced 16 B / lane; 24-byte chunks (16 + 8 B, 8-byte aligned) and 32-byte chunks (2 x 16 B) by consecutive lanes
//   (chunk c by lane c), by lanes in bit-reversed order inside a 1024-chunk window (every lane its own cache line, the
//   order of the count-sorted pair lanes), and 8 B / lane contiguous runs (the staged row write-out).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ size_t scatter(size_t c, int win_log2) {      // bit reversal of the low win_log2 bits
  const size_t lo = c & (((size_t)1 << win_log2) - 1);
  const size_t r = (size_t)(__brev((unsigned)lo) >> (32 - win_log2));
  return (c - lo) + r;
}
template <int MODE>
__global__ void wr(double* __restrict__ x, size_t nchunk, int win_log2) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < nchunk; c += (size_t)gridDim.x * blockDim.x) {
    const double v = (double)c;
    if constexpr (MODE == 0) { double2* q = (double2*)x + c; *q = make_double2(v, v); }                       // 16 B coalesced
    else if constexpr (MODE == 1) { double* q = x + 3 * c; d2u t; t.x = v; t.y = v; *(d2u*)q = t; q[2] = v; }   // 24 B chunk, lane order
    else if constexpr (MODE == 2) { double* q = x + 3 * scatter(c, win_log2); d2u t; t.x = v; t.y = v; *(d2u*)q = t; q[2] = v; }
    else if constexpr (MODE == 3) { double2* q = (double2*)(x + 4 * c); q[0] = make_double2(v, v); q[1] = make_double2(v, v); }   // 32 B chunk, lane order
    else if constexpr (MODE == 4) { double2* q = (double2*)(x + 4 * scatter(c, win_log2)); q[0] = make_double2(v, v); q[1] = make_double2(v, v); }
    else if constexpr (MODE == 5) { x[c] = v; }                                                                  // 8 B coalesced
    else if constexpr (MODE == 6) { double* q = x + 3 * c; q[0] = v; q[1] = v; q[2] = v; }                       // 24 B chunk as 3 x 8 B, lane order
    else if constexpr (MODE == 7) {                    // 24-byte chunks, lane order, through a wave transpose: lane l writes doubles l, l+64, l+128 of the wave's 192
      const size_t w0 = (c & ~(size_t)63) * 3; const int l = (int)(c & 63);
      x[w0 + l] = v; x[w0 + 64 + l] = v; x[w0 + 128 + l] = v;
    }
  }
}
template <int MODE> double run(double* buf, size_t nchunk, int bytes_per_chunk, int win) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  wr<MODE><<<256 * 16, 256>>>(buf, nchunk, win);
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; r++) wr<MODE><<<256 * 16, 256>>>(buf, nchunk, win);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return (double)nchunk * bytes_per_chunk * 3 / (ms * 1e-3) / 1e9;
}
int main() {
  const size_t bytes = (size_t)4 << 30;
  double* buf; CK(hipMalloc((void**)&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  printf("GB/s written (4 GiB per launch)\n");
  printf("16 B/lane coalesced              %8.0f\n", run<0>(buf, bytes / 16, 16, 0));
  printf(" 8 B/lane coalesced              %8.0f\n", run<5>(buf, bytes / 8, 8, 0));
  printf("24 B chunks, lane order (16+8)   %8.0f\n", run<1>(buf, bytes / 24, 24, 0));
  printf("24 B chunks, lane order (3 x 8)  %8.0f\n", run<6>(buf, bytes / 24, 24, 0));
  printf("24 B chunks, wave transpose      %8.0f\n", run<7>(buf, (bytes / 24) & ~(size_t)63, 24, 0));
  for (int w : { 6, 10, 14 }) printf("24 B chunks, scattered 2^%-2d       %8.0f\n", w, run<2>(buf, (bytes / 24) >> w << w, 24, w));
  printf("32 B chunks, lane order (2 x 16) %8.0f\n", run<3>(buf, bytes / 32, 32, 0));
  for (int w : { 6, 10, 14 }) printf("32 B chunks, scattered 2^%-2d       %8.0f\n", w, run<4>(buf, (bytes / 32) >> w << w, 32, w));
  return 0;
}
