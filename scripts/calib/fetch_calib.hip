// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the assembly kernels use
// (the guide only calibrates 16 B/lane): streams a known byte count with 4, 8 and 16 B per lane loads / stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T> __global__ void rd(const T* __restrict__ x, size_t n, double* out) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T v = x[i];
    const double* d = (const double*)&v;
    if constexpr (sizeof(T) == 4) acc += *(const float*)&v; else for (unsigned k = 0; k < sizeof(T) / 8; ++k) acc += d[k];
  }
  if (acc == 12345.678) out[0] = acc;
}
template <class T> __global__ void wr(T* __restrict__ x, size_t n) {
  T v; char* c = (char*)&v; for (unsigned k = 0; k < sizeof(T); ++k) c[k] = (char)k;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = v;
}
// gather like the assembly kernels: 8 B loads at stride-2 doubles (coords of consecutive nodes, one component)
__global__ void rd8s2(const double* __restrict__ x, size_t n, double* out) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * blockDim.x)
    acc += x[2 * i] + x[2 * i + 1];
  if (acc == 12345.678) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)2 << 30;                    // 2 GiB >> 256 MiB Infinity Cache
  void *buf; double* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc((void**)&out, 8)); CK(hipMemset(buf, 0, bytes));
  const int grid = 256 * 16, blk = 256;
  for (int rep = 0; rep < 2; ++rep) {
    rd<float><<<grid, blk>>>((const float*)buf, bytes / 4, out);
    rd<double><<<grid, blk>>>((const double*)buf, bytes / 8, out);
    rd<double2><<<grid, blk>>>((const double2*)buf, bytes / 16, out);
    rd8s2<<<grid, blk>>>((const double*)buf, bytes / 8, out);
    wr<float><<<grid, blk>>>((float*)buf, bytes / 4);
    wr<double><<<grid, blk>>>((double*)buf, bytes / 8);
    wr<double2><<<grid, blk>>>((double2*)buf, bytes / 16);
  }
  CK(hipDeviceSynchronize());
  printf("bytes per kernel: %zu\n", bytes);
  return 0;
}
