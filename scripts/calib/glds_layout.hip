// Where does global_load_lds_{dword,dwordx3,dwordx4} put each lane's bytes?  Prints the LDS dword index at which
// lane l's k-th dword landed (source dword value = 1000 * lane + k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SIZE>
__device__ __forceinline__ void glds(const void* src, void* lds_wave_base) {
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  unsigned keep;
  if constexpr (SIZE == 16)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  else if constexpr (SIZE == 12)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int SIZE>
__global__ void k(const int* src, int* out, int nact) {
  __shared__ int buf[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = -1;
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  if ((int)(threadIdx.x & 63) < nact) glds<SIZE>(src + threadIdx.x * (SIZE / 4), &buf[wv * 64 * 4 + 8]);   // base offset 32 bytes into the wave's region
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = buf[i];
}
template <int SIZE>
void run(int nact) {
  const int n = 128 * 4;
  std::vector<int> h(n);
  for (int l = 0; l < 128; l++) for (int q = 0; q < SIZE / 4; q++) h[l * (SIZE / 4) + q] = 1000 * l + q;
  int *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<SIZE>, dim3(1), dim3(128), 0, 0, d, o, nact);
  std::vector<int> r(2048);
  hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
  printf("SIZE %d active lanes %d: ", SIZE, nact);
  int shown = 0;
  for (int i = 0; i < 2048 && shown < 40; i++) if (r[i] >= 0) { printf("[%d]=%d ", i, r[i]); shown++; }
  printf("\n  wave 1 first: ");
  shown = 0;
  for (int i = 256; i < 2048 && shown < 8; i++) if (r[i] >= 0) { printf("[%d]=%d ", i, r[i]); shown++; }
  printf("\n");
  hipFree(d); hipFree(o);
}
int main() { run<4>(64); run<12>(64); run<16>(64); run<12>(5); run<16>(3); return 0; }
