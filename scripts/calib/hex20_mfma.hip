// hex20_mfma.hip -- the MFMA-FP64 experiment BASELINE.json's north_star asks for ("MFMA FP64 is considered only for
// batched hex20 blocks, justified by rocprof").
//
// Forms the element matrices K_e = sum_ip w B^T D B (60 x 60, 27 Gauss points, Hooke) of a batch of Hex20 elements
// two ways and times both on the same inputs:
//   VALU  the inner loop of the tile kernel's pair phase (fa_tile_impl.h): one lane per node pair a <= b, per Gauss
//         point the two gradients from J^-1 (LDS record) and the 3 x 3 block with the isotropic-form tangent;
//         exploits the sparsity of B (3 of 6 entries per column) and the symmetry of K_e (210 of 400 blocks)
//   MFMA  dense: K_e = L^T B with B the stacked (gauss point, strain component) x 60 matrix and L = w D B, as
//         v_mfma_f64_16x16x4_f64 products (4 x 4 output tiles of 16 x 16, K dimension 6 x 28 = 168), operands
//         staged in LDS two Gauss points at a time; all 16 tiles, and the 10 upper-triangular ones only
// The Gauss-point geometry (J^-1, w det J) is computed beforehand by the same kernel for both (not timed): it is the
// tile kernel's phase 1 in either case.  Prints elements/s of each variant and the largest difference between them.
//
// build: hipcc --offload-arch=gfx950 -O3 scripts/calib/hex20_mfma.hip -o scripts/calib/hex20_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NPE = 20, NIP = 27, ND = 60, REC = 10;
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void geom_kernel(const double* __restrict__ x, const double* __restrict__ dN, const double* __restrict__ wt,
                            double* __restrict__ rec, int ne) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ne * NIP) return;
  const int e = t / NIP, ip = t - e * NIP;
  double J[3][3] = { { 0 } };
  for (int a = 0; a < NPE; a++)
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) J[i][k] += x[(e * NPE + a) * 3 + i] * dN[(ip * NPE + a) * 3 + k];
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
               c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, rd = 1.0 / det;
  double* r = rec + (size_t)t * REC;
  r[0] = c00 * rd; r[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rd; r[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rd;
  r[3] = c01 * rd; r[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rd; r[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rd;
  r[6] = c02 * rd; r[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rd; r[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rd;
  r[9] = wt[ip] * det;
}

// ---- VALU: lane per node pair a <= b (210 of the 256 lanes), persistent over elements ----
__global__ __launch_bounds__(256) void valu_kernel(const double* __restrict__ rec, const double* __restrict__ dN,
                                                   double* __restrict__ K, int ne, double d0, double d1, double d3) {
  __shared__ double sD[NIP * NPE * 3];
  __shared__ double sR[NIP * REC];
  const int tid = threadIdx.x;
  for (int i = tid; i < NIP * NPE * 3; i += 256) sD[i] = dN[i];
  int a = 0, b = 0;
  { int p = tid, row = 0; while (row < NPE && p >= NPE - row) { p -= NPE - row; row++; } a = row; b = row + p; }
  const bool act = tid < NPE * (NPE + 1) / 2;
  for (int e = blockIdx.x; e < ne; e += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < NIP * REC; i += 256) sR[i] = rec[(size_t)e * NIP * REC + i];
    __syncthreads();
    if (!act) continue;
    double Kb[3][3] = { { 0 } };
#pragma unroll 1
    for (int ip = 0; ip < NIP; ip++) {
      const double* r = sR + ip * REC;
      const double w = r[9];
      double wga[3], gb[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        double acca = 0.0, accb = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { acca += sD[(ip * NPE + a) * 3 + k] * r[k * 3 + i]; accb += sD[(ip * NPE + b) * 3 + k] * r[k * 3 + i]; }
        wga[i] = w * acca; gb[i] = accb;
      }
      const double dotp = wga[0] * gb[0] + wga[1] * gb[1] + wga[2] * gb[2];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          if (i == j) Kb[i][j] += (d0 - d3) * (wga[i] * gb[i]) + d3 * dotp;
          else        Kb[i][j] += d1 * (wga[i] * gb[j]) + d3 * (wga[j] * gb[i]);
        }
    }
    double* out = K + (size_t)e * ND * ND;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) out[(3 * a + i) * ND + 3 * b + j] = Kb[i][j];
  }
}

// ---- MFMA: one workgroup (4 waves) per element, dense L^T B with v_mfma_f64_16x16x4_f64 ----
// chunk = two Gauss points = 12 K-rows = 3 MFMA steps; operands of a chunk in LDS as [12][64] (columns padded to 64)
template <bool UPPER>
__global__ __launch_bounds__(256) void mfma_kernel(const double* __restrict__ rec, const double* __restrict__ dN,
                                                   double* __restrict__ K, int ne, double d0, double d1, double d3) {
  __shared__ double sD[NIP * NPE * 3];
  __shared__ double sL[2][12][64], sB[2][12][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < NIP * NPE * 3; i += 256) sD[i] = dN[i];
  for (int i = tid; i < 2 * 12 * 64; i += 256) { (&sL[0][0][0])[i] = 0.0; (&sB[0][0][0])[i] = 0.0; }
  // tiles of this wave: all 16 -> row `wave`, columns 0..3; upper only -> the 10 tiles (r <= c) dealt round-robin
  int tr[4], tc[4], nt = 0;
  if (!UPPER) { for (int c = 0; c < 4; c++) { tr[nt] = wave; tc[nt] = c; nt++; } }
  else { int q = 0; for (int r = 0; r < 4; r++) for (int c = r; c < 4; c++) { if ((q & 3) == wave) { tr[nt] = r; tc[nt] = c; nt++; } q++; } }
  for (int e = blockIdx.x; e < ne; e += gridDim.x) {
    double4_t acc[4];
    for (int t = 0; t < 4; t++) acc[t] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
    const double* re = rec + (size_t)e * NIP * REC;
    for (int ch = 0; ch < 14; ch++) {
      const int buf = ch & 1;
      // fill: thread t < 120 -> (gauss point ch*2 + t/60, column t%60): the 6 entries of B and of L = w D B
      if (tid < 120) {
        const int g = tid / 60, col = tid - g * 60, ip = ch * 2 + g;
        double Bc[6] = { 0, 0, 0, 0, 0, 0 }, Lc[6] = { 0, 0, 0, 0, 0, 0 };
        if (ip < NIP) {
          const int an = col / 3, i = col - an * 3;
          const double* r = re + ip * REC;
          double ga[3];
#pragma unroll
          for (int q = 0; q < 3; q++) ga[q] = sD[(ip * NPE + an) * 3 + 0] * r[0 * 3 + q] + sD[(ip * NPE + an) * 3 + 1] * r[1 * 3 + q] + sD[(ip * NPE + an) * 3 + 2] * r[2 * 3 + q];
          // column 3a+i of B (ShapeUtils.cpp:114-145): xx,yy,zz,xy,yz,zx
          if (i == 0) { Bc[0] = ga[0]; Bc[3] = ga[1]; Bc[5] = ga[2]; }
          else if (i == 1) { Bc[1] = ga[1]; Bc[3] = ga[0]; Bc[4] = ga[2]; }
          else { Bc[2] = ga[2]; Bc[4] = ga[1]; Bc[5] = ga[0]; }
          const double w = r[9], s = Bc[0] + Bc[1] + Bc[2];
#pragma unroll
          for (int c = 0; c < 3; c++) Lc[c] = w * ((d0 - d1) * Bc[c] + d1 * s);
#pragma unroll
          for (int c = 3; c < 6; c++) Lc[c] = w * d3 * Bc[c];
        }
#pragma unroll
        for (int c = 0; c < 6; c++) { sB[buf][g * 6 + c][col] = Bc[c]; sL[buf][g * 6 + c][col] = Lc[c]; }
      }
      __syncthreads();                                 // one barrier per chunk: the other buffer is filled next
#pragma unroll
      for (int ks = 0; ks < 3; ks++) {
        const int kr = ks * 4 + (lane >> 4);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (t < nt) {
            const double av = sL[buf][kr][tr[t] * 16 + (lane & 15)];     // A[i][k] = L[k][row tile + i], lane = 16 k + i
            const double bv = sB[buf][kr][tc[t] * 16 + (lane & 15)];     // B[k][j] = B[k][col tile + j], lane = 16 k + j
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    // D[i][j]: j = lane % 16, i = 4 * (lane / 16) + r
    double* out = K + (size_t)e * ND * ND;
    for (int t = 0; t < nt; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = tr[t] * 16 + 4 * (lane >> 4) + r, j = tc[t] * 16 + (lane & 15);
        if (i < ND && j < ND) out[i * ND + j] = acc[t][r];
      }
  }
}

static void hex20_tables(std::vector<double>& dN, std::vector<double>& w, std::vector<double>& xi) {
  const double g3[3] = { -std::sqrt(0.6), 0.0, std::sqrt(0.6) }, w3[3] = { 5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0 };
  int c[20][3];
  const int px[8] = { -1, 0, 1, 1, 1, 0, -1, -1 }, py[8] = { -1, -1, -1, 0, 1, 1, 1, 0 };
  for (int a = 0; a < 8; a++) { c[a][0] = px[a]; c[a][1] = py[a]; c[a][2] = -1; c[12 + a][0] = px[a]; c[12 + a][1] = py[a]; c[12 + a][2] = 1; }
  for (int a = 0; a < 4; a++) { c[8 + a][0] = px[2 * a]; c[8 + a][1] = py[2 * a]; c[8 + a][2] = 0; }
  dN.assign(NIP * NPE * 3, 0.0); w.assign(NIP, 0.0); xi.assign(NPE * 3, 0.0);
  for (int a = 0; a < NPE; a++) for (int k = 0; k < 3; k++) xi[a * 3 + k] = c[a][k];
  for (int iz = 0; iz < 3; iz++) for (int iy = 0; iy < 3; iy++) for (int ix = 0; ix < 3; ix++) {
    const int ip = (iz * 3 + iy) * 3 + ix;
    const double g[3] = { g3[ix], g3[iy], g3[iz] };
    for (int a = 0; a < NPE; a++) {
      int mid = -1;
      for (int k = 0; k < 3; k++) if (c[a][k] == 0) mid = k;
      double f[3], d[3];
      for (int k = 0; k < 3; k++) f[k] = 1 + c[a][k] * g[k];
      if (mid < 0) {
        const double sum = c[a][0] * g[0] + c[a][1] * g[1] + c[a][2] * g[2] - 2;
        for (int k = 0; k < 3; k++) d[k] = 0.125 * c[a][k] * f[(k + 1) % 3] * f[(k + 2) % 3] * (sum + f[k]);
      } else {
        const int p = (mid + 1) % 3, q = (mid + 2) % 3;
        const double q2 = 1 - g[mid] * g[mid];
        d[mid] = -0.5 * g[mid] * f[p] * f[q]; d[p] = 0.25 * c[a][p] * q2 * f[q]; d[q] = 0.25 * c[a][q] * q2 * f[p];
      }
      for (int k = 0; k < 3; k++) dN[(ip * NPE + a) * 3 + k] = d[k];
    }
    w[ip] = w3[ix] * w3[iy] * w3[iz];
  }
}

int main(int argc, char** argv) {
  const int ne = argc > 1 ? atoi(argv[1]) : 32768, reps = 5;
  std::vector<double> dN, w, xi;
  hex20_tables(dN, w, xi);
  std::vector<double> x((size_t)ne * NPE * 3);
  srand(7);
  for (int e = 0; e < ne; e++)
    for (int a = 0; a < NPE; a++)
      for (int k = 0; k < 3; k++) x[((size_t)e * NPE + a) * 3 + k] = 0.05 * (xi[a * 3 + k] + 0.1 * (rand() / (double)RAND_MAX - 0.5));
  const double E = 210e3, nu = 0.3, aa = E / ((1 + nu) * (1 - 2 * nu)), d0 = aa * (1 - nu), d1 = aa * nu, d3 = aa * 0.5 * (1 - 2 * nu);
  double *dx, *ddN, *dw, *drec, *dK1, *dK2;
  const size_t kbytes = (size_t)ne * ND * ND * sizeof(double);
  CHECK(hipMalloc(&dx, x.size() * 8)); CHECK(hipMalloc(&ddN, dN.size() * 8)); CHECK(hipMalloc(&dw, w.size() * 8));
  CHECK(hipMalloc(&drec, (size_t)ne * NIP * REC * 8)); CHECK(hipMalloc(&dK1, kbytes)); CHECK(hipMalloc(&dK2, kbytes));
  CHECK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(ddN, dN.data(), dN.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemset(dK1, 0, kbytes)); CHECK(hipMemset(dK2, 0, kbytes));
  hipLaunchKernelGGL(geom_kernel, dim3((ne * NIP + 255) / 256), dim3(256), 0, 0, dx, ddN, dw, drec, ne);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  auto time_it = [&](auto launch, const char* name) {
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-34s %8.3f ms for %d elements = %8.2f M elements/s  -> %6.1f ms for the 2 M elements of cfg5\n", name, ms, ne,
           ne / ms * 1e-3, 2.0e6 / (ne / (double)ms));
  };
  time_it([&] { hipLaunchKernelGGL(valu_kernel, dim3(grid), dim3(256), 0, 0, drec, ddN, dK1, ne, d0, d1, d3); }, "VALU, pair lanes (210 blocks)");
  time_it([&] { hipLaunchKernelGGL(mfma_kernel<false>, dim3(grid), dim3(256), 0, 0, drec, ddN, dK2, ne, d0, d1, d3); }, "MFMA f64 16x16x4, 16 tiles");
  std::vector<double> k1((size_t)2 * ND * ND), k2((size_t)2 * ND * ND);
  const int probe[2] = { 0, ne - 1 };
  double worst = 0.0, scale = 0.0;
  for (int q = 0; q < 2; q++) {
    CHECK(hipMemcpy(k1.data() + q * ND * ND, dK1 + (size_t)probe[q] * ND * ND, ND * ND * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(k2.data() + q * ND * ND, dK2 + (size_t)probe[q] * ND * ND, ND * ND * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < ND; i++) for (int j = 0; j < ND; j++) if (j / 3 >= i / 3) {
      worst = std::fmax(worst, std::fabs(k1[q * ND * ND + i * ND + j] - k2[q * ND * ND + i * ND + j]));
      scale = std::fmax(scale, std::fabs(k1[q * ND * ND + i * ND + j]));
    }
  }
  printf("largest |K_valu - K_mfma| over the upper blocks of 2 elements: %.3e (largest entry %.3e, relative %.2e)\n", worst, scale, worst / scale);
  CHECK(hipMemset(dK2, 0, kbytes));
  time_it([&] { hipLaunchKernelGGL(mfma_kernel<true>, dim3(grid), dim3(256), 0, 0, drec, ddN, dK2, ne, d0, d1, d3); }, "MFMA f64 16x16x4, 10 upper tiles");
  worst = 0.0;
  for (int q = 0; q < 2; q++) {
    CHECK(hipMemcpy(k2.data() + q * ND * ND, dK2 + (size_t)probe[q] * ND * ND, ND * ND * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < ND; i++) for (int j = 0; j < ND; j++) if (j / 16 >= i / 16 && j / 3 >= i / 3)
      worst = std::fmax(worst, std::fabs(k1[q * ND * ND + i * ND + j] - k2[q * ND * ND + i * ND + j]));
  }
  printf("upper-tile variant: largest difference %.3e (relative %.2e)\n", worst, worst / scale);
  return 0;
}
