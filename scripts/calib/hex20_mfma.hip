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
  __syncthreads();
  // tiles of this wave: all 16 -> row `wave`, columns 0..3; upper only -> the 10 tiles (r <= c), tile q = wave + 4 t
  auto tile_rc = [&](int t, int& r, int& c) {
    if (!UPPER) { r = wave; c = t; return t < 4; }
    const int q = wave + 4 * t;
    r = (q >= 4) + (q >= 7) + (q >= 9);
    c = r + q - (4 * r - (r * (r - 1)) / 2);
    return q < 10;
  };
  constexpr int NT = UPPER ? 3 : 4;
  for (int e = blockIdx.x; e < ne; e += gridDim.x) {
    double4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
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
          Bc[0] = i == 0 ? ga[0] : 0.0; Bc[1] = i == 1 ? ga[1] : 0.0; Bc[2] = i == 2 ? ga[2] : 0.0;
          Bc[3] = i == 0 ? ga[1] : (i == 1 ? ga[0] : 0.0);
          Bc[4] = i == 1 ? ga[2] : (i == 2 ? ga[1] : 0.0);
          Bc[5] = i == 0 ? ga[2] : (i == 2 ? ga[0] : 0.0);
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
        for (int t = 0; t < NT; t++) {
          int r, c;
          if (tile_rc(t, r, c)) {
            const double av = sL[buf][kr][r * 16 + (lane & 15)];     // A[i][k] = L[k][row tile + i], lane = 16 k + i
            const double bv = sB[buf][kr][c * 16 + (lane & 15)];     // B[k][j] = B[k][col tile + j], lane = 16 k + j
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    // D[i][j]: j = lane % 16, i = 4 * register + lane / 16 (measured: scripts/calib/mfma_f64_layout.hip)
    double* out = K + (size_t)e * ND * ND;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      int tr, tc;
      if (!tile_rc(t, tr, tc)) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = tr * 16 + 4 * r + (lane >> 4), j = tc * 16 + (lane & 15);
        if (i < ND && j < ND) out[i * ND + j] = acc[t][r];
      }
    }
  }
}

// debug: same chunk fill as mfma_kernel, the product taken by plain FMAs from the LDS operands (one element)
__global__ __launch_bounds__(256) void fillcheck_kernel(const double* __restrict__ rec, const double* __restrict__ dN,
                                                        double* __restrict__ K, double d0, double d1, double d3, double* __restrict__ dump) {
  __shared__ double sD[NIP * NPE * 3];
  __shared__ double sL[12][64], sB[12][64];
  const int tid = threadIdx.x;
  for (int i = tid; i < NIP * NPE * 3; i += 256) sD[i] = dN[i];
  for (int i = tid; i < 12 * 64; i += 256) { (&sL[0][0])[i] = 0.0; (&sB[0][0])[i] = 0.0; }
  __syncthreads();
  double acc[15];
  for (int q = 0; q < 15; q++) acc[q] = 0.0;
  for (int ch = 0; ch < 14; ch++) {
    if (tid < 120) {
      const int g = tid / 60, col = tid - g * 60, ip = ch * 2 + g;
      double Bc[6] = { 0, 0, 0, 0, 0, 0 }, Lc[6] = { 0, 0, 0, 0, 0, 0 };
      if (ip < NIP) {
        const int an = col / 3, i = col - an * 3;
        const double* r = rec + ip * REC;
        double ga[3];
        for (int q = 0; q < 3; q++) ga[q] = sD[(ip * NPE + an) * 3 + 0] * r[0 * 3 + q] + sD[(ip * NPE + an) * 3 + 1] * r[1 * 3 + q] + sD[(ip * NPE + an) * 3 + 2] * r[2 * 3 + q];
        Bc[0] = i == 0 ? ga[0] : 0.0; Bc[1] = i == 1 ? ga[1] : 0.0; Bc[2] = i == 2 ? ga[2] : 0.0;
        Bc[3] = i == 0 ? ga[1] : (i == 1 ? ga[0] : 0.0);
        Bc[4] = i == 1 ? ga[2] : (i == 2 ? ga[1] : 0.0);
        Bc[5] = i == 0 ? ga[2] : (i == 2 ? ga[0] : 0.0);
        const double w = r[9], s = Bc[0] + Bc[1] + Bc[2];
        for (int c = 0; c < 3; c++) Lc[c] = w * ((d0 - d1) * Bc[c] + d1 * s);
        for (int c = 3; c < 6; c++) Lc[c] = w * d3 * Bc[c];
      }
      for (int c = 0; c < 6; c++) { sB[g * 6 + c][col] = Bc[c]; sL[g * 6 + c][col] = Lc[c]; }
    }
    __syncthreads();
    if (ch == 0) for (int i = tid; i < 12 * 64; i += 256) { dump[i] = (&sL[0][0])[i]; dump[768 + i] = (&sB[0][0])[i]; }
    for (int q = 0; q < 15; q++) {
      const int idx = tid + 256 * q;
      if (idx < ND * ND) { const int i = idx / ND, j = idx - i * ND; for (int k = 0; k < 12; k++) acc[q] += sL[k][i] * sB[k][j]; }
    }
    __syncthreads();
  }
  for (int q = 0; q < 15; q++) { const int idx = tid + 256 * q; if (idx < ND * ND) K[idx] = acc[q]; }
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
  if (getenv("HOSTONLY")) {                            // emulate the chunk fill on the host for element 0
    std::vector<double> rec(NIP * REC), kh(ND * ND, 0.0), kv(ND * ND, 0.0);
    for (int ip = 0; ip < NIP; ip++) {
      double J[3][3] = { { 0 } };
      for (int a = 0; a < NPE; a++) for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) J[i][k] += x[a * 3 + i] * dN[(ip * NPE + a) * 3 + k];
      const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, rd = 1.0 / det;
      double* r = &rec[ip * REC];
      r[0] = c00 * rd; r[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rd; r[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rd;
      r[3] = c01 * rd; r[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rd; r[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rd;
      r[6] = c02 * rd; r[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rd; r[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rd;
      r[9] = w[ip] * det;
    }
    for (int ip = 0; ip < NIP; ip++) {
      double sL[6][64] = { { 0 } }, sB[6][64] = { { 0 } };
      const double* r = &rec[ip * REC];
      for (int col = 0; col < 60; col++) {
        const int an = col / 3, i = col - an * 3;
        double Bc[6] = { 0 }, Lc[6] = { 0 }, ga[3];
        for (int q = 0; q < 3; q++) ga[q] = dN[(ip * NPE + an) * 3 + 0] * r[0 * 3 + q] + dN[(ip * NPE + an) * 3 + 1] * r[1 * 3 + q] + dN[(ip * NPE + an) * 3 + 2] * r[2 * 3 + q];
        if (i == 0) { Bc[0] = ga[0]; Bc[3] = ga[1]; Bc[5] = ga[2]; }
        else if (i == 1) { Bc[1] = ga[1]; Bc[3] = ga[0]; Bc[4] = ga[2]; }
        else { Bc[2] = ga[2]; Bc[4] = ga[1]; Bc[5] = ga[0]; }
        const double ww = r[9], sum = Bc[0] + Bc[1] + Bc[2];
        for (int c = 0; c < 3; c++) Lc[c] = ww * ((d0 - d1) * Bc[c] + d1 * sum);
        for (int c = 3; c < 6; c++) Lc[c] = ww * d3 * Bc[c];
        for (int c = 0; c < 6; c++) { sB[c][col] = Bc[c]; sL[c][col] = Lc[c]; }
      }
      for (int i = 0; i < ND; i++) for (int j = 0; j < ND; j++) for (int k = 0; k < 6; k++) kh[i * ND + j] += sL[k][i] * sB[k][j];
      // pair formula
      for (int a = 0; a < NPE; a++) for (int b2 = 0; b2 < NPE; b2++) {
        double wga[3], gb[3];
        for (int i = 0; i < 3; i++) { double aa2 = 0, bb = 0; for (int k = 0; k < 3; k++) { aa2 += dN[(ip * NPE + a) * 3 + k] * r[k * 3 + i]; bb += dN[(ip * NPE + b2) * 3 + k] * r[k * 3 + i]; } wga[i] = r[9] * aa2; gb[i] = bb; }
        const double dotp = wga[0] * gb[0] + wga[1] * gb[1] + wga[2] * gb[2];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
          kv[(3 * a + i) * ND + 3 * b2 + j] += (i == j) ? (d0 - d3) * (wga[i] * gb[i]) + d3 * dotp : d1 * (wga[i] * gb[j]) + d3 * (wga[j] * gb[i]);
      }
    }
    double em = 0, sc = 0;
    for (int i = 0; i < ND * ND; i++) { em = std::fmax(em, std::fabs(kh[i] - kv[i])); sc = std::fmax(sc, std::fabs(kv[i])); }
    printf("host: fill emulation against the pair formula: %.3e (largest %.3e)\n", em, sc);
    return 0;
  }
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
  {                                                    // host reference for element 0: dense B^T D B
    std::vector<double> kh(ND * ND, 0.0);
    for (int ip = 0; ip < NIP; ip++) {
      double J[3][3] = { { 0 } }, inv[3][3];
      for (int a = 0; a < NPE; a++) for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) J[i][k] += x[a * 3 + i] * dN[(ip * NPE + a) * 3 + k];
      const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      inv[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det; inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det; inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
      inv[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det; inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
      inv[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det; inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det; inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
      double B[6][ND] = { { 0 } }, D[6][6] = { { 0 } };
      for (int a = 0; a < NPE; a++) {
        double g[3];
        for (int i = 0; i < 3; i++) g[i] = dN[(ip * NPE + a) * 3 + 0] * inv[0][i] + dN[(ip * NPE + a) * 3 + 1] * inv[1][i] + dN[(ip * NPE + a) * 3 + 2] * inv[2][i];
        B[0][3 * a] = g[0]; B[1][3 * a + 1] = g[1]; B[2][3 * a + 2] = g[2];
        B[3][3 * a] = g[1]; B[3][3 * a + 1] = g[0]; B[4][3 * a + 1] = g[2]; B[4][3 * a + 2] = g[1]; B[5][3 * a + 2] = g[0]; B[5][3 * a] = g[2];
      }
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D[i][j] = (i == j) ? d0 : d1;
      for (int i = 3; i < 6; i++) D[i][i] = d3;
      for (int r = 0; r < ND; r++) for (int c2 = 0; c2 < ND; c2++) {
        double acc = 0.0;
        for (int p = 0; p < 6; p++) for (int q = 0; q < 6; q++) acc += B[p][r] * D[p][q] * B[q][c2];
        kh[r * ND + c2] += w[ip] * det * acc;
      }
    }
    double e1m = 0.0, e2m = 0.0, sc = 0.0;
    for (int i = 0; i < ND; i++) for (int j = 0; j < ND; j++) if (j / 3 >= i / 3) {
      e1m = std::fmax(e1m, std::fabs(k1[i * ND + j] - kh[i * ND + j])); e2m = std::fmax(e2m, std::fabs(k2[i * ND + j] - kh[i * ND + j]));
      sc = std::fmax(sc, std::fabs(kh[i * ND + j]));
    }
    {
      double* dk3; CHECK(hipMalloc(&dk3, ND * ND * 8));
      double* ddump; CHECK(hipMalloc(&ddump, 1536 * 8));
      fillcheck_kernel<<<1, 256>>>(drec, ddN, dk3, d0, d1, d3, ddump);
      std::vector<double> k3(ND * ND), dump(1536), hrec(NIP * REC);
      CHECK(hipMemcpy(dump.data(), ddump, 1536 * 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hrec.data(), drec, NIP * REC * 8, hipMemcpyDeviceToHost));
      int shown = 0;
      for (int g = 0; g < 2; g++) for (int col = 0; col < 60; col++) {
        const int an = col / 3, i = col - an * 3, ip = g;
        const double* r = &hrec[ip * REC];
        double Bc[6] = { 0 }, ga[3];
        for (int q = 0; q < 3; q++) ga[q] = dN[(ip * NPE + an) * 3 + 0] * r[0 * 3 + q] + dN[(ip * NPE + an) * 3 + 1] * r[1 * 3 + q] + dN[(ip * NPE + an) * 3 + 2] * r[2 * 3 + q];
        if (i == 0) { Bc[0] = ga[0]; Bc[3] = ga[1]; Bc[5] = ga[2]; }
        else if (i == 1) { Bc[1] = ga[1]; Bc[3] = ga[0]; Bc[4] = ga[2]; }
        else { Bc[2] = ga[2]; Bc[4] = ga[1]; Bc[5] = ga[0]; }
        for (int c2 = 0; c2 < 6; c2++) {
          const double dv = dump[768 + (g * 6 + c2) * 64 + col];
          if (std::fabs(dv - Bc[c2]) > 1e-9 * (1 + std::fabs(Bc[c2])) && shown++ < 12) printf("  sB[g %d][row %d][col %d]: device %.6e host %.6e\n", g, c2, col, dv, Bc[c2]);
        }
      }
      printf("chunk 0 of B: %d mismatching entries\n", shown);
      CHECK(hipMemcpy(k3.data(), dk3, ND * ND * 8, hipMemcpyDeviceToHost));
      double e3 = 0.0;
      for (int i = 0; i < ND * ND; i++) e3 = std::fmax(e3, std::fabs(k3[i] - kh[i]));
      printf("fill check (LDS operands, plain FMAs): off by %.2e relative\n", e3 / 2.133e4);
    }
    printf("element 0 against a host reference (largest entry %.3e): VALU off by %.2e, MFMA off by %.2e (relative)\n", sc, e1m / sc, e2m / sc);
  }
  for (int q = 0; q < 6; q++) { const int i = q * 7, j = 59 - q * 3 < i ? i : 59 - q * 3; printf("  K[%d][%d]: valu %.9e  mfma %.9e\n", i, j, k1[i * ND + j], k2[i * ND + j]); }
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
