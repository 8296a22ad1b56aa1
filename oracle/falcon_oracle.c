/*
 * falcon_oracle.c -- CPU restatement of falcon's K / f_int assembly hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see falcon_oracle.h).  PARITY UNPINNED: no reference
 * fixture pins these numbers; analytic KATs in tests/ do.
 *
 * The loop structure follows the reference on purpose (element-serial, Gauss
 * point inner loop, dense B^T D B, per-entry search scatter into CSR) so the
 * same code doubles as the timed single-core CPU baseline.
 *
 * Reference files restated (paths under /root/reference):
 *   src/fem/solidmech/LinearElasticityModel.cpp:297-430
 *   src/fem/solidmech/PhaseFractureModel.cpp:279-361,502-751
 *   src/fem/solidmech/PhaseFractureExtModel.cpp:488-774
 *   src/fem/solidmech/PhaseFractureExtItModel.cpp:518-797
 *   src/materials/HookeMaterial.cpp:152-166,202-267
 *   src/materials/AmorPhaseMaterial.cpp:77-150,200-232
 *   src/materials/BourdinPhaseMaterial.cpp:131-145
 *   src/materials/MiehePhaseMaterial.cpp:80-149,200-258,342-406
 *   src/util/ShapeUtils.cpp:73-145, TensorUtils.cpp, MathUtils.cpp:23-41, Constants.cpp:9
 *   src/fem/solidmech/MicroPhaseFracture{,Ext,ExtIt}Model.cpp (getMatrix_), PhaseFractureModel.cpp:844-1063 (getArcFunc_)
 *   src/fem/solidmech/GradientDamageModel.cpp:368-588, src/materials/DamageMaterial.cpp:356-440,602-706,886-1032,1044-1633
 *   src/materials/VonMisesStrain.cpp:34-116
 *   src/util/Invariants.cpp:35-224, MathUtils.cpp:112-182 (solveCubicEqn), Constants.cpp:13-18
 */
#include "falcon_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXNPE 20
#define MAXNIP 27
#define MAXSTR 6
#define MAXND  (3 * MAXNPE)

static char g_err[512] = "";
const char* fo_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* src/util/Constants.cpp:9 */
static const int STRAIN_COUNTS[4] = { 0, 1, 4, 6 };

/* ------------------------------------------------------------------------- */
/* Shapes (Jive IShape stand-in; standard isoparametric definitions)         */
/* ------------------------------------------------------------------------- */

typedef struct {
  int    rank, npe, nip;
  double N [MAXNIP][MAXNPE];
  double dN[MAXNIP][MAXNPE][3];
  double w [MAXNIP];
} shape_t;

static void gauss1d(int n, double* x, double* w) {
  if (n == 1) { x[0] = 0.0; w[0] = 2.0; }
  else if (n == 2) { x[0] = -1.0 / sqrt(3.0); x[1] = 1.0 / sqrt(3.0); w[0] = w[1] = 1.0; }
  else { x[0] = -sqrt(0.6); x[1] = 0.0; x[2] = sqrt(0.6); w[0] = w[2] = 5.0 / 9.0; w[1] = 8.0 / 9.0; }
}

/* perimeter orderings (corner-mid-corner-mid), SURVEY 8c */
static const double Q4X[4][2] = { {-1,-1},{1,-1},{1,1},{-1,1} };
static const double Q8X[8][2] = { {-1,-1},{0,-1},{1,-1},{1,0},{1,1},{0,1},{-1,1},{-1,0} };
static const double H8X[8][3] = { {-1,-1,-1},{1,-1,-1},{1,1,-1},{-1,1,-1},
                                  {-1,-1, 1},{1,-1, 1},{1,1, 1},{-1,1, 1} };
/* bottom perimeter (8), vertical mid-edges (4), top perimeter (8) */
static const double H20X[20][3] = {
  {-1,-1,-1},{0,-1,-1},{1,-1,-1},{1,0,-1},{1,1,-1},{0,1,-1},{-1,1,-1},{-1,0,-1},
  {-1,-1,0},{1,-1,0},{1,1,0},{-1,1,0},
  {-1,-1,1},{0,-1,1},{1,-1,1},{1,0,1},{1,1,1},{0,1,1},{-1,1,1},{-1,0,1} };

static void eval_tri3(double x, double y, double* N, double (*d)[3]) {
  N[0] = 1.0 - x - y; N[1] = x; N[2] = y;
  d[0][0] = -1; d[0][1] = -1; d[1][0] = 1; d[1][1] = 0; d[2][0] = 0; d[2][1] = 1;
}
static void eval_tri6(double x, double y, double* N, double (*d)[3]) {
  double L1 = 1.0 - x - y, L2 = x, L3 = y;
  /* dL1 = (-1,-1), dL2 = (1,0), dL3 = (0,1) */
  N[0] = L1 * (2 * L1 - 1); d[0][0] = -(4 * L1 - 1); d[0][1] = -(4 * L1 - 1);
  N[1] = 4 * L1 * L2;       d[1][0] = 4 * (L1 - L2);  d[1][1] = -4 * L2;
  N[2] = L2 * (2 * L2 - 1); d[2][0] = 4 * L2 - 1;     d[2][1] = 0;
  N[3] = 4 * L2 * L3;       d[3][0] = 4 * L3;         d[3][1] = 4 * L2;
  N[4] = L3 * (2 * L3 - 1); d[4][0] = 0;              d[4][1] = 4 * L3 - 1;
  N[5] = 4 * L3 * L1;       d[5][0] = -4 * L3;        d[5][1] = 4 * (L1 - L3);
}
static void eval_quad4(double x, double y, double* N, double (*d)[3]) {
  for (int a = 0; a < 4; a++) {
    double xa = Q4X[a][0], ya = Q4X[a][1];
    N[a] = 0.25 * (1 + xa * x) * (1 + ya * y);
    d[a][0] = 0.25 * xa * (1 + ya * y);
    d[a][1] = 0.25 * ya * (1 + xa * x);
  }
}
static void eval_quad8(double x, double y, double* N, double (*d)[3]) {
  for (int a = 0; a < 8; a++) {
    double xa = Q8X[a][0], ya = Q8X[a][1];
    if (xa != 0 && ya != 0) {
      N[a] = 0.25 * (1 + xa * x) * (1 + ya * y) * (xa * x + ya * y - 1);
      d[a][0] = 0.25 * xa * (1 + ya * y) * (2 * xa * x + ya * y);
      d[a][1] = 0.25 * ya * (1 + xa * x) * (xa * x + 2 * ya * y);
    } else if (xa == 0) {
      N[a] = 0.5 * (1 - x * x) * (1 + ya * y);
      d[a][0] = -x * (1 + ya * y);
      d[a][1] = 0.5 * ya * (1 - x * x);
    } else {
      N[a] = 0.5 * (1 + xa * x) * (1 - y * y);
      d[a][0] = 0.5 * xa * (1 - y * y);
      d[a][1] = -y * (1 + xa * x);
    }
  }
}
static void eval_hex8(double x, double y, double z, double* N, double (*d)[3]) {
  for (int a = 0; a < 8; a++) {
    double xa = H8X[a][0], ya = H8X[a][1], za = H8X[a][2];
    N[a] = 0.125 * (1 + xa * x) * (1 + ya * y) * (1 + za * z);
    d[a][0] = 0.125 * xa * (1 + ya * y) * (1 + za * z);
    d[a][1] = 0.125 * ya * (1 + xa * x) * (1 + za * z);
    d[a][2] = 0.125 * za * (1 + xa * x) * (1 + ya * y);
  }
}
static void eval_hex20(double x, double y, double z, double* N, double (*d)[3]) {
  for (int a = 0; a < 20; a++) {
    double xa = H20X[a][0], ya = H20X[a][1], za = H20X[a][2];
    if (xa != 0 && ya != 0 && za != 0) {
      double s = xa * x + ya * y + za * z - 2;
      double fx = 1 + xa * x, fy = 1 + ya * y, fz = 1 + za * z;
      N[a] = 0.125 * fx * fy * fz * s;
      d[a][0] = 0.125 * xa * fy * fz * (s + fx);
      d[a][1] = 0.125 * ya * fx * fz * (s + fy);
      d[a][2] = 0.125 * za * fx * fy * (s + fz);
    } else if (xa == 0) {
      N[a] = 0.25 * (1 - x * x) * (1 + ya * y) * (1 + za * z);
      d[a][0] = -0.5 * x * (1 + ya * y) * (1 + za * z);
      d[a][1] = 0.25 * ya * (1 - x * x) * (1 + za * z);
      d[a][2] = 0.25 * za * (1 - x * x) * (1 + ya * y);
    } else if (ya == 0) {
      N[a] = 0.25 * (1 + xa * x) * (1 - y * y) * (1 + za * z);
      d[a][0] = 0.25 * xa * (1 - y * y) * (1 + za * z);
      d[a][1] = -0.5 * y * (1 + xa * x) * (1 + za * z);
      d[a][2] = 0.25 * za * (1 + xa * x) * (1 - y * y);
    } else {
      N[a] = 0.25 * (1 + xa * x) * (1 + ya * y) * (1 - z * z);
      d[a][0] = 0.25 * xa * (1 + ya * y) * (1 - z * z);
      d[a][1] = 0.25 * ya * (1 + xa * x) * (1 - z * z);
      d[a][2] = -0.5 * z * (1 + xa * x) * (1 + ya * y);
    }
  }
}

static int shape_init(shape_t* s, int id) {
  memset(s, 0, sizeof *s);
  double gx[3], gw[3];
  switch (id) {
    case FO_TRI3_G1: case FO_TRI3_G3: case FO_TRI6_G3: {
      s->rank = 2; s->npe = (id == FO_TRI6_G3) ? 6 : 3;
      double px[3][2]; double pw[3];
      if (id == FO_TRI3_G1) { s->nip = 1; px[0][0] = px[0][1] = 1.0 / 3.0; pw[0] = 0.5; }
      else { s->nip = 3;
             px[0][0] = 1.0 / 6.0; px[0][1] = 1.0 / 6.0;
             px[1][0] = 2.0 / 3.0; px[1][1] = 1.0 / 6.0;
             px[2][0] = 1.0 / 6.0; px[2][1] = 2.0 / 3.0;
             pw[0] = pw[1] = pw[2] = 1.0 / 6.0; }
      for (int ip = 0; ip < s->nip; ip++) {
        if (s->npe == 3) eval_tri3(px[ip][0], px[ip][1], s->N[ip], s->dN[ip]);
        else             eval_tri6(px[ip][0], px[ip][1], s->N[ip], s->dN[ip]);
        s->w[ip] = pw[ip];
      }
      return 0;
    }
    case FO_QUAD4_G2: case FO_QUAD8_G3: case FO_QUAD8_G2: {
      int n = (id == FO_QUAD8_G3) ? 3 : 2;
      s->rank = 2; s->npe = (id == FO_QUAD4_G2) ? 4 : 8; s->nip = n * n;
      gauss1d(n, gx, gw);
      for (int iy = 0; iy < n; iy++) for (int ix = 0; ix < n; ix++) {
        int ip = iy * n + ix;
        if (s->npe == 4) eval_quad4(gx[ix], gx[iy], s->N[ip], s->dN[ip]);
        else             eval_quad8(gx[ix], gx[iy], s->N[ip], s->dN[ip]);
        s->w[ip] = gw[ix] * gw[iy];
      }
      return 0;
    }
    case FO_HEX8_G2: case FO_HEX20_G3: case FO_HEX20_G2: {
      int n = (id == FO_HEX20_G3) ? 3 : 2;
      s->rank = 3; s->npe = (id == FO_HEX8_G2) ? 8 : 20; s->nip = n * n * n;
      gauss1d(n, gx, gw);
      for (int iz = 0; iz < n; iz++) for (int iy = 0; iy < n; iy++) for (int ix = 0; ix < n; ix++) {
        int ip = (iz * n + iy) * n + ix;
        if (s->npe == 8) eval_hex8 (gx[ix], gx[iy], gx[iz], s->N[ip], s->dN[ip]);
        else             eval_hex20(gx[ix], gx[iy], gx[iz], s->N[ip], s->dN[ip]);
        s->w[ip] = gw[ix] * gw[iy] * gw[iz];
      }
      return 0;
    }
  }
  FAIL("unknown shape id %d", id);
}

int fo_shape_info_get(int shape, fo_shape_info* out) {
  shape_t s;
  if (shape_init(&s, shape)) return -1;
  out->rank = s.rank; out->npe = s.npe; out->nip = s.nip; out->str = STRAIN_COUNTS[s.rank];
  return 0;
}

int fo_shape_tables(int shape, double* N, double* dN, double* w) {
  shape_t s;
  if (shape_init(&s, shape)) return -1;
  for (int ip = 0; ip < s.nip; ip++) {
    w[ip] = s.w[ip];
    for (int a = 0; a < s.npe; a++) {
      N[ip * s.npe + a] = s.N[ip][a];
      for (int k = 0; k < s.rank; k++) dN[(ip * s.npe + a) * s.rank + k] = s.dN[ip][a][k];
    }
  }
  return 0;
}

/* IShape::getShapeGradients stand-in: grads[ip][a][i] = dN_a/dx_i, wip = w * det J */
static int shape_gradients(const shape_t* s, const double (*x)[3],
                           double grads[MAXNIP][MAXNPE][3], double* wip) {
  const int r = s->rank;
  for (int ip = 0; ip < s->nip; ip++) {
    double J[3][3] = { {0} };      /* J[i][k] = sum_a x_a[i] dN_a/dxi_k */
    for (int a = 0; a < s->npe; a++)
      for (int i = 0; i < r; i++)
        for (int k = 0; k < r; k++) J[i][k] += x[a][i] * s->dN[ip][a][k];
    double det, inv[3][3];
    if (r == 2) {
      det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      inv[0][0] =  J[1][1] / det; inv[0][1] = -J[0][1] / det;
      inv[1][0] = -J[1][0] / det; inv[1][1] =  J[0][0] / det;
    } else {
      double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
      double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
      double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
      inv[0][0] = c00 / det;
      inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
      inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
      inv[1][0] = c01 / det;
      inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
      inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
      inv[2][0] = c02 / det;
      inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
      inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
    }
    if (!(det > 0.0)) FAIL("non-positive Jacobian determinant %g", det);
    /* dN/dx_i = sum_k dN/dxi_k * inv[k][i]   (inv = J^{-1}: dxi_k/dx_i) */
    for (int a = 0; a < s->npe; a++)
      for (int i = 0; i < r; i++) {
        double g = 0.0;
        for (int k = 0; k < r; k++) g += s->dN[ip][a][k] * inv[k][i];
        grads[ip][a][i] = g;
      }
    wip[ip] = s->w[ip] * det;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* B matrix: src/util/ShapeUtils.cpp:73-104 (2D, 4 rows) and :114-145 (3D)    */
/* b is str x nd, row-major here: b[c*nd + col]                              */
/* ------------------------------------------------------------------------- */
static void fill_B(int rank, int npe, const double g[MAXNPE][3], double* b) {
  const int nd = rank * npe;
  const int str = STRAIN_COUNTS[rank];
  memset(b, 0, sizeof(double) * str * nd);
  if (rank == 2) {
    for (int a = 0; a < npe; a++) {
      int i = 2 * a, i1 = i + 1;
      b[0 * nd + i ] = g[a][0];
      b[1 * nd + i1] = g[a][1];
      b[3 * nd + i ] = g[a][1];
      b[3 * nd + i1] = g[a][0];
    }
  } else {
    for (int a = 0; a < npe; a++) {
      int i = 3 * a;
      b[0 * nd + i + 0] = g[a][0];
      b[1 * nd + i + 1] = g[a][1];
      b[2 * nd + i + 2] = g[a][2];
      b[3 * nd + i + 0] = g[a][1];
      b[3 * nd + i + 1] = g[a][0];
      b[4 * nd + i + 1] = g[a][2];
      b[4 * nd + i + 2] = g[a][1];
      b[5 * nd + i + 2] = g[a][0];
      b[5 * nd + i + 0] = g[a][2];
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Materials                                                                 */
/* ------------------------------------------------------------------------- */

/* HookeMaterial::computeStiffMat_, src/materials/HookeMaterial.cpp:202-267 */
static int hooke_stiff(int rank, int state, double e, double nu, double* D /* str*str */) {
  const int n = STRAIN_COUNTS[rank];
  memset(D, 0, sizeof(double) * n * n);
#define S(i, j) D[(i) * n + (j)]
  if (rank == 3) {
    const double a = e / ((1.0 + nu) * (1.0 - 2.0 * nu));
    const double b = 0.5 * (1.0 - 2.0 * nu);
    const double c = 1.0 - nu;
    S(0,0) = a * c;
    S(0,1) = S(0,2) = a * nu;
    S(1,1) = S(0,0);
    S(1,2) = S(0,1);
    S(2,2) = S(0,0);
    S(3,3) = a * b;
    S(4,4) = S(3,3);
    S(5,5) = S(3,3);
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) S(i,j) = S(j,i);
  } else if (rank == 2 && state == FO_PLANE_STRAIN) {
    const double a = e / ((1.0 + nu) * (1.0 - 2.0 * nu));
    const double b = 0.5 * (1.0 - 2.0 * nu);
    const double c = 1.0 - nu;
    S(0,0) = a * c;
    S(0,1) = S(1,0) = a * nu;
    S(0,2) = a * nu;
    S(1,1) = a * c;
    S(1,2) = a * nu;
    S(3,3) = a * b;
    S(2,0) = a * nu; S(2,1) = a * nu; S(2,2) = a * c;
  } else if (rank == 2 && state == FO_PLANE_STRESS) {
    const double a = e / (1.0 - nu * nu);
    S(0,0) = a;
    S(0,1) = S(1,0) = a * nu;
    S(1,1) = a;
    S(3,3) = a * 0.5 * (1.0 - nu);
  } else {
    FAIL("hooke: unexpected rank/state %d/%d", rank, state);
  }
#undef S
  return 0;
}

static void matvec(int n, const double* A, const double* x, double* y) {
  for (int i = 0; i < n; i++) {
    double s = 0.0;
    for (int j = 0; j < n; j++) s += A[i * n + j] * x[j];
    y[i] = s;
  }
}
static double dotn(int n, const double* a, const double* b) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

/* src/util/MathUtils.cpp:23-41 */
static double macaulayP(double x) { return 0.5 * (x + fabs(x)); }
static double macaulayN(double x) { return 0.5 * (x - fabs(x)); }
static double heavisideP(double x) { return (x < 0.) ? 0.0 : 1.0; }

/* AmorPhaseMaterial::update, src/materials/AmorPhaseMaterial.cpp:200-232 (+ configure :85-148) */
static void amor_update(int rank, double E, double nu, const double* strain, double gphi,
                        double* stress, double* stiff, double* psi, double* stressP) {
  const int n = STRAIN_COUNTS[rank];
  const double K = E / (3.0 * (1.0 - 2.0 * nu));
  const double G = E / (2.0 * (1.0 + nu));
  double PVol[MAXSTR * MAXSTR] = { 0 }, PDev[MAXSTR * MAXSTR] = { 0 };
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    PVol[i * n + j] = 1.0;
    PDev[i * n + j] = (i == j) ? 2.0 / 3.0 : -1.0 / 3.0;
  }
  for (int i = 3; i < n; i++) PDev[i * n + i] = 0.5;

  double tr = strain[0] + strain[1] + strain[2];
  for (int i = 0; i < n * n; i++) stiff[i] = 2.0 * G * PDev[i];
  if (tr > 0.0) for (int i = 0; i < n * n; i++) stiff[i] += K * PVol[i];
  matvec(n, stiff, strain, stressP);
  *psi = 0.5 * dotn(n, stressP, strain);
  for (int i = 0; i < n * n; i++) stiff[i] *= gphi;
  if (tr <= 0.0) for (int i = 0; i < n * n; i++) stiff[i] += K * PVol[i];
  matvec(n, stiff, strain, stress);
}

/* BourdinPhaseMaterial::update, src/materials/BourdinPhaseMaterial.cpp:131-145 */
static int bourdin_update(int rank, int state, double E, double nu, const double* strain, double gphi,
                          double* stress, double* stiff, double* psi, double* stressP) {
  const int n = STRAIN_COUNTS[rank];
  double D[MAXSTR * MAXSTR];
  if (hooke_stiff(rank, state, E, nu, D)) return -1;
  matvec(n, D, strain, stressP);
  *psi = 0.5 * dotn(n, stressP, strain);
  for (int i = 0; i < n; i++) stress[i] = gphi * stressP[i];
  for (int i = 0; i < n * n; i++) stiff[i] = gphi * D[i];
  return 0;
}

/* jem::numeric::EigenUtils::symSolve stand-in: cyclic Jacobi on a 3x3 symmetric matrix.
 * (Jem's algorithm is not visible; P+ depends only on eigen-spaces, not on the solver.) */
static void sym_eig3(const double A[3][3], double* val, double V[3][3]) {
  double a[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { a[i][j] = A[i][j]; V[i][j] = (i == j); }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (a[p][q] == 0.0) continue;
      double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
      double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) {          /* A <- A J */
        double akp = a[k][p], akq = a[k][q];
        a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; k++) {          /* A <- J^T A */
        double apk = a[p][k], aqk = a[q][k];
        a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; k++) {          /* V <- V J */
        double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
      }
    }
  }
  for (int i = 0; i < 3; i++) val[i] = a[i][i];
}

typedef double quad4_t[3][3][3][3];

/* tensorUtils::tensor2voigt, src/util/TensorUtils.cpp:304-398 (verbatim incl. the
 * duplicated-term shear-shear entries of the 3D branch, :368-369,378-379,388-389) */
static void tensor2voigt(quad4_t A, int strCount, double* M) {
  const int n = strCount;
  memset(M, 0, sizeof(double) * n * n);
#define MM(i, j) M[(i) * n + (j)]
  if (n == 4) {
    MM(0,0) = A[0][0][0][0]; MM(1,0) = A[1][1][0][0]; MM(2,0) = A[2][2][0][0];
    MM(3,0) = 0.5 * (A[0][1][0][0] + A[1][0][0][0]);
    MM(0,1) = A[0][0][1][1]; MM(1,1) = A[1][1][1][1]; MM(2,1) = A[2][2][1][1];
    MM(3,1) = 0.5 * (A[0][1][1][1] + A[1][0][1][1]);
    MM(0,2) = A[0][0][2][2]; MM(1,2) = A[1][1][2][2]; MM(2,2) = A[2][2][2][2];
    MM(3,2) = 0.5 * (A[0][1][2][2] + A[1][0][2][2]);
    MM(0,3) = 0.5 * (A[0][0][0][1] + A[0][0][1][0]);
    MM(1,3) = 0.5 * (A[1][1][0][1] + A[1][1][1][0]);
    MM(2,3) = 0.5 * (A[2][2][0][1] + A[2][2][1][0]);
    MM(3,3) = 0.25 * (A[0][1][0][1] + A[1][0][0][1] + A[0][1][1][0] + A[1][0][1][0]);
  } else {
    MM(0,0) = A[0][0][0][0]; MM(1,0) = A[1][1][0][0]; MM(2,0) = A[2][2][0][0];
    MM(3,0) = 0.5 * (A[0][1][0][0] + A[1][0][0][0]);
    MM(4,0) = 0.5 * (A[1][2][0][0] + A[2][1][0][0]);
    MM(5,0) = 0.5 * (A[2][0][0][0] + A[0][2][0][0]);
    MM(0,1) = A[0][0][1][1]; MM(1,1) = A[1][1][1][1]; MM(2,1) = A[2][2][1][1];
    MM(3,1) = 0.5 * (A[0][1][1][1] + A[1][0][1][1]);
    MM(4,1) = 0.5 * (A[1][2][1][1] + A[2][1][1][1]);
    MM(5,1) = 0.5 * (A[2][0][1][1] + A[0][2][1][1]);
    MM(0,2) = A[0][0][2][2]; MM(1,2) = A[1][1][2][2]; MM(2,2) = A[2][2][2][2];
    MM(3,2) = 0.5 * (A[0][1][2][2] + A[1][0][2][2]);
    MM(4,2) = 0.5 * (A[1][2][2][2] + A[2][1][2][2]);
    MM(5,2) = 0.5 * (A[2][0][2][2] + A[0][2][2][2]);
    MM(0,3) = 0.5 * (A[0][0][0][1] + A[0][0][1][0]);
    MM(1,3) = 0.5 * (A[1][1][0][1] + A[1][1][1][0]);
    MM(2,3) = 0.5 * (A[2][2][0][1] + A[2][2][1][0]);
    MM(3,3) = 0.25 * (A[0][1][0][1] + A[1][0][1][0] + A[0][1][0][1] + A[1][0][1][0]);
    MM(4,3) = 0.25 * (A[1][2][0][1] + A[2][1][1][0] + A[2][1][0][1] + A[1][2][1][0]);
    MM(5,3) = 0.25 * (A[0][2][0][1] + A[2][0][1][0] + A[2][0][0][1] + A[0][2][1][0]);
    MM(0,4) = 0.5 * (A[0][0][1][2] + A[0][0][2][1]);
    MM(1,4) = 0.5 * (A[1][1][1][2] + A[1][1][2][1]);
    MM(2,4) = 0.5 * (A[2][2][1][2] + A[2][2][2][1]);
    MM(3,4) = 0.25 * (A[0][1][1][2] + A[1][0][2][1] + A[0][1][1][2] + A[1][0][2][1]);
    MM(4,4) = 0.25 * (A[1][2][1][2] + A[2][1][2][1] + A[2][1][1][2] + A[1][2][2][1]);
    MM(5,4) = 0.25 * (A[0][2][1][2] + A[2][0][2][1] + A[2][0][1][2] + A[0][2][2][1]);
    MM(0,5) = 0.5 * (A[0][0][0][2] + A[0][0][2][0]);
    MM(1,5) = 0.5 * (A[1][1][0][2] + A[1][1][2][0]);
    MM(2,5) = 0.5 * (A[2][2][0][2] + A[2][2][2][0]);
    MM(3,5) = 0.25 * (A[0][1][0][2] + A[1][0][2][0] + A[0][1][0][2] + A[1][0][2][0]);
    MM(4,5) = 0.25 * (A[1][2][0][2] + A[2][1][2][0] + A[2][1][0][2] + A[1][2][2][0]);
    MM(5,5) = 0.25 * (A[0][2][0][2] + A[2][0][2][0] + A[2][0][0][2] + A[0][2][2][0]);
  }
#undef MM
}

/* MiehePhaseMaterial::getposProj_, src/materials/MiehePhaseMaterial.cpp:342-406 */
static void miehe_pos_proj(quad4_t P, const double T[3][3]) {
  double val[3], V[3][3], epos[3], diag[3];
  memset(P, 0, sizeof(quad4_t));
  sym_eig3(T, val, V);
  for (int i = 0; i < 3; i++) { epos[i] = macaulayP(val[i]); diag[i] = (val[i] > 0.0) ? 1.0 : 0.0; }
  double M[3][3][3];                       /* M[a] = n_a (x) n_a ; V[:,a] is eigenvector a */
  for (int a = 0; a < 3; a++)
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[a][i][j] = V[i][a] * V[j][a];
  for (int r = 0; r < 3; r++)
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++)
        P[i][j][k][l] += M[r][i][j] * M[r][k][l] * diag[r];
  for (int a = 0; a < 3; a++) for (int b = 0; b < a; b++) {
    double theta;
    if (fabs(val[a] - val[b]) <= 1e-13) theta = 0.5 * (diag[a] + diag[b]) / 2.0;
    else                                theta = 0.5 * (epos[a] - epos[b]) / (val[a] - val[b]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) {
        /* otimesu(A,B)_ijkl = A_ik B_jl ; otimesl(A,B)_ijkl = A_il B_jk  (TensorUtils.cpp:108-171) */
        double Gab = M[a][i][k] * M[b][j][l] + M[a][i][l] * M[b][j][k];
        double Gba = M[b][i][k] * M[a][j][l] + M[b][i][l] * M[a][j][k];
        P[i][j][k][l] += theta * (Gab + Gba);
      }
  }
}

/* MiehePhaseMaterial::update, src/materials/MiehePhaseMaterial.cpp:200-258 (+ configure :97-99) */
static void miehe_update(int rank, double E, double nu, const double* c, double gphi,
                         double* stress, double* stiff, double* psi, double* stressP) {
  const int n = STRAIN_COUNTS[rank];
  const double K = E / (3.0 * (1.0 - 2.0 * nu));
  const double G = E / (2.0 * (1.0 + nu));
  const double lmda = K - 0.666666667 * G;       /* literal, MiehePhaseMaterial.cpp:99 */
  double T[3][3] = { {0} };                      /* voigt2tensorStrain, TensorUtils.cpp:467-492 */
  T[0][0] = c[0]; T[1][1] = c[1]; T[2][2] = c[2];
  T[0][1] = T[1][0] = .5 * c[3];
  if (n == 6) { T[1][2] = T[2][1] = .5 * c[4]; T[2][0] = T[0][2] = .5 * c[5]; }

  static __thread quad4_t posP, negP, jac;
  miehe_pos_proj(posP, T);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) {
      double I4 = 0.5 * ((i == k) && (j == l)) + 0.5 * ((i == l) && (j == k));
      negP[i][j][k][l] = I4 - posP[i][j][k][l];
    }
  double sp[3][3], sn[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0.0;
    for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) s += posP[i][j][k][l] * T[k][l];
    sp[i][j] = s; sn[i][j] = T[i][j] - s;
  }
  double tr = T[0][0] + T[1][1] + T[2][2];
  double sigP[3][3], sigN[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double I2 = (i == j) ? 1.0 : 0.0;
    sigP[i][j] = lmda * macaulayP(tr) * I2 + 2.0 * G * sp[i][j];
    sigN[i][j] = lmda * macaulayN(tr) * I2 + 2.0 * G * sn[i][j];
  }
  /* tensor2voigtStress, TensorUtils.cpp:436-460 */
  double vP[MAXSTR], vN[MAXSTR];
  vP[0] = sigP[0][0]; vP[1] = sigP[1][1]; vP[2] = sigP[2][2]; vP[3] = 0.5 * (sigP[0][1] + sigP[1][0]);
  vN[0] = sigN[0][0]; vN[1] = sigN[1][1]; vN[2] = sigN[2][2]; vN[3] = 0.5 * (sigN[0][1] + sigN[1][0]);
  if (n == 6) {
    vP[4] = 0.5 * (sigP[1][2] + sigP[2][1]); vP[5] = 0.5 * (sigP[0][2] + sigP[2][0]);
    vN[4] = 0.5 * (sigN[1][2] + sigN[2][1]); vN[5] = 0.5 * (sigN[0][2] + sigN[2][0]);
  }
  for (int i = 0; i < n; i++) { stress[i] = gphi * vP[i] + vN[i]; stressP[i] = vP[i]; }
  double dd = 0.0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) dd += sp[i][j] * sp[i][j];
  *psi = 0.5 * lmda * pow(macaulayP(tr), 2.0) + G * dd;
  double signpos = heavisideP(tr), signneg = 1.0 - signpos;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) {
      double II = ((i == j) && (k == l)) ? 1.0 : 0.0;
      jac[i][j][k][l] = gphi * (2.0 * G * posP[i][j][k][l] + lmda * signpos * II)
                             + (2.0 * G * negP[i][j][k][l] + lmda * signneg * II);
    }
  tensor2voigt(jac, n, stiff);
}

int fo_material_update(const fo_config* cfg, int rank, const double* strain, double gphi,
                       double* stress, double* stiff, double* psi, double* stressP) {
  const int n = STRAIN_COUNTS[rank];
  switch (cfg->material) {
    case FO_HOOKE:   /* HookeMaterial::update, HookeMaterial.cpp:152-166 */
      if (hooke_stiff(rank, cfg->state, cfg->young, cfg->poisson, stiff)) return -1;
      matvec(n, stiff, strain, stress);
      if (psi) *psi = 0.0;
      return 0;
    case FO_AMOR:
      amor_update(rank, cfg->young, cfg->poisson, strain, gphi, stress, stiff, psi, stressP);
      return 0;
    case FO_BOURDIN:
      return bourdin_update(rank, cfg->state, cfg->young, cfg->poisson, strain, gphi, stress, stiff, psi, stressP);
    case FO_MIEHE:
      miehe_update(rank, cfg->young, cfg->poisson, strain, gphi, stress, stiff, psi, stressP);
      return 0;
  }
  FAIL("unknown material %d", cfg->material);
}

/* ------------------------------------------------------------------------- */
/* Overloaded update(stressP, stressN, stiffP, stiffN, strain) of the micromorphic models */
/* ------------------------------------------------------------------------- */
/* AmorPhaseMaterial.cpp:239-268.  NOTE (verbatim): stiffN is assigned only when trace(strain) <= 0; in tension the
 * caller's array keeps whatever an earlier point left in it (the model's `stiffN` lives outside its element loop,
 * MicroPhaseFractureModel.cpp:535).  stiffN is therefore in/out here. */
static void amor_update_split(int rank, double E, double nu, const double* strain,
                              double* stressP, double* stressN, double* stiffP, double* stiffN, double* psi) {
  const int n = STRAIN_COUNTS[rank];
  const double K = E / (3.0 * (1.0 - 2.0 * nu));
  const double G = E / (2.0 * (1.0 + nu));
  double PVol[MAXSTR * MAXSTR] = { 0 }, PDev[MAXSTR * MAXSTR] = { 0 };
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    PVol[i * n + j] = 1.0;
    PDev[i * n + j] = (i == j) ? 2.0 / 3.0 : -1.0 / 3.0;
  }
  for (int i = 3; i < n; i++) PDev[i * n + i] = 0.5;
  double tr = strain[0] + strain[1] + strain[2];
  for (int i = 0; i < n * n; i++) stiffP[i] = 2.0 * G * PDev[i];
  if (tr > 0.0) { for (int i = 0; i < n * n; i++) stiffP[i] += K * PVol[i]; }
  else          { for (int i = 0; i < n * n; i++) stiffN[i] = K * PVol[i]; }
  matvec(n, stiffP, strain, stressP);
  matvec(n, stiffN, strain, stressN);
  *psi = 0.5 * dotn(n, stressP, strain);
}

/* BourdinPhaseMaterial.cpp:148-166 */
static int bourdin_update_split(int rank, int state, double E, double nu, const double* strain,
                                double* stressP, double* stressN, double* stiffP, double* stiffN, double* psi) {
  const int n = STRAIN_COUNTS[rank];
  if (hooke_stiff(rank, state, E, nu, stiffP)) return -1;
  matvec(n, stiffP, strain, stressP);
  for (int i = 0; i < n; i++) stressN[i] = 0.0;
  for (int i = 0; i < n * n; i++) stiffN[i] = 0.0;
  *psi = 0.5 * dotn(n, stressP, strain);
  return 0;
}

/* MiehePhaseMaterial.cpp:264-324: jacP = 2G P+ + lambda H I(x)I, jacN = 2G P- + lambda (1-H) I(x)I.  The degraded
 * update above is g * (P part) + (N part); evaluating it with g = 1 and g = 0 separates the two exactly as written. */
static void miehe_update_split(int rank, double E, double nu, const double* strain,
                               double* stressP, double* stressN, double* stiffP, double* stiffN, double* psi) {
  const int n = STRAIN_COUNTS[rank];
  double s1[MAXSTR], D1[MAXSTR * MAXSTR], sp[MAXSTR], p1;
  miehe_update(rank, E, nu, strain, 0.0, stressN, stiffN, psi, sp);       /* g = 0: N part */
  miehe_update(rank, E, nu, strain, 1.0, s1, D1, &p1, stressP);           /* g = 1: P + N; stressP_ = vP */
  for (int i = 0; i < n * n; i++) stiffP[i] = D1[i] - stiffN[i];
}

static int material_update_split(const fo_config* cfg, int rank, const double* strain,
                                 double* stressP, double* stressN, double* stiffP, double* stiffN, double* psi) {
  switch (cfg->material) {
    case FO_AMOR:    amor_update_split(rank, cfg->young, cfg->poisson, strain, stressP, stressN, stiffP, stiffN, psi); return 0;
    case FO_BOURDIN: return bourdin_update_split(rank, cfg->state, cfg->young, cfg->poisson, strain, stressP, stressN, stiffP, stiffN, psi);
    case FO_MIEHE:   miehe_update_split(rank, cfg->young, cfg->poisson, strain, stressP, stressN, stiffP, stiffN, psi); return 0;
  }
  FAIL("material %d has no positive/negative split", cfg->material);
}

static __thread int64_t g_micro_stale = 0;
int64_t fo_micro_stale_hits(void) { return g_micro_stale; }

/* getStrain_ (PhaseFractureModel.cpp:1282-1395; LinearElasticityModel has the same loop): the Gauss-point strains of
 * the strain_ cache are extrapolated to the nodes with the transposed shape functions, element by element:
 * ndStrain = sum_ip N(:,ip) strain_(0:str, ipoint), ndWeights = sum_ip N(:,ip); weights[inodes] += ndWeights,
 * table.addBlock(inodes, jcols, ndStrain) (+=).  strain: [nel*nip][srows] (column ipoint of strain_), table:
 * [nnode][str], weights: [nnode]; the caller (Jive) divides the rows by the weights afterwards. */
int fo_strain_table(int shape, int64_t nel, const int32_t* conn, const double* strain, int srows, int str,
                    double* table, double* weights) {
  shape_t s;
  if (shape_init(&s, shape)) return -1;
  int64_t ipoint = 0;
  for (int64_t ie = 0; ie < nel; ie++) {
    double ndStrain[MAXNPE][6], ndWeights[MAXNPE];
    for (int a = 0; a < s.npe; a++) { ndWeights[a] = 0.0; for (int c = 0; c < str; c++) ndStrain[a][c] = 0.0; }
    for (int ip = 0; ip < s.nip; ip++, ipoint++)
      for (int a = 0; a < s.npe; a++) {
        for (int c = 0; c < str; c++) ndStrain[a][c] += s.N[ip][a] * strain[ipoint * srows + c];
        ndWeights[a] += s.N[ip][a];
      }
    for (int a = 0; a < s.npe; a++) {
      const int64_t n = conn[ie * s.npe + a];
      weights[n] += ndWeights[a];
      for (int c = 0; c < str; c++) table[n * str + c] += ndStrain[a][c];
    }
  }
  return 0;
}

/* getStress_ (PhaseFractureModel.cpp:1165-1271): per Gauss point the material is updated with the cached strain and the
 * cached degradation strain_(strCount, ipoint) (`update(stressIp, stiff, strainIp, gphiEx, ipoint)`, all strCount
 * components: 2-D carries zz), and the stresses are extrapolated to the nodes like the strains above.
 * LinearElasticityModel's getStress_ is the same loop with Hooke's update (no degradation row in the cache). */
int fo_stress_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain, int srows,
                    double* table, double* weights) {
  shape_t s;
  if (shape_init(&s, cfg->shape)) return -1;
  const int str = STRAIN_COUNTS[s.rank];
  int64_t ipoint = 0;
  for (int64_t ie = 0; ie < nel; ie++) {
    double ndStress[MAXNPE][6], ndWeights[MAXNPE];
    for (int a = 0; a < s.npe; a++) { ndWeights[a] = 0.0; for (int c = 0; c < str; c++) ndStress[a][c] = 0.0; }
    for (int ip = 0; ip < s.nip; ip++, ipoint++) {
      double stressIp[MAXSTR], stiff[MAXSTR * MAXSTR], sp[MAXSTR], psi;
      const double gphi = (srows > str) ? strain[ipoint * srows + str] : 1.0;
      if (fo_material_update(cfg, s.rank, strain + ipoint * srows, gphi, stressIp, stiff, &psi, sp)) return -1;
      for (int a = 0; a < s.npe; a++) {
        for (int c = 0; c < str; c++) ndStress[a][c] += s.N[ip][a] * stressIp[c];
        ndWeights[a] += s.N[ip][a];
      }
    }
    for (int a = 0; a < s.npe; a++) {
      const int64_t n = conn[ie * s.npe + a];
      weights[n] += ndWeights[a];
      for (int c = 0; c < str; c++) table[n * str + c] += ndStress[a][c];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Fracture constants: PhaseFractureModel.cpp:296-355                        */
/* ------------------------------------------------------------------------- */
typedef struct { double cw, eta, p, a1, a2, a3, psi0; } frac_t;

static int frac_init(const fo_config* cfg, frac_t* f) {
  const double E = cfg->young, gc = cfg->gc, l0 = cfg->l0, sT = cfg->sigma_t;
  switch (cfg->frac_type) {
    case FO_AT1: f->cw = 8.0 / 3.0; f->eta = 1.0; f->p = 2.0; f->a1 = 2.0; f->a2 = -0.5; f->a3 = 0.0;
                 f->psi0 = (3. * gc) / (16. * l0); return 0;
    case FO_AT2: f->cw = 2.0; f->eta = 0.0; f->p = 2.0; f->a1 = 2.0; f->a2 = -0.5; f->a3 = 0.0;
                 f->psi0 = 0.0; return 0;
    case FO_CZM_LINEAR: f->cw = 3.14159265359; f->eta = 2.0; f->p = 2.0;
                 f->a1 = (4.0 * E * gc) / (f->cw * l0 * sT * sT); f->a2 = -0.5; f->a3 = 0.0;
                 f->psi0 = 0.5 * (sT * sT) / E; return 0;
    case FO_CZM_EXPONENTIAL: f->cw = 3.14159265359; f->eta = 2.0; f->p = 2.5;
                 f->a1 = (4.0 * E * gc) / (f->cw * l0 * sT * sT); f->a2 = pow(2.0, 5.0 / 3.0) - 3.0; f->a3 = 0.0;
                 f->psi0 = 0.5 * (sT * sT) / E; return 0;
    case FO_CZM_CORNELISSEN: f->cw = 3.14159265359; f->eta = 2.0; f->p = 2.0;
                 f->a1 = (4.0 * E * gc) / (f->cw * l0 * sT * sT); f->a2 = 1.3868; f->a3 = 0.6567;
                 f->psi0 = 0.5 * (sT * sT) / E; return 0;
    case FO_CZM_HYPERBOLIC: f->cw = 3.14159265359; f->eta = 2.0; f->p = 4.0;
                 f->a1 = (4.0 * E * gc) / (f->cw * l0 * sT * sT); f->a2 = pow(2.0, 7.0 / 3.0) - 4.5; f->a3 = 0.0;
                 f->psi0 = 0.5 * (sT * sT) / E; return 0;
  }
  FAIL("unexpected definition of fracture type %d", cfg->frac_type);
}

double fo_psi0(const fo_config* cfg) { frac_t f; if (frac_init(cfg, &f)) return NAN; return f.psi0; }

/* PhaseFractureModel.cpp:660-676 */
static void degradation(const frac_t* f, double pf, double* g, double* dg, double* ddg) {
  const double p_ = f->p, a1_ = f->a1, a2_ = f->a2, a3_ = f->a3;
  double gphi_n   = pow(1 - pf, p_);
  double gphi_d   = gphi_n + a1_ * pf + a1_ * a2_ * pf * pf + a1_ * a2_ * a3_ * pf * pf * pf;
  *g              = gphi_n / gphi_d + 1.e-7;
  double dgphi_n  = -p_ * (pow(1 - pf, p_ - 1));
  double dgphi_d  = dgphi_n + a1_ + 2.0 * a1_ * a2_ * pf + 3.0 * a1_ * a2_ * a3_ * pf * pf;
  double ddgphi_n = p_ * (p_ - 1) * (pow(1 - pf, p_ - 2));
  double ddgphi_d = ddgphi_n + 2.0 * a1_ * a2_ + 6.0 * a1_ * a2_ * a3_ * pf;
  *dg  = (dgphi_n * gphi_d - gphi_n * dgphi_d) / (gphi_d * gphi_d);
  *ddg = ((ddgphi_n * gphi_d - gphi_n * ddgphi_d) * gphi_d - 2.0 *
          (dgphi_n * gphi_d - gphi_n * dgphi_d) * dgphi_d)
         / (gphi_d * gphi_d * gphi_d);
}

void fo_degradation(const fo_config* cfg, double pf, double* g, double* dg, double* ddg) {
  frac_t f;
  if (frac_init(cfg, &f)) { *g = *dg = *ddg = NAN; return; }
  degradation(&f, pf, g, dg, ddg);
}

int fo_ndof_per_node(int model, int rank) { return (model == FO_LINEAR_ELASTICITY) ? rank : rank + 1; }

/* ------------------------------------------------------------------------- */
/* Canonical DOF numbering + CSR pattern (Jive XDofSpace / FEM MatrixBuilder  */
/* stand-in): dof = node*ndofpn + type; row columns sorted ascending.         */
/* type_pair_mask bit (i*ndofpn + j) set => entry (type i row, type j col).   */
/* ------------------------------------------------------------------------- */
int64_t fo_pattern(int nnode, int64_t nel, int npe, const int32_t* conn, int ndofpn, int rank,
                   int kind, int keep_off_diags, int64_t* row_ptr, int32_t* col_idx) {
  (void)rank;
  /* kind/keep_off_diags -> mask; kind FULL = all pairs; WRITTEN = (u,u),(phi,phi)[,(u,phi),(phi,u)];
   * keep_off_diags: 0 none, 1 both (PhaseFracture), 2 phi-u only (PhaseFractureExt*) */
  unsigned mask = 0;
  for (int i = 0; i < ndofpn; i++) for (int j = 0; j < ndofpn; j++) {
    int on = 1;
    if (kind == FO_PATTERN_WRITTEN && ndofpn == rank + 1) {
      int iu = i < rank, ju = j < rank;
      if (iu != ju) on = (keep_off_diags == 1) || (keep_off_diags == 2 && !iu && ju);
    }
    if (on) mask |= 1u << (i * ndofpn + j);
  }
  int cnt_type[4] = { 0, 0, 0, 0 };       /* entries per neighbour node for a row of type i */
  for (int i = 0; i < ndofpn; i++) for (int j = 0; j < ndofpn; j++)
    if (mask >> (i * ndofpn + j) & 1) cnt_type[i]++;

  int64_t* nptr = (int64_t*)calloc((size_t)nnode + 1, sizeof(int64_t));
  if (!nptr) FAIL("out of memory");
  for (int64_t e = 0; e < nel; e++) for (int a = 0; a < npe; a++) nptr[conn[e * npe + a] + 1]++;
  for (int n = 0; n < nnode; n++) nptr[n + 1] += nptr[n];
  int64_t* nelm = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nptr[nnode] ? nptr[nnode] : 1));
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * ((size_t)nnode + 1));
  if (!nelm || !fill) FAIL("out of memory");
  memcpy(fill, nptr, sizeof(int64_t) * ((size_t)nnode + 1));
  for (int64_t e = 0; e < nel; e++) for (int a = 0; a < npe; a++) nelm[fill[conn[e * npe + a]]++] = e;
  free(fill);

  int32_t nb[1024];
  int64_t pos = 0;
  row_ptr[0] = 0;
  for (int n = 0; n < nnode; n++) {
    int m = 0;
    for (int64_t k = nptr[n]; k < nptr[n + 1]; k++) {
      const int32_t* c = conn + nelm[k] * npe;
      for (int a = 0; a < npe; a++) {
        int32_t v = c[a]; int j = m;
        while (j > 0 && nb[j - 1] > v) j--;
        if (j > 0 && nb[j - 1] == v) continue;
        if (m >= 1024) FAIL("node valence too large");
        memmove(nb + j + 1, nb + j, sizeof(int32_t) * (size_t)(m - j));
        nb[j] = v; m++;
      }
    }
    for (int i = 0; i < ndofpn; i++) {
      if (col_idx)
        for (int k = 0; k < m; k++)
          for (int j = 0; j < ndofpn; j++)
            if (mask >> (i * ndofpn + j) & 1) col_idx[pos++] = nb[k] * ndofpn + j;
      if (!col_idx) pos += (int64_t)m * cnt_type[i];
      row_ptr[(int64_t)n * ndofpn + i + 1] = pos;
    }
  }
  free(nptr); free(nelm);
  return pos;
}

/* MatrixBuilder::addBlock stand-in: K(irows, jcols) += block; per-entry binary search */
static int add_block(const int64_t* row_ptr, const int32_t* col_idx, double* val,
                     int nr, const int64_t* irows, int nc, const int64_t* jcols,
                     const double* block /* nr x nc row-major */) {
  for (int i = 0; i < nr; i++) {
    int64_t lo0 = row_ptr[irows[i]], hi0 = row_ptr[irows[i] + 1];
    for (int j = 0; j < nc; j++) {
      int64_t lo = lo0, hi = hi0; int32_t c = (int32_t)jcols[j];
      while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (col_idx[mid] < c) lo = mid + 1; else hi = mid; }
      if (lo >= hi0 || col_idx[lo] != c) FAIL("addBlock: entry (%lld,%lld) not in pattern",
                                               (long long)irows[i], (long long)jcols[j]);
      val[lo] += block[i * nc + j];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* The hot loop                                                              */
/* ------------------------------------------------------------------------- */
/* ---- DamageMaterial, restated for eqvStrain = Energy (DamageMaterial.cpp:645-650, :700-705) ---- */
static double damage_evolution(const fo_config* c, double K) {          /* damageEvolution_, :802-855, laws :1044-1633 */
  const double Ki = c->kappa_i, Kc = c->kappa_c;
  if (K < Ki) return 0.0;                                               /* first statement of every law */
  switch (c->softening) {
    case FO_SOFT_PERFECT: return 1.0 - Ki / K;
    case FO_SOFT_LINEAR1: { double omega = K > Kc ? 1.0 : Kc * (K - Ki) / (K * (Kc - Ki)); if (omega == 1.0) omega = 0.999; return omega; }
    case FO_SOFT_LINEAR2: return K > Kc ? 0.999 : (K - Ki) / (Kc - Ki);
    case FO_SOFT_EXPONENTIAL1: return 1.0 - (Ki / K) * (1.0 - c->soft_alpha + c->soft_alpha * exp(-c->soft_beta * (K - Ki)));
    case FO_SOFT_EXPONENTIAL2: return 1.0 - (Ki / K) * exp((Ki - K) / (Kc - Ki));
    case FO_SOFT_POWER: {                                               /* powerSoftening, :1389-1413 (Kcc = Kc) */
      const double Kcc = Kc;
      const double omegac = 1.0 - pow(Ki / Kcc, c->soft_beta) * pow((Kc - Kcc) / (Kc - Ki), c->soft_alpha);
      return (K > Kcc) ? omegac : 1.0 - pow(Ki / K, c->soft_beta) * pow((Kc - K) / (Kc - Ki), c->soft_alpha);
    }
    case FO_SOFT_HYPERBOLIC: {                                          /* hyperbolicSoftening, :1350-1364 */
      const double omega = 1.0 - 1.0 / (1.0 + c->soft_b * (K - Ki));
      /* jem::isTiny(1 - omega): Jem's threshold is not visible here; DBL_MIN stands in (only b (K - Ki) > 4e307 gets there) */
      return (fabs(1.0 - omega) < 2.2250738585072014e-308) ? 0.999 : omega;
    }
    case FO_SOFT_EXPONENTIAL2REG: return 0.0;       /* not in the one-argument damageEvolution_ (:802-851): omega stays 0 */
    case FO_SOFT_EXPOENERGY: { const double tem = pow((K - Ki) / Kc, c->soft_s); return 1.0 - exp(-tem); }   /* :1495-1505 */
    case FO_SOFT_POLYNOMIAL: { const double a = (K - Ki) / (Kc - Ki); return 1.0 - (Ki / K) * (1.0 + (pow(a, 2.0) * (2.0 * a - 3.0))); } /* :1530-1546 */
    default: {                                                          /* multilinearSoftening, :1581-1603 */
      const double Kp = c->kappa_p;
      if (K > Kp) return 1.0 - Ki / K;
      if (K < Kp) return 0.0;                                           /* linear2Softening (Kp, Kc, K) */
      return K > Kc ? 0.999 : (K - Kp) / (Kc - Kp);
    }
  }
}
static double damage_derivative(const fo_config* c, double K) {         /* getdOmegadKappa, :712-761 */
  const double Ki = c->kappa_i, Kc = c->kappa_c;
  switch (c->softening) {
    case FO_SOFT_PERFECT: return (K < Ki) ? 0.0 : Ki / (K * K);
    case FO_SOFT_LINEAR1: return (K < Kc) ? (Ki * Kc) / ((Kc - Ki) * K * K) : 0.0;
    case FO_SOFT_LINEAR2: return (K < Kc) ? 1.0 / (Kc - Ki) : 0.0;
    case FO_SOFT_EXPONENTIAL1: {
      const double h = K - Ki, a = Ki / (K * K), b = a * K, cc = c->soft_alpha * exp(-c->soft_beta * h);
      return a * (1.0 - c->soft_alpha + cc) + c->soft_beta * b * cc;
    }
    case FO_SOFT_EXPONENTIAL2: { const double a = Ki - K, b = Kc - Ki, cc = exp(a / b), d = Ki / (K * K); return d * cc + d * K * cc / b; }
    case FO_SOFT_POWER: {                                               /* getDerivPowerSoftening, :1446-1472 */
      if (K > Kc) return 0.0;
      const double a = Ki / K, b = 1.0 / (Kc - Ki), cc = (Kc - K) * b;
      const double d = pow(a, c->soft_beta - 1.0), e = pow(cc, c->soft_alpha - 1.0);
      return d * e * ((a / K) * c->soft_beta * cc + a * c->soft_alpha * b);
    }
    case FO_SOFT_HYPERBOLIC: return c->soft_b / (1.0 + c->soft_b * (K - Ki)) / (1.0 + c->soft_b * (K - Ki));   /* :1370-1380 */
    case FO_SOFT_EXPONENTIAL2REG: return 0.0;       /* not in the one-argument getdOmegadKappa (:712-761) */
    case FO_SOFT_EXPOENERGY: {                                          /* :1512-1523 */
      if (K < Ki) return 0.0;
      const double tem1 = pow((K - Ki) / Kc, c->soft_s), tem2 = pow((K - Ki) / Kc, c->soft_s - 1);
      return exp(-tem1) * tem2 * c->soft_s / Kc;
    }
    case FO_SOFT_POLYNOMIAL: {                      /* :1553-1574, verbatim: d is built from the un-normalised a = Ki - K */
      if (K < Ki) return 0.0;
      const double a = (Ki - K), b = (Kc - Ki), cc = a / b, d = 1.0 + (pow(a, 2.0) * (2.0 * a - 3.0));
      return -Ki / (K * K) * (2.0 * K * (cc / b * (2.0 * cc - 3.0) + pow(cc, 2.0) / b) - d);
    }
    default: {                                                          /* getDerivMultilinearSoftening, :1610-1633 */
      const double Kp = c->kappa_p;
      if (K < Ki) return 0.0;
      if (K > Kp) return Ki / (K * K);
      return (K < Kc) ? 1.0 / (Kc - Kp) : 0.0;                          /* getDerivLinear2Softening (Kp, Kc, K) */
    }
  }
}

/* ---- Mazars equivalent strain (DamageMaterial.cpp:618-642, getdEquidEpsMazars_ :886-1032), with what it calls:
 * Invariants (src/util/Invariants.cpp:35-224) and solveCubicEqn (src/util/MathUtils.cpp:112-182), constants of
 * src/util/Constants.cpp:13-18 -- PI is the reference's truncated literal, EPS = 1e-16 ---- */
static const double REF_PI = 3.14159265;
static void strain_v6(int rank, const double* strain, double* v6) {      /* Invariants::Invariants, isStrain = true */
  v6[0] = strain[0]; v6[1] = strain[1]; v6[2] = strain[2]; v6[3] = 0.5 * strain[3]; v6[4] = 0.0; v6[5] = 0.0;
  if (rank == 3) { v6[4] = 0.5 * strain[4]; v6[5] = 0.5 * strain[5]; }
}
static double inv_I1(const double* v) { return v[0] + v[1] + v[2]; }
static double inv_I2(const double* v) { return v[0] * v[1] + v[1] * v[2] + v[2] * v[0] - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]); }
static double inv_I3(const double* v) {
  return v[0] * v[1] * v[2] + 2.0 * v[3] * v[4] * v[5] - v[0] * v[4] * v[4] - v[1] * v[5] * v[5] - v[2] * v[3] * v[3];
}
static int solve_cubic(double* ans, double a, double b, double c, double d) {   /* MathUtils.cpp:112-182 */
  const double EPS = 1.e-16, RAD_120 = 0.6666666666666667 * REF_PI;
  ans[0] = ans[1] = ans[2] = NAN;
  if (fabs(a) < EPS) {
    if (fabs(b) < EPS) { ans[0] = -d / c; return 1; }
    double D = c * c - 4.0 * b * d;
    if (D < 0.0) return 0;
    D = sqrt(D);
    ans[0] = 0.5 * (-c + D) / b; ans[1] = 0.5 * (-c - D) / b;
    return 2;
  }
  const double kk = 1.0 / a, aa = b * kk, bb = c * kk, cc = d * kk, aa3 = aa / 3.0;
  const double q = (aa * aa - 3.0 * bb) / 9.0;
  const double r = (2.0 * aa * aa * aa - 9.0 * aa * bb + 27.0 * cc) / 54.0;
  double help = r / sqrt(q * q * q);
  if (fabs(help) > 1.0) help = (help < 0) ? -1.0 : 1.0;
  const double phi = acos(help), p = sqrt(q);
  ans[0] = -2.0 * p * cos(phi / 3.0) - aa3;
  ans[1] = -2.0 * p * cos(phi / 3.0 + RAD_120) - aa3;
  ans[2] = -2.0 * p * cos(phi / 3.0 - RAD_120) - aa3;
  return 3;
}
static void principal_values(const double* v6, double* roots) {          /* Invariants::getPrincipalValues */
  const int n = solve_cubic(roots, 1.0, -inv_I1(v6), inv_I2(v6), -inv_I3(v6));
  for (int i = n; i < 3; i++) roots[i] = 0.0;
}
static double mazars_eq(int rank, const double* strain) {                /* DamageMaterial.cpp:618-642 */
  double v6[6], princ[3], s = 0.0;
  strain_v6(rank, strain, v6);
  principal_values(v6, princ);
  for (int i = 0; i < 3; i++) if (princ[i] > 0) s += princ[i] * princ[i];
  return sqrt(s);
}
static void mazars_deriv(const fo_config* c, int rank, const double* strain, double* ret) {   /* :886-1032 */
  const int n = (rank == 3) ? 6 : 4;
  for (int i = 0; i < n; i++) ret[i] = 0.0;
  if (rank == 3) {
    double v[6], princ[3];
    strain_v6(3, strain, v);
    principal_values(v, princ);
    const double I1 = inv_I1(v), I2 = inv_I2(v);
    const double dI1[6] = { 1.0, 1.0, 1.0, 0.0, 0.0, 0.0 };
    const double dI2[6] = { v[1] + v[2], v[2] + v[0], v[0] + v[1], -2. * v[3], -2. * v[4], -2. * v[5] };
    const double dI3[6] = { v[1] * v[2] - v[4] * v[4], v[2] * v[0] - v[5] * v[5], v[0] * v[1] - v[3] * v[3],
                            2.0 * (v[4] * v[5] - v[2] * v[3]), 2.0 * (v[5] * v[3] - v[0] * v[4]), 2.0 * (v[3] * v[4] - v[1] * v[5]) };
    double epsBar = 0.0;
    for (int i = 0; i < 3; i++) if (princ[i] > 0) epsBar += princ[i] * princ[i];
    epsBar = sqrt(epsBar);
    if (epsBar < 1.e-16) return;
    for (int i = 0; i < 3; i++) {
      const double vi = princ[i];
      if (vi > 0) {
        const double temp = 1.0 / (3.0 * vi * vi - 2.0 * I1 * vi + I2);
        for (int k = 0; k < 6; k++) ret[k] += vi * (temp * (vi * vi * dI1[k] - vi * dI2[k] + dI3[k]));
      }
    }
    for (int k = 0; k < 6; k++) ret[k] *= (1.0 / epsBar);
    return;
  }
  const double exx = strain[0], eyy = strain[1], exy = 0.5 * strain[3];
  const double exxMeyy = exx - eyy, exxPeyy = exx + eyy;
  const double poi = c->poisson, poiD = poi / (poi - 1.0);
  double d = sqrt(exxMeyy * exxMeyy + 4.0 * exy * exy);
  double p0 = 0.5 * (exxPeyy + d), p1 = exxPeyy - p0, p2 = (c->state == FO_PLANE_STRESS) ? poiD * exxPeyy : 0.0;
  p0 = p0 < 0. ? 0. : p0; p1 = p1 < 0. ? 0. : p1; p2 = p2 < 0. ? 0. : p2;
  const double den = sqrt(p0 * p0 + p1 * p1 + p2 * p2);
  if (den < 1e-08) return;
  const double denInv = 1. / den;
  const double dedprin0 = p0 * denInv, dedprin1 = p1 * denInv, dedprin2 = p2 * denInv;
  if (d == 0.0) d = 1.0;
  const double fac = 0.5 / d;
  const double de1dexx = 0.5 + fac * exxMeyy, de1deyy = 1.0 - de1dexx, de1dexy = 2.0 * fac * exy;
  const double de2dexx = de1deyy, de2deyy = de1dexx, de2dexy = -de1dexy;
  const double de3dexx = poiD, de3deyy = de3dexx;
  ret[0] = dedprin0 * de1dexx + dedprin1 * de2dexx + dedprin2 * de3dexx;
  ret[1] = dedprin0 * de1deyy + dedprin1 * de2deyy + dedprin2 * de3deyy;
  ret[3] = dedprin0 * de1dexy + dedprin1 * de2dexy;
}

/* ---- von Mises equivalent strain (DamageMaterial.cpp:609-617, :667-684 with VonMisesStrain.cpp:34-116 and
 * Invariants::getJ2 / getGradI1 / getGradJ2, Invariants.cpp:111-202).  Rank 3 only: for a 4-component strain the
 * reference assigns the 6-vector of :683 to a 4-vector.  The gradient is the one with respect to the tensor components
 * (2 v6[3] = gamma_xy for the shear entries), as the reference returns it. ---- */
static void vonmises_abc(const fo_config* cfg, double* a, double* b, double* c) {       /* VonMisesStrain::init */
  const double k = cfg->eta, nu = cfg->poisson;
  *a = (k - 1.0) / (1.0 - 2.0 * nu); *b = 0.5 / k; *c = 12.0 * k / (1.0 + 2.0 * nu + nu * nu);
}
static double inv_J2(const double* v) {
  return (1.0 / 3.0) * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2] - v[0] * v[1] - v[1] * v[2] - v[2] * v[0]) +
         v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
}
static double vonmises_eq(const fo_config* cfg, int rank, const double* strain) {
  double v[6], a, b, c;
  strain_v6(rank, strain, v); vonmises_abc(cfg, &a, &b, &c);
  const double I1 = inv_I1(v), J2 = inv_J2(v);
  return b * (a * I1 + sqrt(a * a * I1 * I1 + c * J2));
}
static void vonmises_deriv(const fo_config* cfg, int rank, const double* strain, double* ret) {
  double v[6], a, b, c;
  strain_v6(rank, strain, v); vonmises_abc(cfg, &a, &b, &c);
  const double I1 = inv_I1(v), J2 = inv_J2(v);
  double d = sqrt(a * a * I1 * I1 + c * J2);
  if (d == 0.0) d = 1.0;
  const double e = b / d, deqdI1 = a * b + e * a * a * I1, deqdJ2 = e * c * 0.5;
  const double gJ2[6] = { (1.0 / 3.0) * (2.0 * v[0] - v[1] - v[2]), (1.0 / 3.0) * (2.0 * v[1] - v[2] - v[0]),
                          (1.0 / 3.0) * (2.0 * v[2] - v[0] - v[1]), 2.0 * v[3], 2.0 * v[4], 2.0 * v[5] };
  for (int i = 0; i < 6; i++) ret[i] = deqdI1 * (i < 3 ? 1.0 : 0.0) + deqdJ2 * gJ2[i];
}
int fo_vonmises(const fo_config* cfg, const double* strain6, double* eq, double* deriv) {   /* test hook */
  *eq = vonmises_eq(cfg, 3, strain6);
  vonmises_deriv(cfg, 3, strain6, deriv);
  return 0;
}

/* test hook: Mazars equivalent strain, its derivative and the principal values of a STRAIN_COUNTS[rank] strain vector */
int fo_mazars(const fo_config* cfg, int rank, const double* strain, double* eq, double* deriv, double* princ) {
  if (rank != 2 && rank != 3) FAIL("rank %d", rank);
  double v6[6];
  strain_v6(rank, strain, v6);
  principal_values(v6, princ);
  *eq = mazars_eq(rank, strain);
  mazars_deriv(cfg, rank, strain, deriv);
  return 0;
}

/* GradientDamageModel::getMatrix_ (GradientDamageModel.cpp:368-588) with DamageMaterial::update (:356-440).
 * hist_pre / hist_new: kappa, negative when the loading flag is set (preHist_.eqveps / preHist_.loading). */
static int assemble_gd(const fo_config* cfg, const double* coords, const int32_t* conn, int64_t el_begin, int64_t el_end,
                       const int64_t* row_ptr, const int32_t* col_idx, const double* state,
                       const double* hist_pre, double* hist_new, double* strain_cache, double* Kval, double* force) {
  shape_t sh;
  if (shape_init(&sh, cfg->shape)) return -1;
  if (cfg->material != FO_DAMAGE) FAIL("GradientDamage needs the Damage material");
  if (cfg->eqv_strain < FO_EQV_ENERGY || cfg->eqv_strain > FO_EQV_VONMISES) FAIL("unexpected definition of equivalent strain!!!");
  if (cfg->eqv_strain == FO_EQV_VONMISES) {
    if (sh.rank != 3) FAIL("eqvStrain = vonMises: rank 3 only (DamageMaterial.cpp:677-683 returns a 6-vector)");
    if (!(cfg->eta > 0.0)) FAIL("eqvStrain = vonMises needs eta > 0");
  }
  const int rank = sh.rank, nodeCount = sh.npe, ipCount = sh.nip;
  const int strCount = STRAIN_COUNTS[rank];
  const int dispCount = nodeCount * rank, eqvCount = nodeCount, ndofpn = rank + 1, srows = strCount + 1;
  const int mbuilder = (Kval != NULL);
  const double c = cfg->length_scale;
  double stiff0[MAXSTR * MAXSTR];
  if (hooke_stiff(rank, cfg->state, cfg->young, cfg->poisson, stiff0)) return -1;   /* giveElasticMatrix, :445 */

  static __thread double grads[MAXNIP][MAXNPE][3];
  static __thread double bd[MAXSTR * MAXND], DB[MAXSTR * MAXND];
  static __thread double elemMat1[MAXND * MAXND], elemMat2[MAXND * MAXNPE], elemMat3[MAXNPE * MAXND], elemMat4[MAXNPE * MAXNPE];
  double xc[MAXNPE][3], ipWeights[MAXNIP], disp[MAXND], eqv[MAXNPE];
  double stiff[MAXSTR * MAXSTR], stress[MAXSTR], strain[MAXSTR], d0e[MAXSTR], deq[MAXSTR];
  double elemForce1[MAXND], elemForce2[MAXNPE];
  int64_t dispDofs[MAXND], eqvDofs[MAXNPE];

  for (int64_t ie = el_begin; ie < el_end; ie++) {
    const int32_t* inodes = conn + ie * nodeCount;
    for (int a = 0; a < nodeCount; a++) {
      for (int i = 0; i < 3; i++) xc[a][i] = (i < rank) ? coords[(int64_t)inodes[a] * rank + i] : 0.0;
      for (int i = 0; i < rank; i++) dispDofs[a * rank + i] = (int64_t)inodes[a] * ndofpn + i;
      eqvDofs[a] = (int64_t)inodes[a] * ndofpn + rank;
    }
    if (shape_gradients(&sh, (const double (*)[3])xc, grads, ipWeights)) return -1;
    for (int i = 0; i < dispCount; i++) disp[i] = state[dispDofs[i]];
    for (int a = 0; a < eqvCount; a++) eqv[a] = state[eqvDofs[a]];
    memset(elemForce1, 0, sizeof elemForce1); memset(elemForce2, 0, sizeof elemForce2);
    memset(elemMat1, 0, sizeof(double) * dispCount * dispCount); memset(elemMat2, 0, sizeof(double) * dispCount * eqvCount);
    memset(elemMat3, 0, sizeof(double) * eqvCount * dispCount); memset(elemMat4, 0, sizeof(double) * eqvCount * eqvCount);

    for (int ip = 0; ip < ipCount; ip++) {
      const int64_t ipoint = ie * ipCount + ip;
      fill_B(rank, nodeCount, (const double (*)[3])grads[ip], bd);
      for (int cc = 0; cc < strCount; cc++) {
        double s = 0.0;
        for (int j = 0; j < dispCount; j++) s += bd[cc * dispCount + j] * disp[j];
        strain[cc] = s;
      }
      const double* Nip = sh.N[ip];
      const double eqvIp = dotn(eqvCount, Nip, eqv);                     /* strain_(strCount, ipoint), :524 */
      if (strain_cache) {
        for (int cc = 0; cc < strCount; cc++) strain_cache[ipoint * srows + cc] = strain[cc];
        strain_cache[ipoint * srows + strCount] = eqvIp;
      }
      /* material_->update (DamageMaterial.cpp:356-440), nonlocal branch: equiStrain = strain[n] */
      const double kappa0 = fabs(hist_pre[ipoint]);
      const int load0 = hist_pre[ipoint] < 0.0;
      const double f = eqvIp - kappa0;
      double kappa; int load;
      if (f == 0.0) { kappa = kappa0; load = load0; }
      else if (f > 0.0) { kappa = eqvIp; load = (kappa < cfg->kappa_i) ? 0 : 1; }
      else { kappa = kappa0; load = 0; }
      const double omega = damage_evolution(cfg, kappa);
      for (int i = 0; i < strCount * strCount; i++) stiff[i] = (1.0 - omega) * stiff0[i];
      matvec(strCount, stiff, strain, stress);
      hist_new[ipoint] = load ? -kappa : kappa;

      const double wip = ipWeights[ip];
      matvec(strCount, stiff0, strain, d0e);                              /* stiff0 eps (:557); = dEpsBardEps for Energy, :700-705 */
      if (cfg->eqv_strain == FO_EQV_MAZARS) mazars_deriv(cfg, rank, strain, deq);   /* getdEpsBardEps (:546) */
      else if (cfg->eqv_strain == FO_EQV_VONMISES) vonmises_deriv(cfg, rank, strain, deq);
      else for (int cc = 0; cc < strCount; cc++) deq[cc] = d0e[cc];
      if (mbuilder) {
        for (int cc = 0; cc < strCount; cc++) for (int j = 0; j < dispCount; j++) {      /* elemMat1 += wip bdt stiff bd */
          double s = 0.0;
          for (int k = 0; k < strCount; k++) s += stiff[cc * strCount + k] * bd[k * dispCount + j];
          DB[cc * dispCount + j] = s;
        }
        for (int i = 0; i < dispCount; i++) for (int j = 0; j < dispCount; j++) {
          double s = 0.0;
          for (int cc = 0; cc < strCount; cc++) s += bd[cc * dispCount + i] * DB[cc * dispCount + j];
          elemMat1[i * dispCount + j] += wip * s;
        }
        for (int a = 0; a < eqvCount; a++) for (int j = 0; j < dispCount; j++) {         /* elemMat3 -= wip (N (x) dEpsBardEps) bd */
          double s = 0.0;
          for (int cc = 0; cc < strCount; cc++) s += deq[cc] * bd[cc * dispCount + j];
          elemMat3[a * dispCount + j] -= wip * (Nip[a] * s);
        }
        if (load) {                                                        /* isLoading(ipoint): newHist_.loading, :549 */
          const double dOmega = damage_derivative(cfg, kappa);
          for (int i = 0; i < dispCount; i++) {
            double s = 0.0;
            for (int cc = 0; cc < strCount; cc++) s += bd[cc * dispCount + i] * d0e[cc];
            for (int b = 0; b < eqvCount; b++) elemMat2[i * eqvCount + b] -= wip * dOmega * (s * Nip[b]);
          }
        }
      }
      for (int a = 0; a < eqvCount; a++) for (int b = 0; b < eqvCount; b++) {            /* elemMat4, always (:563) */
        double s = 0.0;
        for (int i = 0; i < rank; i++) s += grads[ip][a][i] * grads[ip][b][i];
        elemMat4[a * eqvCount + b] += wip * (Nip[a] * Nip[b] + c * s);
      }
      const double equiStrain = (cfg->eqv_strain == FO_EQV_MAZARS) ? mazars_eq(rank, strain)
                              : (cfg->eqv_strain == FO_EQV_VONMISES) ? vonmises_eq(cfg, rank, strain)
                                                                    : 0.5 * dotn(strCount, strain, d0e);   /* getEquiStrain (:602-652) */
      for (int i = 0; i < dispCount; i++) {
        double s = 0.0;
        for (int cc = 0; cc < strCount; cc++) s += bd[cc * dispCount + i] * stress[cc];
        elemForce1[i] += wip * s;
      }
      for (int a = 0; a < eqvCount; a++) elemForce2[a] -= wip * equiStrain * Nip[a];
    }
    for (int a = 0; a < eqvCount; a++) {                                   /* elemForce2 += elemMat4 eqv (:577) */
      double s = 0.0;
      for (int b = 0; b < eqvCount; b++) s += elemMat4[a * eqvCount + b] * eqv[b];
      elemForce2[a] += s;
    }
    if (mbuilder) {
      if (add_block(row_ptr, col_idx, Kval, dispCount, dispDofs, dispCount, dispDofs, elemMat1)) return -1;
      if (add_block(row_ptr, col_idx, Kval, dispCount, dispDofs, eqvCount, eqvDofs, elemMat2)) return -1;
      if (add_block(row_ptr, col_idx, Kval, eqvCount, eqvDofs, dispCount, dispDofs, elemMat3)) return -1;
      if (add_block(row_ptr, col_idx, Kval, eqvCount, eqvDofs, eqvCount, eqvDofs, elemMat4)) return -1;
    }
    for (int i = 0; i < dispCount; i++) force[dispDofs[i]] += elemForce1[i];
    for (int a = 0; a < eqvCount; a++) force[eqvDofs[a]] += elemForce2[a];
  }
  return 0;
}

/* ---- GradientDamageModel's tables (GET_TABLE, GradientDamageModel.cpp:884-948) ---- */

/* DamageMaterial::update (DamageMaterial.cpp:356-437) at one point, the secant matrix left out: `m_entries` is the size
 * of the strain vector the caller hands in.  strCount + 1 entries: the last one is the nonlocal equivalent strain
 * (:374-377); strCount entries: "local damage models", the equivalent strain is computed from the strain (:370-373).
 * hist_pre / hist_new as in assemble_gd (kappa, negative when the loading flag is set); newHist_ is rewritten (:434-435). */
static void damage_update_point(const fo_config* cfg, int rank, const double* stiff0, const double* strain, int m_entries,
                                double hist_pre, double* stress, double* hist_new) {
  const int n = STRAIN_COUNTS[rank];
  double equiStrain;
  if (n == m_entries) {
    double d0e[MAXSTR];
    matvec(n, stiff0, strain, d0e);
    equiStrain = (cfg->eqv_strain == FO_EQV_MAZARS) ? mazars_eq(rank, strain)
               : (cfg->eqv_strain == FO_EQV_VONMISES) ? vonmises_eq(cfg, rank, strain)
                                                      : 0.5 * dotn(n, strain, d0e);       /* getEquiStrain (:602-652) */
  } else equiStrain = strain[n];
  const double kappa0 = fabs(hist_pre);
  const int load0 = hist_pre < 0.0;
  const double f = equiStrain - kappa0;
  double kappa; int load;
  if (f == 0.0) { kappa = kappa0; load = load0; }
  else if (f > 0.0) { kappa = equiStrain; load = (kappa < cfg->kappa_i) ? 0 : 1; }
  else { kappa = kappa0; load = 0; }
  const double omega = damage_evolution(cfg, kappa);
  double stiff[MAXSTR * MAXSTR];
  for (int i = 0; i < n * n; i++) stiff[i] = (1.0 - omega) * stiff0[i];
  matvec(n, stiff, strain, stress);
  *hist_new = load ? -kappa : kappa;
}

static int gd_check(const fo_config* cfg, const shape_t* sh) {
  if (cfg->model != FO_GRADIENT_DAMAGE || cfg->material != FO_DAMAGE) FAIL("GradientDamage tables need the GradientDamage model with the Damage material");
  if (cfg->eqv_strain == FO_EQV_VONMISES && sh->rank != 3) FAIL("eqvStrain = vonMises: rank 3 only");
  return 0;
}

/* getStress_ (GradientDamageModel.cpp:956-1063): per Gauss point `material_->update(stressIp, stiff, strain_(ALL,ipoint),
 * ipoint)` with all strCount + 1 rows of the cache, so the nonlocal branch; N (x) stressIp summed per element, then
 * added to the node rows.  strain: [nip_total][strCount + 1].  hist_new is rewritten, as the reference's update does. */
int fo_gd_stress_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain,
                       const double* hist_pre, double* hist_new, double* table, double* weights) {
  shape_t s;
  if (shape_init(&s, cfg->shape)) return -1;
  if (gd_check(cfg, &s)) return -1;
  const int str = STRAIN_COUNTS[s.rank], srows = str + 1;
  double stiff0[MAXSTR * MAXSTR];
  if (hooke_stiff(s.rank, cfg->state, cfg->young, cfg->poisson, stiff0)) return -1;
  int64_t ipoint = 0;
  for (int64_t ie = 0; ie < nel; ie++) {
    double ndStress[MAXNPE][6], ndWeights[MAXNPE];
    for (int a = 0; a < s.npe; a++) { ndWeights[a] = 0.0; for (int c = 0; c < str; c++) ndStress[a][c] = 0.0; }
    for (int ip = 0; ip < s.nip; ip++, ipoint++) {
      double stressIp[MAXSTR];
      damage_update_point(cfg, s.rank, stiff0, strain + ipoint * srows, srows, hist_pre[ipoint], stressIp, hist_new + ipoint);
      for (int a = 0; a < s.npe; a++) {
        for (int c = 0; c < str; c++) ndStress[a][c] += s.N[ip][a] * stressIp[c];
        ndWeights[a] += s.N[ip][a];
      }
    }
    for (int a = 0; a < s.npe; a++) {
      const int64_t n = conn[ie * s.npe + a];
      weights[n] += ndWeights[a];
      for (int c = 0; c < str; c++) table[n * str + c] += ndStress[a][c];
    }
  }
  return 0;
}

/* getStrain_ (GradientDamageModel.cpp:1071-1177): strCount + 1 columns, the last one "e_eq" (:1135), all rows of the
 * cache extrapolated (:1161).  The column switch (:1097-1131) has no case 4: in 2-D (strCount = 4) the reference throws
 * "unexpected number of strain components: 4", and so does this. */
int fo_gd_strain_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain,
                       double* table, double* weights) {
  shape_t s;
  if (shape_init(&s, cfg->shape)) return -1;
  if (gd_check(cfg, &s)) return -1;
  const int str = STRAIN_COUNTS[s.rank], srows = str + 1;
  if (str != 1 && str != 3 && str != 6) FAIL("unexpected number of strain components: %d", str);
  int64_t ipoint = 0;
  for (int64_t ie = 0; ie < nel; ie++) {
    double ndStrain[MAXNPE][7], ndWeights[MAXNPE];
    for (int a = 0; a < s.npe; a++) { ndWeights[a] = 0.0; for (int c = 0; c < srows; c++) ndStrain[a][c] = 0.0; }
    for (int ip = 0; ip < s.nip; ip++, ipoint++)
      for (int a = 0; a < s.npe; a++) {
        for (int c = 0; c < srows; c++) ndStrain[a][c] += s.N[ip][a] * strain[ipoint * srows + c];
        ndWeights[a] += s.N[ip][a];
      }
    for (int a = 0; a < s.npe; a++) {
      const int64_t n = conn[ie * s.npe + a];
      weights[n] += ndWeights[a];
      for (int c = 0; c < srows; c++) table[n * srows + c] += ndStrain[a][c];
    }
  }
  return 0;
}

/* getDamage_ (GradientDamageModel.cpp:1184-1241): one column "hist"; hist = material_->giveOmega() =
 * damageEvolution_(newHist_.eqveps[i]) (DamageMaterial.cpp:533-545); the ABSOLUTE values of the shape functions weigh
 * both the values and the weights (:1233-1234). */
int fo_gd_damage_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* hist_new,
                       double* table, double* weights) {
  shape_t s;
  if (shape_init(&s, cfg->shape)) return -1;
  if (gd_check(cfg, &s)) return -1;
  int64_t ipoint = 0;
  for (int64_t ie = 0; ie < nel; ie++) {
    double ndHistory[MAXNPE], ndWeights[MAXNPE];
    for (int a = 0; a < s.npe; a++) { ndWeights[a] = 0.0; ndHistory[a] = 0.0; }
    for (int ip = 0; ip < s.nip; ip++, ipoint++) {
      const double h = damage_evolution(cfg, fabs(hist_new[ipoint]));
      for (int a = 0; a < s.npe; a++) { ndHistory[a] += fabs(s.N[ip][a]) * h; ndWeights[a] += fabs(s.N[ip][a]); }
    }
    for (int a = 0; a < s.npe; a++) {
      const int64_t n = conn[ie * s.npe + a];
      weights[n] += ndWeights[a];
      table[n] += ndHistory[a];
    }
  }
  return 0;
}

/* getOutputData_ (GradientDamageModel.cpp:1254-1365): point values [strain.tensor | stress.tensor | damage] (TbFiller's
 * announce order, :1270-1273: 2 strCount + 1 of them).  The damage is giveOmega() taken BEFORE the loop (:1316); the
 * stress comes from `update(stress, stiff, strain, ipoint)` with strain = the first strCount rows of the cache (:1340,
 * :1343): stress and strain have the same size, so the material takes its LOCAL branch and computes the equivalent
 * strain from the strain itself -- and rewrites newHist_ with the outcome.  cols: TbFiller's i2table (:1347). */
int fo_gd_output_data(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain,
                      const double* hist_pre, double* hist_new, int ncols, const int32_t* cols,
                      double* table, double* weights) {
  shape_t s;
  if (shape_init(&s, cfg->shape)) return -1;
  if (gd_check(cfg, &s)) return -1;
  const int str = STRAIN_COUNTS[s.rank], srows = str + 1, ntype = 2 * str + 1;
  for (int j = 0; j < ncols; j++) if (cols[j] < 0 || cols[j] >= ntype) FAIL("column %d out of range", cols[j]);
  double stiff0[MAXSTR * MAXSTR];
  if (hooke_stiff(s.rank, cfg->state, cfg->young, cfg->poisson, stiff0)) return -1;
  const int64_t npoint = nel * s.nip;
  double* hist = (double*)malloc(sizeof(double) * (size_t)(npoint > 0 ? npoint : 1));
  if (!hist) FAIL("out of memory");
  for (int64_t i = 0; i < npoint; i++) hist[i] = damage_evolution(cfg, fabs(hist_new[i]));
  int64_t igpoint = 0;
  for (int64_t ie = 0; ie < nel; ie++) {
    double ndValuesOut[MAXNPE][13], ndWeights[MAXNPE], ipValues[13];
    for (int a = 0; a < s.npe; a++) { ndWeights[a] = 0.0; for (int j = 0; j < ncols; j++) ndValuesOut[a][j] = 0.0; }
    for (int ip = 0; ip < s.nip; ip++, igpoint++) {
      for (int c = 0; c < str; c++) ipValues[c] = strain[igpoint * srows + c];
      ipValues[2 * str] = hist[igpoint];
      damage_update_point(cfg, s.rank, stiff0, ipValues, str, hist_pre[igpoint], ipValues + str, hist_new + igpoint);
      for (int a = 0; a < s.npe; a++) {
        for (int j = 0; j < ncols; j++) ndValuesOut[a][j] += s.N[ip][a] * ipValues[cols[j]];
        ndWeights[a] += s.N[ip][a];
      }
    }
    for (int a = 0; a < s.npe; a++) {
      const int64_t n = conn[ie * s.npe + a];
      weights[n] += ndWeights[a];
      for (int j = 0; j < ncols; j++) table[n * ncols + j] += ndValuesOut[a][j];
    }
  }
  free(hist);
  return 0;
}

int fo_assemble(const fo_config* cfg,
                int nnode, const double* coords,
                int64_t nel, const int32_t* conn,
                int64_t el_begin, int64_t el_end,
                const int64_t* row_ptr, const int32_t* col_idx,
                const double* state, const double* state0, const double* state00,
                double dt_, double dt0_,
                const double* hist_pre, double* hist_new,
                double* strain_cache,
                double* Kval, double* force, double* err_ext) {
  (void)nnode; (void)nel;
  if (cfg->model == FO_GRADIENT_DAMAGE)
    return assemble_gd(cfg, coords, conn, el_begin, el_end, row_ptr, col_idx, state, hist_pre, hist_new, strain_cache,
                       Kval, force);
  shape_t sh;
  if (shape_init(&sh, cfg->shape)) return -1;
  const int rank = sh.rank, nodeCount = sh.npe, ipCount = sh.nip;
  const int strCount = STRAIN_COUNTS[rank];
  const int dispCount = nodeCount * rank, phiCount = nodeCount;
  const int model = cfg->model;
  const int isPF = (model != FO_LINEAR_ELASTICITY);
  const int ndofpn = fo_ndof_per_node(model, rank);
  const int srows = isPF ? strCount + 1 : strCount;
  const int mbuilder = (Kval != NULL);

  frac_t fr; memset(&fr, 0, sizeof fr);
  if (isPF && frac_init(cfg, &fr)) return -1;
  if (isPF && cfg->material == FO_HOOKE) FAIL("phase-field model needs a PhaseFractureMaterial");
  if (!isPF && cfg->material != FO_HOOKE) FAIL("LinearElasticity is restated with Hooke only");
  const double gc_ = cfg->gc, l0_ = cfg->l0, cw_ = fr.cw, eta_ = fr.eta, beta_ = cfg->beta, Psi0_ = fr.psi0;

  static __thread double grads[MAXNIP][MAXNPE][3];
  double xc[MAXNPE][3], ipWeights[MAXNIP];
  double disp[MAXND], phi[MAXNPE], phi0[MAXNPE], phiX[MAXNPE];
  double stiff[MAXSTR * MAXSTR], stress[MAXSTR], strain[MAXSTR], stressP[MAXSTR];
  double elemForce1[MAXND], elemForce2[MAXNPE];
  static __thread double elemMat1[MAXND * MAXND], elemMat2[MAXND * MAXNPE], elemMat3[MAXNPE * MAXND],
                         elemMat4[MAXNPE * MAXNPE];
  static __thread double bd[MAXSTR * MAXND], DB[MAXSTR * MAXND];
  double sPB[MAXND], BeBe[MAXNPE * MAXNPE];
  int64_t dispDofs[MAXND], phiDofs[MAXNPE];
  double errExt = 0.0;
  /* MicroPhaseFractureModel.cpp:533-538,577: these live outside the element loop, so values left by an earlier
   * Gauss point stay visible (Amor's stiffN in tension; dphidStrain in the `d > 1` branch).  Jem leaves new arrays
   * uninitialised; zero is assumed here. */
  double stiffP[MAXSTR * MAXSTR] = { 0 }, stiffN[MAXSTR * MAXSTR] = { 0 }, stressN[MAXSTR] = { 0 }, dphidStrain[MAXSTR] = { 0 };
  g_micro_stale = 0;

  for (int64_t ie = el_begin; ie < el_end; ie++) {
    const int32_t* inodes = conn + ie * nodeCount;
    for (int a = 0; a < nodeCount; a++) {
      for (int i = 0; i < 3; i++) xc[a][i] = (i < rank) ? coords[(int64_t)inodes[a] * rank + i] : 0.0;
      /* XDofSpace::getDofIndices: node-major, type-minor */
      for (int i = 0; i < rank; i++) dispDofs[a * rank + i] = (int64_t)inodes[a] * ndofpn + i;
      phiDofs[a] = (int64_t)inodes[a] * ndofpn + rank;
    }
    if (shape_gradients(&sh, (const double (*)[3])xc, grads, ipWeights)) return -1;

    for (int i = 0; i < dispCount; i++) disp[i] = state[dispDofs[i]];
    if (isPF) for (int a = 0; a < phiCount; a++) phi[a] = state[phiDofs[a]];
    if ((model == FO_PHASE_FRACTURE_EXT || model == FO_PHASE_FRACTURE_EXTIT))
      for (int a = 0; a < phiCount; a++) { phi0[a] = state0[phiDofs[a]]; phiX[a] = state00[phiDofs[a]]; }
    if (model == FO_MICRO_EXT || model == FO_MICRO_EXTIT)      /* phiEx = select(stateExt_, phiDofs) */
      for (int a = 0; a < phiCount; a++) phiX[a] = state00[phiDofs[a]];

    memset(elemForce1, 0, sizeof elemForce1); memset(elemForce2, 0, sizeof elemForce2);
    if (mbuilder) {
      memset(elemMat1, 0, sizeof(double) * dispCount * dispCount);
      memset(elemMat2, 0, sizeof(double) * dispCount * phiCount);
      memset(elemMat3, 0, sizeof(double) * phiCount * dispCount);
      memset(elemMat4, 0, sizeof(double) * phiCount * phiCount);
    }

    for (int ip = 0; ip < ipCount; ip++) {
      const int64_t ipoint = ie * ipCount + ip;          /* initializeIPMPMap_, PhaseFractureModel.cpp:1070-1093 */
      fill_B(rank, nodeCount, (const double (*)[3])grads[ip], bd);
      for (int c = 0; c < strCount; c++) {               /* matmul(strain, bd, disp) */
        double s = 0.0;
        for (int j = 0; j < dispCount; j++) s += bd[c * dispCount + j] * disp[j];
        strain[c] = s;
      }
      if (strain_cache) for (int c = 0; c < strCount; c++) strain_cache[ipoint * srows + c] = strain[c];
      const double wip = ipWeights[ip];

      if (!isPF) {
        /* LinearElasticityModel.cpp:411-420 */
        if (fo_material_update(cfg, rank, strain, 0.0, stress, stiff, NULL, NULL)) return -1;
      } else if (model == FO_MICRO_PHASE_FRACTURE) {
        /* MicroPhaseFractureModel.cpp:656-830 */
        const double* Nip = sh.N[ip];
        const double p_ = fr.p, a1_ = fr.a1, a2_ = fr.a2, a3_ = fr.a3;
        const double alpha = beta_ * gc_ / l0_;
        double Psi;
        if (material_update_split(cfg, rank, strain, stressP, stressN, stiffP, stiffN, &Psi)) return -1;
        Psi = (Psi > Psi0_) ? Psi : Psi0_;
        const double pf = dotn(phiCount, Nip, phi);
        const double d0 = hist_pre[ipoint];
        double d = d0;
        double gphi_n = pow(1.0 - d, p_);
        double gphi_d = gphi_n + a1_ * d + a1_ * a2_ * d * d + a1_ * a2_ * a3_ * d * d * d;
        double dgphi_n = -p_ * (pow(1.0 - d, p_ - 1));
        double dgphi_d = dgphi_n + a1_ + 2.0 * a1_ * a2_ * d + 3.0 * a1_ * a2_ * a3_ * d * d;
        double ddgphi_n = p_ * (p_ - 1.0) * (pow(1.0 - d, p_ - 2.0));
        double ddgphi_d = ddgphi_n + 2.0 * a1_ * a2_ + 6.0 * a1_ * a2_ * a3_ * d;
        double dgphi = (dgphi_n * gphi_d - gphi_n * dgphi_d) / (gphi_d * gphi_d);
        double ddgphi = ((ddgphi_n * gphi_d - gphi_n * ddgphi_d) * gphi_d - 2.0 *
                         (dgphi_n * gphi_d - gphi_n * dgphi_d) * dgphi_d) / (gphi_d * gphi_d * gphi_d);
        double dw = eta_ + 2.0 * (1.0 - eta_) * d;
        double ddw = 2.0 * (1.0 - eta_);
        double Res = dgphi * Psi + gc_ / (cw_ * l0_) * dw + alpha * (d - pf);
        double J = ddgphi * Psi + gc_ / (cw_ * l0_) * ddw + alpha;
        int iter = 0;
        while (fabs(Res) > 1.e-4 && iter < 100) {        /* jem::abs (using namespace jem, MicroPhaseFractureModel.h:58) */
          iter += 1;
          d -= Res / J;
          gphi_n = pow(1.0 - d, p_);
          gphi_d = gphi_n + a1_ * d + a1_ * a2_ * d * d + a1_ * a2_ * a3_ * d * d * d;
          dgphi_n = -p_ * (pow(1.0 - d, p_ - 1.0));
          dgphi_d = dgphi_n + a1_ + 2.0 * a1_ * a2_ * d + 3.0 * a1_ * a2_ * a3_ * d * d;
          ddgphi_n = p_ * (p_ - 1.0) * (pow(1.0 - d, p_ - 2.0));
          ddgphi_d = ddgphi_n + 2.0 * a1_ * a2_ + 6.0 * a1_ * a2_ * a3_ * d;
          dgphi = (dgphi_n * gphi_d - gphi_n * dgphi_d) / (gphi_d * gphi_d);
          ddgphi = ((ddgphi_n * gphi_d - gphi_n * ddgphi_d) * gphi_d - 2.0 *
                    (dgphi_n * gphi_d - gphi_n * dgphi_d) * dgphi_d) / (gphi_d * gphi_d * gphi_d);
          dw = eta_ + 2.0 * (1.0 - eta_) * d;
          ddw = 2.0 * (1.0 - eta_);
          Res = dgphi * Psi + gc_ / (cw_ * l0_) * dw + alpha * (d - pf);
          J = ddgphi * Psi + gc_ / (cw_ * l0_) * ddw + alpha;
        }
        double dphidd = 0.0;
        if (d > d0) {
          gphi_n = pow(1.0 - d, p_);                      /* gphi_d is NOT recomputed here (commented out, :744) */
          dgphi_n = -p_ * (pow(1.0 - d, p_ - 1.0));
          dgphi_d = dgphi_n + a1_ + 2.0 * a1_ * a2_ * d + 3.0 * a1_ * a2_ * a3_ * d * d;
          ddgphi_n = p_ * (p_ - 1.0) * (pow(1.0 - d, p_ - 2.0));
          ddgphi_d = ddgphi_n + 2.0 * a1_ * a2_ + 6.0 * a1_ * a2_ * a3_ * d;
          dgphi = (dgphi_n * gphi_d - gphi_n * dgphi_d) / (gphi_d * gphi_d);
          ddgphi = ((ddgphi_n * gphi_d - gphi_n * ddgphi_d) * gphi_d - 2.0 *
                    (dgphi_n * gphi_d - gphi_n * dgphi_d) * dgphi_d) / (gphi_d * gphi_d * gphi_d);
          ddw = 2.0 * (1.0 - eta_);
          const double den = ddgphi * Psi + gc_ / (cw_ * l0_) * ddw + alpha;
          for (int c = 0; c < strCount; c++) dphidStrain[c] = -(dgphi) / den * stressP[c];
          dphidd = alpha / den;
        } else if (d < 0.0) {
          d = 0.0;
          for (int c = 0; c < strCount; c++) dphidStrain[c] = 0.0;
          dphidd = 0.0;
        } else if (d > 1.0) {
          d = 1.0;                                        /* dphidStrain keeps its previous contents (:772-776) */
          dphidd = 0.0;
          g_micro_stale++;
        } else {
          d = d0;
          for (int c = 0; c < strCount; c++) dphidStrain[c] = 0.0;
          dphidd = 0.0;
        }
        gphi_n = pow(1.0 - d, p_);
        gphi_d = gphi_n + a1_ * d + a1_ * a2_ * d * d + a1_ * a2_ * a3_ * d * d * d;
        dgphi_n = -p_ * (pow(1.0 - d, p_ - 1.0));
        dgphi_d = dgphi_n + a1_ + 2.0 * a1_ * a2_ * d + 3.0 * a1_ * a2_ * a3_ * d * d;
        const double gphi = gphi_n / gphi_d;             /* no floor here (:795) */
        dgphi = (dgphi_n * gphi_d - gphi_n * dgphi_d) / (gphi_d * gphi_d);
        if (strain_cache) strain_cache[ipoint * srows + strCount] = gphi;
        hist_new[ipoint] = d;
        /* stiff = gphi*stiffP + stiffN (:804) plus the second elemMat1 term w*dgphi*bdt*(stressP (x) dphidStrain)*bd (:816),
         * folded into one matrix for the common B^T D B product below */
        for (int c = 0; c < strCount; c++) {
          stress[c] = gphi * stressP[c] + stressN[c];
          for (int k = 0; k < strCount; k++)
            stiff[c * strCount + k] = gphi * stiffP[c * strCount + k] + stiffN[c * strCount + k]
                                      + dgphi * (stressP[c] * dphidStrain[k]);
        }
        for (int a = 0; a < phiCount; a++) for (int b = 0; b < phiCount; b++) {
          double s = 0.0;
          for (int i = 0; i < rank; i++) s += grads[ip][a][i] * grads[ip][b][i];
          BeBe[a * phiCount + b] = s;
        }
        const double c2 = 2.0 * gc_ * l0_ / cw_;
        if (mbuilder) {
          for (int a = 0; a < phiCount; a++) for (int b = 0; b < phiCount; b++)
            elemMat4[a * phiCount + b] += wip * (alpha * (1.0 - dphidd) * (Nip[a] * Nip[b]) + c2 * BeBe[a * phiCount + b]);
          if (cfg->keep_off_diags) {
            for (int j = 0; j < dispCount; j++) {          /* matmul(dphidStrain, bd) */
              double s = 0.0;
              for (int c = 0; c < strCount; c++) s += dphidStrain[c] * bd[c * dispCount + j];
              sPB[j] = s;
            }
            for (int i = 0; i < dispCount; i++) for (int b = 0; b < phiCount; b++) {
              double s = 0.0;
              for (int c = 0; c < strCount; c++) s += bd[c * dispCount + i] * (stressP[c] * Nip[b]);
              elemMat2[i * phiCount + b] += wip * dgphi * dphidd * s;
            }
            for (int a = 0; a < phiCount; a++) for (int j = 0; j < dispCount; j++)
              elemMat3[a * dispCount + j] -= wip * alpha * (Nip[a] * sPB[j]);
          }
        }
        for (int a = 0; a < phiCount; a++) {
          double s = 0.0;
          for (int b = 0; b < phiCount; b++) s += BeBe[a * phiCount + b] * phi[b];
          elemForce2[a] += wip * (alpha * (pf - d) * Nip[a] + c2 * s);
        }
      } else if (model == FO_MICRO_EXT || model == FO_MICRO_EXTIT) {
        /* MicroPhaseFractureExtModel.cpp:656-940; ExtIt differs by errExt_ only (MicroPhaseFractureExtItModel.cpp:712) */
        const double* Nip = sh.N[ip];
        const double p_ = fr.p, a1_ = fr.a1, a2_ = fr.a2, a3_ = fr.a3;
        const double alpha = beta_ * gc_ / l0_;
        const double pf = dotn(phiCount, Nip, phi);
        double pfEx = dotn(phiCount, Nip, phiX);
        if (model == FO_MICRO_EXTIT) errExt += pow(pf - pfEx, 2.0);
        if (pfEx < 0.0) pfEx = 0.0; else if (pfEx > 1.0) pfEx = 1.0;
        double Psi;
        if (material_update_split(cfg, rank, strain, stressP, stressN, stiffP, stiffN, &Psi)) return -1;
        Psi = (Psi > Psi0_) ? Psi : Psi0_;
        const double d0 = hist_pre[ipoint];
        double d = d0;
        double gphi_n, gphi_d, dgphi_n, dgphi_d, ddgphi_n, ddgphi_d, dgphi, ddgphi, dw, ddw, Res, J;
#define MICRO_TERMS(target)                                                                              \
        gphi_n = pow(1.0 - d, p_);                                                                       \
        gphi_d = gphi_n + a1_ * d + a1_ * a2_ * d * d + a1_ * a2_ * a3_ * d * d * d;                     \
        dgphi_n = -p_ * (pow(1.0 - d, p_ - 1.0));                                                        \
        dgphi_d = dgphi_n + a1_ + 2.0 * a1_ * a2_ * d + 3.0 * a1_ * a2_ * a3_ * d * d;                   \
        ddgphi_n = p_ * (p_ - 1.0) * (pow(1.0 - d, p_ - 2.0));                                           \
        ddgphi_d = ddgphi_n + 2.0 * a1_ * a2_ + 6.0 * a1_ * a2_ * a3_ * d;                               \
        dgphi = (dgphi_n * gphi_d - gphi_n * dgphi_d) / (gphi_d * gphi_d);                               \
        ddgphi = ((ddgphi_n * gphi_d - gphi_n * ddgphi_d) * gphi_d - 2.0 *                               \
                  (dgphi_n * gphi_d - gphi_n * dgphi_d) * dgphi_d) / (gphi_d * gphi_d * gphi_d);         \
        dw = eta_ + 2.0 * (1.0 - eta_) * d;                                                              \
        ddw = 2.0 * (1.0 - eta_);                                                                        \
        Res = dgphi * Psi + gc_ / (cw_ * l0_) * dw + alpha * (d - (target));                             \
        J = ddgphi * Psi + gc_ / (cw_ * l0_) * ddw + alpha;
        MICRO_TERMS(pfEx)
        int iter = 0;
        while (fabs(Res) > 1.e-4 && iter < 100) { iter += 1; d -= Res / J; MICRO_TERMS(pfEx) }
        if (d < 0.0) d = 0.0; else if (d > 1.0) d = 1.0;
        d = (d > d0) ? d : d0;
        gphi_n = pow(1.0 - d, p_);
        gphi_d = gphi_n + a1_ * d + a1_ * a2_ * d * d + a1_ * a2_ * a3_ * d * d * d;
        const double gphi = gphi_n / gphi_d + 1.e-10;
        if (strain_cache) strain_cache[ipoint * srows + strCount] = gphi;
        hist_new[ipoint] = d;
        for (int c = 0; c < strCount; c++) {
          stress[c] = gphi * stressP[c] + stressN[c];
          for (int k = 0; k < strCount; k++) stiff[c * strCount + k] = gphi * stiffP[c * strCount + k] + stiffN[c * strCount + k];
        }
        MICRO_TERMS(pf)                                   /* second solve, for the current micromorphic variable */
        iter = 0;
        while (fabs(Res) > 1.e-4 && iter < 100) { iter += 1; d -= Res / J; MICRO_TERMS(pf) }
#undef MICRO_TERMS
        double dphidd = 0.0;
        if (d > d0) {
          const double den = ddgphi * Psi + gc_ / (cw_ * l0_) * ddw + alpha;
          for (int c = 0; c < strCount; c++) dphidStrain[c] = -(dgphi) / den * stressP[c];
          dphidd = alpha / den;
        } else if (d < 0.0) {
          d = 0.0;
          for (int c = 0; c < strCount; c++) dphidStrain[c] = 0.0;
        } else if (d > 1.0) {
          d = 1.0;                                        /* dphidStrain keeps its previous contents */
          g_micro_stale++;
        } else {
          d = d0;
          for (int c = 0; c < strCount; c++) dphidStrain[c] = 0.0;
        }
        for (int a = 0; a < phiCount; a++) for (int b = 0; b < phiCount; b++) {
          double s = 0.0;
          for (int i = 0; i < rank; i++) s += grads[ip][a][i] * grads[ip][b][i];
          BeBe[a * phiCount + b] = s;
        }
        const double c2 = 2.0 * gc_ * l0_ / cw_;
        if (mbuilder) {
          for (int j = 0; j < dispCount; j++) {            /* matmul(dphidStrain, bd) */
            double s = 0.0;
            for (int c = 0; c < strCount; c++) s += dphidStrain[c] * bd[c * dispCount + j];
            sPB[j] = s;
          }
          for (int a = 0; a < phiCount; a++) for (int j = 0; j < dispCount; j++)
            elemMat3[a * dispCount + j] -= wip * alpha * (Nip[a] * sPB[j]);
          for (int a = 0; a < phiCount; a++) for (int b = 0; b < phiCount; b++)
            elemMat4[a * phiCount + b] += wip * (alpha * (1.0 - dphidd) * (Nip[a] * Nip[b]) + c2 * BeBe[a * phiCount + b]);
        }
        for (int a = 0; a < phiCount; a++) {
          double s = 0.0;
          for (int b = 0; b < phiCount; b++) s += BeBe[a * phiCount + b] * phi[b];
          elemForce2[a] += wip * (alpha * (pf - d) * Nip[a] + c2 * s);
        }
      } else {
        const double* Nip = sh.N[ip];
        double pf = dotn(phiCount, Nip, phi);
        double gphi, dgphi, ddgphi, gmat;
        double pf0 = 0.0;
        if (model == FO_PHASE_FRACTURE) {
          degradation(&fr, pf, &gphi, &dgphi, &ddgphi);
          gmat = gphi;
        } else {
          /* PhaseFractureExtModel.cpp:656-711 / PhaseFractureExtItModel.cpp (diff: pfEx source, errExt_) */
          pf0 = dotn(phiCount, Nip, phi0);
          double pfEx;
          if (model == FO_PHASE_FRACTURE_EXT) {
            double pf00 = dotn(phiCount, Nip, phiX);
            if (dt0_ < 1.e-15) pfEx = pf0;
            else               pfEx = -pf00 * dt_ / dt0_ + pf0 * (dt_ + dt0_) / dt0_;
          } else {
            pfEx = dotn(phiCount, Nip, phiX);
            errExt += pow(pf - pfEx, 2.0);
          }
          if (pfEx < 0.0) pfEx = 0.0; else if (pfEx > 1.0) pfEx = 1.0;
          double d1, d2;
          degradation(&fr, pfEx, &gmat, &d1, &d2);       /* only g(pfEx)+1e-7 is used */
          double gdummy;
          degradation(&fr, pf, &gdummy, &dgphi, &ddgphi);
        }
        if (strain_cache) strain_cache[ipoint * srows + strCount] = gmat;
        double dw  = eta_ + 2.0 * (1 - eta_) * pf;
        double ddw = 2.0 * (1 - eta_);
        double Psi;
        if (fo_material_update(cfg, rank, strain, gmat, stress, stiff, &Psi, stressP)) return -1;
        double loading = 1.0;
        if (model == FO_PHASE_FRACTURE) {
          const double H0 = hist_pre[ipoint];            /* PhaseFractureModel.cpp:696-703 */
          if (Psi < H0) loading = 0.0;
          Psi = (Psi > H0) ? Psi : H0;
          hist_new[ipoint] = Psi;
        } else {
          Psi = (Psi > Psi0_) ? Psi : Psi0_;             /* PhaseFractureExtModel.cpp:726 */
        }
        /* be^T be  (mc2.matmul(bet, be)) */
        for (int a = 0; a < phiCount; a++) for (int b = 0; b < phiCount; b++) {
          double s = 0.0;
          for (int i = 0; i < rank; i++) s += grads[ip][a][i] * grads[ip][b][i];
          BeBe[a * phiCount + b] = s;
        }
        if (mbuilder) {
          const double c1 = gc_ / (cw_ * l0_) * ddw + ddgphi * Psi, c2 = 2.0 * gc_ * l0_ / cw_;
          for (int a = 0; a < phiCount; a++) for (int b = 0; b < phiCount; b++)
            elemMat4[a * phiCount + b] += wip * (c1 * (Nip[a] * Nip[b]) + c2 * BeBe[a * phiCount + b]);
          const int k2 = (model == FO_PHASE_FRACTURE) && cfg->keep_off_diags;
          const int k3 = (model == FO_PHASE_FRACTURE) ? cfg->keep_off_diags : 1;
          if (k2 || k3) {
            for (int j = 0; j < dispCount; j++) {          /* matmul(stressP, bd) */
              double s = 0.0;
              for (int c = 0; c < strCount; c++) s += stressP[c] * bd[c * dispCount + j];
              sPB[j] = s;
            }
            if (k2) for (int i = 0; i < dispCount; i++) for (int b = 0; b < phiCount; b++) {
              /* bdt * (stressP (x) Nip)  == (bdt*stressP) (x) Nip */
              double s = 0.0;
              for (int c = 0; c < strCount; c++) s += bd[c * dispCount + i] * (stressP[c] * Nip[b]);
              elemMat2[i * phiCount + b] += wip * dgphi * s;
            }
            if (k3) for (int a = 0; a < phiCount; a++) for (int j = 0; j < dispCount; j++)
              elemMat3[a * dispCount + j] += wip * loading * dgphi * (Nip[a] * sPB[j]);
          }
        }
        {
          const double cf = gc_ / (cw_ * l0_) * dw + dgphi * Psi, c2 = 2.0 * gc_ * l0_ / cw_;
          for (int a = 0; a < phiCount; a++) {
            double s = 0.0;
            for (int b = 0; b < phiCount; b++) s += BeBe[a * phiCount + b] * phi[b];
            elemForce2[a] += wip * (Nip[a] * cf + c2 * s);
          }
        }
        if ((model == FO_PHASE_FRACTURE_EXT || model == FO_PHASE_FRACTURE_EXTIT) && (pf0 - pf > 1.e-15)) {   /* PhaseFractureExtModel.cpp:752-758 */
          const double cp = gc_ / (cw_ * l0_) * beta_;
          if (mbuilder) for (int a = 0; a < phiCount; a++) for (int b = 0; b < phiCount; b++)
            elemMat4[a * phiCount + b] += wip * (cp * (Nip[a] * Nip[b]));
          for (int a = 0; a < phiCount; a++) elemForce2[a] -= wip * (cp * (pf0 - pf)) * Nip[a];
        }
      }

      if (mbuilder) {
        /* elemMat1 += wip * bdt * stiff * bd   (dense 3-chain) */
        for (int c = 0; c < strCount; c++) for (int j = 0; j < dispCount; j++) {
          double s = 0.0;
          for (int k = 0; k < strCount; k++) s += stiff[c * strCount + k] * bd[k * dispCount + j];
          DB[c * dispCount + j] = s;
        }
        for (int i = 0; i < dispCount; i++) for (int j = 0; j < dispCount; j++) {
          double s = 0.0;
          for (int c = 0; c < strCount; c++) s += bd[c * dispCount + i] * DB[c * dispCount + j];
          elemMat1[i * dispCount + j] += wip * s;
        }
      }
      for (int i = 0; i < dispCount; i++) {              /* elemForce1 += wip * bdt * stress */
        double s = 0.0;
        for (int c = 0; c < strCount; c++) s += bd[c * dispCount + i] * stress[c];
        elemForce1[i] += wip * s;
      }
    }

    if (mbuilder) {
      if (add_block(row_ptr, col_idx, Kval, dispCount, dispDofs, dispCount, dispDofs, elemMat1)) return -1;
      if (isPF) {
        const int plain = (model == FO_PHASE_FRACTURE) || (model == FO_MICRO_PHASE_FRACTURE);
        const int k2 = plain && cfg->keep_off_diags;
        const int k3 = plain ? cfg->keep_off_diags : 1;
        if (add_block(row_ptr, col_idx, Kval, phiCount, phiDofs, phiCount, phiDofs, elemMat4)) return -1;
        if (k2 && add_block(row_ptr, col_idx, Kval, dispCount, dispDofs, phiCount, phiDofs, elemMat2)) return -1;
        if (k3 && add_block(row_ptr, col_idx, Kval, phiCount, phiDofs, dispCount, dispDofs, elemMat3)) return -1;
      }
    }
    for (int i = 0; i < dispCount; i++) force[dispDofs[i]] += elemForce1[i];
    if (isPF) for (int a = 0; a < phiCount; a++) force[phiDofs[a]] += elemForce2[a];
  }
  if (err_ext) *err_ext = errExt;
  return 0;
}

/* PhaseFractureModel::getArcFunc_ (PhaseFractureModel.cpp:844-1063): the dissipation-based arc-length function of
 * the element loop.  jac10[ndof] is zeroed and filled (:870, :1055-1056); *fvalue = FE_DISSIPATION (:1046, :1059);
 * jac11 is 0 (:871).  Psi / stressP are givePsi() / giveDerivPsi() of the update with the CURRENT degradation (no
 * history maximum here, :1033-1038). */
int fo_arc_func(const fo_config* cfg, int nnode, const double* coords, int64_t nel, const int32_t* conn,
                const double* state, const double* state0, double* jac10, double* fvalue_out) {
  shape_t sh;
  if (shape_init(&sh, cfg->shape)) return -1;
  /* MicroPhaseFractureModel::getArcFunc_ (MicroPhaseFractureModel.cpp:945-1166) is the same loop, line for line */
  if (cfg->model != FO_PHASE_FRACTURE && cfg->model != FO_MICRO_PHASE_FRACTURE)
    FAIL("getArcFunc_ is restated for PhaseFractureModel and MicroPhaseFractureModel");
  const int rank = sh.rank, nodeCount = sh.npe, ipCount = sh.nip;
  const int strCount = STRAIN_COUNTS[rank];
  const int dispCount = nodeCount * rank, phiCount = nodeCount;
  const int ndofpn = rank + 1;
  frac_t fr; memset(&fr, 0, sizeof fr);
  if (frac_init(cfg, &fr)) return -1;
  const double p_ = fr.p, a1_ = fr.a1, a2_ = fr.a2, a3_ = fr.a3;

  static __thread double grads[MAXNIP][MAXNPE][3];
  static __thread double bd[MAXSTR * MAXND];
  double xc[MAXNPE][3], ipWeights[MAXNIP];
  double disp[MAXND], phi[MAXNPE], phi0[MAXNPE];
  double stiff[MAXSTR * MAXSTR], stress[MAXSTR], strain[MAXSTR], stressP[MAXSTR];
  double elemjac10_1[MAXND], elemjac10_2[MAXNPE];
  int64_t dispDofs[MAXND], phiDofs[MAXNPE];
  double fvalue = 0.0;

  for (int64_t i = 0; i < (int64_t)nnode * ndofpn; i++) jac10[i] = 0.0;
  for (int64_t ie = 0; ie < nel; ie++) {
    const int32_t* inodes = conn + ie * nodeCount;
    for (int a = 0; a < nodeCount; a++) {
      for (int i = 0; i < 3; i++) xc[a][i] = (i < rank) ? coords[(int64_t)inodes[a] * rank + i] : 0.0;
      for (int i = 0; i < rank; i++) dispDofs[a * rank + i] = (int64_t)inodes[a] * ndofpn + i;
      phiDofs[a] = (int64_t)inodes[a] * ndofpn + rank;
    }
    if (shape_gradients(&sh, (const double (*)[3])xc, grads, ipWeights)) return -1;
    for (int i = 0; i < dispCount; i++) disp[i] = state[dispDofs[i]];
    for (int a = 0; a < phiCount; a++) { phi[a] = state[phiDofs[a]]; phi0[a] = state0[phiDofs[a]]; }
    memset(elemjac10_1, 0, sizeof elemjac10_1); memset(elemjac10_2, 0, sizeof elemjac10_2);
    for (int ip = 0; ip < ipCount; ip++) {
      fill_B(rank, nodeCount, (const double (*)[3])grads[ip], bd);
      for (int c = 0; c < strCount; c++) {
        double s = 0.0;
        for (int j = 0; j < dispCount; j++) s += bd[c * dispCount + j] * disp[j];
        strain[c] = s;
      }
      const double* Nip = sh.N[ip];
      const double pf = dotn(phiCount, Nip, phi), pf0 = dotn(phiCount, Nip, phi0);
      double gphi, gphi0, dgphi;
      {
        const double gphi_n = pow(1 - pf, p_);
        const double gphi_d = gphi_n + a1_ * pf + a1_ * a2_ * pf * pf + a1_ * a2_ * a3_ * pf * pf * pf;
        gphi = gphi_n / gphi_d + 1.e-7;
        const double dgphi_n = -p_ * (pow(1 - pf, p_ - 1));
        const double dgphi_d = dgphi_n + a1_ + 2.0 * a1_ * a2_ * pf + 3.0 * a1_ * a2_ * a3_ * pf * pf;
        dgphi = (dgphi_n * gphi_d - gphi_n * dgphi_d) / (gphi_d * gphi_d);
      }
      {
        const double gphi_n = pow(1 - pf0, p_);
        const double gphi_d = gphi_n + a1_ * pf0 + a1_ * a2_ * pf0 * pf0 + a1_ * a2_ * a3_ * pf0 * pf0 * pf0;
        gphi0 = gphi_n / gphi_d + 1.e-7;
      }
      double Psi;
      if (fo_material_update(cfg, rank, strain, gphi, stress, stiff, &Psi, stressP)) return -1;
      const double wip = ipWeights[ip];
      fvalue += wip * (gphi0 - gphi) * Psi;
      for (int i = 0; i < dispCount; i++) {              /* elemjac10_1 += wip (gphi0 - gphi) bdt stressP */
        double s = 0.0;
        for (int c = 0; c < strCount; c++) s += bd[c * dispCount + i] * stressP[c];
        elemjac10_1[i] += wip * (gphi0 - gphi) * s;
      }
      for (int a = 0; a < phiCount; a++) elemjac10_2[a] -= wip * Nip[a] * dgphi * Psi;
    }
    for (int i = 0; i < dispCount; i++) jac10[dispDofs[i]] += elemjac10_1[i];
    for (int a = 0; a < phiCount; a++) jac10[phiDofs[a]] += elemjac10_2[a];
  }
  *fvalue_out = fvalue;
  return 0;
}
