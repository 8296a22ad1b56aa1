/*
 * falcon_oracle.h -- CPU restatement of falcon's element-level K / f_int assembly.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference (ritukeshbharali/falcon) cannot be built here
 * (needs the external Jem/Jive 3.0 libraries) and ships no golden numeric
 * vectors (its tests are smoke tests, tests/do_tests.sh:47-59).  This oracle is
 * a restatement of the reference's arithmetic, file:line cited per function,
 * pinned only by analytic known-answer tests (tests/test_oracle_kat.py).
 * Jive-internal pieces (shape functions, Gauss rules, DOF numbering, CSR
 * pattern, matmul summation order, EigenUtils::symSolve) are restated from
 * their published/standard definitions and are not pinned by any reference
 * fixture.
 */
#ifndef FALCON_ORACLE_H
#define FALCON_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shape + integration scheme ids (Jive shape.type / shape.intScheme strings) */
enum {
  FO_TRI3_G1   = 0, /* Triangle3, Gauss1                    */
  FO_TRI3_G3   = 1, /* Triangle3, Gauss3                    */
  FO_QUAD4_G2  = 2, /* Quad4,     Gauss2*Gauss2             */
  FO_QUAD8_G3  = 3, /* Quad8,     Gauss3*Gauss3             */
  FO_QUAD8_G2  = 4, /* Quad8,     Gauss2*Gauss2             */
  FO_TRI6_G3   = 5, /* Triangle6, Gauss3                    */
  FO_HEX8_G2   = 6, /* Hex8,      Gauss2*Gauss2*Gauss2      */
  FO_HEX20_G3  = 7, /* Hex20,     Gauss3*Gauss3*Gauss3      */
  FO_HEX20_G2  = 8, /* Hex20,     Gauss2*Gauss2*Gauss2      */
  FO_SHAPE_COUNT = 9
};

/* model ids: ModelFactory names in src/fem/solidmech/FalconSolidMechModels.cpp:10-22 */
enum {
  FO_LINEAR_ELASTICITY   = 0,
  FO_PHASE_FRACTURE      = 1,
  FO_PHASE_FRACTURE_EXT  = 2,
  FO_PHASE_FRACTURE_EXTIT= 3,
  FO_MICRO_PHASE_FRACTURE= 4,  /* MicroPhaseFractureModel.cpp:502-851 (micromorphic: local damage d by Newton) */
  FO_MICRO_EXT           = 5,  /* MicroPhaseFractureExtModel.cpp:523-952: g from the extrapolated micromorphic variable */
  FO_MICRO_EXTIT         = 6,  /* MicroPhaseFractureExtItModel.cpp: the same + errExt_ (pass stateExt_ as state00 for both) */
  FO_GRADIENT_DAMAGE     = 7   /* GradientDamageModel.cpp:368-588 with DamageMaterial (eqvStrain = Energy); the history
                                  arrays carry kappa with the loading flag of DamageMaterial.cpp:356-440 in the sign bit */
};

/* material ids: newMaterial, src/materials/Material.cpp:233-244 */
enum { FO_HOOKE = 0, FO_AMOR = 1, FO_BOURDIN = 2, FO_MIEHE = 3, FO_DAMAGE = 4 };

/* DamageMaterial softening laws restated (DamageMaterial.cpp:1044-1633) */
enum { FO_SOFT_PERFECT = 0, FO_SOFT_LINEAR1 = 1, FO_SOFT_LINEAR2 = 2, FO_SOFT_EXPONENTIAL1 = 3, FO_SOFT_EXPONENTIAL2 = 4,
       FO_SOFT_POWER = 5, FO_SOFT_HYPERBOLIC = 6, FO_SOFT_EXPONENTIAL2REG = 7, FO_SOFT_EXPOENERGY = 8, FO_SOFT_POLYNOMIAL = 9,
       FO_SOFT_MULTILINEAR = 10 };   /* the order of softeningLawDefs, DamageMaterial.cpp:134-144 */

/* DamageMaterial equivalent strains restated: "Energy" (:644-649, :698-703) and "Mazars" (:618-642, :886-1032); the
 * vonMises one returns a 6-vector for 2-D strains (:677-683) whose use in GradientDamageModel.cpp:548 is out of bounds */
enum { FO_EQV_ENERGY = 0, FO_EQV_MAZARS = 1, FO_EQV_VONMISES = 2 /* rank 3 only */ };

/* 2D problem state (HookeMaterial.cpp:92-103) */
enum { FO_PLANE_STRAIN = 0, FO_PLANE_STRESS = 1 };

/* fracture types (PhaseFractureModel.cpp:67-72) */
enum { FO_AT1 = 0, FO_AT2 = 1, FO_CZM_LINEAR = 2, FO_CZM_EXPONENTIAL = 3,
       FO_CZM_CORNELISSEN = 4, FO_CZM_HYPERBOLIC = 5 };

/* canonical pattern kinds */
enum { FO_PATTERN_FULL = 0,   /* every dof-type pair of every node pair sharing an element */
       FO_PATTERN_WRITTEN = 1 /* only the blocks the model writes (u-u, phi-phi[, u-phi, phi-u]) */ };

typedef struct {
  int    shape;          /* FO_* shape id            */
  int    model;          /* FO_* model id            */
  int    material;       /* FO_* material id         */
  int    state;          /* FO_PLANE_*  (rank 2)     */
  double young, poisson;
  int    frac_type;      /* FO_AT1...                */
  double gc, l0, sigma_t, beta;
  int    keep_off_diags; /* PhaseFracture only       */
  /* GradientDamage / Damage (DamageMaterial.cpp:157-268) */
  int    softening;
  double kappa_i, kappa_c, soft_alpha, soft_beta, length_scale;
  int    eqv_strain;     /* FO_EQV_ENERGY / FO_EQV_MAZARS (DamageMaterial.cpp:275-308) */
  double soft_b, soft_s, kappa_p;   /* hyperbolic b (:237), expoEnergy s (:257), multilinear kappaP (:271) */
  double eta;            /* vonMises: ratio of compressive to tensile strength (:297) */
} fo_config;

typedef struct {
  int rank, npe, nip, str;
} fo_shape_info;

int  fo_shape_info_get(int shape, fo_shape_info* out);
/* N[nip*npe] (ip-major), dN[nip*npe*rank], w[nip] on the reference element */
int  fo_shape_tables(int shape, double* N, double* dN, double* w);

/* dofs per node for a model */
int  fo_ndof_per_node(int model, int rank);
/* number of Gauss points of the last fo_assemble call (this thread) that took the `d > 1` branch of
 * MicroPhaseFractureModel.cpp:772-776, where the reference uses a dphidStrain left over from an earlier point */
int64_t fo_micro_stale_hits(void);
/* getStrain_ (PhaseFractureModel.cpp:1282-1395): nodal sums of the cached Gauss-point strains and of the weights */
int fo_strain_table(int shape, int64_t nel, const int32_t* conn, const double* strain, int srows, int str,
                    double* table, double* weights);
/* getStress_ (PhaseFractureModel.cpp:1165-1271): material update from the cached strain / degradation, nodal sums */
int fo_stress_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain, int srows,
                    double* table, double* weights);

/* Psi0 = initial history value, PhaseFractureModel.cpp:296-359 */
/* GradientDamageModel's tables (GradientDamageModel.cpp:956-1365): getStress_, getStrain_ (3-D only: the reference
 * throws in 2-D), getDamage_, getOutputData_.  strain: [nip_total][strCount + 1]; hist_*: kappa, negative = loading. */
int fo_gd_stress_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain,
                       const double* hist_pre, double* hist_new, double* table, double* weights);
int fo_gd_strain_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain,
                       double* table, double* weights);
int fo_gd_damage_table(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* hist_new,
                       double* table, double* weights);
int fo_gd_output_data(const fo_config* cfg, int64_t nel, const int32_t* conn, const double* strain,
                      const double* hist_pre, double* hist_new, int ncols, const int32_t* cols,
                      double* table, double* weights);

double fo_psi0(const fo_config* cfg);

/* degradation function g, g', g'' at phi (PhaseFractureModel.cpp:660-676); g carries the +1e-7 */
void fo_degradation(const fo_config* cfg, double pf, double* g, double* dg, double* ddg);

/* single material-point update; strain[str] -> stress[str], stiff[str*str] (row-major), psi, stressP[str]
 * gphi ignored for Hooke. */
int  fo_material_update(const fo_config* cfg, int rank, const double* strain, double gphi,
                        double* stress, double* stiff, double* psi, double* stressP);

/* canonical CSR pattern: node-major dofs (dof = node*ndofpn + type), rows column-sorted.
 * kind FO_PATTERN_FULL: every dof-type pair; FO_PATTERN_WRITTEN: (u,u),(phi,phi) plus,
 * by keep_off_diags: 0 nothing, 1 (u,phi)+(phi,u) [PhaseFracture keepOffDiags], 2 (phi,u) only
 * [PhaseFractureExt*].  Call with col_idx == NULL to get row_ptr only
 * (row_ptr has nnode*ndofpn+1 entries).  Returns nnz. */
int64_t fo_pattern(int nnode, int64_t nel, int npe, const int32_t* conn, int ndofpn, int rank,
                   int kind, int keep_off_diags, int64_t* row_ptr, int32_t* col_idx);

/* Mazars equivalent strain of DamageMaterial (test hook): eq, d eq / d strain [STRAIN_COUNTS], principal values [3] */
int fo_mazars(const fo_config* cfg, int rank, const double* strain, double* eq, double* deriv, double* princ);
int fo_vonmises(const fo_config* cfg, const double* strain6, double* eq, double* deriv);

/* Element-serial assembly (LinearElasticityModel.cpp:297-430, PhaseFractureModel.cpp:502-751,
 * PhaseFractureExtModel.cpp:488-774, PhaseFractureExtItModel.cpp:518-797).
 * K_values may be NULL (GET_INT_VECTOR).  K and f are ACCUMULATED into.
 * state0/state00: Ext;  state0/state_ext: ExtIt.  hist_pre/hist_new: PhaseFracture (nel*nip).
 * strain_cache: (str[+1]) x (nel*nip), column = ipoint, may be NULL.
 * err_ext: ExtIt output (sum (pf-pfEx)^2), may be NULL.
 * el_begin/el_end: element range (for bounded timing samples and partition tests). */
int  fo_assemble(const fo_config* cfg,
                 int nnode, const double* coords,      /* nnode*rank, node-major */
                 int64_t nel, const int32_t* conn,     /* nel*npe */
                 int64_t el_begin, int64_t el_end,
                 const int64_t* row_ptr, const int32_t* col_idx,
                 const double* state, const double* state0, const double* state00,
                 double dt, double dt0,
                 const double* hist_pre, double* hist_new,
                 double* strain_cache,
                 double* K_values, double* f_int, double* err_ext);

/* ---- constraints and residual (falcon_oracle_cons.c) ----
 * u[slave[k]] = rval[k] + sum_{q in mptr[k]..mptr[k+1]} mcoef[q] * u[mdof[q]]; rval / mptr may be NULL */
typedef struct {
  int64_t nslave; const int64_t* slave; const double* rval;
  const int64_t* mptr; const int64_t* mdof; const double* mcoef;
} fo_cons;
void fo_eval_masters(double* vec, const fo_cons* c);
void fo_eval_slaves(double* vec, const fo_cons* c);
int  fo_cons_eliminate(int64_t ndof, const int64_t* row_ptr, const int32_t* col_idx, double* K, double* rhs,
                       const fo_cons* c);
void fo_residual(int64_t n, const fo_cons* c, const double* fext, const double* fint, const double* du,
                 const double* u, const double* lower, const double* upper, double* r, double* r2, double* out);

/* PhaseFractureModel::getArcFunc_ (PhaseFractureModel.cpp:844-1063): jac10[nnode*ndofpn] (zeroed, then filled) and
 * the dissipation increment FE_DISSIPATION; jac11 = 0.  PhaseFracture only. */
int fo_arc_func(const fo_config* cfg, int nnode, const double* coords, int64_t nel, const int32_t* conn,
                const double* state, const double* state0, double* jac10, double* fvalue);

const char* fo_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
