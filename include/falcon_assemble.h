/*
 * falcon_assemble.h -- C ABI of libfalcon_assemble.so: MI355X (gfx950) element-level
 * tangent-stiffness / internal-force assembly, the drop-in for the loop behind falcon's
 * Model::takeAction(GET_MATRIX0 / GET_INT_VECTOR) handlers.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the
 * ritukeshbharali/falcon checkout).  Plain pointers and sizes only; no torch / Jive types.
 * All functions return FA_OK (0) or a negative fa_status; fa_last_error() gives the text
 * (the Jive shim rethrows it as jem::Error, cf. PhaseFractureModel.cpp:239-246).
 *
 * Threading: like the reference (takeAction is only ever called from the solver's thread), a
 * context must not be used from two threads at once.
 */
#ifndef FALCON_ASSEMBLE_H
#define FALCON_ASSEMBLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fa_ctx fa_ctx;
typedef int32_t fa_status;

enum { FA_OK = 0, FA_ERR_ARG = -1, FA_ERR_UNSUPPORTED = -2, FA_ERR_HIP = -3, FA_ERR_NOMEM = -4 };

/* shape.type / shape.intScheme of the model's "shape" property (IShapeFactory::newInstance,
 * PhaseFractureModel.cpp:124-128) */
enum { FA_TRI3_G1 = 0, FA_TRI3_G3 = 1, FA_QUAD4_G2 = 2, FA_QUAD8_G3 = 3, FA_QUAD8_G2 = 4, FA_TRI6_G3 = 5,
       FA_HEX8_G2 = 6, FA_HEX20_G3 = 7, FA_HEX20_G2 = 8 };

/* ModelFactory names, src/fem/solidmech/FalconSolidMechModels.cpp:10-22 */
enum { FA_LINEAR_ELASTICITY = 0, FA_PHASE_FRACTURE = 1, FA_PHASE_FRACTURE_EXT = 2, FA_PHASE_FRACTURE_EXTIT = 3,
       FA_MICRO_PHASE_FRACTURE = 4, /* "MicroPhaseFracture": micromorphic variant, MicroPhaseFractureModel.cpp:502-851;
                                      history = local damage d (preHist_.phasef_), penalty default 250.
                                      With FA_AMOR_PHASE (all three micromorphic models) the reference's split update reads
                                      stiffN from an array it fills at compression points only and never resets
                                      (AmorPhaseMaterial.cpp:256-263, MicroPhaseFractureModel.cpp:535): a point in tension
                                      gets K PVol once the element loop has passed a compression point and the array's
                                      initial contents before.  Jem does not initialise that array -- the reference is
                                      undefined there; this library ASSUMES ZERO.  Consequences the caller must know: K and
                                      f_int depend on the serial element numbering (through the first compression point of
                                      the call) and on the partition (every context / rank finds its own first compression
                                      point), exactly as the reference's would.  FA_VERBOSE=1 prints this at fa_create.
                                      The first compression point is found once per call by a pre-pass and every kernel
                                      follows it (points before it are tension, the point itself compression, whatever the
                                      last bits of a kernel's own trace say).  A point whose trace is rounding noise
                                      elsewhere keeps the sign its kernel computes -- the tangent jumps by O(K) there in the
                                      reference too; FA_MICRO_AMOR_MASK=1 at fa_create makes the pre-pass classify every
                                      point (all kernel paths then agree on K as well; +30 % on the cfg3 mesh). */
       FA_MICRO_PHASE_FRACTURE_EXT = 5,   /* "MicroPhaseFractureExt", MicroPhaseFractureExtModel.cpp:523-952: g from
                                      the damage solved for the extrapolated micromorphic variable; the caller passes
                                      stateExt_ (:553-566) as state00; state0 / dt / dt0 are not read; penalty 6500 */
       FA_MICRO_PHASE_FRACTURE_EXTIT = 6, /* "MicroPhaseFractureExtIt", MicroPhaseFractureExtItModel.cpp: the same plus
                                      errExt_ = sum (pf - pfEx)^2 (:712) returned through err_ext */
       FA_GRADIENT_DAMAGE = 7             /* "GradientDamage", GradientDamageModel.cpp:368-588: implicit
                                      gradient-enhanced damage, dofs = displacements + the nonlocal equivalent strain
                                      (where the phase-field models have phi); material FA_DAMAGE; history = kappa with
                                      the loading flag (DamageMaterial.cpp:356-440) in its sign bit; all four blocks
                                      K1..K4 are written (pattern as with keepOffDiags) */ };

/* material.type, newMaterial, src/materials/Material.cpp:233-244 */
enum { FA_HOOKE = 0, FA_AMOR_PHASE = 1, FA_BOURDIN_PHASE = 2, FA_MIEHE_PHASE = 3,
       FA_DAMAGE = 4 /* "Damage", src/materials/DamageMaterial.cpp, with eqvStrain = "Energy" (:644-649, :698-703) or
                        "Mazars" (:618-642, :886-1032; principal strains from the reference's trigonometric cubic solve
                        with its PI = 3.14159265) or, for 3-D shapes, "vonMises" (:609-617, :667-684, VonMisesStrain.cpp).
                        "vonMises" is refused in 2-D: its derivative is a 6-vector whatever the strain size (:677-683),
                        used out of bounds at GradientDamageModel.cpp:548 */ };

/* Device-side limits of FA_DAMAGE, checked by fa_create and stricter than DamageMaterial::configure (which takes any
 * value >= 0): kappaI > 0 always (omega divides by kappa); kappaC > kappaI for linear1 / linear2 / exponential2 / power /
 * polynomial; b >= 0 (hyperbolic); kappaC > 0 and s >= 0 (expoEnergy); kappaC > kappaP >= 0 (multilinear); eta > 0 with
 * eqvStrain = vonMises.  A property set the reference accepts at configure and that violates one of these is refused
 * with FA_ERR_ARG here (the reference would divide by zero later, at the first update). */
/* eqvStrain of the Damage material */
enum { FA_EQV_ENERGY = 0, FA_EQV_MAZARS = 1, FA_EQV_VONMISES = 2 /* 3-D shapes only, needs eta */ };

/* softening (DamageMaterial.cpp:182-273; the laws are the free functions of :1044-1633), numbered like
 * softeningLawDefs (:134-144).  "exponential2Reg" is accepted by the reference but absent from the damageEvolution_ /
 * getdOmegadKappa overloads GradientDamage reaches (:802-851, :712-761): omega = 0 throughout, as there. */
enum { FA_SOFT_PERFECT = 0, FA_SOFT_LINEAR1 = 1 /* kappaC */, FA_SOFT_LINEAR2 = 2 /* kappaC */,
       FA_SOFT_EXPONENTIAL1 = 3 /* alpha, beta */, FA_SOFT_EXPONENTIAL2 = 4 /* kappaC */,
       FA_SOFT_POWER = 5 /* kappaC, alpha, beta */, FA_SOFT_HYPERBOLIC = 6 /* b */, FA_SOFT_EXPONENTIAL2REG = 7,
       FA_SOFT_EXPOENERGY = 8 /* kappaC, s */, FA_SOFT_POLYNOMIAL = 9 /* kappaC */,
       FA_SOFT_MULTILINEAR = 10 /* kappaP, kappaC */ };

/* material.state, src/materials/HookeMaterial.cpp:92-103 */
enum { FA_PLANE_STRAIN = 0, FA_PLANE_STRESS = 1 };

/* fractureType, PhaseFractureModel.cpp:67-72 */
enum { FA_AT1 = 0, FA_AT2 = 1, FA_CZM_LINEAR = 2, FA_CZM_EXPONENTIAL = 3, FA_CZM_CORNELISSEN = 4,
       FA_CZM_HYPERBOLIC = 5 };

/* canonical CSR pattern built when the caller passes none (stand-in for Jive's "FEM" MatrixBuilder,
 * SURVEY.md 3.4): FULL = every dof-type pair of every node pair sharing an element; WRITTEN = only the
 * blocks this model writes. */
enum { FA_PATTERN_FULL = 0, FA_PATTERN_WRITTEN = 1 };

/* fa_assemble flags */
enum {
  FA_ACCUMULATE   = 0,  /* K += , f += : the reference's semantics (addBlock / select(force,dofs) +=,
                           PhaseFractureModel.cpp:738-749); the caller zeroes beforehand */
  FA_OVERWRITE    = 1,  /* K = , f = over the whole pattern (== caller's zero-fill followed by +=) */
  FA_STORE_STRAIN = 2,  /* also materialise the strain_ cache (PhaseFractureModel.cpp:651,678) */
  FA_FORCE_GENERIC = 4, /* testing: use the generic atomic-scatter kernel even for a canonical pattern */
  FA_PART_INTERFACE = 16, /* after fa_set_interface: only the tiles that own an interface node (their K rows, their f_int entries) */
  FA_PART_INTERIOR  = 32, /* ... only the other tiles; the two calls together are one full pass (any order, any two streams) */
  FA_DETERMINISTIC = 8  /* bitwise run-to-run reproducible K/f: every entry summed in ascending element order (the ordered
                           row kernel; the force-vector kernel for GET_INT_VECTOR passes of Quad4 / Triangle3 / Hex8).  Exists for
                           Triangle3/Gauss1, Quad4 and Hex8 on the canonical node-major pattern only: on every other
                           context (higher-order shapes, a custom dof map or pattern) the call is refused with
                           FA_ERR_UNSUPPORTED -- those kernels add f_int (and, for non-canonical patterns, K) with
                           atomics whose order varies from run to run.  Without the flag a GET_INT_VECTOR pass of these
                           shapes runs on ELEMENT tiles (every Gauss point evaluated once, f_int rows added with global
                           atomics: 14-49 % faster, equal to the ordered sum to rounding, ~1e-16 relative) */
};

typedef struct {
  int32_t shape, model, material, state;      /* enums above                                    */
  double  young, poisson;                      /* material.young / material.poisson              */
  int32_t frac_type;                           /* fractureType                                   */
  double  gc, l0, sigma_t, beta;               /* griffithEnergy, lengthScale, tensileStrength, penalty
                                                  (defaults 2.7, 0.015, 0, 6500: PhaseFractureModel.cpp:253-263,
                                                  PhaseFractureExtModel.cpp:247) */
  int32_t keep_off_diags;                      /* keepOffDiags (PhaseFracture only)              */

  int64_t nnode;  const double*  coords;       /* host, nnode*rank, node-major (nodes_.getSomeCoords)   */
  int64_t nel;    const int32_t* conn;         /* host, nel*npe, 0-based node indices of the model's
                                                  element group in group order (elems_.getElemNodes)     */
  /* CSR pattern as the caller's matrix holds it (SparseMatrix::getRowOffsets/getColumnIndices,
   * src/solvers/linear/cuDSSSolver.cpp:639-663); NULL => canonical pattern of kind pattern_kind.
   * DOF numbering is node-major, type-minor: dof = node*ndofpn + type (XDofSpace::addDofs order,
   * PhaseFractureModel.cpp:159-179), unless node_dofs is given. */
  const int64_t* row_ptr;  const int32_t* col_idx;
  const int64_t* node_dofs;                    /* host, nnode*ndofpn global dof index per (node,type), or NULL */
  int64_t ndof;                                /* total dofs of the matrix (0 => nnode*ndofpn)          */
  int32_t pattern_kind;
  int32_t device;                              /* HIP device ordinal                                    */
  /* FA_GRADIENT_DAMAGE / FA_DAMAGE only (DamageMaterial.cpp:157-268): softening law and its parameters, kappaI (or
   * ft / young), kappaC, alpha, beta, and the gradient parameter c = lengthScale (GradientDamageModel.cpp:438) */
  int32_t softening;
  double  kappa_i, kappa_c, soft_alpha, soft_beta, length_scale;
  int32_t eqv_strain;                          /* FA_EQV_* (eqvStrain, DamageMaterial.cpp:275-308) */
  double  soft_b, soft_s, kappa_p;             /* hyperbolic "b" (:237), expoEnergy "s" (:257), multilinear "kappaP" (:271) */
  double  eta;                                 /* FA_EQV_VONMISES: "eta" (:297), compressive / tensile strength */
} fa_problem;

typedef struct {
  /* all DEVICE pointers unless noted */
  const double* state;    /* StateVector::get       (PhaseFractureModel.cpp:406)           */
  const double* state0;   /* StateVector::getOld    (PhaseFractureExtModel.cpp:389); Ext, ExtIt */
  const double* state00;  /* StateVector::getOldOld (Ext) or stateExt_ (ExtIt, PhaseFractureExtItModel.cpp:407) */
  double  dt, dt0;        /* XProps::STEP_SIZE / STEP_SIZE_0 as the model holds them (dt_, dt0_) */
  double* K_values;       /* nnz, NULL => force only (GET_INT_VECTOR)                       */
  double* f_int;          /* ndof                                                           */
  double* err_ext;        /* 1 double, ExtIt: errExt_ = sum (pf-pfEx)^2 (PhaseFractureExtItModel.cpp:693); may be NULL */
  int32_t flags;          /* FA_ACCUMULATE | FA_OVERWRITE | FA_STORE_STRAIN ...             */
  void*   stream;         /* hipStream_t, NULL => default stream                            */
} fa_assemble_args;

/* Replaces the model constructor + configure (PhaseFractureModel.cpp:82-267,279-361): element group,
 * shape, DOF space, material, fracture constants, history allocation (preHist := Psi0). */
fa_status fa_create(const fa_problem* p, fa_ctx** out);
/* Model destructor (PhaseFractureModel.cpp:270-271). */
void      fa_destroy(fa_ctx* ctx);
const char* fa_last_error(void);

/* sizes: dofs_->dofCount(), nnz of the pattern, ipCount*ielemCount, shape_->ipointCount() etc. */
int64_t   fa_ndof(const fa_ctx* ctx);
int64_t   fa_nnz(const fa_ctx* ctx);
int64_t   fa_nip_total(const fa_ctx* ctx);
int32_t   fa_ndof_per_node(const fa_ctx* ctx);
int32_t   fa_strain_rows(const fa_ctx* ctx);
/* 1 if the coalesced LDS-staged kernels (patch / row) are used, 0 if an atomic-scatter kernel is (generic
 * kernel; cooperative element kernel of the higher-order shapes) */
int32_t   fa_uses_row_kernel(const fa_ctx* ctx);
/* set-up statistics of the patch kernel: number of node patches and Gauss-point redundancy
 * (element evaluations / elements; halo elements are evaluated by every patch they touch).  No reference counterpart. */
int32_t   fa_patch_stats(const fa_ctx* ctx, double* redundancy);
/* Psi0_ (PhaseFractureModel.cpp:296-355) */
double    fa_psi0(const fa_ctx* ctx);

/* The element tables the kernels integrate with (IShapeFactory::newInstance + shape_->getShapeFunctions(),
 * reference-space gradients and weights behind shape_->getShapeGradients, PhaseFractureModel.cpp:124-128,602).
 * Host only, no GPU needed.  N[nip*npe], dN[nip*npe*rank], w[nip]; any output pointer may be NULL. */
fa_status fa_shape_info(int32_t shape, int32_t* rank, int32_t* npe, int32_t* nip);
fa_status fa_shape_tables(int32_t shape, double* N, double* dN, double* w);

/* CSR pattern hand-off (what the solver wrappers pull: cuDSSSolver.cpp:639-663).  Host copies
 * (either pointer may be NULL) and device-resident pointers. */
fa_status fa_pattern_host(const fa_ctx* ctx, int64_t* row_ptr, int32_t* col_idx);
fa_status fa_pattern_device(const fa_ctx* ctx, const int64_t** row_ptr, const int32_t** col_idx);

/* getMatrix_(mbuilder|nullptr, force, state[, state0, state00]): GET_MATRIX0 / GET_INT_VECTOR
 * (LinearElasticityModel.cpp:297-430, PhaseFractureModel.cpp:502-751, PhaseFractureExtModel.cpp:488-774,
 * PhaseFractureExtItModel.cpp:518-797).  Asynchronous on args->stream.  newHist_ is rewritten on every
 * call (PhaseFractureModel.cpp:703). */
fa_status fa_assemble(fa_ctx* ctx, const fa_assemble_args* args);

/* fa_assemble for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same kernels and same results
 * as fa_assemble, but the call allocates nothing: the per-workgroup partial sums of errExt_ (ExtIt models) go to a
 * file-scope device array of the library -- 131072 doubles = 1 MiB, held by every GPU the library runs on until the
 * process ends -- instead of the context's hipMalloc'ed buffer.  Limits: a mesh whose launch has more than 131072
 * workgroups is refused with FA_ERR_ARG when err_ext is requested (about 8 M Quad4 or 2.5 M Hex8 elements; models
 * without errExt_ have no limit); FA_STORE_STRAIN is accepted only once an earlier fa_assemble call has created the
 * strain cache.  All calls on one GPU share the array: calls that overlap on one GPU (different streams, different
 * contexts) race on it -- serialise them. */
fa_status fa_assemble_noalloc(fa_ctx* ctx, const fa_assemble_args* args);

/* Interface-first assembly for a partitioned run (one process per GPU; the reference's counterpart is Jive's
 * MPI decomposition, build/main.cpp:127-130,177-178, which assembles a subdomain in one piece and exchanges afterwards).
 * node_ranges: nranges pairs [begin, end) of node indices on a partition interface (HOST array).  Afterwards a pass can
 * be issued as two fa_assemble calls -- flags | FA_PART_INTERFACE on one stream, flags | FA_PART_INTERIOR on another --:
 * the first computes the K rows and f_int entries of every node tile that owns an interface node (at least the interface
 * rows; whole tiles, so usually a few node layers more), the second everything else, so that the exchange of the
 * interface rows (RCCL over xGMI) runs while the interior is still being assembled.  Every row of K and entry of f_int is
 * written by exactly one of the two calls; history / strain cache / errExt_ likewise (errExt_ is summed by the INTERIOR
 * call: order it after the INTERFACE call when err_ext is requested).  Available on the node-tile kernels (Triangle3,
 * Quad4, Hex8 and the higher-order shapes on the canonical pattern); FA_ERR_UNSUPPORTED on the generic / cooperative
 * kernels and for MicroPhaseFracture* + AmorPhase (one first compression point per pass).  Set-up copies the tile
 * descriptors back once: O(tiles). */
fa_status fa_set_interface(fa_ctx* ctx, int32_t nranges, const int64_t* node_ranges);

/* Same with HOST vectors (what a Jive shim holds): H2D state, assemble, D2H K/f, synchronous.
 * K_values/f_int are accumulated into or overwritten on the host according to flags. */
fa_status fa_assemble_host(fa_ctx* ctx, const double* state, const double* state0, const double* state00,
                           double dt, double dt0, double* K_values, double* f_int, double* err_ext,
                           int32_t flags);

/* Actions::COMMIT: newHist_.H0_.swap(preHist_.H0_) (PhaseFractureModel.cpp:468-473). */
fa_status fa_commit(fa_ctx* ctx);

/* History (Flex<double> H0_, PhaseFractureModel.h:292-298): which = 0 preHist_, 1 newHist_.  Host copies
 * (checkpoint / restart) and the device pointer. */
fa_status fa_get_history(const fa_ctx* ctx, int32_t which, double* host_out);
fa_status fa_set_history(fa_ctx* ctx, int32_t which, const double* host_in);
double*   fa_history_device(fa_ctx* ctx, int32_t which);
/* MicroPhaseFracture*: number of Gauss points of the last fa_assemble call that took the `d > 1` branch of the local
 * damage update (MicroPhaseFractureModel.cpp:772-776).  There the reference keeps a dphidStrain left over from an earlier
 * Gauss point of its serial loop; the kernels use 0.  The branch needs a committed d0 > 1, so the count is 0 in any
 * physical run -- a non-zero value says that K of this call differs from the reference's.  Waits for the call to finish.
 * *out = 0 for every other model. */
fa_status fa_micro_stale_hits(const fa_ctx* ctx, int64_t* out);
/* strain_ cache, (str[+1]) x nip_total, column = ipoint (PhaseFractureModel.cpp:187); filled by calls
 * made with FA_STORE_STRAIN. */
fa_status fa_get_strain_cache(const fa_ctx* ctx, double* host_out);

/* getStrain_ (PhaseFractureModel.cpp:1282-1395): the Gauss-point strains of the strain_ cache are
 * extrapolated to the nodes with the transposed shape functions: table[node][c] += sum over the node's elements and
 * their Gauss points of N_node(ip) strain_(c, ipoint), weights[node] += sum N_node(ip) (Jive divides the table rows by
 * the weights afterwards).  table: device [nnode][str] (str = 4 in 2-D: xx, yy, zz, xy; 6 in 3-D), weights: device
 * [nnode]; both are added to, like table.addBlock / `select(weights, inodes) +=`.  Needs a strain cache filled by a
 * call with FA_STORE_STRAIN.  One lane per node, elements ascending: same summation order as the reference's loop. */
fa_status fa_strain_table(fa_ctx* ctx, double* table, double* weights, void* stream);
/* getStress_ (PhaseFractureModel.cpp:1165-1271; LinearElasticityModel's is the same loop with Hooke): like
 * fa_strain_table, but every Gauss point's stress comes from the material update with the cached strain and the cached
 * degradation (`update(stressIp, stiff, strainIp, gphiEx, ipoint)`), all str components (2-D: xx, yy, zz, xy).
 *
 * FA_GRADIENT_DAMAGE (GradientDamageModel.cpp:884-1241, GET_TABLE "strain" / "stress" / "damage"):
 *  - fa_strain_table follows GradientDamageModel::getStrain_ (:1071-1177): str + 1 columns, the last one the nonlocal
 *    equivalent strain "e_eq" (table: [nnode][7] in 3-D).  The reference's column switch has no case for str = 4, so its
 *    2-D call throws "unexpected number of strain components: 4" (:1124-1130); fa_strain_table returns
 *    FA_ERR_UNSUPPORTED with that message.
 *  - fa_stress_table follows getStress_ (:956-1063): `material_->update(stressIp, stiff, strain_(ALL,ipoint), ipoint)`
 *    with all str + 1 cached rows, i.e. DamageMaterial's nonlocal branch (DamageMaterial.cpp:374-377): kappa from the
 *    cached nonlocal strain and preHist_, stress = (1 - omega) D eps.  Like the reference's update it REWRITES newHist_
 *    (:434-435) -- the values the last fa_assemble wrote, unless fa_commit / fa_set_history came in between.
 *  - fa_damage_table follows getDamage_ (:1184-1241): one column "hist" = omega(newHist_ kappa) (giveOmega,
 *    DamageMaterial.cpp:533-545), values and weights both weighed with |N_node(ip)| (:1233-1234).  table: [nnode]. */
fa_status fa_stress_table(fa_ctx* ctx, double* table, double* weights, void* stream);
fa_status fa_damage_table(fa_ctx* ctx, double* table, double* weights, void* stream);

/* getOutputData_ (PhaseFractureModel.cpp:1408-1508): per Gauss point the values [strain.tensor | stress.tensor]
 * (2 str entries: 8 in 2-D, 12 in 3-D; TbFiller's announce order, :1424-1425), the stress from the material update with
 * the cached strain and degradation (:1482-1486); the columns cols[0..ncols) of them (TbFiller's filter, i2table,
 * :1438-1439,:1492; host array) are extrapolated with the shape functions and ADDED to table (device, [nnode][ncols]),
 * the shape-function sums to weights (device, [nnode]) -- once per element, as :1497-1506 do.  Needs a strain cache
 * (FA_STORE_STRAIN).  The context keeps a scratch of nnode * (2 str + 2) doubles after the first call.
 * FA_GRADIENT_DAMAGE (GradientDamageModel.cpp:1254-1365): the point values are [strain.tensor | stress.tensor | damage]
 * (2 str + 1 entries, :1270-1273).  damage = giveOmega() of newHist_ as it is BEFORE the call (:1316); the stress comes
 * from `update(stress, stiff, strain, ipoint)` with a strain of stress size (:1340-1343), which sends DamageMaterial down
 * its LOCAL branch (equivalent strain computed from the strain, DamageMaterial.cpp:370-373) and rewrites newHist_ with
 * that outcome (:434-435) -- reproduced, so a following fa_damage_table / fa_commit sees what the reference's would. */
fa_status fa_output_data(fa_ctx* ctx, int32_t ncols, const int32_t* cols, double* table, double* weights, void* stream);

/* getArcFunc_ (PhaseFractureModel.cpp:844-1063; ArclenActions::GET_ARC_FUNC of the dissipation-based
 * arc-length solver, TSArclenModule.cpp:612-641): the same Gauss-point loop as getMatrix_ with other outputs --
 *     jac10 (device, ndof doubles; zeroed and filled, :870,:1055-1056) = d(dissipation)/d(state)
 *     fe_dissipation (device, 1 double) = sum wip (g(phi0) - g(phi)) Psi        (XProps::FE_DISSIPATION, :1046,:1059)
 * jac11 is 0 (:871).  state / state0: StateVector::get / getOld (:860-861), device.  PhaseFracture and MicroPhaseFracture
 * (MicroPhaseFractureModel.cpp:945-1166 is the same loop line for line: nodal phase field in the degradation).  One lane per
 * node over its elements, no atomics; bitwise repeatable. */
fa_status fa_arc_func(fa_ctx* ctx, const double* state, const double* state0, double* jac10, double* fe_dissipation,
                      void* stream);

/* Host consumers of f_int / state: LodiModel sums the internal force and the state over the dofs
 * of a node group on every GET_MATRIX0 / GET_INT_VECTOR (src/io/models/LodiModel.cpp:263-283: load[i] +=
 * fint1[idofs(i,j)], disp[i] += state[idofs(i,j)], entries < 0 skipped).  With a device-resident f_int that is one
 * small reduction instead of a copy of the whole vector: out[0] = sum_k vec[idx[k]] over the n device-resident
 * indices (idx[k] < 0 skipped), summed in a fixed order (bitwise repeatable).  vec, idx, out: device pointers.  The
 * caller divides by the node count and does the allsum over ranks as LodiModel does (:266,:288-289). */
fa_status fa_gather_sum(fa_ctx* ctx, const double* vec, const int64_t* idx, int64_t n, double* out, void* stream);

/* ---- hand-off of the device-resident K to a linear solver ---- */

/* The pattern with 32-bit offsets: the types the solver wrappers hand to their libraries
 * (cuDSSSolver.cpp:655-663: JEM_PRECHECK(msize <= maxOf<int>()), offsets_ = castTo<int>(getRowOffsets()),
 * indices_ = castTo<int>(getColumnIndices()); :692-694 CUDA_R_32I).  The 32-bit offsets are made on the first call and
 * kept by the context; FA_ERR_UNSUPPORTED when nnz does not fit.  Device pointers. */
fa_status fa_pattern_device32(fa_ctx* ctx, const int32_t** row_ptr, const int32_t** col_idx);
/* HIP device ordinal of the context (fa_problem.device) */
int32_t   fa_device(const fa_ctx* ctx);

/* Constraints in the form jive::util::Constraints holds them (the reference creates them in DirichletModel.cpp:287,
 * DispArclenBCs.cpp:358,440,494, PeriodicModel.cpp:287, GradientDamageModel.cpp:843, StaggeredModule.cpp:767-1031):
 *     u[slave_dofs[k]] = rvalues[k] + sum_{q = master_ptr[k]}^{master_ptr[k+1]-1} master_coefs[q] * u[master_dofs[q]]
 * HOST arrays; rvalues NULL => 0; master_ptr NULL => no masters (plain Dirichlet conditions).  A master must not be a
 * slave (the compressed form, Constraints::compress).
 * The elimination writes (j, master) for every stored (j, slave), (master, q) for every stored (slave, q) and the slaves'
 * diagonal.  Plain Dirichlet conditions never need entries the pattern lacks; periodic / tied / arc-length constraints
 * usually do.  Like StdConstrainer, which builds a new matrix structure (cuDSSSolver.cpp:108-121 getOutputMatrix), the
 * constraint set then owns an OUTPUT pattern: the context's pattern plus that fill-in (merged into the rows by column
 * where the row is column-sorted, appended to the row otherwise).  fa_cons_fill_in says how many entries were added (0:
 * the output pattern IS the context's and fa_cons_eliminate works in place), fa_cons_nnz the size of the output
 * matrix, fa_cons_pattern_* hand it out (64- and 32-bit offsets, device; host copy), fa_cons_eliminate_out fills it.
 * Set-up cost: one device scan of the nnz column indices (HBM speed, no copy of the pattern to the host) plus host work
 * and memory proportional to the rows the constraints touch; with fill-in, nnz + fill int32 of device memory for the
 * output columns and ndof + 1 int64 for its offsets.  The constraint set is tied to the context's pattern and device;
 * destroy it before the context. */
typedef struct fa_cons fa_cons;
fa_status fa_cons_create(fa_ctx* ctx, int64_t nslave, const int64_t* slave_dofs, const double* rvalues,
                         const int64_t* master_ptr, const int64_t* master_dofs, const double* master_coefs,
                         fa_cons** out);
void      fa_cons_destroy(fa_cons* cons);
int64_t   fa_cons_fill_in(const fa_cons* cons);
int64_t   fa_cons_nnz(const fa_cons* cons);
fa_status fa_cons_pattern_device(const fa_cons* cons, const int64_t** row_ptr, const int32_t** col_idx);
fa_status fa_cons_pattern_device32(fa_cons* cons, const int32_t** row_ptr, const int32_t** col_idx);
fa_status fa_cons_pattern_host(const fa_cons* cons, int64_t* row_ptr, int32_t* col_idx);
/* new right-hand values (they change with every load step, DispArclenBCs.cpp:494); host array of nslave doubles */
fa_status fa_cons_set_rvalues(fa_cons* cons, const double* rvalues, void* stream);

/* What StdConstrainer does to the system before the solver library sees it (cuDSSSolver.cpp:108-121: matrix_ =
 * conman_->getOutputMatrix(); :293 conman_->getRhs(f, rhs)), in place on the device arrays:
 *     K   := T^t K T + I_S        (slave rows / columns folded into the masters, then zeroed; unit diagonal)
 *     rhs := T^t (rhs - K c)      (c = rvalues on the slaves; rhs[slaves] = 0)
 * with u = T u + c the constraints.  K_values: nnz doubles as fa_assemble wrote them; rhs: ndof doubles or NULL.
 * Only the entries in slave rows / columns and their fold destinations are touched: O(boundary) per call.  No atomics;
 * bitwise repeatable.  After the solve, fa_cons_eval_slaves gives the slave entries of the solution (:416 getLhs). */
fa_status fa_cons_eliminate(const fa_cons* cons, double* K_values, double* rhs, void* stream);
/* The same into the output pattern: K_out (fa_cons_nnz doubles, device) = K_values copied to its places in the output
 * pattern, fill-in zero, then eliminated there; K_values is left as it is, rhs is updated in place.  Works with and
 * without fill-in (without: a plain copy, or in place when K_out == K_values); with fill-in fa_cons_eliminate itself
 * answers FA_ERR_ARG.  The copy reads and writes every entry once: 16 bytes per non-zero on top of the O(boundary)
 * elimination -- what the reference's constrainer does on the host every time the matrix changes. */
fa_status fa_cons_eliminate_out(const fa_cons* cons, const double* K_values, double* K_out, double* rhs, void* stream);
/* jive::util::evalMasterDofs (BFGSModule.cpp:443, LodiModel.cpp:265, TSArclenModule.cpp:520,641): vec[master] +=
 * coef * vec[slave], vec[slave] = 0.  Device vector of ndof doubles. */
fa_status fa_cons_eval_masters(const fa_cons* cons, double* vec, void* stream);
/* jive::util::evalSlaveDofs: vec[slave] = rvalue + sum coef * vec[master]. */
fa_status fa_cons_eval_slaves(const fa_cons* cons, double* vec, void* stream);

/* BFGSModule::Work_::updateResidual (BFGSModule.cpp:438-489) on device vectors of ndof doubles:
 *     r = fext - fint;  evalMasterDofs(r);
 *     unbounded (lower == upper == NULL): rnorm = norm2(r), enorm = r . du
 *     bounded: r2[i] = -r[i] clipped by the active bounds of u (:452-479), rnorm = norm2(r2), enorm = r2 . du
 * out (device, 3 doubles) = { rnorm, enorm, rnorm^2 }: rnorm is not yet divided by rscale (:485-488); rnorm^2 is what a
 * multi-rank caller sums over ranks.  cons may hold zero slaves.  Fixed-order two-stage sums: bitwise repeatable. */
fa_status fa_residual(const fa_cons* cons, const double* fext, const double* fint, const double* du, const double* u,
                      const double* lower, const double* upper, double* r, double* r2, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
