/*
 * falcon_oracle_cons.c -- CPU restatement of the constraint handling and residual evaluation around the K / f_int
 * hot path.  TEST INFRASTRUCTURE ONLY (see falcon_oracle.h).
 *
 * PARITY UNPINNED: jive::util::Constraints, jive::util::evalMasterDofs / evalSlaveDofs and
 * jive::solver::StdConstrainer are Jive classes, absent from the reference checkout.  What is restated here is their
 * documented effect, anchored on the reference's call sites:
 *   - src/solvers/linear/cuDSSSolver.cpp:108-121 (StdConstrainer wraps the matrix; matrix_ = getOutputMatrix()),
 *     :293-294 (getRhs / initLhs), :416 (getLhs);
 *   - src/solvers/implicit/BFGSModule.cpp:438-489 (updateResidual: axpy, evalMasterDofs, norms, bound projection);
 *   - the constraint forms the reference creates: DirichletModel.cpp:287 (idof = val), DispArclenBCs.cpp:440
 *     (idof = 1.0 * master), PeriodicModel.cpp:287 (idof = rval + coef * jdof), GradientDamageModel.cpp:843
 *     (idof = sum coeffs * jdofs).
 * A constraint is   u[slave] = rval + sum_k coef[k] * u[master[k]]   (masters are never slaves), i.e. u = T u + c with
 * T the identity on the free dofs.  The constrained system solved in place of K du = r is
 *       (T^t K T + I_S) du = T^t (r - K c),   rhs[S] = 0,
 * and the solution's slave entries are evaluated afterwards (evalSlaveDofs).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "falcon_oracle.h"

static int64_t find_entry(const int64_t* row_ptr, const int32_t* col_idx, int64_t r, int64_t c) {
  for (int64_t p = row_ptr[r]; p < row_ptr[r + 1]; p++) if (col_idx[p] == c) return p;
  return -1;
}

/* evalMasterDofs: the slave entries are folded into their masters and zeroed (T^t vec) */
void fo_eval_masters(double* vec, const fo_cons* c) {
  for (int64_t k = 0; k < c->nslave; k++) {
    const double t = vec[c->slave[k]];
    vec[c->slave[k]] = 0.0;
    if (c->mptr) for (int64_t q = c->mptr[k]; q < c->mptr[k + 1]; q++) vec[c->mdof[q]] += c->mcoef[q] * t;
  }
}

/* evalSlaveDofs: vec[slave] = rval + sum coef * vec[master] */
void fo_eval_slaves(double* vec, const fo_cons* c) {
  for (int64_t k = 0; k < c->nslave; k++) {
    double t = c->rval ? c->rval[k] : 0.0;
    if (c->mptr) for (int64_t q = c->mptr[k]; q < c->mptr[k + 1]; q++) t += c->mcoef[q] * vec[c->mdof[q]];
    vec[c->slave[k]] = t;
  }
}

/* StdConstrainer's effect on the CSR matrix and the right-hand side, in place.  Returns -1 (fo_last_error unchanged)
 * when the pattern lacks an entry the elimination has to write (fill-in). */
int fo_cons_eliminate(int64_t ndof, const int64_t* row_ptr, const int32_t* col_idx, double* K, double* rhs,
                      const fo_cons* c) {
  int64_t* sl = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ndof > 0 ? ndof : 1));   /* dof -> slave index or -1 */
  if (!sl) return -2;
  for (int64_t i = 0; i < ndof; i++) sl[i] = -1;
  for (int64_t k = 0; k < c->nslave; k++) sl[c->slave[k]] = k;
  int rc = 0;
  /* rhs -= K c  (c holds rval on the slave dofs) */
  if (rhs && c->rval)
    for (int64_t i = 0; i < ndof; i++)
      for (int64_t p = row_ptr[i]; p < row_ptr[i + 1]; p++) {
        const int64_t k = sl[col_idx[p]];
        if (k >= 0) rhs[i] -= K[p] * c->rval[k];
      }
  /* K <- K T: the slave columns are folded into the master columns */
  for (int64_t i = 0; i < ndof && !rc; i++)
    for (int64_t p = row_ptr[i]; p < row_ptr[i + 1]; p++) {
      const int64_t k = sl[col_idx[p]];
      if (k < 0) continue;
      const double v = K[p];
      K[p] = 0.0;
      if (c->mptr) for (int64_t q = c->mptr[k]; q < c->mptr[k + 1]; q++) {
        const int64_t d = find_entry(row_ptr, col_idx, i, c->mdof[q]);
        if (d < 0) { rc = -1; break; }
        K[d] += c->mcoef[q] * v;
      }
    }
  /* K <- T^t K, rhs <- T^t rhs: the slave rows are folded into the master rows; unit diagonal on the slaves */
  for (int64_t k = 0; k < c->nslave && !rc; k++) {
    const int64_t s = c->slave[k];
    for (int64_t p = row_ptr[s]; p < row_ptr[s + 1]; p++) {
      const double v = K[p];
      K[p] = (col_idx[p] == s) ? 1.0 : 0.0;
      if (sl[col_idx[p]] >= 0) continue;               /* a slave column: already zero */
      if (c->mptr) for (int64_t q = c->mptr[k]; q < c->mptr[k + 1]; q++) {
        const int64_t d = find_entry(row_ptr, col_idx, c->mdof[q], col_idx[p]);
        if (d < 0) { rc = -1; break; }
        K[d] += c->mcoef[q] * v;
      }
    }
    if (find_entry(row_ptr, col_idx, s, s) < 0) rc = -1;
  }
  if (rhs && !rc) fo_eval_masters(rhs, c);
  free(sl);
  return rc;
}

/* BFGSModule::Work_::updateResidual (BFGSModule.cpp:438-489): r = fext - fint, evalMasterDofs, then either the plain
 * norms or, with bounds, those of the projected vector r2.  out[0] = rnorm (not yet divided by rscale, :485-488),
 * out[1] = enorm, out[2] = rnorm^2.  lower/upper NULL => unbounded. */
void fo_residual(int64_t n, const fo_cons* c, const double* fext, const double* fint, const double* du,
                 const double* u, const double* lower, const double* upper, double* r, double* r2, double* out) {
  for (int64_t i = 0; i < n; i++) r[i] = fext[i] - 1.0 * fint[i];
  if (c) fo_eval_masters(r, c);
  double s0 = 0.0, s1 = 0.0;
  if (!lower || !upper) {
    for (int64_t i = 0; i < n; i++) { s0 += r[i] * r[i]; s1 += r[i] * du[i]; }
  } else {
    const double epsilon = sqrt(2.220446049250313e-16);   /* Float::EPSILON */
    for (int64_t i = 0; i < n; i++) {
      double lb = lower[i], ub = upper[i];
      const double x = u[i];
      double f = -r[i];
      lb += epsilon * fabs(lb);
      ub -= epsilon * fabs(ub);
      if (x <= lb) f = (0.0 < f) ? 0.0 : f;              /* min(0, f) */
      if (x >= ub) f = (0.0 > f) ? 0.0 : f;              /* max(0, f) */
      r2[i] = f;
    }
    for (int64_t i = 0; i < n; i++) { s0 += r2[i] * r2[i]; s1 += r2[i] * du[i]; }
  }
  out[0] = sqrt(s0); out[1] = s1; out[2] = s0;
}
