/* CPU restatement (plain C) of the reference's single-threaded Chebyshev step on a
 * SparseMatrixCSC{ComplexF64,Int64}.  TEST / BASELINE INFRASTRUCTURE ONLY: used by
 * tests/ as a second checker and by bench.py's `cpu_baseline` leg (kind "port");
 * never linked into or called by the product path.
 *
 * Follows, operation by operation:
 *   cheby!            src/cheby.jl:150-213   (copyto!/lmul!/axpy! sequence kept)
 *   mul!(y, A, x)     SparseArrays' serial CSC kernel that LinearAlgebra dispatches to
 *                     for SparseMatrixCSC: y = 0; for each column j, y[rowval] += nzval*x[j]
 * Indices are 0-based int64 here (Julia: 1-based Int64).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef double _Complex c128;

static void csc_mul(int64_t n, const int64_t* colptr, const int64_t* rowval, const c128* nzval,
                    const c128* x, c128* y) {
  for (int64_t i = 0; i < n; ++i) y[i] = 0.0;
  for (int64_t j = 0; j < n; ++j) {
    const c128 xj = x[j];
    for (int64_t p = colptr[j]; p < colptr[j + 1]; ++p) y[rowval[p]] += nzval[p] * xj;
  }
}

/* returns the number of mat-vecs, or -1 on a bad argument */
int qp_ref_cheby_csc(int64_t n, const int64_t* colptr, const int64_t* rowval, const c128* nzval,
                     c128* psi, c128* v0, c128* v1, c128* v2, const double* a, int n_coeffs,
                     double Delta, double E_min, double dt) {
  if (n_coeffs < 2 || !(Delta > 0)) return -1;
  const double beta = (Delta / 2) + E_min;                       /* :156 */
  c128 c = (dt > 0) ? (-2.0 * I) / Delta : (2.0 * I) / Delta;    /* :158-162 */
  int nmv = 0;
  memcpy(v0, psi, (size_t)n * sizeof(c128));                     /* copyto!(v0, Psi)  :171 */
  for (int64_t i = 0; i < n; ++i) psi[i] *= a[0];                /* lmul!(a[1], Psi)  :172 */
  csc_mul(n, colptr, rowval, nzval, v0, v1);                     /* mul!(v1, H, v0)   :176 */
  ++nmv;
  for (int64_t i = 0; i < n; ++i) v1[i] += -beta * v0[i];        /* axpy!             :178 */
  for (int64_t i = 0; i < n; ++i) v1[i] *= c;                    /* lmul!             :179 */
  for (int64_t i = 0; i < n; ++i) psi[i] += a[1] * v1[i];        /* axpy!             :182 */
  c *= 2;                                                        /*                   :184 */
  for (int k = 2; k < n_coeffs; ++k) {                           /*                   :186 */
    csc_mul(n, colptr, rowval, nzval, v1, v2);                   /*                   :190 */
    ++nmv;
    for (int64_t i = 0; i < n; ++i) v2[i] += -beta * v1[i];      /*                   :192 */
    for (int64_t i = 0; i < n; ++i) v2[i] *= c;                  /*                   :193 */
    for (int64_t i = 0; i < n; ++i) v2[i] += v0[i];              /*                   :202 */
    for (int64_t i = 0; i < n; ++i) psi[i] += a[k] * v2[i];      /*                   :205 */
    c128* t = v0;                                                /* rotate            :207 */
    v0 = v1;
    v1 = v2;
    v2 = t;
  }
  const c128 ph = cexp(-I * beta * dt);                          /*                   :211 */
  for (int64_t i = 0; i < n; ++i) psi[i] *= ph;
  return nmv;
}
