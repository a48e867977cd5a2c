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
#include <stdlib.h>
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

/* All-cores variant (CPU baseline): the same cheby! arithmetic with a row-parallel
 * CSR mat-vec and the shift / scale / recurrence / accumulate passes of a term fused into the
 * row loop -- what a tuned multi-threaded CPU implementation of the path would do.  H is
 * passed in CSR (for the Hermitian benchmark matrix that is the conjugate of its CSC).
 * Summation order differs from the serial port (row-wise dot products), so it agrees with it to
 * rounding, not bit for bit.  Returns the number of mat-vecs. */
int qp_ref_cheby_csr_omp(int64_t n, const int64_t* rowptr, const int64_t* col, const c128* val, c128* psi,
                         c128* v0, c128* v1, const double* a, int n_coeffs, double Delta, double E_min,
                         double dt) {
  if (n_coeffs < 2 || !(Delta > 0)) return -1;
  const double beta = (Delta / 2) + E_min;
  c128 c = (dt > 0) ? (-2.0 * I) / Delta : (2.0 * I) / Delta;
  const c128 ph = cexp(-I * beta * dt);
  int nmv = 0;
  /* term 1: v0 = Psi; v1 = c (H v0 - beta v0); Psi = a0 v0 + a1 v1 */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) v0[i] = psi[i];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    c128 s = 0.0;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) s += val[p] * v0[col[p]];
    const c128 t = c * (s - beta * v0[i]);
    v1[i] = t;
    psi[i] = a[0] * v0[i] + a[1] * t;
  }
  ++nmv;
  c *= 2;
  for (int k = 2; k < n_coeffs; ++k) {
    /* v2 = c (H v1 - beta v1) + v0, written over v0 (row-local); Psi += a_k v2 */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      c128 s = 0.0;
      for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) s += val[p] * v1[col[p]];
      const c128 t = c * (s - beta * v1[i]) + v0[i];
      v0[i] = t;
      psi[i] += a[k] * t;
    }
    ++nmv;
    c128* tmp = v0;
    v0 = v1;
    v1 = tmp;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) psi[i] *= ph;
  return nmv;
}

/* The all-cores variant on buffers of its own, every page first touched by the thread that later
 * works on it (static schedule over the rows, the schedule of the compute loops): on a multi-socket
 * host the matrix and the vectors are then spread over the memory controllers instead of sitting
 * on the node of the thread that built them. */
typedef struct {
  int64_t n;
  int64_t *rowptr, *col;
  c128 *val, *psi, *v0, *v1;
} qp_ref_omp;

void qp_ref_omp_free(qp_ref_omp* h) {
  if (!h) return;
  free(h->rowptr);
  free(h->col);
  free(h->val);
  free(h->psi);
  free(h->v0);
  free(h->v1);
  free(h);
}

qp_ref_omp* qp_ref_omp_setup(int64_t n, const int64_t* rowptr, const int64_t* col, const c128* val, const c128* psi) {
  qp_ref_omp* h = (qp_ref_omp*)calloc(1, sizeof(qp_ref_omp));
  if (!h) return NULL;
  const int64_t nnz = rowptr[n];
  h->n = n;
  h->rowptr = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
  h->col = (int64_t*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int64_t));
  h->val = (c128*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(c128));
  h->psi = (c128*)malloc((size_t)(n > 0 ? n : 1) * sizeof(c128));
  h->v0 = (c128*)malloc((size_t)(n > 0 ? n : 1) * sizeof(c128));
  h->v1 = (c128*)malloc((size_t)(n > 0 ? n : 1) * sizeof(c128));
  if (!h->rowptr || !h->col || !h->val || !h->psi || !h->v0 || !h->v1) {
    qp_ref_omp_free(h);
    return NULL;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    h->rowptr[i] = rowptr[i];
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
      h->col[p] = col[p];
      h->val[p] = val[p];
    }
    h->psi[i] = psi[i];
    h->v0[i] = 0.0;
    h->v1[i] = 0.0;
  }
  h->rowptr[n] = nnz;
  return h;
}

int qp_ref_omp_step(qp_ref_omp* h, const double* a, int n_coeffs, double Delta, double E_min, double dt) {
  return qp_ref_cheby_csr_omp(h->n, h->rowptr, h->col, h->val, h->psi, h->v0, h->v1, a, n_coeffs, Delta, E_min, dt);
}

void qp_ref_omp_get_psi(const qp_ref_omp* h, c128* out) { memcpy(out, h->psi, (size_t)h->n * sizeof(c128)); }
