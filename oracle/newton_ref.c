/* CPU restatement (plain C) of the hot loops of the reference's single-threaded Newton /
 * restarted-Arnoldi step on a SparseMatrixCSC{ComplexF64,Int64}.  TEST / BASELINE
 * INFRASTRUCTURE ONLY: used by tests/ as a second checker and by bench.py's `cpu_baseline`
 * leg of config C3 (kind "port"); never linked into or called by the product path.
 *
 * Follows, operation by operation:
 *   arnoldi!          src/arnoldi.jl:60-100  (fill!, copyto!, mul!, the SEQUENTIAL modified
 *                     Gram-Schmidt loop dot -> axpy per basis vector, norm, lmul!)
 *   mul!(y, A, x)     SparseArrays' serial CSC kernel: y = 0; for each column j,
 *                     y[rowval] += nzval * x[j]
 *   the two tall-skinny combinations of newton!   src/newton.jl:346-352 (Psi += P_i q_i)
 *                     and :363-367 (v = R_0 v + R_i q_i), one axpy per basis vector
 *   norm / lmul!      src/newton.jl:271-272
 * The restart loop around them (Hessenberg eigenvalues, Leja ordering, divided differences:
 * a few hundred scalars per restart) is the NumPy oracle's own, see oracle/ref_c.py: newton_csc.
 * Indices are 0-based int64 here (Julia: 1-based Int64).  Hess is column-major with leading
 * dimension ldh, as the Julia matrix is.
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

static c128 dotc(int64_t n, const c128* x, const c128* y) { /* dot(x, y) = sum conj(x_i) y_i */
  c128 s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += conj(x[i]) * y[i];
  return s;
}

static double nrm2(int64_t n, const c128* x) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += creal(x[i]) * creal(x[i]) + cimag(x[i]) * cimag(x[i]);
  return sqrt(s);
}

double qp_ref_norm(int64_t n, const c128* x) { return nrm2(n, x); }

void qp_ref_scale(int64_t n, c128* x, double re, double im) { /* lmul!(c, x) */
  const c128 c = re + im * I;
  for (int64_t i = 0; i < n; ++i) x[i] *= c;
}

/* arnoldi!(Hess, q, m, psi, H, dt; extended, norm_min)  src/arnoldi.jl:60-100.
 * q: (m + 1) vectors of length n, contiguous (vector k at q + k n).  Returns the effective m
 * (reduced on breakdown, after Hess[j+1, j] was written, as the reference does), -1 on a bad
 * argument.  n_matvec (may be NULL) receives the number of mat-vecs done. */
int qp_ref_arnoldi_csc(int64_t n, const int64_t* colptr, const int64_t* rowval, const c128* nzval, c128* q,
                       int m, const c128* psi, double dt, int extended, double norm_min, c128* Hess, int ldh,
                       int* n_matvec) {
  const int dim = extended ? m + 1 : m;
  if (m < 1 || ldh < dim || dt == 0.0) return -1;
  for (int j = 0; j < ldh; ++j)                                   /* fill!(Hess, 0)          :78 */
    for (int i = 0; i < ldh; ++i) Hess[(int64_t)j * ldh + i] = 0.0;
  memcpy(q, psi, (size_t)n * sizeof(c128));                       /* copyto!(q[1], psi)      :79 */
  int nmv = 0;
  for (int j = 0; j < m; ++j) {                                   /*                         :80 */
    c128* w = q + (int64_t)(j + 1) * n;
    csc_mul(n, colptr, rowval, nzval, q + (int64_t)j * n, w);     /* mul!(q[j+1], H, q[j])   :82 */
    ++nmv;
    for (int i = 0; i <= j; ++i) {                                /*                         :84 */
      const c128* qi = q + (int64_t)i * n;
      const c128 h = dt * dotc(n, qi, w);                         /* Hess[i,j] = dt <q_i, w> :85 */
      Hess[(int64_t)j * ldh + i] = h;
      const c128 f = -h / dt;
      for (int64_t k = 0; k < n; ++k) w[k] += f * qi[k];          /* axpy!(-Hess/dt, q_i, w) :86 */
    }
    if (j + 1 < m || extended) {                                  /*                         :88 */
      const double h = nrm2(n, w);
      Hess[(int64_t)j * ldh + (j + 1)] = dt * h;                  /*                         :90 */
      if (h < norm_min) {                                         /* dimensionality exhausted :91-95 */
        m = j + 1;
        break;
      }
      const double s = 1.0 / h;
      for (int64_t k = 0; k < n; ++k) w[k] *= s;                  /* lmul!(1/h, q[j+1])      :96 */
    }
  }
  if (n_matvec) *n_matvec = nmv;
  return m;
}

/* dst = c0 dst + sum_{i < m} coef[i] q_i   (c0 = 0: dst is overwritten, not read), one axpy per
 * basis vector in index order -- src/newton.jl:346-352 with q = arnoldi_vecs, coef = P, and
 * :363-367 with q = arnoldi_vecs[2:], coef = R[2:], c0 = R[1]. */
void qp_ref_lincomb(int64_t n, c128* dst, double c0_re, double c0_im, const c128* q, const c128* coef, int m) {
  const c128 c0 = c0_re + c0_im * I;
  if (c0_re == 0.0 && c0_im == 0.0) {
    for (int64_t k = 0; k < n; ++k) dst[k] = 0.0;                 /* fill!(Psi, 0)           :347 */
  } else if (!(c0_re == 1.0 && c0_im == 0.0)) {
    for (int64_t k = 0; k < n; ++k) dst[k] *= c0;                 /* lmul!(R[1], v)          :364 */
  }
  for (int i = 0; i < m; ++i) {
    const c128* qi = q + (int64_t)i * n;
    const c128 c = coef[i];
    for (int64_t k = 0; k < n; ++k) dst[k] += c * qi[k];          /* axpy!                   :351 / :366 */
  }
}
