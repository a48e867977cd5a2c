// Host-side small dense numerics of the prop_step! path (m <= ~200): Bessel
// coefficients, Hessenberg eigenvalues, Leja ordering, Newton divided differences,
// and the bit-exact index work at the boundary (CSC -> CSR, row partition).
// No HIP in this file: it is also linked into CPU-only test builds.
#include "qprop_internal.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

namespace qp {

// ---- profiler ranges (qprop_internal.h) -------------------------------------------------------------------------
namespace {
using PushFn = int (*)(const char*);
using PopFn = int (*)();
PushFn g_push = nullptr;
PopFn g_pop = nullptr;
std::once_flag g_roctx_once;
void resolve_roctx() {
  for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    g_push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
    g_pop = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
    if (g_push && g_pop) return;
    g_push = nullptr;
    g_pop = nullptr;
  }
}
}  // namespace
void range_push(const char* name) {
  std::call_once(g_roctx_once, resolve_roctx);
  if (g_push) (void)g_push(name);
}
void range_pop() {
  if (g_pop) (void)g_pop();
}
bool ranges_enabled_by_env() {
  static const bool on = [] {
    const char* e = std::getenv("QP_ROCTX");
    return e && e[0] == '1';
  }();
  return on;
}

using cplx = std::complex<double>;

// ---------------------------------------------------------------------------
// Cheby coefficients -- src/cheby.jl:25-39.  SpecialFunctions.besselj(Int, Float64)
// is openlibm's jn; glibc's jn is the same fdlibm lineage.
// ---------------------------------------------------------------------------
std::vector<double> cheby_coeffs(double Delta, double dt, double limit) {
  double alpha = std::fabs(0.5 * Delta * dt);
  std::vector<double> coeffs;
  double a = ::jn(0, alpha);
  coeffs.push_back(a);
  double eps = std::fabs(a);
  int i = 1;
  while (eps > limit) {
    a = 2.0 * ::jn(i, alpha);
    coeffs.push_back(a);
    eps = std::fabs(a);
    ++i;
    if (i > (1 << 24)) break;  // guard against limit <= 0
  }
  return coeffs;
}

// ---------------------------------------------------------------------------
// Eigenvalues of a complex upper-Hessenberg matrix: shifted QR with Givens
// rotations (Wilkinson shift, exceptional shifts at 10/20 iterations), eigenvalues
// only, active-window updates.  Stands in for LAPACK `eigvals` at
// src/arnoldi.jl:165; output sorted by (real, imag) like Julia's default sortby.
// A is column-major n x n (lda = n) and is destroyed.
// ---------------------------------------------------------------------------
static inline double abs1(const cplx& z) { return std::fabs(z.real()) + std::fabs(z.imag()); }

bool hessenberg_eigvals_inplace(int n, cplx* A, cplx* w) {
  auto H = [&](int i, int j) -> cplx& { return A[(size_t)j * n + i]; };
  const double eps = 2.220446049250313e-16;
  std::vector<cplx> cs(n), sn(n);
  int en = n - 1;
  int its = 0;
  const int max_its = 60;
  double hnorm = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= std::min(j + 1, n - 1); ++i) hnorm += abs1(H(i, j));
  if (hnorm == 0.0) hnorm = 1.0;
  while (en >= 0) {
    // find l: smallest index such that the block [l..en] is unreduced
    int l = en;
    for (; l > 0; --l) {
      double s = abs1(H(l - 1, l - 1)) + abs1(H(l, l));
      if (s == 0.0) s = hnorm;
      if (abs1(H(l, l - 1)) <= eps * s) {
        H(l, l - 1) = 0.0;
        break;
      }
    }
    if (l == en) {
      w[en] = H(en, en);
      --en;
      its = 0;
      continue;
    }
    if (its >= max_its) return false;
    cplx mu;
    if (its == 10 || its == 20) {
      // exceptional shift
      mu = H(en, en) + cplx(std::fabs(H(en, en - 1).real()) +
                                (en >= 2 ? std::fabs(H(en - 1, en - 2).real()) : 0.0),
                            0.0);
    } else {
      // Wilkinson shift: eigenvalue of trailing 2x2 closer to H(en,en)
      cplx a = H(en - 1, en - 1), b = H(en - 1, en), c = H(en, en - 1), d = H(en, en);
      cplx tr2 = 0.5 * (a - d);
      cplx disc = std::sqrt(tr2 * tr2 + b * c);
      cplx e1 = d + tr2 + disc, e2 = d + tr2 - disc;  // (a+d)/2 +- disc
      mu = (std::abs(e1 - d) < std::abs(e2 - d)) ? e1 : e2;
    }
    ++its;
    for (int i = l; i <= en; ++i) H(i, i) -= mu;
    // QR factorisation of the active block by Givens rotations (from the left)
    for (int k = l; k < en; ++k) {
      cplx a = H(k, k), b = H(k + 1, k);
      double r = std::sqrt(std::norm(a) + std::norm(b));
      cplx c, s;
      if (r == 0.0) {
        c = 1.0;
        s = 0.0;
      } else {
        c = a / r;
        s = b / r;
      }
      cs[k] = c;
      sn[k] = s;
      // rows k, k+1, columns k..en:  [ conj(c) conj(s) ; -s c ]
      for (int j = k; j <= en; ++j) {
        cplx x = H(k, j), y = H(k + 1, j);
        H(k, j) = std::conj(c) * x + std::conj(s) * y;
        H(k + 1, j) = -s * x + c * y;
      }
      H(k + 1, k) = 0.0;
    }
    // multiply by the rotations from the right: columns k, k+1, rows l..min(k+1,en)
    for (int k = l; k < en; ++k) {
      cplx c = cs[k], s = sn[k];
      int imax = std::min(k + 1, en);
      for (int i = l; i <= imax; ++i) {
        cplx x = H(i, k), y = H(i, k + 1);
        H(i, k) = x * c + y * s;
        H(i, k + 1) = -x * std::conj(s) + y * std::conj(c);
      }
    }
    for (int i = l; i <= en; ++i) H(i, i) += mu;
  }
  std::sort(w, w + n, [](const cplx& x, const cplx& y) {
    if (x.real() != y.real()) return x.real() < y.real();
    return x.imag() < y.imag();
  });
  return true;
}

// eigenvalues of the leading j x j block -- one round of the loop of src/arnoldi.jl:150-168
int diagonalize_hessenberg_block(const cplx* Hess, int ldh, int j, cplx* out) {
  static thread_local std::vector<cplx> work;   // reused between calls (no allocation in a restart loop)
  auto Hm = [&](int r, int c) { return Hess[(size_t)c * ldh + r]; };
  if (j == 1) {
    out[0] = Hm(0, 0);
  } else if (j == 2) {
    cplx a = Hm(0, 0), c = Hm(1, 0), b = Hm(0, 1), d = Hm(1, 1);
    cplx s = std::sqrt(a * a + 4.0 * b * c - 2.0 * a * d + d * d);
    out[0] = 0.5 * (a + d - s);
    out[1] = 0.5 * (a + d + s);
  } else {
    work.assign((size_t)j * j, cplx(0));
    for (int c = 0; c < j; ++c)
      for (int r = 0; r < j; ++r) work[(size_t)c * j + r] = Hm(r, c);
    if (!hessenberg_eigvals_inplace(j, work.data(), out)) return QP_E_INTERNAL;
  }
  return QP_OK;
}

// diagonalize_hessenberg_matrix -- src/arnoldi.jl:143-170
int diagonalize_hessenberg(const cplx* Hess, int ldh, int m, bool accumulate, cplx* out) {
  int offset = 0;
  for (int j = accumulate ? 1 : m; j <= m; ++j) {
    const int st = diagonalize_hessenberg_block(Hess, ldh, j, out + offset);
    if (st != QP_OK) return st;
    offset += j;
  }
  return QP_OK;
}

// ---------------------------------------------------------------------------
// extend_leja! -- src/newton.jl:97-148 (zero-based; `leja` must hold n + n_use)
// ---------------------------------------------------------------------------
// The reference ranks the candidates by  p_i = prod_j |z_i - leja_j|^(1/(n + n_use))  (the exponent only keeps the
// product inside the floating-point range, :127,:135), a left-to-right chain of one hypot and one pow per factor, and
// takes the first maximum (strict >).  Those pow calls were the largest part of the host time that the device waits
// for at the end of every Arnoldi sweep of newton! (0.10 of 1.1 ms per sweep at m = 20).  Here the ranking is done in
// two tiers that together make the reference's choice:
//   1. every candidate's  P_i = prod_j |z_i - leja_j|^2  is kept as mantissa x 2^exponent (one multiplication and
//      one frexp per factor, exact to a few ulp, no overflow / underflow for any number of factors or any distance: the
//      difference is scaled by a power of two BEFORE it is squared).  Any increasing function of the product ranks
//      alike, so the candidate with the largest P_i is the reference's choice -- unless another candidate comes
//      within the rounding noise of the reference's own chain;
//   2. candidates whose P_i lies within 1e-8 (relative) of the largest -- Ritz values of nested Hessenberg blocks that
//      have converged to the same eigenvalue, exact duplicates -- are re-ranked by the reference's arithmetic itself
//      (the hypot / pow chain, factor by factor, first maximum wins among them in index order).  A candidate outside
//      that set is below the maximum by more than 1e-8 / (2 (n + n_use)) in the reference's p, orders of magnitude
//      above the chain's rounding error, so it cannot be the reference's choice.
// The selection is therefore the reference's (the oracle's) index for index -- tests/test_cabi_host.py, including
// adversarial inputs with duplicated and nearly equal candidates -- at the cost of the fast path.
// prod_folded (optional, n > 0): the products over the first n Leja points, which a caller may have built
// while the candidates were arriving (leja_fold_candidate).
// frexp / ldexp on the bit pattern (normal numbers; anything else takes the library call): the two Leja loops run
// one of each per factor and candidate, at the end of every Arnoldi sweep while the device waits
static inline double frexp_bits(double x, int* e) {
  uint64_t b;
  std::memcpy(&b, &x, sizeof b);
  const int ex = (int)((b >> 52) & 0x7ff);
  if (ex == 0 || ex == 0x7ff) return std::frexp(x, e);
  *e = ex - 1022;
  b = (b & ~(0x7ffULL << 52)) | (1022ULL << 52);
  std::memcpy(&x, &b, sizeof b);
  return x;
}
static inline double scale_pow2(double x, int k) {   // x * 2^k, exact
  if (k < -1000 || k > 1000) return std::ldexp(x, k);
  const uint64_t b = (uint64_t)(k + 1023) << 52;
  double f;
  std::memcpy(&f, &b, sizeof f);
  return x * f;
}
static inline void scaled_mul_dist2(ScaledProd& p, cplx d) {
  // |d|^2 with the range of |d| taken out first: scale by 2^-k, k = exponent of the larger component
  const double ax = std::fabs(d.real()), ay = std::fabs(d.imag());
  const double big = ax > ay ? ax : ay;
  if (!(big > 0.0)) {   // zero distance (or NaN): the product is zero
    p.m = (big == 0.0) ? 0.0 : big;
    return;
  }
  int k = 0;
  (void)frexp_bits(big, &k);
  const double sx = scale_pow2(d.real(), -k), sy = scale_pow2(d.imag(), -k);   // larger component in [0.5, 1)
  int de = 0;
  p.m = frexp_bits(p.m * (sx * sx + sy * sy), &de);   // m stays in [0.5, 1) (or 0)
  p.e += de + 2 * k;
}
static inline bool scaled_greater(const ScaledProd& a, const ScaledProd& b) {   // a > b, both >= 0
  if (a.m == 0.0 || b.m == 0.0) return a.m > b.m;
  return a.e != b.e ? a.e > b.e : a.m > b.m;
}
// a >= b (1 - tol)?  (b > 0)
static inline bool scaled_near(const ScaledProd& a, const ScaledProd& b, double tol) {
  if (a.m == 0.0) return false;
  const int64_t de = (int64_t)a.e - (int64_t)b.e;
  if (de > 2) return true;
  if (de < -2) return false;
  return scale_pow2(a.m, (int)de) >= b.m * (1.0 - tol);
}

ScaledProd leja_fold_candidate(const cplx* leja, int n, cplx z) {
  ScaledProd p{1.0, 0};
  for (int j = 0; j < n; ++j) scaled_mul_dist2(p, z - leja[j]);
  return p;
}

// the reference's product for one candidate, operation by operation (src/newton.jl:132-136)
static inline double leja_reference_product(const cplx* leja, int n_have, cplx z, double exponent) {
  double p = 1.0;
  // abs(::ComplexF64) is hypot (libm's, as the oracle's NumPy uses; libstdc++'s std::abs(complex) may take a scaled
  // sqrt instead, one ulp apart, and exact ties between candidates are decided by exactly these bits)
  for (int j = 0; j < n_have; ++j) {
    const cplx d = z - leja[j];
    p = p * std::pow(std::hypot(d.real(), d.imag()), exponent);
  }
  return p;
}

void extend_leja(cplx* leja, int n, cplx* newpoints, int n_new, int n_use, const ScaledProd* prod_folded) {
  int u = n_new - 1;
  int i_add_start = 0;
  if (n == 0) {
    cplx z_last = newpoints[u];
    for (int i = 0; i < u; ++i) {
      if (std::hypot(newpoints[i].real(), newpoints[i].imag()) > std::hypot(z_last.real(), z_last.imag())) {
        newpoints[u] = newpoints[i];
        newpoints[i] = z_last;
        z_last = newpoints[u];
      }
    }
    leja[0] = newpoints[u];
    i_add_start = 1;
  }
  const double exponent = 1.0 / (double)(n + n_use);                                   // :127
  constexpr double kNearTie = 1e-8;
  // one running product per candidate; every new Leja point appends one factor to each (the reference
  // recomputes the whole product for every new point: O(m^3 n) pow calls)
  std::vector<ScaledProd> prod((size_t)std::max(n_new, 1), ScaledProd{1.0, 0});
  int n_done = 0;  // number of Leja points already folded into prod[]
  if (prod_folded && n > 0) {
    std::copy(prod_folded, prod_folded + n_new, prod.begin());
    n_done = n;
  }
  for (int i_add = i_add_start; i_add < n_use; ++i_add) {
    const int n_have = n + i_add;
    for (int i = 0; i <= u - i_add; ++i) {
      ScaledProd p = prod[i];
      for (int j = n_done; j < n_have; ++j) scaled_mul_dist2(p, newpoints[i] - leja[j]);
      prod[i] = p;
    }
    n_done = n_have;
    ScaledProd p_max{0.0, 0};
    int i_max = 0;
    for (int i = 0; i <= u - i_add; ++i) {
      if (scaled_greater(prod[i], p_max)) {  // strict: first maximum wins (src/newton.jl:137-140)
        p_max = prod[i];
        i_max = i;
      }
    }
    if (p_max.m > 0.0) {
      // tier 2: anything within the noise of the reference's chain is decided by that chain
      bool tie = false;
      for (int i = 0; i <= u - i_add && !tie; ++i) tie = (i != i_max) && scaled_near(prod[i], p_max, kNearTie);
      if (tie) {
        double pr_max = 0.0;   // (every tied candidate's product is positive: the loop below is the reference's :130-141)
        int ir_max = i_max;
        for (int i = 0; i <= u - i_add; ++i) {
          if (!scaled_near(prod[i], p_max, kNearTie)) continue;
          const double pr = leja_reference_product(leja, n_have, newpoints[i], exponent);
          if (pr > pr_max) {
            pr_max = pr;
            ir_max = i;
          }
        }
        i_max = ir_max;
      }
    }
    leja[n + i_add] = newpoints[i_max];
    // remove the used point by replacing it with the last unused point (:145)
    newpoints[i_max] = newpoints[u - i_add];
    prod[i_max] = prod[u - i_add];
  }
}

cplx eval_func(int func_id, qp_func_cb cb, void* user, cplx z) {
  switch (func_id) {
    case QP_FUNC_EXPMI: return std::exp(cplx(0, -1) * z);
    case QP_FUNC_EXP: return std::exp(z);
    default: {
      qp_c128 zi{z.real(), z.imag()}, zo{0, 0};
      cb(&zi, &zo, user);
      return cplx(zo.re, zo.im);
    }
  }
}

// extend_newton_coeffs! -- src/newton.jl:176-214 (zero-based; `a` must hold n_leja)
int extend_newton_coeffs(cplx* a, int n_a, const cplx* leja, int func_id, qp_func_cb cb,
                         void* user, int n_leja, double radius) {
  int m = n_leja - n_a;
  int n0 = n_a;
  if (!(radius > 0)) return QP_E_BAD_ARG;
  if (n_a == 0) {
    a[0] = eval_func(func_id, cb, user, leja[0]);
    n0 = 1;
  }
  for (int k = n0; k < n_a + m; ++k) {
    cplx d = 1.0, pn = 0.0;
    for (int n = 1; n <= k - 1; ++n) {
      cplx zd = leja[k] - leja[n - 1];
      d = d * zd / radius;
      pn = pn + a[n] * d;
    }
    cplx zd = leja[k] - leja[k - 1];
    d = d * zd / radius;
    if (!(std::abs(d) > 1e-200)) return QP_E_DIVDIFF_UNDERFLOW;
    a[k] = (eval_func(func_id, cb, user, leja[k]) - a[0] - pn) / d;
  }
  return QP_OK;
}

// ---------------------------------------------------------------------------
// CSC -> CSR (stable counting sort; columns end up ascending within each row because
// CSC is traversed column by column)
// ---------------------------------------------------------------------------
int csc_to_csr(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowval,
               const qp_c128* nzval, int base, int64_t* rowptr, int32_t* col, qp_c128* vals) {
  // the pointer array is validated before anything is indexed through it: it must start at the index
  // base, be monotone and stay inside [0, nnz]
  if (colptr[0] != base) return QP_E_BAD_ARG;
  int64_t nnz = colptr[ncols] - base;
  if (nnz < 0) return QP_E_BAD_ARG;
  for (int64_t c = 0; c < ncols; ++c)
    if (colptr[c + 1] < colptr[c] || colptr[c + 1] - base > nnz) return QP_E_BAD_ARG;
  std::fill(rowptr, rowptr + nrows + 1, (int64_t)0);
  for (int64_t p = 0; p < nnz; ++p) {
    int64_t r = rowval[p] - base;
    if (r < 0 || r >= nrows) return QP_E_BAD_ARG;
    rowptr[r + 1]++;
  }
  for (int64_t r = 0; r < nrows; ++r) rowptr[r + 1] += rowptr[r];
  std::vector<int64_t> next(rowptr, rowptr + nrows);
  for (int64_t c = 0; c < ncols; ++c) {
    for (int64_t p = colptr[c] - base; p < colptr[c + 1] - base; ++p) {
      int64_t r = rowval[p] - base;
      int64_t dst = next[r]++;
      col[dst] = (int32_t)c;
      vals[dst] = nzval[p];
    }
  }
  return QP_OK;
}

void partition_rows(const int64_t* rowptr, int64_t nrows, int nparts, int balance,
                    int64_t* bounds) {
  bounds[0] = 0;
  if (balance == 0) {
    int64_t base = nrows / nparts, rem = nrows % nparts;
    for (int r = 0; r < nparts; ++r) bounds[r + 1] = bounds[r] + base + (r < rem ? 1 : 0);
  } else {
    int64_t nnz = rowptr[nrows];
    for (int k = 1; k < nparts; ++k) {
      int64_t target = (int64_t)(((__int128)k * nnz) / nparts);
      bounds[k] = std::lower_bound(rowptr, rowptr + nrows + 1, target) - rowptr;
    }
    bounds[nparts] = nrows;
  }
}

}  // namespace qp

// ---------------------------------------------------------------------------
// C ABI of the host-only entry points
// ---------------------------------------------------------------------------
using qp::cplx;

extern "C" {

int qp_cheby_coeffs(double Delta, double dt, double limit, double* out, int cap, int* n_out) {
  QP_TRY
  if (!n_out) return qp::fail(QP_E_BAD_ARG, "qp_cheby_coeffs: n_out is NULL");
  auto c = qp::cheby_coeffs(Delta, dt, limit);
  *n_out = (int)c.size();
  if ((int)c.size() > cap || !out)
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_coeffs: need room for %d coefficients", (int)c.size());
  std::memcpy(out, c.data(), c.size() * sizeof(double));
  return QP_OK;
  QP_CATCH
}

int qp_hessenberg_eigvals(const qp_c128* Hess, int ldh, int m, int accumulate, qp_c128* out) {
  QP_TRY
  if (!Hess || !out || m < 1 || ldh < m) return qp::fail(QP_E_BAD_ARG, "qp_hessenberg_eigvals: bad args");
  int st = qp::diagonalize_hessenberg(reinterpret_cast<const cplx*>(Hess), ldh, m, accumulate != 0,
                                      reinterpret_cast<cplx*>(out));
  if (st != QP_OK) return qp::fail(st, "Hessenberg QR did not converge");
  return QP_OK;
  QP_CATCH
}

int qp_extend_leja(qp_c128* leja, int n, qp_c128* newpoints, int n_newpoints, int n_use) {
  QP_TRY
  if (!leja || !newpoints || n < 0 || n_use < 1 || n_newpoints < n_use)
    return qp::fail(QP_E_BAD_ARG, "qp_extend_leja: bad args");
  // through the pre-folded form that newton! uses (the head of every candidate's product built first,
  // as the restart loop does while the Hessenberg columns arrive): same products, same selection
  const cplx* lj = reinterpret_cast<const cplx*>(leja);
  const cplx* np_ = reinterpret_cast<const cplx*>(newpoints);
  std::vector<qp::ScaledProd> head((size_t)n_newpoints, qp::ScaledProd{1.0, 0});
  if (n > 0)
    for (int i = 0; i < n_newpoints; ++i) head[i] = qp::leja_fold_candidate(lj, n, np_[i]);
  qp::extend_leja(reinterpret_cast<cplx*>(leja), n, reinterpret_cast<cplx*>(newpoints), n_newpoints,
                  n_use, n > 0 ? head.data() : nullptr);
  return QP_OK;
  QP_CATCH
}

int qp_extend_newton_coeffs(qp_c128* a, int n_a, const qp_c128* leja, int func_id, qp_func_cb cb,
                            void* user, int n_leja, double radius) {
  QP_TRY
  if (!a || !leja || n_a < 0 || n_leja < n_a) return qp::fail(QP_E_BAD_ARG, "qp_extend_newton_coeffs: bad args");
  if (func_id == QP_FUNC_CALLBACK && !cb) return qp::fail(QP_E_BAD_ARG, "callback func is NULL");
  int st = qp::extend_newton_coeffs(reinterpret_cast<cplx*>(a), n_a, reinterpret_cast<const cplx*>(leja),
                                    func_id, cb, user, n_leja, radius);
  if (st == QP_E_DIVDIFF_UNDERFLOW) return qp::fail(st, "Divided differences too small");
  if (st != QP_OK) return qp::fail(st, "qp_extend_newton_coeffs failed");
  return QP_OK;
  QP_CATCH
}

int qp_csc_to_csr_host(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowval,
                       const qp_c128* nzval, int index_base, int64_t* rowptr_out, int32_t* col_out,
                       qp_c128* vals_out) {
  QP_TRY
  if (!colptr || !rowval || !nzval || !rowptr_out || !col_out || !vals_out || nrows < 0 || ncols < 0 ||
      ncols > INT32_MAX)
    return qp::fail(QP_E_BAD_ARG, "qp_csc_to_csr_host: bad args");
  int st = qp::csc_to_csr(nrows, ncols, colptr, rowval, nzval, index_base, rowptr_out, col_out, vals_out);
  if (st != QP_OK) return qp::fail(st, "qp_csc_to_csr_host: row index out of range");
  return QP_OK;
  QP_CATCH
}

int qp_partition_rows_host(const int64_t* rowptr, int64_t nrows, int nparts, int balance,
                           int64_t* bounds_out) {
  QP_TRY
  if (!rowptr || !bounds_out || nparts < 1 || nrows < 0) return qp::fail(QP_E_BAD_ARG, "qp_partition_rows_host: bad args");
  qp::partition_rows(rowptr, nrows, nparts, balance, bounds_out);
  return QP_OK;
  QP_CATCH
}

}  // extern "C"
