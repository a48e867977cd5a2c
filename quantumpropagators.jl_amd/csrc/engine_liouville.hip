// Matrix-free Liouvillian: the superoperator of liouvillian(H, c_ops;
// convention) (src/generators.jl:473-631) applied to the column-major vec(rho) as n x n
// complex GEMMs on the fp64 matrix cores (rocBLAS zgemm, a plain library GEMM), instead of
// an n^2 x n^2 sparse matrix with 2 n^3 entries for dense H:
//   L rho = M_L rho - rho M_R + s_d sum_k A_k rho A_k^+
//   M_L = s_h H - (s_d / 2) G,  M_R = s_h H + (s_d / 2) G,  G = sum_k A_k^+ A_k,  H = sum_l c_l H_l
//   (s_h, s_d) = (1, i) for :TDSE, (i, 1) for :LvN   (ham_to_superop :473-490, lindblad_to_superop :493-512)
#include <dlfcn.h>

#include <rocblas/rocblas.h>

#include "engine.h"

namespace {

struct RocblasApi {
  void* handle = nullptr;
  rocblas_status (*create_handle)(rocblas_handle*) = nullptr;
  rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
  rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
  rocblas_status (*zgemm)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int,
                          const rocblas_double_complex*, const rocblas_double_complex*, rocblas_int,
                          const rocblas_double_complex*, rocblas_int, const rocblas_double_complex*,
                          rocblas_double_complex*, rocblas_int) = nullptr;
};

// The rocBLAS of this process: QP_ROCBLAS_PATH if set (the Python binding points it at the
// copy PyTorch-ROCm ships, which is then the one already loaded), else the system one.
int rocblas_load(RocblasApi* api) {
  const char* env = getenv("QP_ROCBLAS_PATH");
  const char* names[] = {env, "librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"};
  void* h = nullptr;
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return qp::fail(QP_E_INTERNAL, "librocblas.so not found (set QP_ROCBLAS_PATH): %s", dlerror());
  api->handle = h;
  api->create_handle = reinterpret_cast<decltype(api->create_handle)>(dlsym(h, "rocblas_create_handle"));
  api->destroy_handle = reinterpret_cast<decltype(api->destroy_handle)>(dlsym(h, "rocblas_destroy_handle"));
  api->set_stream = reinterpret_cast<decltype(api->set_stream)>(dlsym(h, "rocblas_set_stream"));
  api->zgemm = reinterpret_cast<decltype(api->zgemm)>(dlsym(h, "rocblas_zgemm"));
  if (!api->create_handle || !api->destroy_handle || !api->set_stream || !api->zgemm)
    return qp::fail(QP_E_INTERNAL, "librocblas.so does not export the expected entry points");
  return QP_OK;
}

#define QP_ROCBLAS(expr)                                                                          \
  do {                                                                                            \
    rocblas_status r__ = (expr);                                                                  \
    if (r__ != rocblas_status_success) return qp::fail(QP_E_INTERNAL, "rocBLAS: %s failed (status %d)", #expr, (int)r__); \
  } while (0)

struct Liouville {
  qp_ctx* ctx = nullptr;
  int64_t n = 0;
  int nterms = 0, nc = 0;
  cplx s_h = 1.0, s_d = cplx(0, 1);
  RocblasApi api;
  rocblas_handle blas = nullptr;
  std::vector<double2*> H;     // nterms dense n x n (column-major)
  double2** H_dev = nullptr;   // device array of the plane pointers
  std::vector<double2*> A;     // nc Lindblad operators
  double2* G = nullptr;        // sum_k A_k^+ A_k
  double2* ML = nullptr;       // s_h H - (s_d / 2) G, including the operator's scale
  double2* MR = nullptr;       // s_h H + (s_d / 2) G, including the operator's scale
  double2* T = nullptr;        // n x n workspace: A_k rho
  double2* scratch = nullptr;  // n^2 workspace for the unfused Chebyshev term
  cplx scale = 1.0;            // the operator's scale, applied to the dissipator GEMMs
};

// ML / MR = sum_l c_l H_l  -/+  g G   (c_l already carry s_h and the scale, g = scale s_d / 2)
__global__ __launch_bounds__(qp::kThreads) void liouville_combine_kernel(double2* __restrict__ ML, double2* __restrict__ MR,
                                                                         const double2* const* __restrict__ H,
                                                                         qp::CoefBlock c, int nterms,
                                                                         const double2* __restrict__ G, double2 g,
                                                                         int64_t n2) {
  for (int64_t p = (int64_t)blockIdx.x * qp::kThreads + threadIdx.x; p < n2; p += (int64_t)gridDim.x * qp::kThreads) {
    double2 h = make_double2(0.0, 0.0);
    for (int l = 0; l < nterms; ++l) {
      const double2 v = H[l][p];
      h.x += c.c[l].x * v.x - c.c[l].y * v.y;
      h.y += c.c[l].x * v.y + c.c[l].y * v.x;
    }
    double2 gg = make_double2(0.0, 0.0);
    if (G) {
      const double2 v = G[p];
      gg = make_double2(g.x * v.x - g.y * v.y, g.x * v.y + g.y * v.x);
    }
    ML[p] = make_double2(h.x - gg.x, h.y - gg.y);
    MR[p] = make_double2(h.x + gg.x, h.y + gg.y);
  }
}

inline const rocblas_double_complex* rc(const double2* p) { return reinterpret_cast<const rocblas_double_complex*>(p); }
inline rocblas_double_complex* rc(double2* p) { return reinterpret_cast<rocblas_double_complex*>(p); }
inline rocblas_double_complex rz(cplx z) { return rocblas_double_complex(z.real(), z.imag()); }

int liouville_refresh(qp_operator* op) {
  Liouville* L = static_cast<Liouville*>(op->mf);
  qp_ctx* ctx = op->ctx;
  const int drift = op->nops - op->ncoeffs;
  if (op->nops > qp::kCoefBlock) return qp::fail(QP_E_BAD_ARG, "at most %d Hamiltonian terms", qp::kCoefBlock);
  qp::CoefBlock cb;
  for (int l = 0; l < op->nops; ++l) {
    cplx c = op->scale * L->s_h;
    if (l >= drift) c *= op->coeffs[l - drift];
    cb.c[l] = d2(c);
  }
  L->scale = op->scale;
  const int64_t n2 = L->n * L->n;
  const int grid = (int)std::min<int64_t>((n2 + qp::kThreads - 1) / qp::kThreads, 4096);
  hipLaunchKernelGGL(liouville_combine_kernel, dim3(grid), dim3(qp::kThreads), 0, ctx->stream, L->ML, L->MR, L->H_dev, cb,
                     op->nops, L->nc > 0 ? L->G : nullptr, d2(op->scale * L->s_d * 0.5), n2);
  QP_HIP(hipGetLastError());
  ctx->stats.n_launch++;
  return QP_OK;
}

// y = beta y + alpha L x,  x and y the column-major n x n density matrices
int liouville_apply(hipStream_t s, void* self, const double2* x, double2* y, double2 alpha, double2 beta, Stats* st) {
  Liouville* L = static_cast<Liouville*>(self);
  const rocblas_int n = (rocblas_int)L->n;
  QP_ROCBLAS(L->api.set_stream(L->blas, s));
  const cplx a(alpha.x, alpha.y);
  const rocblas_double_complex al = rz(a), mal = rz(-a), be = rz(cplx(beta.x, beta.y)), one = rz(1.0), zero = rz(0.0);
  // Y = beta Y + alpha M_L X;  Y -= alpha X M_R
  QP_ROCBLAS(L->api.zgemm(L->blas, rocblas_operation_none, rocblas_operation_none, n, n, n, &al, rc(L->ML), n, rc(x), n, &be, rc(y), n));
  QP_ROCBLAS(L->api.zgemm(L->blas, rocblas_operation_none, rocblas_operation_none, n, n, n, &mal, rc(x), n, rc(L->MR), n, &one, rc(y), n));
  // Y += alpha scale s_d A_k X A_k^+
  const rocblas_double_complex ad = rz(a * L->scale * L->s_d);
  for (int k = 0; k < L->nc; ++k) {
    QP_ROCBLAS(L->api.zgemm(L->blas, rocblas_operation_none, rocblas_operation_none, n, n, n, &one, rc(L->A[k]), n, rc(x), n, &zero, rc(L->T), n));
    QP_ROCBLAS(L->api.zgemm(L->blas, rocblas_operation_none, rocblas_operation_conjugate_transpose, n, n, n, &ad, rc(L->T), n, rc(L->A[k]), n, &one, rc(y), n));
  }
  if (st) {
    st->n_launch += 2 + 2 * L->nc;
    st->n_matvec++;
  }
  return QP_OK;
}

double2* liouville_scratch(void* self) { return static_cast<Liouville*>(self)->scratch; }

void liouville_free(qp_operator* op) {
  Liouville* L = static_cast<Liouville*>(op->mf);
  if (!L) return;
  for (auto p : L->H) (void)hipFree(p);
  for (auto p : L->A) (void)hipFree(p);
  if (L->H_dev) (void)hipFree(L->H_dev);
  if (L->G) (void)hipFree(L->G);
  if (L->ML) (void)hipFree(L->ML);
  if (L->MR) (void)hipFree(L->MR);
  if (L->T) (void)hipFree(L->T);
  if (L->scratch) (void)hipFree(L->scratch);
  if (L->blas) (void)L->api.destroy_handle(L->blas);
  delete L;
  op->mf = nullptr;
}

}  // namespace

extern "C" {

int qp_liouvillian_create(qp_ctx* ctx, int64_t n, const qp_c128* const* H_terms, int nterms, int ncoeffs,
                          const qp_c128* const* c_ops, int nc, int convention, qp_operator** out) {
  QP_TRY
  if (!ctx || !out || n < 1 || nterms < 0 || nc < 0 || ncoeffs < 0 || ncoeffs > nterms || (nterms > 0 && !H_terms) ||
      (nc > 0 && !c_ops) || nterms + nc == 0)
    return qp::fail(QP_E_BAD_ARG, "qp_liouvillian_create: bad arguments (need at least one of H and c_ops)");
  if (convention != QP_CONV_TDSE && convention != QP_CONV_LVN) return qp::fail(QP_E_BAD_ARG, "convention must be :TDSE or :LvN");
  if (nterms > qp::kCoefBlock) return qp::fail(QP_E_BAD_ARG, "at most %d Hamiltonian terms", qp::kCoefBlock);
  if (n > 30000) return qp::fail(QP_E_BAD_ARG, "Hilbert space dimension %lld too large for a dense matrix-free Liouvillian", (long long)n);
  QP_CHECK(use(ctx));
  auto op = std::make_unique<qp_operator>();
  auto L = std::make_unique<Liouville>();
  op->ctx = ctx;
  L->ctx = ctx;
  L->n = n;
  L->nterms = nterms;
  L->nc = nc;
  L->s_h = (convention == QP_CONV_TDSE) ? cplx(1.0) : cplx(0, 1);
  L->s_d = (convention == QP_CONV_TDSE) ? cplx(0, 1) : cplx(1.0);
  const size_t n2 = (size_t)n * n;
  op->mf_free = liouville_free;
  op->mf_refresh = liouville_refresh;
  op->mf = L.get();
  Liouville* Lp = L.release();   // owned by op->mf from here on (freed by liouville_free)
  struct Guard {
    qp_operator* op;
    bool armed = true;
    ~Guard() {
      if (armed && op->mf_free) op->mf_free(op);
    }
  } guard{op.get()};
  QP_CHECK(rocblas_load(&Lp->api));
  QP_ROCBLAS(Lp->api.create_handle(&Lp->blas));
  for (int l = 0; l < nterms; ++l) {
    if (!H_terms[l]) return qp::fail(QP_E_BAD_ARG, "H term %d is NULL", l);
    double2* d = nullptr;
    QP_CHECK(dev_alloc(&d, n2));
    Lp->H.push_back(d);
    QP_HIP(hipMemcpy(d, H_terms[l], n2 * sizeof(double2), hipMemcpyHostToDevice));
  }
  QP_CHECK(dev_alloc(&Lp->H_dev, (size_t)std::max(nterms, 1)));
  if (nterms > 0) QP_HIP(hipMemcpy(Lp->H_dev, Lp->H.data(), nterms * sizeof(double2*), hipMemcpyHostToDevice));
  QP_CHECK(dev_alloc(&Lp->ML, n2));
  QP_CHECK(dev_alloc(&Lp->MR, n2));
  QP_CHECK(dev_alloc(&Lp->T, n2));
  QP_CHECK(dev_alloc(&Lp->scratch, n2 * 1));   // vec(rho) has n^2 entries
  if (nc > 0) {
    QP_CHECK(dev_alloc(&Lp->G, n2));
    const rocblas_double_complex one = rz(1.0), zero = rz(0.0);
    QP_ROCBLAS(Lp->api.set_stream(Lp->blas, ctx->stream));
    for (int k = 0; k < nc; ++k) {
      if (!c_ops[k]) return qp::fail(QP_E_BAD_ARG, "Lindblad operator %d is NULL", k);
      double2* d = nullptr;
      QP_CHECK(dev_alloc(&d, n2));
      Lp->A.push_back(d);
      QP_HIP(hipMemcpy(d, c_ops[k], n2 * sizeof(double2), hipMemcpyHostToDevice));
      // G += A_k^+ A_k
      QP_ROCBLAS(Lp->api.zgemm(Lp->blas, rocblas_operation_conjugate_transpose, rocblas_operation_none, (rocblas_int)n,
                               (rocblas_int)n, (rocblas_int)n, &one, rc(d), (rocblas_int)n, rc(d), (rocblas_int)n,
                               k == 0 ? &zero : &one, rc(Lp->G), (rocblas_int)n));
    }
  }
  DevMatrix& A = op->A;
  A.format = QP_FMT_MATFREE;
  A.nrows = A.ncols = (int64_t)n2;
  A.nnz = INT64_MAX / 4;   // never "small": the persistent kernels need stored entries
  A.stored = 0;
  A.matfree = Lp;
  A.matfree_apply = liouville_apply;
  A.matfree_scratch = liouville_scratch;
  op->nops = nterms;
  op->ncoeffs = ncoeffs;
  op->coeffs.assign((size_t)ncoeffs, cplx(1.0));
  QP_CHECK(liouville_refresh(op.get()));
  guard.armed = false;
  *out = op.release();
  return QP_OK;
  QP_CATCH
}

}  // extern "C"
