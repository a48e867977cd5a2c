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
  double2* T = nullptr;        // n x n workspace: A_k rho (library path)
  double2* Tk = nullptr;       // nc x (n x n) workspace: all A_k rho (fused path)
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

// ---------------------------------------------------------------------------
// Hand-written fp64 matrix-core kernel for the sizes where a chain of library GEMMs is bound
// by its launches (default: n <= 320):  Y = beta Y + sum_j alpha_j P_j op_j(Q_j)  in ONE launch, all
// matrices n x n, column-major.  v_mfma_f64_16x16x4_f64: lane l holds A[l & 15][l >> 4] and
// B[l >> 4][l & 15]; D register r of lane l is D[(l >> 4) + 4 r][l & 15].  A complex
// product is four real ones (Re += ar br - ai bi, Im += ar bi + ai br).
//   * one workgroup = one 16 x 16 tile of Y (BM = 1); its four wavefronts split the inner
//     dimension in four and are summed through LDS in wave order (deterministic);
//   * operands go from L2 straight into registers in the MFMA lane layout (the matrices are at
//     most 4 MB each), software-pipelined D k-steps ahead across the flattened (term, k) loop;
//   * `batched`: workgroup z computes only term z into Y + z n^2 (the K products A_k rho).
// ---------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kMaxTerms = 10;
struct GemmTerm {
  const double2* P;
  const double2* Q;
  double2 alpha;
  int conjT;   // 0: Q, 1: Q^H
};
struct GemmTerms {
  GemmTerm t[kMaxTerms];
  int n_terms;
};

template <int BM>
__global__ __launch_bounds__(256) void zgemm_sum_kernel(double2* __restrict__ Y, int n, double2 beta, GemmTerms terms,
                                                        int batched) {
  constexpr int D = (BM == 1) ? 8 : 3;   // k-steps of prefetch
  __shared__ double red[3][BM * BM][2][4][64];   // partial tiles of waves 1..3
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.x * 16 * BM, col0 = blockIdx.y * 16 * BM;
  const int first = batched ? blockIdx.z : 0;
  const int nt = batched ? 1 : terms.n_terms;
  double2* __restrict__ Yz = Y + (batched ? (size_t)blockIdx.z * n * n : 0);
  // this wave's share of the inner dimension, in k-steps of 4
  const int ksteps = (n + 3) / 4;
  const int per = (ksteps + 3) / 4;
  const int sbeg = wave * per;
  const int send = min(ksteps, sbeg + per);
  const int nsteps = max(send - sbeg, 0);
  const int total = nt * nsteps;
  const int li = lane & 15, lk = lane >> 4;

  v4d cr[BM][BM], ci[BM][BM];
#pragma unroll
  for (int a = 0; a < BM; ++a)
#pragma unroll
    for (int b = 0; b < BM; ++b) cr[a][b] = ci[a][b] = v4d{0.0, 0.0, 0.0, 0.0};

  double2 fa[D][BM], fb[D][BM];
  auto load = [&](int slot, int flat) {
    const int t = first + flat / nsteps;
    const int k = (sbeg + flat % nsteps) * 4 + lk;
    const GemmTerm& tm = terms.t[t];
    const bool kin = k < n;
#pragma unroll
    for (int a = 0; a < BM; ++a) {
      const int r = row0 + a * 16 + li;
      const bool ok = kin && r < n;
      const double2 v = tm.P[ok ? (size_t)k * n + r : 0];
      fa[slot][a] = ok ? v : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int b = 0; b < BM; ++b) {
      const int c = col0 + b * 16 + li;
      const bool ok = kin && c < n;
      double2 v;
      if (tm.conjT) {   // op(Q)[k][c] = conj(Q[c][k])
        v = tm.Q[ok ? (size_t)k * n + c : 0];
        v.y = -v.y;
      } else {
        v = tm.Q[ok ? (size_t)c * n + k : 0];
      }
      fb[slot][b] = ok ? v : make_double2(0.0, 0.0);
    }
  };
  auto compute = [&](int slot, int flat) {
    const double2 al = terms.t[first + flat / nsteps].alpha;
#pragma unroll
    for (int a = 0; a < BM; ++a) {
      const double2 x = fa[slot][a];
      const double ar = al.x * x.x - al.y * x.y, ai = al.x * x.y + al.y * x.x;   // alpha_j folded into the A fragment
#pragma unroll
      for (int b = 0; b < BM; ++b) {
        const double2 y = fb[slot][b];
        cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, y.x, cr[a][b], 0, 0, 0);
        cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, y.y, cr[a][b], 0, 0, 0);
        ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, y.y, ci[a][b], 0, 0, 0);
        ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, y.x, ci[a][b], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < total) load(d, d);
  for (int s = 0; s < total; s += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (s + d < total) {
        compute(d, s + d);
        if (s + d + D < total) load(d, s + d + D);
      }
    }
  }
  // sum the four k-quarters in wave order
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < BM; ++a)
#pragma unroll
      for (int b = 0; b < BM; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[wave - 1][a * BM + b][0][r][lane] = cr[a][b][r];
          red[wave - 1][a * BM + b][1][r][lane] = ci[a][b][r];
        }
  }
  __syncthreads();
  if (wave == 0) {
    const bool bz = (beta.x == 0.0 && beta.y == 0.0) || batched;
#pragma unroll
    for (int a = 0; a < BM; ++a)
#pragma unroll
      for (int b = 0; b < BM; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double vr = cr[a][b][r], vi = ci[a][b][r];
          for (int w = 0; w < 3; ++w) {
            vr += red[w][a * BM + b][0][r][lane];
            vi += red[w][a * BM + b][1][r][lane];
          }
          const int row = row0 + a * 16 + lk + 4 * r, col = col0 + b * 16 + li;
          if (row < n && col < n) {
            double2* y = Yz + (size_t)col * n + row;
            if (!bz) {
              const double2 o = *y;
              vr += beta.x * o.x - beta.y * o.y;
              vi += beta.x * o.y + beta.y * o.x;
            }
            *y = make_double2(vr, vi);
          }
        }
  }
}

int launch_zgemm_sum(hipStream_t s, double2* Y, int n, double2 beta, const GemmTerms& terms, int batched, Stats* st) {
  if (terms.n_terms == 0) return QP_OK;
  // 16 x 16 tiles: as many workgroups as the matrix offers.  (A 32 x 32 tile per wavefront halves
  // the operand traffic but measured slower than the library chain from n = 320 on, where the
  // library is used anyway: profiles/r01/liouville_matrix_free.txt.)
  const int tiles = (n + 15) / 16;
  const dim3 grid(tiles, tiles, batched ? terms.n_terms : 1);
  hipLaunchKernelGGL(zgemm_sum_kernel<1>, grid, dim3(256), 0, s, Y, n, beta, terms, batched);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
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
  if (L->n <= L->ctx->tun.liouville_fused_n && 2 + L->nc <= kMaxTerms) {
    // two launches of the fused matrix-core kernel: T_k = A_k X (batched), then
    // Y = beta Y + alpha (M_L X - X M_R) + alpha scale s_d sum_k T_k A_k^+
    const cplx a(alpha.x, alpha.y);
    if (L->nc > 0) {
      GemmTerms t1;
      t1.n_terms = L->nc;
      for (int k = 0; k < L->nc; ++k) t1.t[k] = GemmTerm{L->A[k], x, make_double2(1.0, 0.0), 0};
      QP_CHECK(launch_zgemm_sum(s, L->Tk, n, make_double2(0.0, 0.0), t1, 1, st));
    }
    GemmTerms t2;
    t2.n_terms = 2 + L->nc;
    t2.t[0] = GemmTerm{L->ML, x, d2(a), 0};
    t2.t[1] = GemmTerm{x, L->MR, d2(-a), 0};
    for (int k = 0; k < L->nc; ++k) t2.t[2 + k] = GemmTerm{L->Tk + (size_t)k * L->n * L->n, L->A[k], d2(a * L->scale * L->s_d), 1};
    QP_CHECK(launch_zgemm_sum(s, y, n, beta, t2, 0, st));
    if (st) st->n_matvec++;
    return QP_OK;
  }
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
  if (L->Tk) (void)hipFree(L->Tk);
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
  op->A.tun = &ctx->tun;
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
  if (nc > 0) QP_CHECK(dev_alloc(&Lp->Tk, n2 * (size_t)nc));
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
