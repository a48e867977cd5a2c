// C ABI of libqprop_hip.so: handles, the Operator lazy sum, BLAS-1, and the host-side
// drivers of cheby!, arnoldi!, newton!, ritzvals/specrange that enqueue the HIP kernels
// of kernels.hip.  Host logic follows the reference line by line (citations inline);
// device work is stream-ordered, with host synchronisation only where the reference
// algorithm needs a scalar on the host (once per Newton restart, once per Arnoldi call).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <memory>

#include <rccl/rccl.h>

#include "device.h"

using qp::cplx;
using qp::DevMatrix;
using qp::kRB;
using qp::kRedBlocks;
using qp::Stats;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
namespace qp {
static thread_local std::string g_last_error;
void set_error(const char* msg) { g_last_error = msg ? msg : ""; }
int fail(int status, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return status;
}
}  // namespace qp

// ---------------------------------------------------------------------------
// handle types
// ---------------------------------------------------------------------------
struct qp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Stats stats;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double2* d_part = nullptr;   // kRedBlocks partials for qp_dot / qp_norm
  double2* h_part = nullptr;   // pinned mirror
};

struct qp_state {
  qp_ctx* ctx;
  double2* d;
  int64_t n;
  bool own;
};

struct qp_matrix {  // canonical host CSR (the result of the boundary's index work)
  qp_ctx* ctx;
  int64_t nrows, ncols, nnz;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> col;
  std::vector<cplx> vals;
};

struct HostLayoutData {
  int format = QP_FMT_RBCSR;
  std::vector<int64_t> bptr;   // RBCSR: all entries; HRB: upper section (c >= r)
  std::vector<int64_t> lptr;   // HRB: lower section (c < r)
  std::vector<int32_t> nlow;   // HRB: number of lower entries per row
  std::vector<int64_t> cmeta, lcmeta;  // per block: (byte offset of the column section << 1) | is16
  int64_t stored = 0, lstored = 0;
};

struct qp_operator {
  qp_ctx* ctx = nullptr;
  DevMatrix A;
  HostLayoutData layout;
  bool hermitian_planes = false;
  // CSR-ordered mirror of the current values for the batched (SpMM) path, built lazily
  int64_t* m_rowptr = nullptr;
  int32_t* m_cols = nullptr;
  int64_t* m_map = nullptr;     // position in A.vals (>= 0) or -(position)-1 for a conj-transposed value
  double2* m_vals = nullptr;
  uint64_t vals_epoch = 1, m_epoch = 0;
  int nops = 0, ncoeffs = 0;
  std::vector<int64_t> u_rowptr;  // union pattern (host), for get_csr and plane scatter
  std::vector<int32_t> u_col;
  std::vector<double2*> planes;   // device value planes, one per term, layout of A.vals
  double2** planes_dev = nullptr;
  double2* combined = nullptr;    // device, allocated on first non-trivial coefficient set
  bool planes_real = false;       // every value of every term has a zero imaginary part
  double* real_vals = nullptr;    // device copy of the real parts of the current values (see DevMatrix::vals_r)
  const double2* real_of = nullptr;  // the complex array real_vals was extracted from, when still valid
  std::vector<cplx> coeffs;
  cplx scale = 1.0;
};

struct qp_split {   // boundary / interior partition of an operator's row blocks (multi-GPU overlap)
  qp_operator* op = nullptr;          // identity check only; never dereferenced at destroy time
  int device = 0;
  int32_t* bmap_boundary = nullptr;
  int32_t* bmap_interior = nullptr;
  int32_t* mirror = nullptr;          // 64 * n_boundary entries: slab position or -1
  int64_t n_boundary = 0, n_interior = 0, nsend = 0;
  hipEvent_t ev_b = nullptr, ev_i = nullptr;
  // in-launch hand-off boundary(m) -> interior(m+1) (see SyncArgs in device.h)
  unsigned* counter = nullptr;        // [0] signal counter, [1] spin-timeout flag
  unsigned signals_issued = 0;
  unsigned wait_from_wg = 0;          // interior workgroups at or beyond this position poll
};

struct qp_krylov {
  qp_ctx* ctx;
  int64_t n;
  int nvec;
  double2* Q = nullptr;         // nvec vectors of length n, contiguous
  double2* hess_dev = nullptr;  // nvec x nvec column major
  double* norms_dev = nullptr;  // nvec
  double2* part = nullptr;      // 2 x kRedBlocks ping-pong partials
  double2* md_part = nullptr;   // kRedBlocks x 2 nvec multidot partials (low-sync MGS)
  double2* gram = nullptr;      // nvec x nvec Gram rows <q_i|q_k>, k < i
  int gram_rows = 0;            // rows 0 .. gram_rows-1 of `gram` describe the current basis
  double2* hcoef = nullptr;     // 2 nvec reduced inner products of the current column
  double2* q(int i) const { return Q + (size_t)i * n; }
};

struct qp_cheby {
  qp_ctx* ctx;
  int64_t n;
  double2* bufA = nullptr;
  double2* acc = nullptr;
  double* chk_part = nullptr;  // per-workgroup triples (allocated on demand)
  double* chk_out = nullptr;   // per-term triples
  int chk_wg = 0, chk_terms = 0;
  // hipGraph of one step's launches: replayed while the key (everything a launch argument
  // is derived from) stays the same, rebuilt when it changes
  struct GraphKey {
    const void *vals = nullptr, *cols = nullptr, *rowptr = nullptr, *psi = nullptr;
    int format = -1, variant = -1, n_coeffs = 0;
    double dt = 0, Delta = 0, E_min = 0;
    uint64_t a_hash = 0;
    bool operator==(const GraphKey& o) const {
      return vals == o.vals && cols == o.cols && rowptr == o.rowptr && psi == o.psi && format == o.format &&
             variant == o.variant && n_coeffs == o.n_coeffs && dt == o.dt && Delta == o.Delta && E_min == o.E_min &&
             a_hash == o.a_hash;
    }
  };
  GraphKey gkey, gpending;
  hipGraphExec_t gexec = nullptr;
  Stats gstats;   // what one replay adds to the context's counters
};

struct qp_newton {
  qp_ctx* ctx;
  int64_t n;
  int m_max;
  qp_krylov* q = nullptr;
  double2* v = nullptr;
  double2* npart = nullptr;  // kRedBlocks |psi|^2 partials
  double2* h_npart = nullptr;
  std::vector<cplx> a, leja;
  double radius = 0;
  int n_a = 0, n_leja = 0, restarts = 0;
};

namespace {

inline double2 d2(cplx z) { return make_double2(z.real(), z.imag()); }
inline double2 d2(qp_c128 z) { return make_double2(z.re, z.im); }
inline cplx cx(qp_c128 z) { return cplx(z.re, z.im); }

int use(qp_ctx* ctx) {
  QP_HIP(hipSetDevice(ctx->device));
  return QP_OK;
}

template <class T>
int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc((void**)p, count * sizeof(T));
  if (e != hipSuccess) return qp::fail(QP_E_ALLOC, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
  return QP_OK;
}

#define QP_CHECK(expr)           \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != QP_OK) return rc__; \
  } while (0)

// sum kRedBlocks partials on the host in index order
cplx sum_partials(const double2* h) {
  double re = 0, im = 0;
  for (int i = 0; i < kRedBlocks; ++i) {
    re += h[i].x;
    im += h[i].y;
  }
  return cplx(re, im);
}

int dot_sync(qp_ctx* ctx, const double2* x, const double2* y, int64_t n, cplx* out) {
  QP_CHECK(qp::launch_dot_partials(ctx->stream, x, y, ctx->d_part, n, &ctx->stats));
  QP_HIP(hipMemcpyAsync(ctx->h_part, ctx->d_part, kRedBlocks * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
  QP_HIP(hipStreamSynchronize(ctx->stream));
  *out = sum_partials(ctx->h_part);
  return QP_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------
extern "C" {

const char* qp_last_error(void) { return qp::g_last_error.c_str(); }

const char* qp_status_name(int s) {
  switch (s) {
    case QP_OK: return "QP_OK";
    case QP_E_BAD_ARG: return "QP_E_BAD_ARG";
    case QP_E_HIP: return "QP_E_HIP";
    case QP_E_DT_MISMATCH: return "QP_E_DT_MISMATCH";
    case QP_E_TOO_FEW_COEFFS: return "QP_E_TOO_FEW_COEFFS";
    case QP_E_NORMALIZATION: return "QP_E_NORMALIZATION";
    case QP_E_MAX_RESTARTS: return "QP_E_MAX_RESTARTS";
    case QP_E_DIVDIFF_UNDERFLOW: return "QP_E_DIVDIFF_UNDERFLOW";
    case QP_E_NO_DEVICE: return "QP_E_NO_DEVICE";
    case QP_E_ALLOC: return "QP_E_ALLOC";
    case QP_E_INTERNAL: return "QP_E_INTERNAL";
    case QP_E_M_MAX: return "QP_E_M_MAX";
    default: return "QP_E_UNKNOWN";
  }
}

int qp_version(void) { return 100; }

int qp_tuning_set(const char* key, int value) {
  if (!key) return qp::fail(QP_E_BAD_ARG, "key is NULL");
  if (std::strcmp(key, "rbcsr_variant") == 0) {
    qp::g_rbcsr_variant = value;
    return QP_OK;
  }
  if (std::strcmp(key, "arnoldi_mode") == 0) {
    qp::g_arnoldi_mode = value;
    return QP_OK;
  }
  if (std::strcmp(key, "hrb_lower_last") == 0) {
    qp::g_hrb_lower_last = value;
    return QP_OK;
  }
  if (std::strcmp(key, "real_vals") == 0) {
    qp::g_real_vals = value;
    return QP_OK;
  }
  if (std::strcmp(key, "stencil") == 0) {
    qp::g_stencil = value;
    return QP_OK;
  }
  if (std::strcmp(key, "acc_defer") == 0) {
    qp::g_acc_defer = value;
    return QP_OK;
  }
  if (std::strcmp(key, "cheby_graph") == 0) {
    qp::g_cheby_graph = value;
    return QP_OK;
  }
  if (std::strcmp(key, "small_nnz") == 0) {
    qp::g_small_nnz = value;
    return QP_OK;
  }
  if (std::strcmp(key, "spmm_tile") == 0) {
    qp::g_spmm_tile = value;
    return QP_OK;
  }
  if (std::strcmp(key, "split_mode") == 0) {
    qp::g_split_mode = value;
    return QP_OK;
  }
  return qp::fail(QP_E_BAD_ARG, "unknown tuning key %s", key);
}

int qp_device_count(int* n_out) {
  QP_TRY
  if (!n_out) return qp::fail(QP_E_BAD_ARG, "n_out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *n_out = n;
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
int qp_ctx_create(int device, void* stream, qp_ctx** out) {
  QP_TRY
  if (!out) return qp::fail(QP_E_BAD_ARG, "qp_ctx_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return qp::fail(QP_E_NO_DEVICE,
                    "no HIP device visible (%s): libqprop_hip has no CPU fallback for the prop_step! path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  }
  if (device < 0 || device >= n) return qp::fail(QP_E_BAD_ARG, "device %d out of range [0,%d)", device, n);
  QP_HIP(hipSetDevice(device));
  auto ctx = std::make_unique<qp_ctx>();
  ctx->device = device;
  if (stream == QP_STREAM_NULL) {
    ctx->stream = nullptr;   // HIP's null stream
  } else if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    QP_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  QP_HIP(hipEventCreate(&ctx->ev0));
  QP_HIP(hipEventCreate(&ctx->ev1));
  QP_CHECK(dev_alloc(&ctx->d_part, kRedBlocks));
  QP_HIP(hipHostMalloc((void**)&ctx->h_part, kRedBlocks * sizeof(double2), hipHostMallocDefault));
  *out = ctx.release();
  return QP_OK;
  QP_CATCH
}

int qp_ctx_destroy(qp_ctx* ctx) {
  QP_TRY
  if (!ctx) return QP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_part) (void)hipFree(ctx->d_part);
  if (ctx->h_part) (void)hipHostFree(ctx->h_part);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return QP_OK;
  QP_CATCH
}

int qp_sync(qp_ctx* ctx) {
  QP_TRY
  if (!ctx) return qp::fail(QP_E_BAD_ARG, "ctx is NULL");
  QP_CHECK(use(ctx));
  QP_HIP(hipStreamSynchronize(ctx->stream));
  return QP_OK;
  QP_CATCH
}

int qp_stats_get(qp_ctx* ctx, qp_stats* out) {
  if (!ctx || !out) return qp::fail(QP_E_BAD_ARG, "qp_stats_get: NULL argument");
  out->n_matvec = ctx->stats.n_matvec;
  out->n_cheby_steps = ctx->stats.n_cheby_steps;
  out->n_newton_steps = ctx->stats.n_newton_steps;
  out->n_restarts = ctx->stats.n_restarts;
  out->n_kernel_launches = ctx->stats.n_launch;
  out->spmv_bytes = ctx->stats.spmv_bytes;
  out->n_graph_launches = ctx->stats.n_graph_launch;
  return QP_OK;
}

int qp_stats_reset(qp_ctx* ctx) {
  if (!ctx) return qp::fail(QP_E_BAD_ARG, "ctx is NULL");
  ctx->stats = Stats();
  return QP_OK;
}

int qp_timer_begin(qp_ctx* ctx) {
  QP_TRY
  if (!ctx) return qp::fail(QP_E_BAD_ARG, "ctx is NULL");
  QP_CHECK(use(ctx));
  QP_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return QP_OK;
  QP_CATCH
}

int qp_timer_end(qp_ctx* ctx, double* elapsed_ms_out) {
  QP_TRY
  if (!ctx || !elapsed_ms_out) return qp::fail(QP_E_BAD_ARG, "qp_timer_end: NULL argument");
  QP_CHECK(use(ctx));
  QP_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  QP_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0;
  QP_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *elapsed_ms_out = ms;
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// matrices: canonicalise to host CSR (bit-exact index work)
// ---------------------------------------------------------------------------
int qp_matrix_create(qp_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* ptr,
                     const int64_t* idx, const void* vals, int val_dtype, int layout, int index_base,
                     int format, qp_matrix** out) {
  QP_TRY
  (void)format;
  if (!ctx || !out || !ptr || (nnz > 0 && (!idx || !vals)))
    return qp::fail(QP_E_BAD_ARG, "qp_matrix_create: NULL argument");
  if (nrows < 0 || ncols < 0 || nnz < 0 || ncols > INT32_MAX)
    return qp::fail(QP_E_BAD_ARG, "qp_matrix_create: bad shape %lld x %lld, nnz %lld", (long long)nrows,
                    (long long)ncols, (long long)nnz);
  if (index_base != 0 && index_base != 1) return qp::fail(QP_E_BAD_ARG, "index_base must be 0 or 1");
  if (val_dtype != QP_VAL_C128 && val_dtype != QP_VAL_F64) return qp::fail(QP_E_BAD_ARG, "bad val_dtype");
  auto m = std::make_unique<qp_matrix>();
  m->ctx = ctx;
  m->nrows = nrows;
  m->ncols = ncols;
  m->nnz = nnz;
  m->rowptr.resize(nrows + 1);
  m->col.resize(nnz);
  m->vals.resize(nnz);
  std::vector<qp_c128> cv;
  const qp_c128* v128 = nullptr;
  if (val_dtype == QP_VAL_F64) {
    cv.resize(nnz);
    const double* r = static_cast<const double*>(vals);
    for (int64_t p = 0; p < nnz; ++p) cv[p] = qp_c128{r[p], 0.0};
    v128 = cv.data();
  } else {
    v128 = static_cast<const qp_c128*>(vals);
  }
  if (layout == QP_LAYOUT_CSC) {
    if (ptr[ncols] - index_base != nnz) return qp::fail(QP_E_BAD_ARG, "colptr[end] does not match nnz");
    int st = qp::csc_to_csr(nrows, ncols, ptr, idx, v128, index_base, m->rowptr.data(), m->col.data(),
                            reinterpret_cast<qp_c128*>(m->vals.data()));
    if (st != QP_OK) return qp::fail(st, "qp_matrix_create: row index out of range");
  } else if (layout == QP_LAYOUT_CSR) {
    if (ptr[nrows] - index_base != nnz) return qp::fail(QP_E_BAD_ARG, "rowptr[end] does not match nnz");
    for (int64_t r = 0; r <= nrows; ++r) m->rowptr[r] = ptr[r] - index_base;
    for (int64_t r = 0; r < nrows; ++r)
      if (m->rowptr[r + 1] < m->rowptr[r]) return qp::fail(QP_E_BAD_ARG, "rowptr not monotone at row %lld", (long long)r);
    for (int64_t p = 0; p < nnz; ++p) {
      int64_t c = idx[p] - index_base;
      if (c < 0 || c >= ncols) return qp::fail(QP_E_BAD_ARG, "column index out of range at %lld", (long long)p);
      m->col[p] = (int32_t)c;
      m->vals[p] = cplx(v128[p].re, v128[p].im);
    }
    // canonical form: columns ascending within each row (stable)
    std::vector<std::pair<int32_t, cplx>> tmp;
    for (int64_t r = 0; r < nrows; ++r) {
      int64_t a = m->rowptr[r], b = m->rowptr[r + 1];
      bool sorted = true;
      for (int64_t p = a + 1; p < b; ++p)
        if (m->col[p] < m->col[p - 1]) { sorted = false; break; }
      if (sorted) continue;
      tmp.resize(b - a);
      for (int64_t p = a; p < b; ++p) tmp[p - a] = {m->col[p], m->vals[p]};
      std::stable_sort(tmp.begin(), tmp.end(), [](auto& x, auto& y) { return x.first < y.first; });
      for (int64_t p = a; p < b; ++p) { m->col[p] = tmp[p - a].first; m->vals[p] = tmp[p - a].second; }
    }
  } else {
    return qp::fail(QP_E_BAD_ARG, "bad layout");
  }
  *out = m.release();
  return QP_OK;
  QP_CATCH
}

int qp_matrix_destroy(qp_matrix* m) {
  delete m;
  return QP_OK;
}

int qp_matrix_info(const qp_matrix* m, int64_t* nrows, int64_t* ncols, int64_t* nnz, int* format,
                   int64_t* stored_nnz) {
  if (!m) return qp::fail(QP_E_BAD_ARG, "matrix is NULL");
  if (nrows) *nrows = m->nrows;
  if (ncols) *ncols = m->ncols;
  if (nnz) *nnz = m->nnz;
  if (format) *format = QP_FMT_CSR;
  if (stored_nnz) *stored_nnz = m->nnz;
  return QP_OK;
}

int qp_matrix_get_csr(const qp_matrix* m, int64_t* rowptr, int32_t* col, qp_c128* vals) {
  if (!m || !rowptr || !col || !vals) return qp::fail(QP_E_BAD_ARG, "qp_matrix_get_csr: NULL argument");
  std::memcpy(rowptr, m->rowptr.data(), (m->nrows + 1) * sizeof(int64_t));
  std::memcpy(col, m->col.data(), m->nnz * sizeof(int32_t));
  std::memcpy(vals, m->vals.data(), m->nnz * sizeof(qp_c128));
  return QP_OK;
}

// ---------------------------------------------------------------------------
// Operator: union pattern + value planes in HBM
// ---------------------------------------------------------------------------
static int operator_free_device(qp_operator* op) {
  for (auto p : op->planes) (void)hipFree(p);
  op->planes.clear();
  if (op->planes_dev) (void)hipFree(op->planes_dev);
  if (op->combined) (void)hipFree(op->combined);
  if (op->real_vals) (void)hipFree(op->real_vals);
  op->real_vals = nullptr;
  op->real_of = nullptr;
  op->A.vals_r = nullptr;
  if (op->A.bptr) (void)hipFree(op->A.bptr);
  if (op->A.rowptr) (void)hipFree(op->A.rowptr);
  if (op->A.cols) (void)hipFree(op->A.cols);
  if (op->A.cmeta) (void)hipFree(op->A.cmeta);
  if (op->A.lcmeta) (void)hipFree(op->A.lcmeta);
  op->A.cmeta = op->A.lcmeta = nullptr;
  if (op->A.lptr) (void)hipFree(op->A.lptr);
  if (op->A.lcols) (void)hipFree(op->A.lcols);
  if (op->A.lpos) (void)hipFree(op->A.lpos);
  if (op->m_rowptr) (void)hipFree(op->m_rowptr);
  if (op->m_cols) (void)hipFree(op->m_cols);
  if (op->m_map) (void)hipFree(op->m_map);
  if (op->m_vals) (void)hipFree(op->m_vals);
  op->m_rowptr = nullptr;
  op->m_cols = nullptr;
  op->m_map = nullptr;
  op->m_vals = nullptr;
  op->m_epoch = 0;
  op->planes_dev = nullptr;
  op->combined = nullptr;
  op->A.bptr = op->A.rowptr = op->A.lptr = nullptr;
  op->A.cols = op->A.lcols = op->A.lpos = nullptr;
  op->A.vals = nullptr;
  return QP_OK;
}

static int operator_free(qp_operator* op) {
  if (!op) return QP_OK;
  (void)hipSetDevice(op->ctx->device);
  (void)hipStreamSynchronize(op->ctx->stream);
  operator_free_device(op);
  delete op;
  return QP_OK;
}

// ---- host-side layout of the two row-block formats --------------------------------
// Within a 64-row block, entry k of row r sits at  base + 64 k + (r % 64); column
// indices (and the lower section's positions) are packed four k per lane.
static inline int64_t rb_val_pos(const std::vector<int64_t>& bptr, int64_t r, int64_t k) {
  return bptr[r / kRB] + k * kRB + (r % kRB);
}
static inline int64_t rb_quad_pos(const std::vector<int64_t>& bptr, int64_t r, int64_t k) {
  return bptr[r / kRB] + (k >> 2) * (4 * kRB) + (r % kRB) * 4 + (k & 3);
}

using HostLayout = HostLayoutData;

// is this canonical CSR exactly Hermitian (bitwise conj-symmetric values, symmetric
// pattern, real diagonal, strictly increasing columns)?
// Columns >= n (ghost columns of a row-partitioned operator in local numbering) are
// outside the square part and always carry their values.
static bool csr_is_hermitian(int64_t n, const std::vector<int64_t>& rp, const std::vector<int32_t>& col,
                             const std::vector<cplx>& vals) {
  int64_t nlower = 0, nupper = 0;
  for (int64_t r = 0; r < n; ++r) {
    for (int64_t p = rp[r]; p < rp[r + 1]; ++p) {
      const int64_t c = col[p];
      if (p > rp[r] && col[p - 1] >= c) return false;
      if (c == r) {
        if (vals[p].imag() != 0.0) return false;
      } else if (c >= n) {
        continue;
      } else if (c > r) {
        ++nupper;
      } else {
        ++nlower;
        const int32_t* b = col.data() + rp[c];
        const int32_t* e = col.data() + rp[c + 1];
        const int32_t* it = std::lower_bound(b, e, (int32_t)r);
        if (it == e || *it != r) return false;
        const cplx t = vals[it - col.data()];
        if (!(t.real() == vals[p].real() && t.imag() == -vals[p].imag())) return false;
      }
    }
  }
  return nlower == nupper;
}

// Encode the quad-packed column sections of all blocks: per block either int32 columns or,
// if every entry is within +-32767 of its row, int16 deltas to the row (2 bytes of index
// traffic per entry instead of 4).  `get(r, k, &is_pad)` returns the column of entry k of
// row r in this section (pad entries: any valid column).
extern "C++" {
// 32-byte record of one slot of a *stencil* lower section (see below)
struct LowerStencilSlot {
  int32_t delta, cb0;
  int64_t pb0, pb1, pad;
};
static_assert(sizeof(LowerStencilSlot) == 32, "layout shared with kernels.hip");

// mode of a block's column section (low two bits of its meta word, the rest is the byte offset)
enum { kColInt32 = 0, kColInt16 = 1, kColStencil = 2 };

// `special(b, w, out)`: a chance to emit a block in the stencil encoding (returns true and
// appends its bytes) before the per-entry encodings are tried.
template <class GetCol, class Special>
static void encode_col_sections(int64_t nrows, int64_t nblocks, const std::vector<int64_t>& ptr, GetCol get,
                                Special special, std::vector<char>& bytes, std::vector<int64_t>& meta) {
  meta.assign((size_t)nblocks, 0);
  bytes.clear();
  for (int64_t b = 0; b < nblocks; ++b) {
    const int64_t w = (ptr[b + 1] - ptr[b]) / kRB;
    while (bytes.size() % 32) bytes.push_back(0);
    const size_t start = bytes.size();
    if (w > 0 && special(b, w, bytes)) {
      meta[b] = ((int64_t)start << 2) | kColStencil;
      continue;
    }
    bool ok16 = true;
    for (int64_t l = 0; l < kRB && ok16; ++l) {
      const int64_t r = b * kRB + l;
      if (r >= nrows) break;
      for (int64_t k = 0; k < w; ++k) {
        bool pad = false;
        const int64_t c = get(r, k, &pad);
        if (!pad && (c - r > 32767 || r - c > 32767)) { ok16 = false; break; }
      }
    }
    meta[b] = ((int64_t)bytes.size() << 2) | (ok16 ? kColInt16 : kColInt32);
    const size_t esz = ok16 ? 2 : 4;
    const size_t off = bytes.size();
    bytes.resize(off + (size_t)w * kRB * esz, 0);
    for (int64_t l = 0; l < kRB; ++l) {
      const int64_t r = b * kRB + l;
      const int64_t rc = std::min(r, nrows - 1);   // the kernel decodes deltas against the clamped row
      for (int64_t k = 0; k < w; ++k) {
        bool pad = (r >= nrows);
        int64_t c = pad ? rc : get(r, k, &pad);
        if (pad && ok16) c = rc;
        const size_t q = (size_t)(k >> 2) * (4 * kRB) + (size_t)l * 4 + (k & 3);   // quad-packed slot
        if (ok16) {
          const int16_t d = (int16_t)(c - rc);
          std::memcpy(&bytes[off + q * 2], &d, 2);
        } else {
          const int32_t c32 = (int32_t)c;
          std::memcpy(&bytes[off + q * 4], &c32, 4);
        }
      }
    }
  }
  while (bytes.size() % 32) bytes.push_back(0);
}

// Stencil blocks: every row of the 64-row block has its k-th entry at the same distance
// delta_k from the diagonal (grids, lattices, tensor-product operators: most blocks of a
// banded H).  The section then stores w int32 deltas for the whole block instead of w x 64
// per-lane indices: the index stream disappears from HBM traffic (wave-uniform loads).
// Pad entries (value 0) take the block's delta too, so row + delta must stay a valid column.
template <class GetCol>
static bool try_stencil_upper(int64_t nrows, int64_t ncols, int64_t b, int64_t w, GetCol get, std::vector<char>& out) {
  std::vector<int32_t> delta((size_t)w, 0);
  for (int64_t k = 0; k < w; ++k) {
    bool have = false;
    int64_t d = 0;
    for (int64_t l = 0; l < kRB; ++l) {
      const int64_t r = b * kRB + l;
      if (r >= nrows) break;
      bool pad = false;
      const int64_t c = get(r, k, &pad);
      if (pad) continue;
      if (!have) {
        d = c - r;
        have = true;
      } else if (c - r != d) {
        return false;
      }
    }
    if (!have) d = 0;   // a slot of pure padding (width rounded up to a quad): column = row
    if (d > INT32_MAX || d < INT32_MIN) return false;
    // every lane (pad entries and the clamped rows of a partial last block included) must
    // land on a valid column
    const int64_t r_lo = b * kRB, r_hi = std::min(b * kRB + kRB - 1, nrows - 1);
    if (r_lo + d < 0 || r_hi + d >= ncols) return false;
    delta[(size_t)k] = (int32_t)d;
  }
  const size_t off = out.size();
  out.resize(off + (size_t)w * 4);
  std::memcpy(&out[off], delta.data(), (size_t)w * 4);
  return true;
}
}  // extern "C++"

static int64_t decode_col(const std::vector<char>& bytes, const std::vector<int64_t>& meta, int64_t nrows, int64_t r,
                          int64_t k, bool lower = false) {
  const int64_t m = meta[r / kRB];
  const size_t off = (size_t)(m >> 2);
  const int mode = (int)(m & 3);
  if (mode == kColStencil) {
    int32_t d;
    std::memcpy(&d, &bytes[off + (size_t)k * (lower ? sizeof(LowerStencilSlot) : 4)], 4);
    return std::min(r, nrows - 1) + d;
  }
  const size_t q = (size_t)(k >> 2) * (4 * kRB) + (size_t)(r % kRB) * 4 + (k & 3);
  if (mode == kColInt16) {
    int16_t d;
    std::memcpy(&d, &bytes[off + q * 2], 2);
    return std::min(r, nrows - 1) + d;
  }
  int32_t c;
  std::memcpy(&c, &bytes[off + q * 4], 4);
  return c;
}

// position in the upper value array of the conj-transposed value of lower entry k of row r,
// for a block whose lower section is in the stencil encoding
static int64_t decode_lower_stencil_pos(const std::vector<char>& bytes, const std::vector<int64_t>& meta, int64_t nrows,
                                        int64_t r, int64_t k) {
  const int64_t m = meta[r / kRB];
  LowerStencilSlot e;
  std::memcpy(&e, &bytes[(size_t)(m >> 2) + (size_t)k * sizeof(LowerStencilSlot)], sizeof(e));
  const int64_t c = std::min(r, nrows - 1) + e.delta;
  return ((c >> 6) == e.cb0 ? e.pb0 : e.pb1) + (c & 63);
}

// Build every device array of `op` for `format` from the union pattern (op->u_rowptr /
// u_col) and the per-term values given in union-CSR order.
static int operator_build_device(qp_operator* op, int format, const std::vector<std::vector<cplx>>& planes_csr) {
  qp_ctx* ctx = op->ctx;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  DevMatrix& A = op->A;
  const int64_t nrows = A.nrows;
  const int64_t nnz = ur[nrows];
  const int nops = (int)planes_csr.size();
  A.nblocks = (nrows + kRB - 1) / kRB;
  A.format = format;
  HostLayout& Lh = op->layout;
  Lh = HostLayout();
  Lh.format = format;

  if (format == QP_FMT_CSR) {
    A.stored = nnz;
    QP_CHECK(dev_alloc(&A.rowptr, ur.size()));
    QP_HIP(hipMemcpy(A.rowptr, ur.data(), ur.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    QP_CHECK(dev_alloc(&A.cols, (size_t)nnz));
    QP_HIP(hipMemcpy(A.cols, uc.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    double mean = nrows > 0 ? (double)nnz / (double)nrows : 1.0;
    int T = 2;
    while (T < 64 && T < mean) T *= 2;
    A.lanes_per_row = T;
  } else {
    const bool hrb = (format == QP_FMT_HRB);
    Lh.bptr.assign(A.nblocks + 1, 0);
    if (hrb) {
      Lh.lptr.assign(A.nblocks + 1, 0);
      Lh.nlow.assign(nrows, 0);
      for (int64_t r = 0; r < nrows; ++r) {
        const int32_t* b = uc.data() + ur[r];
        const int32_t* e = uc.data() + ur[r + 1];
        Lh.nlow[r] = (int32_t)(std::lower_bound(b, e, (int32_t)r) - b);
      }
    }
    for (int64_t b = 0; b < A.nblocks; ++b) {
      int64_t wu = 0, wl = 0;
      for (int64_t r = b * kRB; r < std::min(nrows, (b + 1) * kRB); ++r) {
        const int64_t len = ur[r + 1] - ur[r];
        const int64_t nl = hrb ? Lh.nlow[r] : 0;
        wu = std::max(wu, len - nl);
        wl = std::max(wl, nl);
      }
      wu = (wu + 3) & ~(int64_t)3;
      wl = (wl + 3) & ~(int64_t)3;
      Lh.bptr[b + 1] = Lh.bptr[b] + wu * kRB;
      if (hrb) Lh.lptr[b + 1] = Lh.lptr[b] + wl * kRB;
    }
    Lh.stored = Lh.bptr[A.nblocks] + kRB;   // + one block of slack: padded lower entries read vals[0..63]
    Lh.lstored = hrb ? Lh.lptr[A.nblocks] : 0;
    A.stored = Lh.stored;
    A.lstored = Lh.lstored;
    // upper (or full) column indices
    {
      std::vector<char> cbytes;
      auto get_upper = [&](int64_t r, int64_t k, bool* pad) -> int64_t {
        const int64_t nl = hrb ? Lh.nlow[r] : 0;
        const int64_t len = ur[r + 1] - ur[r] - nl;
        if (k < len) return uc[ur[r] + nl + k];
        *pad = true;
        return (ur[r + 1] > ur[r]) ? uc[ur[r]] : 0;
      };
      encode_col_sections(nrows, A.nblocks, Lh.bptr, get_upper,
                          [&](int64_t b, int64_t w, std::vector<char>& out) {
                            return qp::g_stencil != 0 && try_stencil_upper(nrows, A.ncols, b, w, get_upper, out);
                          },
                          cbytes, Lh.cmeta);
      A.colbytes = (int64_t)cbytes.size();
      QP_CHECK(dev_alloc(reinterpret_cast<char**>(&A.cols), cbytes.size()));
      QP_HIP(hipMemcpy(A.cols, cbytes.data(), cbytes.size(), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(&A.cmeta, Lh.cmeta.size()));
      QP_HIP(hipMemcpy(A.cmeta, Lh.cmeta.data(), Lh.cmeta.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    QP_CHECK(dev_alloc(&A.bptr, Lh.bptr.size()));
    QP_HIP(hipMemcpy(A.bptr, Lh.bptr.data(), Lh.bptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    if (hrb) {
      // lower section: (column, position of the conj-transposed value in the upper section)
      std::vector<int32_t> lpos((size_t)std::max<int64_t>(A.lstored, 1), -1);
      if (Lh.stored >= (int64_t)INT32_MAX) return qp::fail(QP_E_BAD_ARG, "Hermitian-packed format needs < 2^31 stored values per GPU");
      for (int64_t r = 0; r < nrows; ++r) {
        const int64_t nl = Lh.nlow[r];
        for (int64_t k = 0; k < nl; ++k) {
          const int64_t c = uc[ur[r] + k];
          const int32_t* b = uc.data() + ur[c];
          const int32_t* e = uc.data() + ur[c + 1];
          const int64_t kk = (std::lower_bound(b, e, (int32_t)r) - b) - Lh.nlow[c];  // index of (c,r) among row c's upper entries
          lpos[rb_quad_pos(Lh.lptr, r, k)] = (int32_t)rb_val_pos(Lh.bptr, c, kk);
        }
      }
      std::vector<char> lbytes;
      // stencil lower block: every row has a real entry in every slot, at a block-wide
      // distance delta_k, and the conj-transposed values sit at one slot per column block
      // (at most two column blocks per slot): position = pb(column block) + column % 64
      auto try_stencil_lower = [&](int64_t b, int64_t w, std::vector<char>& out) -> bool {
        if (qp::g_stencil == 0) return false;
        std::vector<LowerStencilSlot> slots((size_t)w);
        for (int64_t k = 0; k < w; ++k) {
          LowerStencilSlot e{0, 0, -1, -1, 0};
          bool have = false;
          for (int64_t l = 0; l < kRB; ++l) {
            const int64_t r = b * kRB + l;
            if (r >= nrows) break;
            if (k >= Lh.nlow[r]) return false;
            const int64_t c = uc[ur[r] + k];
            const int64_t base = (int64_t)lpos[rb_quad_pos(Lh.lptr, r, k)] - (c & 63);
            if (!have) {
              e.delta = (int32_t)(c - r);
              e.cb0 = (int32_t)(c >> 6);
              e.pb0 = base;
              have = true;
            } else if (c - r != e.delta) {
              return false;
            }
            if ((c >> 6) == e.cb0) {
              if (base != e.pb0) return false;
            } else if ((c >> 6) == e.cb0 + 1) {
              if (e.pb1 < 0) e.pb1 = base;
              else if (base != e.pb1) return false;
            } else {
              return false;
            }
          }
          if (!have) return false;
          if (e.pb1 < 0) e.pb1 = e.pb0;
          slots[(size_t)k] = e;
        }
        const size_t off = out.size();
        out.resize(off + (size_t)w * sizeof(LowerStencilSlot));
        std::memcpy(&out[off], slots.data(), (size_t)w * sizeof(LowerStencilSlot));
        return true;
      };
      encode_col_sections(nrows, A.nblocks, Lh.lptr,
                          [&](int64_t r, int64_t k, bool* pad) -> int64_t {
                            if (k < Lh.nlow[r]) return uc[ur[r] + k];
                            *pad = true;          // padded: any valid column, value masked by pos < 0
                            return r;
                          },
                          try_stencil_lower, lbytes, Lh.lcmeta);
      A.lcolbytes = (int64_t)lbytes.size();
      QP_CHECK(dev_alloc(&A.lptr, Lh.lptr.size()));
      QP_HIP(hipMemcpy(A.lptr, Lh.lptr.data(), Lh.lptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(reinterpret_cast<char**>(&A.lcols), std::max<size_t>(lbytes.size(), 16)));
      if (!lbytes.empty()) QP_HIP(hipMemcpy(A.lcols, lbytes.data(), lbytes.size(), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(&A.lcmeta, Lh.lcmeta.size()));
      QP_HIP(hipMemcpy(A.lcmeta, Lh.lcmeta.data(), Lh.lcmeta.size() * sizeof(int64_t), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(&A.lpos, lpos.size()));
      QP_HIP(hipMemcpy(A.lpos, lpos.data(), lpos.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  }

  // ---- value planes ----
  op->planes_real = true;
  for (const auto& pv : planes_csr)
    for (const cplx& v : pv)
      if (v.imag() != 0.0) {
        op->planes_real = false;
        break;
      }
  std::vector<cplx> hplane((size_t)std::max<int64_t>(A.stored, 1));
  for (int l = 0; l < nops; ++l) {
    std::fill(hplane.begin(), hplane.end(), cplx(0.0));
    const auto& pv = planes_csr[l];
    for (int64_t r = 0; r < nrows; ++r) {
      const int64_t nl = (format == QP_FMT_HRB) ? op->layout.nlow[r] : 0;
      for (int64_t k = nl; k < ur[r + 1] - ur[r]; ++k) {
        const int64_t pos = (format == QP_FMT_CSR) ? ur[r] + k : rb_val_pos(op->layout.bptr, r, k - nl);
        hplane[pos] = pv[ur[r] + k];
      }
    }
    double2* dp = nullptr;
    QP_CHECK(dev_alloc(&dp, (size_t)A.stored));
    op->planes.push_back(dp);
    QP_HIP(hipMemcpy(dp, hplane.data(), (size_t)A.stored * sizeof(double2), hipMemcpyHostToDevice));
  }
  QP_CHECK(dev_alloc(&op->planes_dev, (size_t)nops));
  QP_HIP(hipMemcpy(op->planes_dev, op->planes.data(), nops * sizeof(double2*), hipMemcpyHostToDevice));
  A.vals = op->planes[0];
  (void)ctx;
  return QP_OK;
}

// current per-term values (device planes) back in union-CSR order
static int operator_download_planes(qp_operator* op, std::vector<std::vector<cplx>>& planes_csr) {
  const DevMatrix& A = op->A;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  const int64_t nnz = ur[A.nrows];
  QP_HIP(hipStreamSynchronize(op->ctx->stream));
  std::vector<cplx> hv((size_t)std::max<int64_t>(A.stored, 1));
  planes_csr.assign(op->planes.size(), std::vector<cplx>());
  for (size_t l = 0; l < op->planes.size(); ++l) {
    QP_HIP(hipMemcpy(hv.data(), op->planes[l], (size_t)A.stored * sizeof(double2), hipMemcpyDeviceToHost));
    auto& out = planes_csr[l];
    out.assign((size_t)nnz, cplx(0));
    for (int64_t r = 0; r < A.nrows; ++r) {
      const int64_t nl = (A.format == QP_FMT_HRB) ? op->layout.nlow[r] : 0;
      for (int64_t k = 0; k < ur[r + 1] - ur[r]; ++k) {
        if (A.format == QP_FMT_CSR) {
          out[ur[r] + k] = hv[ur[r] + k];
        } else if (k >= nl) {
          out[ur[r] + k] = hv[rb_val_pos(op->layout.bptr, r, k - nl)];
        } else {  // lower entry of a Hermitian-packed operator: conj of its transpose
          const int64_t c = uc[ur[r] + k];
          const int32_t* b = uc.data() + ur[c];
          const int32_t* e = uc.data() + ur[c + 1];
          const int64_t kk = (std::lower_bound(b, e, (int32_t)r) - b) - op->layout.nlow[c];
          out[ur[r] + k] = std::conj(hv[rb_val_pos(op->layout.bptr, c, kk)]);
        }
      }
    }
  }
  return QP_OK;
}

static int choose_format(qp_operator* op, int requested, bool hermitian) {
  const auto& ur = op->u_rowptr;
  const int64_t nrows = op->A.nrows, nnz = ur[nrows];
  const int64_t nblocks = (nrows + kRB - 1) / kRB;
  int64_t rb_stored = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    int64_t w = 0;
    for (int64_t r = b * kRB; r < std::min(nrows, (b + 1) * kRB); ++r) w = std::max(w, ur[r + 1] - ur[r]);
    rb_stored += ((w + 3) & ~(int64_t)3) * kRB;
  }
  // the row-block kernels stream ~1.6x faster than the sub-wave CSR kernel at equal bytes
  // (profiles/r01/kbench_*): accept up to 50 % padding before falling back
  const bool rb_ok = (double)rb_stored <= 1.5 * (double)nnz + 1024.0;
  if (requested == QP_FMT_AUTO) {
    if (!rb_ok) return QP_FMT_CSR;
    // few, long rows (small dense generators: the reference's test and benchmark sizes): a
    // row block gives one wavefront 64 rows to walk entry by entry -- too few wavefronts to
    // hide the latency.  One wavefront per row instead (CSR kernel, 64 lanes per row).
    if (nblocks < 2048 && nnz >= 32 * nrows) return QP_FMT_CSR;
    if (!hermitian) return QP_FMT_RBCSR;
    // Hermitian packing pays only if the transposed values are still in the XCD's L2
    // (4 MiB) when the lower entry is processed: the rows stream in order, so require
    // (row - col) * bytes-per-row <= 2 MiB for at least 85 % of the lower entries.
    // Measured (profiles/r01/kbench): banded 53.6 vs 72.1 us per term, scattered 103.8 vs 93.4.
    const auto& uc = op->u_col;
    const double row_bytes = 14.0 * (double)nnz / (double)std::max<int64_t>(nrows, 1) + 80.0;
    const int64_t maxdist = (int64_t)(2.0 * 1048576.0 / row_bytes);
    int64_t nlow = 0, nnear = 0;
    for (int64_t r = 0; r < nrows; ++r)
      for (int64_t p = ur[r]; p < ur[r + 1] && uc[p] < r; ++p) {
        ++nlow;
        if (r - uc[p] <= maxdist) ++nnear;
      }
    return ((double)nnear >= 0.85 * (double)nlow) ? QP_FMT_HRB : QP_FMT_RBCSR;
  }
  if (requested == QP_FMT_HRB && !hermitian) return -1;
  return requested;
}

int qp_operator_create(qp_ctx* ctx, qp_matrix* const* ops, int nops, int ncoeffs, int format,
                       qp_operator** out) {
  QP_TRY
  if (!ctx || !ops || !out || nops < 1 || ncoeffs < 0 || ncoeffs > nops)
    return qp::fail(QP_E_BAD_ARG, "qp_operator_create: bad arguments");
  for (int l = 0; l < nops; ++l) {
    if (!ops[l]) return qp::fail(QP_E_BAD_ARG, "ops[%d] is NULL", l);
    if (ops[l]->nrows != ops[0]->nrows || ops[l]->ncols != ops[0]->ncols)
      return qp::fail(QP_E_BAD_ARG, "ops[%d] shape differs from ops[0]", l);
  }
  if (format < QP_FMT_AUTO || format > QP_FMT_HRB) return qp::fail(QP_E_BAD_ARG, "bad device format %d", format);
  QP_CHECK(use(ctx));
  std::unique_ptr<qp_operator, int (*)(qp_operator*)> op(new qp_operator(), operator_free);
  op->ctx = ctx;
  op->nops = nops;
  op->ncoeffs = ncoeffs;
  op->coeffs.assign(ncoeffs, cplx(1.0));
  const int64_t nrows = ops[0]->nrows, ncols = ops[0]->ncols;
  op->A.nrows = nrows;
  op->A.ncols = ncols;

  // ---- union sparsity pattern (sorted merge per row) ----
  auto& ur = op->u_rowptr;
  auto& uc = op->u_col;
  ur.assign(nrows + 1, 0);
  if (nops == 1) {
    ur = ops[0]->rowptr;
    uc = ops[0]->col;
  } else {
    std::vector<int32_t> merged;
    for (int64_t r = 0; r < nrows; ++r) {
      merged.clear();
      for (int l = 0; l < nops; ++l)
        merged.insert(merged.end(), ops[l]->col.begin() + ops[l]->rowptr[r], ops[l]->col.begin() + ops[l]->rowptr[r + 1]);
      std::sort(merged.begin(), merged.end());
      merged.erase(std::unique(merged.begin(), merged.end()), merged.end());
      uc.insert(uc.end(), merged.begin(), merged.end());
      ur[r + 1] = (int64_t)uc.size();
    }
  }
  op->A.nnz = ur[nrows];

  // ---- per-term values in union order (duplicates within a row are summed, as Julia's sparse() does) ----
  std::vector<std::vector<cplx>> planes_csr(nops);
  for (int l = 0; l < nops; ++l) {
    const qp_matrix* M = ops[l];
    auto& pv = planes_csr[l];
    pv.assign((size_t)op->A.nnz, cplx(0));
    for (int64_t r = 0; r < nrows; ++r) {
      int64_t k = 0;
      for (int64_t p = M->rowptr[r]; p < M->rowptr[r + 1]; ++p) {
        while (uc[ur[r] + k] != M->col[p]) ++k;
        pv[ur[r] + k] += M->vals[p];
      }
    }
  }
  bool hermitian = (ncols >= nrows) && (format == QP_FMT_AUTO || format == QP_FMT_HRB);
  for (int l = 0; hermitian && l < nops; ++l) hermitian = csr_is_hermitian(nrows, ur, uc, planes_csr[l]);
  op->hermitian_planes = hermitian;
  const int fmt = choose_format(op.get(), format, hermitian);
  if (fmt < 0) return qp::fail(QP_E_BAD_ARG, "QP_FMT_HRB requested but the operator terms are not exactly Hermitian");
  QP_CHECK(operator_build_device(op.get(), fmt, planes_csr));
  planes_csr.clear();
  qp_operator* raw = op.release();
  std::vector<qp_c128> ones(ncoeffs, qp_c128{1.0, 0.0});
  int rc = qp_operator_set_coeffs(raw, ones.data(), ncoeffs);
  if (rc != QP_OK) {
    operator_free(raw);
    return rc;
  }
  *out = raw;
  return QP_OK;
  QP_CATCH
}

static int operator_refresh(qp_operator* op) {
  qp_ctx* ctx = op->ctx;
  op->vals_epoch++;
  const int drift = op->nops - op->ncoeffs;  // src/generators.jl:635
  std::vector<double2> eff(op->nops);
  bool all_one = true, all_real = true;
  for (int l = 0; l < op->nops; ++l) {
    cplx c = op->scale;
    if (l >= drift) c *= op->coeffs[l - drift];
    eff[l] = d2(c);
    if (!(c == cplx(1.0))) all_one = false;
    if (c.imag() != 0.0) all_real = false;
  }
  if (op->A.format == QP_FMT_HRB && !all_real) {
    // a complex combination of Hermitian terms is not Hermitian: leave the packed format
    // (slow path, once): re-lay the planes out as full row-block CSR
    std::vector<std::vector<cplx>> planes_csr;
    QP_CHECK(operator_download_planes(op, planes_csr));
    operator_free_device(op);
    const int fmt = choose_format(op, QP_FMT_AUTO, false);
    QP_CHECK(operator_build_device(op, fmt, planes_csr));
  }
  if (op->nops == 1 && all_one) {
    op->A.vals = op->planes[0];
  } else {
    if (!op->combined) QP_CHECK(dev_alloc(&op->combined, (size_t)op->A.stored));
    QP_CHECK(qp::launch_combine_planes(ctx->stream, op->combined, op->planes_dev, eff.data(), op->nops, op->A.stored,
                                       &ctx->stats));
    op->A.vals = op->combined;
    op->real_of = nullptr;   // rewritten
  }
  // real terms with real coefficients: the mat-vec kernels stream a real copy (8 instead of 16
  // bytes per value); everything else keeps reading the complex array
  op->A.vals_r = nullptr;
  if (qp::g_real_vals && op->planes_real && all_real && op->A.stored > 0) {
    if (!op->real_vals) QP_CHECK(dev_alloc(&op->real_vals, (size_t)op->A.stored));
    if (op->real_of != op->A.vals) {
      QP_CHECK(qp::launch_real_part(ctx->stream, op->real_vals, op->A.vals, op->A.stored, &ctx->stats));
      op->real_of = (op->A.vals == op->combined) ? nullptr : op->A.vals;   // a plane never changes
    }
    op->A.vals_r = op->real_vals;
  }
  return QP_OK;
}

int qp_operator_set_coeffs(qp_operator* op, const qp_c128* coeffs, int ncoeffs) {
  QP_TRY
  if (!op || (ncoeffs > 0 && !coeffs)) return qp::fail(QP_E_BAD_ARG, "qp_operator_set_coeffs: NULL argument");
  if (ncoeffs != op->ncoeffs) return qp::fail(QP_E_BAD_ARG, "expected %d coefficients, got %d", op->ncoeffs, ncoeffs);
  QP_CHECK(use(op->ctx));
  for (int i = 0; i < ncoeffs; ++i) op->coeffs[i] = cx(coeffs[i]);
  return operator_refresh(op);
  QP_CATCH
}

int qp_operator_set_scale(qp_operator* op, qp_c128 scale) {
  QP_TRY
  if (!op) return qp::fail(QP_E_BAD_ARG, "operator is NULL");
  QP_CHECK(use(op->ctx));
  op->scale = cx(scale);
  return operator_refresh(op);
  QP_CATCH
}

int qp_operator_destroy(qp_operator* op) {
  QP_TRY
  return operator_free(op);
  QP_CATCH
}

int qp_operator_info(const qp_operator* op, int64_t* nrows, int64_t* ncols, int64_t* nnz, int* format) {
  if (!op) return qp::fail(QP_E_BAD_ARG, "operator is NULL");
  if (nrows) *nrows = op->A.nrows;
  if (ncols) *ncols = op->A.ncols;
  if (nnz) *nnz = op->A.nnz;
  if (format) *format = op->A.format;
  return QP_OK;
}

int qp_operator_layout_info(const qp_operator* op, int64_t out[5]) {
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_layout_info: NULL argument");
  for (int i = 0; i < 5; ++i) out[i] = 0;
  const DevMatrix& A = op->A;
  out[4] = A.stored;
  if (A.format == QP_FMT_CSR) return QP_OK;
  const HostLayout& Lh = op->layout;
  out[0] = A.nblocks;
  int64_t idx_bytes = A.colbytes + A.lcolbytes;
  for (int64_t b = 0; b < A.nblocks; ++b) {
    if ((Lh.cmeta[b] & 3) == kColStencil) out[1]++;
    if (A.format == QP_FMT_HRB) {
      if ((Lh.lcmeta[b] & 3) == kColStencil) out[2]++;
      else idx_bytes += (Lh.lptr[b + 1] - Lh.lptr[b]) * (int64_t)sizeof(int32_t);
    }
  }
  out[3] = idx_bytes;
  return QP_OK;
}

// download the *device* copy (current combined values and indices) back as canonical CSR
int qp_operator_get_csr(qp_operator* op, int64_t* rowptr, int32_t* col, qp_c128* vals) {
  QP_TRY
  if (!op || !rowptr || !col || !vals) return qp::fail(QP_E_BAD_ARG, "qp_operator_get_csr: NULL argument");
  QP_CHECK(use(op->ctx));
  const DevMatrix& A = op->A;
  const auto& ur = op->u_rowptr;
  QP_HIP(hipStreamSynchronize(op->ctx->stream));
  std::vector<cplx> hv((size_t)std::max<int64_t>(A.stored, 1));
  QP_HIP(hipMemcpy(hv.data(), A.vals, (size_t)A.stored * sizeof(double2), hipMemcpyDeviceToHost));
  if (A.format == QP_FMT_CSR) {
    std::vector<int64_t> rp(A.nrows + 1);
    std::vector<int32_t> hc((size_t)std::max<int64_t>(A.nnz, 1));
    QP_HIP(hipMemcpy(rp.data(), A.rowptr, rp.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    QP_HIP(hipMemcpy(hc.data(), A.cols, (size_t)A.nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::memcpy(rowptr, rp.data(), rp.size() * sizeof(int64_t));
    std::memcpy(col, hc.data(), (size_t)A.nnz * sizeof(int32_t));
    std::memcpy(vals, hv.data(), (size_t)A.nnz * sizeof(qp_c128));
    return QP_OK;
  }
  std::vector<int64_t> bptr(A.nblocks + 1), cmeta((size_t)A.nblocks), lptr, lcmeta;
  std::vector<char> cbytes((size_t)std::max<int64_t>(A.colbytes, 1)), lbytes;
  QP_HIP(hipMemcpy(bptr.data(), A.bptr, bptr.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
  QP_HIP(hipMemcpy(cmeta.data(), A.cmeta, cmeta.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
  QP_HIP(hipMemcpy(cbytes.data(), A.cols, (size_t)A.colbytes, hipMemcpyDeviceToHost));
  std::vector<int32_t> lp;
  if (A.format == QP_FMT_HRB) {
    lptr.resize(A.nblocks + 1);
    lcmeta.resize((size_t)A.nblocks);
    lbytes.resize((size_t)std::max<int64_t>(A.lcolbytes, 1));
    lp.resize((size_t)std::max<int64_t>(A.lstored, 1));
    QP_HIP(hipMemcpy(lptr.data(), A.lptr, lptr.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    QP_HIP(hipMemcpy(lcmeta.data(), A.lcmeta, lcmeta.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (A.lcolbytes > 0) QP_HIP(hipMemcpy(lbytes.data(), A.lcols, (size_t)A.lcolbytes, hipMemcpyDeviceToHost));
    QP_HIP(hipMemcpy(lp.data(), A.lpos, lp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  for (int64_t r = 0; r <= A.nrows; ++r) rowptr[r] = ur[r];
  for (int64_t r = 0; r < A.nrows; ++r) {
    const int64_t nl = (A.format == QP_FMT_HRB) ? op->layout.nlow[r] : 0;
    for (int64_t k = 0; k < ur[r + 1] - ur[r]; ++k) {
      cplx v;
      int64_t c;
      if (k >= nl) {
        c = decode_col(cbytes, cmeta, A.nrows, r, k - nl);
        v = hv[rb_val_pos(bptr, r, k - nl)];
      } else {
        c = decode_col(lbytes, lcmeta, A.nrows, r, k, true);
        const bool stencil = (lcmeta[r / kRB] & 3) == kColStencil;
        v = std::conj(hv[stencil ? decode_lower_stencil_pos(lbytes, lcmeta, A.nrows, r, k) : lp[rb_quad_pos(lptr, r, k)]]);
      }
      col[ur[r] + k] = (int32_t)c;
      vals[ur[r] + k] = qp_c128{v.real(), v.imag()};
    }
  }
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// states and BLAS-1
// ---------------------------------------------------------------------------
int qp_state_create(qp_ctx* ctx, int64_t n, qp_state** out) {
  QP_TRY
  if (!ctx || !out || n < 0) return qp::fail(QP_E_BAD_ARG, "qp_state_create: bad arguments");
  QP_CHECK(use(ctx));
  auto s = std::make_unique<qp_state>();
  s->ctx = ctx;
  s->n = n;
  s->own = true;
  QP_CHECK(dev_alloc(&s->d, (size_t)n));
  QP_HIP(hipMemsetAsync(s->d, 0, (size_t)n * sizeof(double2), ctx->stream));
  *out = s.release();
  return QP_OK;
  QP_CATCH
}

int qp_state_wrap(qp_ctx* ctx, void* device_ptr, int64_t n, qp_state** out) {
  QP_TRY
  if (!ctx || !out || !device_ptr || n < 0) return qp::fail(QP_E_BAD_ARG, "qp_state_wrap: bad arguments");
  if ((uintptr_t)device_ptr % 16 != 0) return qp::fail(QP_E_BAD_ARG, "device pointer must be 16-byte aligned");
  auto s = std::make_unique<qp_state>();
  s->ctx = ctx;
  s->d = static_cast<double2*>(device_ptr);
  s->n = n;
  s->own = false;
  *out = s.release();
  return QP_OK;
  QP_CATCH
}

int qp_state_destroy(qp_state* s) {
  QP_TRY
  if (!s) return QP_OK;
  if (s->own) {
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    (void)hipFree(s->d);
  }
  delete s;
  return QP_OK;
  QP_CATCH
}

int qp_state_upload(qp_state* s, const qp_c128* host) {
  QP_TRY
  if (!s || !host) return qp::fail(QP_E_BAD_ARG, "qp_state_upload: NULL argument");
  QP_CHECK(use(s->ctx));
  QP_HIP(hipMemcpyAsync(s->d, host, (size_t)s->n * sizeof(double2), hipMemcpyHostToDevice, s->ctx->stream));
  QP_HIP(hipStreamSynchronize(s->ctx->stream));
  return QP_OK;
  QP_CATCH
}

int qp_state_download(const qp_state* s, qp_c128* host) {
  QP_TRY
  if (!s || !host) return qp::fail(QP_E_BAD_ARG, "qp_state_download: NULL argument");
  QP_CHECK(use(s->ctx));
  QP_HIP(hipMemcpyAsync(host, s->d, (size_t)s->n * sizeof(double2), hipMemcpyDeviceToHost, s->ctx->stream));
  QP_HIP(hipStreamSynchronize(s->ctx->stream));
  return QP_OK;
  QP_CATCH
}

void* qp_state_ptr(const qp_state* s) { return s ? s->d : nullptr; }
int64_t qp_state_len(const qp_state* s) { return s ? s->n : -1; }

int qp_copy(qp_state* dst, const qp_state* src) {
  QP_TRY
  if (!dst || !src || dst->n != src->n) return qp::fail(QP_E_BAD_ARG, "qp_copy: length mismatch");
  QP_CHECK(use(dst->ctx));
  if (dst->d != src->d)
    QP_HIP(hipMemcpyAsync(dst->d, src->d, (size_t)dst->n * sizeof(double2), hipMemcpyDeviceToDevice, dst->ctx->stream));
  return QP_OK;
  QP_CATCH
}

int qp_scal(qp_state* x, qp_c128 alpha) {
  QP_TRY
  if (!x) return qp::fail(QP_E_BAD_ARG, "state is NULL");
  QP_CHECK(use(x->ctx));
  return qp::launch_scal(x->ctx->stream, x->d, d2(alpha), x->n, &x->ctx->stats);
  QP_CATCH
}

int qp_axpy(qp_c128 alpha, const qp_state* x, qp_state* y) {
  QP_TRY
  if (!x || !y || x->n != y->n) return qp::fail(QP_E_BAD_ARG, "qp_axpy: length mismatch");
  QP_CHECK(use(y->ctx));
  return qp::launch_axpy(y->ctx->stream, d2(alpha), x->d, y->d, y->n, &y->ctx->stats);
  QP_CATCH
}

int qp_fill(qp_state* x, qp_c128 alpha) {
  QP_TRY
  if (!x) return qp::fail(QP_E_BAD_ARG, "state is NULL");
  QP_CHECK(use(x->ctx));
  return qp::launch_fill(x->ctx->stream, x->d, d2(alpha), x->n, &x->ctx->stats);
  QP_CATCH
}

int qp_dot(const qp_state* x, const qp_state* y, qp_c128* out) {
  QP_TRY
  if (!x || !y || !out || x->n != y->n) return qp::fail(QP_E_BAD_ARG, "qp_dot: bad arguments");
  QP_CHECK(use(x->ctx));
  cplx r;
  QP_CHECK(dot_sync(x->ctx, x->d, y->d, x->n, &r));
  *out = qp_c128{r.real(), r.imag()};
  return QP_OK;
  QP_CATCH
}

int qp_norm(const qp_state* x, double* out) {
  QP_TRY
  if (!x || !out) return qp::fail(QP_E_BAD_ARG, "qp_norm: bad arguments");
  QP_CHECK(use(x->ctx));
  cplx r;
  QP_CHECK(dot_sync(x->ctx, x->d, x->d, x->n, &r));
  *out = std::sqrt(r.real());
  return QP_OK;
  QP_CATCH
}

int qp_mul(qp_operator* op, const qp_state* x, qp_state* y, qp_c128 alpha, qp_c128 beta) {
  QP_TRY
  if (!op || !x || !y) return qp::fail(QP_E_BAD_ARG, "qp_mul: NULL argument");
  if (x->n != op->A.ncols || y->n != op->A.nrows)
    return qp::fail(QP_E_BAD_ARG, "qp_mul: shape mismatch (op %lld x %lld, x %lld, y %lld)", (long long)op->A.nrows,
                    (long long)op->A.ncols, (long long)x->n, (long long)y->n);
  if (x->d == y->d) return qp::fail(QP_E_BAD_ARG, "qp_mul: x and y must not alias");
  QP_CHECK(use(op->ctx));
  qp::PlainEpi e;
  e.y = y->d;
  e.alpha = d2(alpha);
  e.beta = d2(beta);
  e.beta_zero = (beta.re == 0.0 && beta.im == 0.0);
  return qp::launch_spmv_plain(op->ctx->stream, op->A, x->d, e, &op->ctx->stats);
  QP_CATCH
}

int qp_dot_op(const qp_state* x, qp_operator* op, const qp_state* y, qp_state* tmp, qp_c128* out) {
  QP_TRY
  if (!x || !op || !y || !tmp || !out) return qp::fail(QP_E_BAD_ARG, "qp_dot_op: NULL argument");
  QP_CHECK(qp_mul(op, y, tmp, qp_c128{1, 0}, qp_c128{0, 0}));
  return qp_dot(x, tmp, out);
  QP_CATCH
}

// ---------------------------------------------------------------------------
// Chebyshev
// ---------------------------------------------------------------------------
int qp_cheby_create(qp_ctx* ctx, int64_t n, qp_cheby** out) {
  QP_TRY
  if (!ctx || !out || n < 0) return qp::fail(QP_E_BAD_ARG, "qp_cheby_create: bad arguments");
  QP_CHECK(use(ctx));
  auto w = std::make_unique<qp_cheby>();
  w->ctx = ctx;
  w->n = n;
  QP_CHECK(dev_alloc(&w->bufA, (size_t)n));
  QP_CHECK(dev_alloc(&w->acc, (size_t)n));
  *out = w.release();
  return QP_OK;
  QP_CATCH
}

int qp_cheby_destroy(qp_cheby* w) {
  QP_TRY
  if (!w) return QP_OK;
  (void)hipSetDevice(w->ctx->device);
  (void)hipStreamSynchronize(w->ctx->stream);
  if (w->bufA) (void)hipFree(w->bufA);
  if (w->acc) (void)hipFree(w->acc);
  if (w->chk_part) (void)hipFree(w->chk_part);
  if (w->chk_out) (void)hipFree(w->chk_out);
  if (w->gexec) (void)hipGraphExecDestroy(w->gexec);
  delete w;
  return QP_OK;
  QP_CATCH
}

// terms whose epilogue updates the Psi accumulator: every third one counted from the last
// (the epilogue of term m holds v_{m-2}, v_{m-1}, v_m of the row).  The first update must
// still see Psi = v_0, which term 2 overwrites in place, so it is forced to term <= 2.
static void acc_schedule(const double* a, int n_coeffs, bool defer, qp_acc_defer* out) {
  const int nterms = n_coeffs - 1;
  std::vector<char> upd((size_t)nterms + 1, defer ? 0 : 1);
  if (defer) {
    int m0 = nterms;
    for (; m0 >= 1; m0 -= 3) upd[m0] = 1;
    if (m0 + 3 == 3) upd[1] = 1;
  }
  int last_upd = 0;
  for (int m = 1; m <= nterms; ++m) {
    qp_acc_defer& d = out[m - 1];
    d = qp_acc_defer{0, 0, 0.0, 0.0};
    if (upd[m]) {
      d.n_defer = m - last_upd - 1;
      d.a_d1 = (d.n_defer >= 1) ? a[m - 1] : 0.0;
      d.a_d2 = (d.n_defer == 2) ? a[m - 2] : 0.0;
      last_upd = m;
    } else {
      d.skip = 1;
    }
  }
}

static void set_defer(qp::ChebyEpi& e, const qp_acc_defer* d) {
  if (!d) return;
  e.acc_skip = d->skip ? 1 : 0;
  e.n_defer = d->skip ? 0 : d->n_defer;
  e.a_d1 = d->a_d1;
  e.a_d2 = d->a_d2;
}

int qp_acc_schedule_host(const double* a, int n_coeffs, qp_acc_defer* out) {
  if (!a || !out || n_coeffs < 2) return qp::fail(QP_E_BAD_ARG, "qp_acc_schedule_host: bad arguments");
  acc_schedule(a, n_coeffs, true, out);
  return QP_OK;
}

int qp_cheby_term(qp_operator* op, const qp_state* x, int64_t xoff, const qp_state* v0, qp_state* vout,
                  const qp_state* acc_in, qp_state* acc_out, qp_c128 c, double beta, double a_prev, double a,
                  qp_c128 phase, const qp_acc_defer* defer) {
  QP_TRY
  const bool skip = defer && defer->skip;
  if (!op || !x || (!acc_out && !skip)) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: NULL argument");
  if (defer && !skip && (defer->n_defer < 0 || defer->n_defer > 2 || (defer->n_defer > 0 && !v0 && (defer->n_defer == 2 || !acc_in))))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: deferred accumulation needs v0");
  const int64_t nr = op->A.nrows;
  if (x->n != op->A.ncols || xoff < 0 || xoff + nr > x->n) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: x shape / offset mismatch");
  if ((v0 && v0->n != nr) || (vout && vout->n != nr) || (acc_in && acc_in->n != nr) || (acc_out && acc_out->n != nr))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: local vector length mismatch");
  auto overlaps = [&](const qp_state* s) { return s && s->d < x->d + x->n && x->d < s->d + s->n; };
  if (overlaps(vout) || overlaps(acc_out)) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term: outputs must not overlap the gathered x");
  QP_CHECK(use(op->ctx));
  qp::ChebyEpi e;
  e.xloc = x->d + xoff;
  e.v0 = v0 ? v0->d : nullptr;
  e.vout = vout ? vout->d : nullptr;
  e.acc_in = (acc_in && !skip) ? acc_in->d : nullptr;
  e.acc_out = (acc_out && !skip) ? acc_out->d : nullptr;
  e.c = d2(c);
  e.beta = beta;
  e.a_prev = a_prev;
  e.a = a;
  e.phase = d2(phase);
  e.apply_phase = !(phase.re == 1.0 && phase.im == 0.0);
  e.check_partials = nullptr;
  set_defer(e, defer);
  return qp::launch_spmv_cheby(op->ctx->stream, op->A, x->d, e, &op->ctx->stats);
  QP_CATCH
}

// the launches of one cheby! call (src/cheby.jl:171-211): n_coeffs - 1 fused mat-vec + term
// kernels and, when the result does not land in Psi's buffer, one copy
static int cheby_step_launches(qp_cheby* w, qp_operator* op, qp_state* psi, const double* a, int n_coeffs, double beta,
                               cplx c, cplx phase, bool check_normalization) {
  qp_ctx* ctx = op->ctx;
  const DevMatrix& A = op->A;
  const int nterms = n_coeffs - 1;
  const int nwg = qp::spmv_grid_size(A);
  double2* P = psi->d;
  double2* B = w->bufA;
  double2* ACC = w->acc;
  double2* result = nullptr;
  std::vector<qp_acc_defer> sched((size_t)nterms);
  acc_schedule(a, n_coeffs, qp::g_acc_defer != 0, sched.data());
  bool updated = false;   // has any term written the accumulator yet?
  for (int m = 1; m <= nterms; ++m) {
    const bool last = (m == nterms);
    qp::ChebyEpi e;
    const double2* x;
    if (m == 1) {
      // v0 = Psi; Psi = a1 v0; v1 = c (H v0 - beta v0); Psi += a2 v1     :171-182
      x = P;
      e.v0 = nullptr;
      e.vout = last ? nullptr : B;
      e.acc_in = nullptr;
      e.acc_out = ACC;
      result = ACC;
    } else {
      // v2 = c (H v1 - beta v1) + v0; Psi += a_i v2; rotate            :186-207
      double2* xb = (m % 2 == 0) ? B : P;   // holds v1 (gathered)
      double2* ob = (m % 2 == 0) ? P : B;   // holds v0, overwritten in place by v2
      x = xb;
      e.v0 = ob;
      e.vout = last ? nullptr : ob;
      e.acc_in = updated ? ACC : nullptr;
      e.acc_out = (last && xb == B) ? P : ACC;  // P may be written only while it is not gathered
      result = e.acc_out;
    }
    e.a_prev = updated ? 0.0 : a[0];
    set_defer(e, &sched[m - 1]);
    if (sched[m - 1].skip) {
      e.acc_in = nullptr;
      e.acc_out = nullptr;
    } else {
      updated = true;
    }
    e.xloc = x;
    e.c = d2(c);
    e.beta = beta;
    e.a = a[m];
    e.phase = d2(phase);
    e.apply_phase = last ? 1 : 0;
    // the reference checks terms i >= 3 only (inside the loop at :186)
    e.check_partials = (check_normalization && m >= 2) ? w->chk_part : nullptr;
    QP_CHECK(qp::launch_spmv_cheby(ctx->stream, A, x, e, &ctx->stats));
    if (e.check_partials)
      QP_CHECK(qp::launch_reduce_triples(ctx->stream, w->chk_part, nwg, w->chk_out + 3 * (m - 1), &ctx->stats));
    if (m == 1) c *= 2.0;  // :184
  }
  if (result != P)
    QP_HIP(hipMemcpyAsync(P, result, (size_t)psi->n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
  return QP_OK;
}

int qp_cheby_step(qp_cheby* w, qp_operator* op, qp_state* psi, const double* a, int n_coeffs, double Delta,
                  double E_min, double dt, double wrk_dt, double limit, int check_normalization) {
  QP_TRY
  if (!w || !op || !psi || !a) return qp::fail(QP_E_BAD_ARG, "qp_cheby_step: NULL argument");
  if (op->A.nrows != op->A.ncols || psi->n != op->A.nrows || w->n != psi->n)
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_step: shape mismatch");
  // @assert abs(dt) ~ abs(wrk.dt)   (isapprox, rtol = sqrt(eps))   src/cheby.jl:157
  {
    const double x = std::fabs(dt), y = std::fabs(wrk_dt);
    if (!(std::fabs(x - y) <= 1.4901161193847656e-08 * std::max(x, y)))
      return qp::fail(QP_E_DT_MISMATCH, "wrk was initialized for dt=%g, not dt=abs(%g)", wrk_dt, dt);
  }
  if (n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(Delta > 0)) return qp::fail(QP_E_BAD_ARG, "Delta must be positive");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  const double beta = (Delta / 2) + E_min;                        // :156
  cplx c = (dt > 0) ? cplx(0, -2.0) / Delta : cplx(0, 2.0) / Delta;  // :158-162
  const cplx phase = std::exp(cplx(0, -1) * beta * dt);            // :211
  const int nterms = n_coeffs - 1;
  const DevMatrix& A = op->A;
  const int nwg = qp::spmv_grid_size(A);
  if (check_normalization) {
    if (w->chk_wg < nwg) {
      if (w->chk_part) QP_HIP(hipFree(w->chk_part));
      QP_CHECK(dev_alloc(&w->chk_part, (size_t)3 * nwg));
      w->chk_wg = nwg;
    }
    if (w->chk_terms < nterms) {
      if (w->chk_out) QP_HIP(hipFree(w->chk_out));
      QP_CHECK(dev_alloc(&w->chk_out, (size_t)3 * nterms));
      w->chk_terms = nterms;
    }
  }
  // launch-bound systems (a term takes less than its launch): replay the step as a hipGraph
  bool done = false;
  if (qp::g_cheby_graph && !check_normalization && ctx->stream != nullptr && ctx->stream != hipStreamLegacy) {
    qp_cheby::GraphKey key;
    key.vals = A.vals_r ? (const void*)A.vals_r : (const void*)A.vals;
    key.cols = A.cols;
    key.rowptr = A.format == QP_FMT_CSR ? (const void*)A.rowptr : (const void*)A.bptr;
    key.psi = psi->d;
    key.format = A.format;
    key.variant = qp::g_rbcsr_variant;
    key.n_coeffs = n_coeffs;
    key.dt = dt;
    key.Delta = Delta;
    key.E_min = E_min;
    uint64_t h = 1469598103934665603ull;   // FNV-1a over the coefficient bits
    for (int i = 0; i < n_coeffs; ++i) {
      uint64_t bits;
      std::memcpy(&bits, &a[i], 8);
      h = (h ^ bits) * 1099511628211ull;
    }
    key.a_hash = h;
    auto replay = [&]() -> int {
      QP_HIP(hipGraphLaunch(w->gexec, ctx->stream));
      ctx->stats.n_graph_launch++;
      ctx->stats.n_matvec += w->gstats.n_matvec;
      ctx->stats.n_launch += w->gstats.n_launch;
      ctx->stats.spmv_bytes += w->gstats.spmv_bytes;
      return QP_OK;
    };
    if (w->gexec && key == w->gkey) {
      QP_CHECK(replay());
      done = true;
    } else if (key == w->gpending && (int64_t)spmv_grid_size(A) <= qp::g_cheby_graph) {
      // second identical call in a row: record it
      hipGraph_t graph = nullptr;
      const Stats before = ctx->stats;
      QP_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      const int rc = cheby_step_launches(w, op, psi, a, n_coeffs, beta, c, phase, false);
      w->gstats = Stats();
      w->gstats.n_matvec = ctx->stats.n_matvec - before.n_matvec;
      w->gstats.n_launch = ctx->stats.n_launch - before.n_launch;
      w->gstats.spmv_bytes = ctx->stats.spmv_bytes - before.spmv_bytes;
      ctx->stats = before;
      const hipError_t ec = hipStreamEndCapture(ctx->stream, &graph);
      if (rc != QP_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      QP_HIP(ec);
      if (w->gexec) {
        (void)hipGraphExecDestroy(w->gexec);
        w->gexec = nullptr;
      }
      const hipError_t ei = hipGraphInstantiate(&w->gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      QP_HIP(ei);
      w->gkey = key;
      QP_CHECK(replay());
      done = true;
    } else {
      w->gpending = key;
    }
  }
  if (!done) QP_CHECK(cheby_step_launches(w, op, psi, a, n_coeffs, beta, c, phase, check_normalization != 0));
  ctx->stats.n_cheby_steps++;
  if (check_normalization && nterms >= 2) {
    std::vector<double> h((size_t)3 * nterms);
    QP_HIP(hipMemcpyAsync(h.data(), w->chk_out, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipStreamSynchronize(ctx->stream));
    for (int m = 2; m <= nterms; ++m) {
      const double* t = &h[3 * (m - 1)];
      const double map_norm = std::hypot(t[0], t[1]) / (2 * t[2]);   // :195
      if (!(map_norm <= 1.0 + limit))
        return qp::fail(QP_E_NORMALIZATION, "Incorrect normalization (E_min=%g, Delta=%g)", E_min, Delta);
    }
  }
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// batched states (BASELINE configs[4]): panel X[i*b + s]
// ---------------------------------------------------------------------------
static int operator_csr_mirror(qp_operator* op, bool gather = true) {
  qp_ctx* ctx = op->ctx;
  const DevMatrix& A = op->A;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  const int64_t nnz = A.nnz;
  if (!op->m_rowptr) {
    std::vector<int64_t> map((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t r = 0; r < A.nrows; ++r) {
      const int64_t nl = (A.format == QP_FMT_HRB) ? op->layout.nlow[r] : 0;
      for (int64_t k = 0; k < ur[r + 1] - ur[r]; ++k) {
        int64_t m;
        if (A.format == QP_FMT_CSR) {
          m = ur[r] + k;
        } else if (k >= nl) {
          m = rb_val_pos(op->layout.bptr, r, k - nl);
        } else {
          const int64_t c = uc[ur[r] + k];
          const int32_t* b = uc.data() + ur[c];
          const int32_t* e = uc.data() + ur[c + 1];
          const int64_t kk = (std::lower_bound(b, e, (int32_t)r) - b) - op->layout.nlow[c];
          m = -rb_val_pos(op->layout.bptr, c, kk) - 1;
        }
        map[ur[r] + k] = m;
      }
    }
    QP_CHECK(dev_alloc(&op->m_rowptr, ur.size()));
    QP_CHECK(dev_alloc(&op->m_cols, (size_t)nnz));
    QP_CHECK(dev_alloc(&op->m_map, (size_t)nnz));
    QP_CHECK(dev_alloc(&op->m_vals, (size_t)nnz));
    QP_HIP(hipMemcpy(op->m_rowptr, ur.data(), ur.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    QP_HIP(hipMemcpy(op->m_cols, uc.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    QP_HIP(hipMemcpy(op->m_map, map.data(), (size_t)nnz * sizeof(int64_t), hipMemcpyHostToDevice));
    op->m_epoch = 0;
  }
  if (gather && op->m_epoch != op->vals_epoch) {
    QP_CHECK(qp::launch_gather_csr_vals(ctx->stream, op->m_vals, A.vals, op->m_map, nnz, &ctx->stats));
    op->m_epoch = op->vals_epoch;
  }
  return QP_OK;
}

int qp_cheby_step_batched(qp_cheby* w, qp_operator* op, qp_state* psi, int batch, const double* a, int n_coeffs,
                          double Delta, double E_min, double dt, double wrk_dt) {
  QP_TRY
  if (!w || !op || !psi || !a || batch < 1) return qp::fail(QP_E_BAD_ARG, "qp_cheby_step_batched: bad arguments");
  const int64_t n = op->A.nrows;
  if (op->A.nrows != op->A.ncols || psi->n != n * batch || w->n != psi->n)
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_step_batched: shape mismatch (operator %lld, batch %d, panel %lld)",
                    (long long)n, batch, (long long)psi->n);
  {
    const double x = std::fabs(dt), y = std::fabs(wrk_dt);
    if (!(std::fabs(x - y) <= 1.4901161193847656e-08 * std::max(x, y)))
      return qp::fail(QP_E_DT_MISMATCH, "wrk was initialized for dt=%g, not dt=abs(%g)", wrk_dt, dt);
  }
  if (n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(Delta > 0)) return qp::fail(QP_E_BAD_ARG, "Delta must be positive");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  QP_CHECK(operator_csr_mirror(op));
  const double beta = (Delta / 2) + E_min;
  cplx c = (dt > 0) ? cplx(0, -2.0) / Delta : cplx(0, 2.0) / Delta;
  const cplx phase = std::exp(cplx(0, -1) * beta * dt);
  const int nterms = n_coeffs - 1;
  double2* P = psi->d;
  double2* B = w->bufA;
  double2* ACC = w->acc;
  double2* result = nullptr;
  std::vector<qp_acc_defer> sched((size_t)nterms);
  acc_schedule(a, n_coeffs, qp::g_acc_defer != 0, sched.data());
  bool updated = false;
  for (int m = 1; m <= nterms; ++m) {   // same buffer rotation as qp_cheby_step, element = (row, state)
    const bool last = (m == nterms);
    qp::ChebyEpi e;
    const double2* x;
    if (m == 1) {
      x = P;
      e.v0 = nullptr;
      e.vout = last ? nullptr : B;
      e.acc_in = nullptr;
      e.acc_out = ACC;
      result = ACC;
    } else {
      double2* xb = (m % 2 == 0) ? B : P;
      double2* ob = (m % 2 == 0) ? P : B;
      x = xb;
      e.v0 = ob;
      e.vout = last ? nullptr : ob;
      e.acc_in = updated ? ACC : nullptr;
      e.acc_out = (last && xb == B) ? P : ACC;
      result = e.acc_out;
    }
    e.a_prev = updated ? 0.0 : a[0];
    set_defer(e, &sched[(size_t)m - 1]);
    if (sched[(size_t)m - 1].skip) {
      e.acc_in = nullptr;
      e.acc_out = nullptr;
    } else {
      updated = true;
    }
    e.xloc = x;
    e.c = d2(c);
    e.beta = beta;
    e.a = a[m];
    e.phase = d2(phase);
    e.apply_phase = last ? 1 : 0;
    e.check_partials = nullptr;
    QP_CHECK(qp::launch_spmm_cheby(ctx->stream, op->m_rowptr, op->m_cols, op->m_vals, x, n, op->A.nnz, batch, e,
                                   &ctx->stats));
    if (m == 1) c *= 2.0;
  }
  if (result != P) QP_HIP(hipMemcpyAsync(P, result, (size_t)psi->n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
  ctx->stats.n_cheby_steps++;
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// boundary / interior split of one fused term (overlap of the multi-GPU exchange)
// ---------------------------------------------------------------------------
int qp_split_create(qp_operator* op, const int64_t* send_rows, int64_t nsend, qp_split** out) {
  QP_TRY
  if (!op || !out || nsend < 0 || (nsend > 0 && !send_rows)) return qp::fail(QP_E_BAD_ARG, "qp_split_create: bad arguments");
  const DevMatrix& A = op->A;
  if (A.format != QP_FMT_RBCSR && A.format != QP_FMT_HRB)
    return qp::fail(QP_E_BAD_ARG, "qp_split_create needs a row-block device format (got %d)", A.format);
  QP_CHECK(use(op->ctx));
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  std::vector<char> is_boundary((size_t)A.nblocks, 0);
  std::vector<int32_t> slot_of_row((size_t)A.nrows, -1);
  for (int64_t i = 0; i < nsend; ++i) {
    const int64_t r = send_rows[i];
    if (r < 0 || r >= A.nrows) return qp::fail(QP_E_BAD_ARG, "send row %lld out of range", (long long)r);
    if (slot_of_row[r] >= 0) return qp::fail(QP_E_BAD_ARG, "send row %lld listed twice", (long long)r);
    slot_of_row[r] = (int32_t)i;
    is_boundary[r / kRB] = 1;
  }
  for (int64_t r = 0; r < A.nrows; ++r)   // rows that read a ghost column must wait for the exchange
    if (ur[r + 1] > ur[r] && uc[ur[r + 1] - 1] >= A.nrows) is_boundary[r / kRB] = 1;
  // interior blocks that exchange data with boundary rows inside the local block (they gather
  // from boundary rows, or boundary rows gather from them) are listed last: only they have to
  // wait for the boundary launch of the previous term
  std::vector<char> adjacent((size_t)A.nblocks, 0);
  for (int64_t r = 0; r < A.nrows; ++r) {
    const bool rb_ = is_boundary[r / kRB];
    for (int64_t p = ur[r]; p < ur[r + 1]; ++p) {
      const int64_t c = uc[p];
      if (c >= A.nrows) continue;
      const bool cb = is_boundary[c / kRB];
      if (rb_ && !cb) adjacent[c / kRB] = 1;
      if (!rb_ && cb) adjacent[r / kRB] = 1;
    }
  }
  std::vector<int32_t> bb, bi, bi_adj;
  for (int64_t b = 0; b < A.nblocks; ++b) {
    if (is_boundary[b]) bb.push_back((int32_t)b);
    else if (adjacent[b]) bi_adj.push_back((int32_t)b);
    else bi.push_back((int32_t)b);
  }
  const unsigned wait_from_wg = (unsigned)(bi.size() / (qp::kThreads / 64));
  bi.insert(bi.end(), bi_adj.begin(), bi_adj.end());
  std::vector<int32_t> mirror(bb.size() * kRB + 1, -1);
  for (size_t k = 0; k < bb.size(); ++k)
    for (int l = 0; l < kRB; ++l) {
      const int64_t r = (int64_t)bb[k] * kRB + l;
      if (r < A.nrows) mirror[k * kRB + l] = slot_of_row[r];
    }
  auto sp = std::make_unique<qp_split>();
  sp->op = op;
  sp->device = op->ctx->device;
  sp->n_boundary = (int64_t)bb.size();
  sp->n_interior = (int64_t)bi.size();
  sp->nsend = nsend;
  QP_CHECK(dev_alloc(&sp->bmap_boundary, bb.size()));
  QP_CHECK(dev_alloc(&sp->bmap_interior, bi.size()));
  QP_CHECK(dev_alloc(&sp->mirror, mirror.size()));
  if (!bb.empty()) QP_HIP(hipMemcpy(sp->bmap_boundary, bb.data(), bb.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!bi.empty()) QP_HIP(hipMemcpy(sp->bmap_interior, bi.data(), bi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  QP_HIP(hipMemcpy(sp->mirror, mirror.data(), mirror.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  QP_HIP(hipEventCreateWithFlags(&sp->ev_b, hipEventDisableTiming));
  QP_HIP(hipEventCreateWithFlags(&sp->ev_i, hipEventDisableTiming));
  QP_CHECK(dev_alloc(&sp->counter, 2));
  QP_HIP(hipMemset(sp->counter, 0, 2 * sizeof(unsigned)));
  sp->wait_from_wg = wait_from_wg;
  *out = sp.release();
  return QP_OK;
  QP_CATCH
}

int qp_split_destroy(qp_split* sp) {
  QP_TRY
  if (!sp) return QP_OK;
  (void)hipSetDevice(sp->device);
  (void)hipDeviceSynchronize();
  if (sp->bmap_boundary) (void)hipFree(sp->bmap_boundary);
  if (sp->bmap_interior) (void)hipFree(sp->bmap_interior);
  if (sp->mirror) (void)hipFree(sp->mirror);
  if (sp->counter) (void)hipFree(sp->counter);
  if (sp->ev_b) (void)hipEventDestroy(sp->ev_b);
  if (sp->ev_i) (void)hipEventDestroy(sp->ev_i);
  delete sp;
  return QP_OK;
  QP_CATCH
}

int qp_split_info(const qp_split* sp, int64_t* n_boundary_blocks, int64_t* n_interior_blocks) {
  if (!sp) return qp::fail(QP_E_BAD_ARG, "split is NULL");
  if (n_boundary_blocks) *n_boundary_blocks = sp->n_boundary;
  if (n_interior_blocks) *n_interior_blocks = sp->n_interior;
  return QP_OK;
}

/* synchronises the device; returns QP_E_INTERNAL if an in-launch wait ever timed out */
int qp_split_check(qp_split* sp) {
  QP_TRY
  if (!sp) return qp::fail(QP_E_BAD_ARG, "split is NULL");
  QP_HIP(hipSetDevice(sp->device));
  QP_HIP(hipDeviceSynchronize());
  unsigned h[2] = {0, 0};
  QP_HIP(hipMemcpy(h, sp->counter, sizeof(h), hipMemcpyDeviceToHost));
  if (h[1] != 0) return qp::fail(QP_E_INTERNAL, "an interior launch timed out waiting for its boundary launch");
  return QP_OK;
  QP_CATCH
}

int qp_cheby_term_split(qp_operator* op, qp_split* sp, void* boundary_stream, int first, const qp_state* x,
                        int64_t xoff, const qp_state* v0, qp_state* vout, const qp_state* acc_in, qp_state* acc_out,
                        qp_state* slab, qp_c128 c, double beta, double a_prev, double a, qp_c128 phase,
                        const qp_acc_defer* defer) {
  QP_TRY
  const bool skip = defer && defer->skip;
  if (!op || !sp || sp->op != op || !boundary_stream || !x || (!acc_out && !skip))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: bad arguments");
  if (defer && !skip && (defer->n_defer < 0 || defer->n_defer > 2 || (defer->n_defer > 0 && !v0 && (defer->n_defer == 2 || !acc_in))))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: deferred accumulation needs v0");
  const int64_t nr = op->A.nrows;
  if (x->n != op->A.ncols || xoff < 0 || xoff + nr > x->n) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: x shape / offset mismatch");
  if ((v0 && v0->n != nr) || (vout && vout->n != nr) || (acc_in && acc_in->n != nr) || (acc_out && acc_out->n != nr))
    return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: local vector length mismatch");
  if (slab && slab->n < sp->nsend) return qp::fail(QP_E_BAD_ARG, "qp_cheby_term_split: slab too small");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  hipStream_t S_c = ctx->stream, S_x = (hipStream_t)boundary_stream;
  qp::ChebyEpi e;
  e.xloc = x->d + xoff;
  e.v0 = v0 ? v0->d : nullptr;
  e.vout = vout ? vout->d : nullptr;
  e.acc_in = (acc_in && !skip) ? acc_in->d : nullptr;
  e.acc_out = (acc_out && !skip) ? acc_out->d : nullptr;
  e.c = d2(c);
  e.beta = beta;
  e.a_prev = a_prev;
  e.a = a;
  e.phase = d2(phase);
  e.apply_phase = !(phase.re == 1.0 && phase.im == 0.0);
  e.check_partials = nullptr;
  set_defer(e, defer);
  qp::RowSet rb{sp->bmap_boundary, sp->n_boundary, false};
  qp::RowSet ri{sp->bmap_interior, sp->n_interior, true};
  const bool flag_mode = (qp::g_split_mode == 1);
  if (first && flag_mode) {
    // the caller joined both streams: restart the signal counter (keeps it far from wrap)
    QP_HIP(hipMemsetAsync(sp->counter, 0, sizeof(unsigned), S_c));
    sp->signals_issued = 0;
    QP_HIP(hipEventRecord(sp->ev_i, S_c));
    QP_HIP(hipStreamWaitEvent(S_x, sp->ev_i, 0));
  }
  if (!first) {
    // boundary(m) overwrites rows that interior(m-1) gathered from, and vice versa.  The side
    // stream takes a queue-level event wait (its idle time is hidden); the main stream either
    // does the same (mode 0) or lets only the adjacent workgroups of the interior launch poll
    // the boundary launch's completion counter (mode 1: no idle gap between interior launches)
    QP_HIP(hipStreamWaitEvent(S_x, sp->ev_i, 0));
    if (!flag_mode) QP_HIP(hipStreamWaitEvent(S_c, sp->ev_b, 0));
  }
  if (flag_mode) {
    ri.sync.wait = sp->counter;
    ri.sync.wait_target = sp->signals_issued;      // every boundary workgroup launched so far
    ri.sync.wait_from_wg = sp->wait_from_wg;
    ri.sync.timeout_flag = sp->counter + 1;
    rb.sync.signal = sp->counter;
    sp->signals_issued += (unsigned)((sp->n_boundary + qp::kThreads / 64 - 1) / (qp::kThreads / 64));
  }
  qp::ChebyEpi eb = e;
  if (slab && vout) {   // the slab carries the new term vector (what the next term gathers)
    eb.mirror = sp->mirror;
    eb.slab = slab->d;
  }
  if (sp->n_boundary > 0) QP_CHECK(qp::launch_spmv_cheby(S_x, op->A, x->d, eb, &ctx->stats, &rb));
  if (!flag_mode) QP_HIP(hipEventRecord(sp->ev_b, S_x));
  if (sp->n_interior > 0) QP_CHECK(qp::launch_spmv_cheby(S_c, op->A, x->d, e, &ctx->stats, &ri));
  QP_HIP(hipEventRecord(sp->ev_i, S_c));
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// Arnoldi
// ---------------------------------------------------------------------------
int qp_krylov_create(qp_ctx* ctx, int64_t n, int nvec, qp_krylov** out) {
  QP_TRY
  if (!ctx || !out || n < 0 || nvec < 2) return qp::fail(QP_E_BAD_ARG, "qp_krylov_create: bad arguments");
  QP_CHECK(use(ctx));
  auto q = std::make_unique<qp_krylov>();
  q->ctx = ctx;
  q->n = n;
  q->nvec = nvec;
  QP_CHECK(dev_alloc(&q->Q, (size_t)n * nvec));
  QP_CHECK(dev_alloc(&q->hess_dev, (size_t)nvec * nvec));
  QP_CHECK(dev_alloc(&q->norms_dev, (size_t)nvec));
  QP_CHECK(dev_alloc(&q->part, (size_t)2 * kRedBlocks));
  QP_CHECK(dev_alloc(&q->md_part, (size_t)kRedBlocks * 2 * nvec));
  QP_CHECK(dev_alloc(&q->gram, (size_t)nvec * nvec));
  // rows that were never computed read as NaN: a use of a stale Gram row is loud, not subtle
  QP_HIP(hipMemsetAsync(q->gram, 0xFF, sizeof(double2) * (size_t)nvec * nvec, ctx->stream));
  QP_CHECK(dev_alloc(&q->hcoef, (size_t)2 * nvec));
  *out = q.release();
  return QP_OK;
  QP_CATCH
}

int qp_krylov_destroy(qp_krylov* q) {
  QP_TRY
  if (!q) return QP_OK;
  (void)hipSetDevice(q->ctx->device);
  (void)hipStreamSynchronize(q->ctx->stream);
  if (q->Q) (void)hipFree(q->Q);
  if (q->hess_dev) (void)hipFree(q->hess_dev);
  if (q->norms_dev) (void)hipFree(q->norms_dev);
  if (q->part) (void)hipFree(q->part);
  if (q->md_part) (void)hipFree(q->md_part);
  if (q->gram) (void)hipFree(q->gram);
  if (q->hcoef) (void)hipFree(q->hcoef);
  delete q;
  return QP_OK;
  QP_CATCH
}

int qp_krylov_download(const qp_krylov* q, int i, qp_c128* host) {
  QP_TRY
  if (!q || !host || i < 0 || i >= q->nvec) return qp::fail(QP_E_BAD_ARG, "qp_krylov_download: bad arguments");
  QP_CHECK(use(q->ctx));
  QP_HIP(hipMemcpyAsync(host, q->q(i), (size_t)q->n * sizeof(double2), hipMemcpyDeviceToHost, q->ctx->stream));
  QP_HIP(hipStreamSynchronize(q->ctx->stream));
  return QP_OK;
  QP_CATCH
}

namespace {

// q[j+1] = H q[j], then modified Gram-Schmidt against q[0..j] with the fused
// axpy->dot passes; leaves |q[j+1]|^2 partials in part[(j+1)&1].  hess column `hcol`
// (device, length >= j+1) receives dt*<q_i|q_j+1>.
int arnoldi_column(qp_operator* op, qp_krylov* q, int j, double dt, double2* hcol) {
  qp_ctx* ctx = op->ctx;
  qp::PlainEpi pe;
  pe.y = q->q(j + 1);
  pe.alpha = make_double2(1.0, 0.0);
  pe.beta = make_double2(0.0, 0.0);
  pe.beta_zero = 1;
  QP_CHECK(qp::launch_spmv_plain(ctx->stream, op->A, q->q(j), pe, &ctx->stats));  // src/arnoldi.jl:82
  if (qp::g_arnoldi_mode == 1 && q->gram_rows >= j) {
    // low-synchronisation MGS: same coefficients (to rounding), 3 launches per column;
    // leaves |q[j+1]|^2 partials in part[(j+1)&1] like the sequential path.  Needs the Gram
    // rows of the earlier basis vectors, which only this path maintains (a basis built by
    // the persistent small-system kernel or by sequential passes continues sequentially).
    q->gram_rows = j + 1;
    return qp::launch_mgs_lowsync(ctx->stream, q->Q, q->n, j, q->q(j + 1), q->md_part, q->gram, q->nvec, hcol,
                                  q->hcoef, q->part + (size_t)((j + 1) & 1) * kRedBlocks, dt, q->n, &ctx->stats);
  }
  q->gram_rows = std::min(q->gram_rows, j);
  for (int i = 0; i <= j + 1; ++i) {                                              // :84-87
    qp::MgsArgs a;
    a.w = q->q(j + 1);
    a.q_prev = (i > 0) ? q->q(i - 1) : nullptr;
    a.q_cur = (i <= j) ? q->q(i) : nullptr;
    a.part_in = q->part + (size_t)((i + 1) & 1) * kRedBlocks;
    a.part_out = q->part + (size_t)(i & 1) * kRedBlocks;
    a.hess_prev = (i > 0) ? hcol + (i - 1) : nullptr;
    a.dt = dt;
    a.n = q->n;
    QP_CHECK(qp::launch_mgs_pass(ctx->stream, a, &ctx->stats));
  }
  return QP_OK;
}

}  // namespace

__global__ void norm_guard_scale_kernel(double2* __restrict__ w, const double2* __restrict__ part_in, double2* hess_slot,
                                        double* norm_slot, double dt, double norm_min, int64_t n);

int qp_arnoldi(qp_operator* op, qp_krylov* q, int m, const qp_state* psi, double dt, int extended, double norm_min,
               qp_c128* Hess, int ldh, int* m_out) {
  QP_TRY
  if (!op || !q || !psi || !Hess || !m_out) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi: NULL argument");
  const int dim = extended ? m + 1 : m;
  if (m < 1 || ldh < dim || q->nvec < m + 1) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi: Hess/q too small for m=%d", m);
  if (op->A.nrows != op->A.ncols || psi->n != op->A.nrows || q->n != psi->n) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi: shape mismatch");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  const int ldd = q->nvec;
  std::memset(Hess, 0, sizeof(qp_c128) * (size_t)ldh * ldh);                                      // :78
  QP_HIP(hipMemsetAsync(q->hess_dev, 0, sizeof(double2) * (size_t)ldd * ldd, ctx->stream));
  QP_HIP(hipMemsetAsync(q->norms_dev, 0, sizeof(double) * (size_t)ldd, ctx->stream));
  qp::SmallArgs plan;
  bool small = false;
  if (op->A.nnz <= qp::g_small_nnz && qp::small_arnoldi_fits(q->n, m)) {
    int64_t maxrow = 0;
    for (int64_t r = 0; r < q->n; ++r) maxrow = std::max<int64_t>(maxrow, op->u_rowptr[r + 1] - op->u_rowptr[r]);
    small = qp::small_plan(q->n, maxrow, &plan);
  }
  if (small) {
    // all m columns in one persistent single-workgroup launch (kernels.hip: arnoldi_small_kernel)
    QP_CHECK(operator_csr_mirror(op, false));
    qp::SmallArnoldiArgs a;
    a.n = q->n;
    a.lanes = plan.lanes;
    a.ent = plan.ent;
    a.rows_per_group = plan.rows_per_group;
    a.rowptr = op->m_rowptr;
    a.cols = op->m_cols;
    a.map = op->m_map;
    a.vals = op->A.vals;
    a.start = psi->d;
    a.Q = q->Q;
    a.hess = q->hess_dev;
    a.norms = q->norms_dev;
    a.ldd = ldd;
    a.m = m;
    a.extended = extended;
    a.dt = dt;
    a.norm_min = norm_min;
    QP_CHECK(qp::launch_arnoldi_small(ctx->stream, a, &ctx->stats));
    q->gram_rows = 0;
  } else {
    QP_HIP(hipMemcpyAsync(q->q(0), psi->d, (size_t)q->n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));  // :79
    for (int j = 0; j < m; ++j) {
      double2* hcol = q->hess_dev + (size_t)j * ldd;
      QP_CHECK(arnoldi_column(op, q, j, dt, hcol));
      if ((j + 1 < m) || extended) {                                                               // :88-97
        hipLaunchKernelGGL(norm_guard_scale_kernel, dim3(2048), dim3(qp::kThreads), 0, ctx->stream, q->q(j + 1),
                           q->part + (size_t)((j + 1) & 1) * kRedBlocks, hcol + (j + 1), q->norms_dev + j, dt, norm_min,
                           q->n);
        QP_HIP(hipGetLastError());
        ctx->stats.n_launch++;
      }
    }
  }
  std::vector<cplx> hh((size_t)ldd * ldd);
  std::vector<double> hn(ldd);
  QP_HIP(hipMemcpyAsync(hh.data(), q->hess_dev, hh.size() * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
  QP_HIP(hipMemcpyAsync(hn.data(), q->norms_dev, hn.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  QP_HIP(hipStreamSynchronize(ctx->stream));
  int m_eff = m;
  for (int j = 0; j < m; ++j) {
    if (((j + 1 < m) || extended) && hn[j] < norm_min) {  // dimensionality exhausted  :91-95
      m_eff = j + 1;
      break;
    }
  }
  for (int j = 0; j < m_eff; ++j) {
    const int rows = std::min(j + 2, dim);
    for (int i = 0; i < rows; ++i) {
      cplx v = hh[(size_t)j * ldd + i];
      Hess[(size_t)j * ldh + i] = qp_c128{v.real(), v.imag()};
    }
  }
  *m_out = m_eff;
  return QP_OK;
  QP_CATCH
}

int qp_arnoldi_extend(qp_operator* op, qp_krylov* q, int m, double dt, double norm_min, qp_c128* Hess, int ldh,
                      int* extended_out) {
  QP_TRY
  if (!op || !q || !Hess) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi_extend: NULL argument");
  if (m < 2 || ldh < m || q->nvec < m + 1) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi_extend: Hess/q too small for m=%d", m);
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  if (extended_out) *extended_out = 0;
  cplx n2;
  QP_CHECK(dot_sync(ctx, q->q(m - 1), q->q(m - 1), q->n, &n2));
  const double h = std::sqrt(n2.real());                                   // src/arnoldi.jl:116
  if (h < norm_min) return QP_OK;                                          // :117
  Hess[(size_t)(m - 2) * ldh + (m - 1)] = qp_c128{dt * h, 0.0};            // :118
  const double inv = 1.0 / h;
  QP_CHECK(qp::launch_scal(ctx->stream, q->q(m - 1), make_double2(inv, 0.0), q->n, &ctx->stats));  // :119
  double2* hcol = q->hess_dev;  // scratch column
  QP_CHECK(arnoldi_column(op, q, m - 1, dt, hcol));                        // :120-124
  std::vector<cplx> hc(m);
  QP_HIP(hipMemcpyAsync(hc.data(), hcol, (size_t)m * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
  QP_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < m; ++i) Hess[(size_t)(m - 1) * ldh + i] = qp_c128{hc[i].real(), hc[i].imag()};
  if (extended_out) *extended_out = 1;
  return QP_OK;
  QP_CATCH
}

}  // extern "C"

// ---------------------------------------------------------------------------
// building blocks of a row-partitioned Arnoldi / Newton (the caller owns the collectives)
// ---------------------------------------------------------------------------
extern "C" {

int qp_krylov_vec(qp_krylov* q, int i, qp_state** out) {
  QP_TRY
  if (!q || !out || i < 0 || i >= q->nvec) return qp::fail(QP_E_BAD_ARG, "qp_krylov_vec: bad arguments");
  return qp_state_wrap(q->ctx, q->q(i), q->n, out);
  QP_CATCH
}

int qp_krylov_multidot(qp_krylov* q, int j, qp_state* reduced) {
  QP_TRY
  if (!q || !reduced || j < 0 || j + 1 >= q->nvec || reduced->n < 2 * (j + 1))
    return qp::fail(QP_E_BAD_ARG, "qp_krylov_multidot: bad arguments");
  QP_CHECK(use(q->ctx));
  return qp::launch_mgs_multidot(q->ctx->stream, q->Q, q->n, j, q->q(j + 1), q->md_part, reduced->d, q->n, &q->ctx->stats);
  QP_CATCH
}

int qp_krylov_project(qp_krylov* q, int j, double dt, const qp_state* reduced, qp_state* hess_col,
                      qp_state* norm_partials) {
  QP_TRY
  if (!q || !reduced || !hess_col || !norm_partials || j < 0 || j + 1 >= q->nvec || reduced->n < 2 * (j + 1) ||
      hess_col->n < j + 1 || norm_partials->n < kRedBlocks)
    return qp::fail(QP_E_BAD_ARG, "qp_krylov_project: bad arguments");
  QP_CHECK(use(q->ctx));
  return qp::launch_mgs_project(q->ctx->stream, q->Q, q->n, j, q->q(j + 1), reduced->d, q->gram, q->nvec, hess_col->d,
                                norm_partials->d, dt, q->n, &q->ctx->stats);
  QP_CATCH
}

int qp_krylov_normalize(qp_krylov* q, int j, double dt, double norm_min, const qp_state* norm_partials,
                        qp_state* hess_norm) {
  QP_TRY
  if (!q || !norm_partials || !hess_norm || j < 0 || j + 1 >= q->nvec || norm_partials->n < kRedBlocks || hess_norm->n < 2)
    return qp::fail(QP_E_BAD_ARG, "qp_krylov_normalize: bad arguments");
  QP_CHECK(use(q->ctx));
  hipLaunchKernelGGL(norm_guard_scale_kernel, dim3(2048), dim3(qp::kThreads), 0, q->ctx->stream, q->q(j + 1),
                     norm_partials->d, hess_norm->d, reinterpret_cast<double*>(hess_norm->d + 1), dt, norm_min, q->n);
  QP_HIP(hipGetLastError());
  q->ctx->stats.n_launch++;
  return QP_OK;
  QP_CATCH
}

int qp_combine(qp_state* out, int use_out, qp_c128 s0, qp_krylov* q, int first, int m, const qp_c128* coefs,
               qp_state* norm_partials) {
  QP_TRY
  if (!out || !q || !coefs || first < 0 || m < 1 || first + m > q->nvec || out->n != q->n ||
      (norm_partials && norm_partials->n < kRedBlocks))
    return qp::fail(QP_E_BAD_ARG, "qp_combine: bad arguments");
  QP_CHECK(use(q->ctx));
  return qp::launch_combine_vecs(q->ctx->stream, out->d, use_out, d2(s0), q->q(first), q->n, m,
                                 reinterpret_cast<const double2*>(coefs), norm_partials ? norm_partials->d : nullptr,
                                 q->n, &q->ctx->stats);
  QP_CATCH
}

}  // extern "C"

// norm + guarded scale: lmul!(1/h) only when h >= norm_min (src/arnoldi.jl:89-96); the
// raw norm is kept so that the host can detect breakdown also for dt < 0.
__global__ __launch_bounds__(qp::kThreads) void norm_guard_scale_kernel(double2* __restrict__ w,
                                                                        const double2* __restrict__ part_in,
                                                                        double2* hess_slot, double* norm_slot, double dt,
                                                                        double norm_min, int64_t n) {
  __shared__ double2 lds[qp::kThreads / 64];
  double2 v = part_in[threadIdx.x];
  for (int o = 32; o > 0; o >>= 1) {
    v.x += __shfl_down(v.x, o, 64);
  }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  const double s2 = lds[0].x + lds[1].x + lds[2].x + lds[3].x;
  const double h = sqrt(s2);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *hess_slot = make_double2(dt * h, 0.0);
    *norm_slot = h;
  }
  if (h < norm_min) return;
  const double inv = 1.0 / h;
  for (int64_t i = (int64_t)blockIdx.x * qp::kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * qp::kThreads) {
    double2 t = w[i];
    t.x *= inv;
    t.y *= inv;
    w[i] = t;
  }
}

// ---------------------------------------------------------------------------
// Newton
// ---------------------------------------------------------------------------
extern "C" {

int qp_newton_create(qp_ctx* ctx, int64_t n, int m_max, qp_newton** out) {
  QP_TRY
  if (!ctx || !out || n < 0) return qp::fail(QP_E_BAD_ARG, "qp_newton_create: bad arguments");
  if (m_max <= 2) return qp::fail(QP_E_M_MAX, "Newton propagation requires m_max > 2");          // src/newton.jl:38-40
  if (m_max >= n) {                                                                              // :41-46
    m_max = (int)n - 1;
    if (m_max <= 2) return qp::fail(QP_E_M_MAX, "Newton propagation requires state dimension > 2");
  }
  QP_CHECK(use(ctx));
  auto w = std::make_unique<qp_newton>();
  w->ctx = ctx;
  w->n = n;
  w->m_max = m_max;
  QP_CHECK(qp_krylov_create(ctx, n, m_max + 1, &w->q));
  QP_CHECK(dev_alloc(&w->v, (size_t)n));
  QP_CHECK(dev_alloc(&w->npart, (size_t)kRedBlocks));
  QP_HIP(hipHostMalloc((void**)&w->h_npart, kRedBlocks * sizeof(double2), hipHostMallocDefault));
  w->a.assign((size_t)10 * m_max + 1, cplx(0));      // :50-51
  w->leja.assign((size_t)10 * m_max + 1, cplx(0));
  *out = w.release();
  return QP_OK;
  QP_CATCH
}

int qp_newton_destroy(qp_newton* w) {
  QP_TRY
  if (!w) return QP_OK;
  (void)hipSetDevice(w->ctx->device);
  (void)hipStreamSynchronize(w->ctx->stream);
  qp_krylov_destroy(w->q);
  if (w->v) (void)hipFree(w->v);
  if (w->npart) (void)hipFree(w->npart);
  if (w->h_npart) (void)hipHostFree(w->h_npart);
  delete w;
  return QP_OK;
  QP_CATCH
}

int qp_newton_get_coeffs(const qp_newton* w, qp_c128* a, qp_c128* leja, int cap) {
  if (!w) return qp::fail(QP_E_BAD_ARG, "newton workspace is NULL");
  if (cap < w->n_a) return qp::fail(QP_E_BAD_ARG, "need room for %d coefficients", w->n_a);
  for (int i = 0; i < w->n_a; ++i) {
    if (a) a[i] = qp_c128{w->a[i].real(), w->a[i].imag()};
    if (leja) leja[i] = qp_c128{w->leja[i].real(), w->leja[i].imag()};
  }
  return QP_OK;
}

int qp_newton_step(qp_newton* w, qp_operator* op, qp_state* psi, double dt, int func_id, qp_func_cb cb, void* user,
                   double norm_min, double relerr, int max_restarts, qp_newton_stats* stats) {
  QP_TRY
  if (!w || !op || !psi) return qp::fail(QP_E_BAD_ARG, "qp_newton_step: NULL argument");
  if (op->A.nrows != op->A.ncols || psi->n != op->A.nrows || w->n != psi->n) return qp::fail(QP_E_BAD_ARG, "qp_newton_step: shape mismatch");
  if (func_id == QP_FUNC_CALLBACK && !cb) return qp::fail(QP_E_BAD_ARG, "callback func is NULL");
  if (func_id < 0 || func_id > QP_FUNC_CALLBACK) return qp::fail(QP_E_BAD_ARG, "bad func_id");
  if (dt == 0.0) return qp::fail(QP_E_BAD_ARG, "dt must be non-zero");   // src/newton.jl:263
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  const int m_max = w->m_max;
  int m = m_max;                                                        // :253
  std::fill(w->a.begin(), w->a.end(), cplx(0));                         // :254-255
  std::fill(w->leja.begin(), w->leja.end(), cplx(0));
  const int ldh = m_max + 1;
  std::vector<cplx> Hess((size_t)ldh * ldh, cplx(0));
  std::vector<cplx> R(m + 1), P(m + 1), Rn(m + 1), ritz;
  int n_a = 0, n_leja = 0, s = 0, n_matvec = 0;
  double last_relerr = 0, norm_psi = 0;
  const size_t bytes = (size_t)w->n * sizeof(double2);
  qp_state vstate{ctx, w->v, w->n, false};
  QP_HIP(hipMemcpyAsync(w->v, psi->d, bytes, hipMemcpyDeviceToDevice, ctx->stream));  // :268
  cplx n2;
  QP_CHECK(dot_sync(ctx, w->v, w->v, w->n, &n2));
  double beta = std::sqrt(n2.real());                                                // :271
  QP_CHECK(qp::launch_scal(ctx->stream, w->v, make_double2(1.0 / beta, 0.0), w->n, &ctx->stats));  // :272
  double ms_arnoldi = 0, ms_eig = 0, ms_leja = 0, ms_coeffs = 0, ms_poly = 0, ms_update = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  while (true) {                                                                     // :274
    int m_req = m;
    auto t0 = now();
    QP_CHECK(qp_arnoldi(op, w->q, m_req, &vstate, dt, 1, norm_min, reinterpret_cast<qp_c128*>(Hess.data()), ldh, &m));
    ms_arnoldi += ms_since(t0);
    n_matvec += m_req;
    if (m == 1 && s == 0) {                                                          // :289-295
      const cplx lam = beta * Hess[0];
      const cplx f = qp::eval_func(func_id, cb, user, lam);
      QP_CHECK(qp::launch_scal(ctx->stream, psi->d, d2(f), psi->n, &ctx->stats));
      break;
    }
    ritz.assign((size_t)m * (m + 1) / 2, cplx(0));
    t0 = now();
    if (qp::diagonalize_hessenberg(Hess.data(), ldh, m, true, ritz.data()) != QP_OK)  // :297
      return qp::fail(QP_E_INTERNAL, "Hessenberg QR did not converge");
    ms_eig += ms_since(t0);
    if (s == 0) {                                                                    // :301-303, :67-70
      double rmax = 0;
      for (auto& z : ritz) rmax = std::max(rmax, std::abs(z));
      w->radius = 1.2 * rmax;
    }
    const int n_s = n_leja;                                                          // :307
    if ((int)w->leja.size() < n_leja + m) w->leja.resize((size_t)2 * (n_leja + m), cplx(0));  // :105-110
    t0 = now();
    qp::extend_leja(w->leja.data(), n_leja, ritz.data(), (int)ritz.size(), m);
    ms_leja += ms_since(t0);
    n_leja += m;
    if ((int)w->a.size() < n_leja) w->a.resize((size_t)2 * n_leja, cplx(0));         // :187-192
    {
      t0 = now();
      int st = qp::extend_newton_coeffs(w->a.data(), n_a, w->leja.data(), func_id, cb, user, n_leja, w->radius);  // :314
      ms_coeffs += ms_since(t0);
      if (st == QP_E_DIVDIFF_UNDERFLOW) return qp::fail(st, "Divided differences too small");
      if (st != QP_OK) return qp::fail(st, "extend_newton_coeffs failed (radius=%g)", w->radius);
      n_a = n_leja;
    }
    // Newton polynomial in the extended Hessenberg matrix                           :328-343
    t0 = now();
    const int mp = m + 1;
    R.assign(mp, cplx(0));
    P.assign(mp, cplx(0));
    Rn.assign(mp, cplx(0));
    R[0] = beta;
    P[0] = w->a[n_s] * beta;
    auto apply = [&](cplx z) {
      for (int i = 0; i < mp; ++i) {
        cplx acc = 0;
        for (int k = 0; k < mp; ++k) acc += Hess[(size_t)k * ldh + i] * R[k];
        Rn[i] = (acc - z * R[i]) / w->radius;
      }
      std::swap(R, Rn);
    };
    for (int k = 1; k <= m - 1; ++k) {
      apply(w->leja[n_s + k - 1]);
      for (int i = 0; i < mp; ++i) P[i] += w->a[n_s + k] * R[i];
    }
    ms_poly += ms_since(t0);
    t0 = now();
    // Psi = (s == 0 ? 0 : Psi) + sum_i P_i q_i                                      :346-352
    QP_CHECK(qp::launch_combine_vecs(ctx->stream, psi->d, s == 0 ? 0 : 1, make_double2(1.0, 0.0), w->q->q(0), w->n, m,
                                     reinterpret_cast<const double2*>(P.data()), w->npart, w->n, &ctx->stats));
    QP_HIP(hipMemcpyAsync(w->h_npart, w->npart, kRedBlocks * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    // starting vector of the next restart                                            :356-367
    apply(w->leja[n_s + m - 1]);
    double b2 = 0;
    for (int i = 0; i < mp; ++i) {
      const double ab = std::abs(R[i]);
      b2 += ab * ab;
    }
    beta = std::sqrt(b2);
    for (int i = 0; i < mp; ++i) R[i] *= (1.0 / beta);
    QP_CHECK(qp::launch_combine_vecs(ctx->stream, w->v, 1, d2(R[0]), w->q->q(1), w->n, m,
                                     reinterpret_cast<const double2*>(R.data() + 1), nullptr, w->n, &ctx->stats));
    QP_HIP(hipStreamSynchronize(ctx->stream));
    norm_psi = std::sqrt(sum_partials(w->h_npart).real());
    ms_update += ms_since(t0);
    last_relerr = beta * std::abs(w->a[n_a - 1]) / (1 + norm_psi);                    // :370
    if (last_relerr < relerr) break;
    s += 1;
    if (s > max_restarts) {                                                           // :375
      w->restarts = s;
      return qp::fail(QP_E_MAX_RESTARTS, "newton!: s=%d exceeds max_restarts=%d (relerr=%g)", s, max_restarts, last_relerr);
    }
  }
  w->restarts = s;
  w->n_leja = n_leja;
  w->n_a = n_a;
  ctx->stats.n_newton_steps++;
  ctx->stats.n_restarts += s;
  if (stats) {
    stats->restarts = s;
    stats->n_a = n_a;
    stats->n_leja = n_leja;
    stats->m_last = m;
    stats->n_matvec = n_matvec;
    stats->radius = w->radius;
    stats->last_relerr = last_relerr;
    stats->norm_psi = norm_psi;
    stats->ms_arnoldi = ms_arnoldi;
    stats->ms_eig = ms_eig;
    stats->ms_leja = ms_leja;
    stats->ms_coeffs = ms_coeffs;
    stats->ms_poly = ms_poly;
    stats->ms_update = ms_update;
  }
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// propagate step loop (src/propagate.jl:283-344)
// ---------------------------------------------------------------------------
// Small systems: the whole time grid in one persistent single-workgroup launch
// (kernels.hip: cheby_propagate_small_kernel).  Same arguments as qp_propagate, method 0.
static int propagate_cheby_small(qp_operator* op, qp_state* psi, const qp_prop_spec* spec, qp::SmallArgs a,
                                 const double* dts,
                                 const qp_c128* coeff_table, int ncoeffs, int nsteps, qp_operator* const* observables,
                                 int nobs, qp_c128* expvals_out, qp_c128* states_out) {
  qp_ctx* ctx = op->ctx;
  qp_cheby* w = spec->cheby;
  const int64_t n = psi->n;
  const size_t rows = (size_t)nsteps + 1;
  if (!w || !spec->a) return qp::fail(QP_E_BAD_ARG, "qp_propagate: NULL Chebychev workspace");
  if (op->A.nrows != op->A.ncols || n != op->A.nrows || w->n != n) return qp::fail(QP_E_BAD_ARG, "qp_propagate: shape mismatch");
  if (spec->n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(spec->Delta > 0)) return qp::fail(QP_E_BAD_ARG, "Delta must be positive");
  for (int k = 0; k < nsteps; ++k) {
    const double x = std::fabs(dts[k]), y = std::fabs(spec->wrk_dt);   // src/cheby.jl:157
    if (!(std::fabs(x - y) <= 1.4901161193847656e-08 * std::max(x, y)))
      return qp::fail(QP_E_DT_MISMATCH, "wrk was initialized for dt=%g, not dt=abs(%g)", spec->wrk_dt, dts[k]);
    if ((dts[k] > 0) != (dts[0] > 0)) return qp::fail(QP_E_BAD_ARG, "qp_propagate: time steps change sign");
  }
  QP_CHECK(operator_csr_mirror(op));
  for (int o = 0; o < nobs; ++o) QP_CHECK(operator_csr_mirror(observables[o]));
  const double dt = dts[0];
  const double beta = (spec->Delta / 2) + spec->E_min;
  const cplx c = (dt > 0) ? cplx(0, -2.0) / spec->Delta : cplx(0, 2.0) / spec->Delta;

  a.n = n;
  a.nnz = op->A.nnz;
  a.rowptr = op->m_rowptr;
  a.cols = op->m_cols;
  a.map = op->m_map;
  a.planes = op->planes_dev;
  a.nops = op->nops;
  a.ncoeffs = ncoeffs;
  a.scale = d2(op->scale);
  a.nsteps = nsteps;
  a.n_coeffs = spec->n_coeffs;
  a.c = d2(c);
  a.beta = beta;
  a.phase = d2(std::exp(cplx(0, -1) * beta * dt));
  a.psi = psi->d;
  a.nobs = nobs;
  a.check = spec->check_normalization ? 1 : 0;
  a.limit = spec->limit;

  // one staging buffer: [table | a | obs descriptors | fail | expvals | work | states]
  std::vector<void*> owned;
  struct Free {
    std::vector<void*>& v;
    ~Free() {
      for (void* p : v) (void)hipFree(p);
    }
  } guard{owned};
  auto upload = [&](const void* src, size_t bytes, void** out) -> int {
    void* d = nullptr;
    QP_HIP(hipMalloc(&d, std::max<size_t>(bytes, 16)));
    owned.push_back(d);
    if (src && bytes) QP_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = d;
    return QP_OK;
  };
  void* p = nullptr;
  QP_CHECK(upload(coeff_table, sizeof(qp_c128) * (size_t)nsteps * ncoeffs, &p));
  a.table = static_cast<const double2*>(p);
  QP_CHECK(upload(spec->a, sizeof(double) * (size_t)spec->n_coeffs, &p));
  a.a = static_cast<const double*>(p);
  std::vector<qp::SmallObs> hobs((size_t)nobs);
  for (int o = 0; o < nobs; ++o) hobs[o] = qp::SmallObs{observables[o]->m_rowptr, observables[o]->m_cols, observables[o]->m_vals};
  QP_CHECK(upload(hobs.data(), sizeof(qp::SmallObs) * (size_t)nobs, &p));
  a.obs = static_cast<const qp::SmallObs*>(p);
  QP_CHECK(upload(nullptr, sizeof(int) * 4, &p));
  a.fail = static_cast<int*>(p);
  QP_HIP(hipMemsetAsync(a.fail, 0, sizeof(int) * 4, ctx->stream));
  QP_CHECK(upload(nullptr, sizeof(double2) * rows * (size_t)nobs, &p));
  a.expvals = static_cast<double2*>(p);
  if (states_out) {
    QP_CHECK(upload(nullptr, sizeof(double2) * rows * (size_t)n, &p));
    a.states = static_cast<double2*>(p);
  }
  QP_CHECK(qp::launch_cheby_propagate_small(ctx->stream, a, &ctx->stats));
  ctx->stats.n_cheby_steps += nsteps;
  int fail[4] = {0, 0, 0, 0};
  QP_HIP(hipMemcpyAsync(fail, a.fail, sizeof(fail), hipMemcpyDeviceToHost, ctx->stream));
  if (nobs > 0)
    QP_HIP(hipMemcpyAsync(expvals_out, a.expvals, sizeof(double2) * rows * (size_t)nobs, hipMemcpyDeviceToHost, ctx->stream));
  if (states_out)
    QP_HIP(hipMemcpyAsync(states_out, a.states, sizeof(double2) * rows * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  QP_HIP(hipStreamSynchronize(ctx->stream));
  // leave the operator as the step-by-step loop would: holding the last interval's values
  if (ncoeffs > 0) QP_CHECK(qp_operator_set_coeffs(op, coeff_table + (size_t)(nsteps - 1) * ncoeffs, ncoeffs));
  if (fail[0])
    return qp::fail(QP_E_NORMALIZATION, "Incorrect normalization (E_min=%g, Delta=%g) in step %d, term %d", spec->E_min,
                    spec->Delta, fail[1] + 1, fail[2] + 1);
  return QP_OK;
}

int qp_propagate(qp_operator* op, qp_state* psi, const qp_prop_spec* spec, const double* dts,
                 const qp_c128* coeff_table, int ncoeffs, int nsteps, qp_operator* const* observables, int nobs,
                 qp_c128* expvals_out, qp_c128* states_out) {
  QP_TRY
  if (!op || !psi || !spec || !dts || nsteps < 0 || nobs < 0 || (nobs > 0 && (!observables || !expvals_out)))
    return qp::fail(QP_E_BAD_ARG, "qp_propagate: bad arguments");
  if (ncoeffs != op->ncoeffs || (ncoeffs > 0 && nsteps > 0 && !coeff_table))
    return qp::fail(QP_E_BAD_ARG, "qp_propagate: expected %d coefficients per step", op->ncoeffs);
  if (spec->method != 0 && spec->method != 1) return qp::fail(QP_E_BAD_ARG, "qp_propagate: bad method");
  for (int o = 0; o < nobs; ++o)
    if (!observables[o] || observables[o]->A.nrows != psi->n || observables[o]->A.ncols != psi->n)
      return qp::fail(QP_E_BAD_ARG, "qp_propagate: observable %d has the wrong shape", o);
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  const int64_t n = psi->n;
  const size_t rows = (size_t)nsteps + 1;
  if (spec->method == 0 && nsteps > 0 && op->A.nnz <= qp::g_small_nnz && op->nops <= 64) {
    qp::SmallArgs plan;
    int64_t maxrow = 0;
    for (int64_t r = 0; r < n; ++r) maxrow = std::max<int64_t>(maxrow, op->u_rowptr[r + 1] - op->u_rowptr[r]);
    if (qp::small_plan(n, maxrow, &plan))
      return propagate_cheby_small(op, psi, spec, plan, dts, coeff_table, ncoeffs, nsteps, observables, nobs,
                                   expvals_out, states_out);
  }
  // device staging, released at the end: observable partials and the state history
  double2* d_part = nullptr;
  double2* d_tmp = nullptr;
  double2* d_states = nullptr;
  struct Free {
    double2 *&a, *&b, *&c;
    ~Free() {
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
      if (c) (void)hipFree(c);
    }
  } guard{d_part, d_tmp, d_states};
  if (nobs > 0) {
    QP_CHECK(dev_alloc(&d_part, rows * nobs * kRedBlocks));
    QP_CHECK(dev_alloc(&d_tmp, (size_t)n));
  }
  if (states_out) QP_CHECK(dev_alloc(&d_states, rows * (size_t)n));
  auto record = [&](size_t row) -> int {
    for (int o = 0; o < nobs; ++o) {   // <psi|O|psi> = dot(psi, O psi)
      qp::PlainEpi e;
      e.y = d_tmp;
      e.alpha = make_double2(1.0, 0.0);
      e.beta = make_double2(0.0, 0.0);
      e.beta_zero = 1;
      QP_CHECK(qp::launch_spmv_plain(ctx->stream, observables[o]->A, psi->d, e, &ctx->stats));
      QP_CHECK(qp::launch_dot_partials(ctx->stream, psi->d, d_tmp, d_part + (row * nobs + o) * kRedBlocks, n, &ctx->stats));
    }
    if (d_states)
      QP_HIP(hipMemcpyAsync(d_states + row * (size_t)n, psi->d, (size_t)n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
    return QP_OK;
  };
  QP_CHECK(record(0));
  for (int k = 0; k < nsteps; ++k) {
    if (ncoeffs > 0) QP_CHECK(qp_operator_set_coeffs(op, coeff_table + (size_t)k * ncoeffs, ncoeffs));
    if (spec->method == 0) {
      QP_CHECK(qp_cheby_step(spec->cheby, op, psi, spec->a, spec->n_coeffs, spec->Delta, spec->E_min, dts[k],
                             spec->wrk_dt, spec->limit, spec->check_normalization));
    } else {
      QP_CHECK(qp_newton_step(spec->newton, op, psi, dts[k], spec->func_id, spec->cb, spec->user, spec->norm_min,
                              spec->relerr, spec->max_restarts, nullptr));
    }
    QP_CHECK(record((size_t)k + 1));
  }
  if (nobs > 0) {
    std::vector<cplx> hp(rows * nobs * kRedBlocks);
    QP_HIP(hipMemcpyAsync(hp.data(), d_part, hp.size() * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < rows * nobs; ++i) {
      const cplx v = sum_partials(reinterpret_cast<const double2*>(hp.data() + i * kRedBlocks));
      expvals_out[i] = qp_c128{v.real(), v.imag()};
    }
  }
  if (states_out) {
    QP_HIP(hipMemcpyAsync(states_out, d_states, rows * (size_t)n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipStreamSynchronize(ctx->stream));
  }
  return QP_OK;
  QP_CATCH
}

// ---------------------------------------------------------------------------
// SpectralRange
// ---------------------------------------------------------------------------
int qp_ritzvals(qp_operator* op, const qp_state* state, int m_min, int m_max, double prec, double norm_min,
                qp_c128* out, int* n_out) {
  QP_TRY
  if (!op || !state || !out || !n_out) return qp::fail(QP_E_BAD_ARG, "qp_ritzvals: NULL argument");
  if (m_max <= m_min) return qp::fail(QP_E_BAD_ARG, "m_max=%d must be larger than m_min=%d", m_max, m_min);  // src/specrad.jl:171-173
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  int m = std::max(5, std::min(m_min, m_max - 1));                         // :174
  if (m_max < m) return qp::fail(QP_E_BAD_ARG, "m_max=%d too small (need >= %d)", m_max, m);
  const int ldh = m_max;
  std::vector<cplx> Hess((size_t)ldh * ldh, cplx(0));
  qp_krylov* q = nullptr;
  QP_CHECK(qp_krylov_create(ctx, state->n, m_max + 1, &q));
  std::unique_ptr<qp_krylov, int (*)(qp_krylov*)> guard(q, qp_krylov_destroy);
  std::vector<cplx> ev;
  auto stats3 = [&](double& lo, double& hi, double& im) {
    lo = ev[0].real();
    hi = ev[0].real();
    im = std::fabs(ev[0].imag());
    for (auto& z : ev) {
      lo = std::min(lo, z.real());
      hi = std::max(hi, z.real());
      im = std::max(im, std::fabs(z.imag()));
    }
  };
  auto diag = [&](int mm) -> int {
    ev.assign(mm, cplx(0));
    return qp::diagonalize_hessenberg(Hess.data(), ldh, mm, false, ev.data());
  };
  int m0 = m - 1;
  QP_CHECK(qp_arnoldi(op, q, m0, state, 1.0, 0, norm_min, reinterpret_cast<qp_c128*>(Hess.data()), ldh, &m0));  // :182
  if (diag(m0) != QP_OK) return qp::fail(QP_E_INTERNAL, "Hessenberg QR did not converge");
  double lo0, hi0, im0;
  stats3(lo0, hi0, im0);
  if (m0 == m - 1) {
    int ext = 0;
    QP_CHECK(qp_arnoldi_extend(op, q, m, 1.0, norm_min, reinterpret_cast<qp_c128*>(Hess.data()), ldh, &ext));  // :190
    if (diag(m) != QP_OK) return qp::fail(QP_E_INTERNAL, "Hessenberg QR did not converge");
    double lo, hi, im;
    stats3(lo, hi, im);
    double er_lo = (lo0 != 0.0) ? std::fabs(1.0 - lo / lo0) : 0.0;
    double er_hi = (hi0 != 0.0) ? std::fabs(1.0 - hi / hi0) : 0.0;
    double ei = (im0 != 0.0) ? std::fabs(1.0 - im / im0) : 0.0;
    while ((er_lo > prec) || (er_hi > prec) || ((im0 > 1e-14) && ei > prec)) {   // :198
      lo0 = lo;
      hi0 = hi;
      im0 = im;
      m = m + 1;
      // quirk kept: the reference discards extend_arnoldi!'s return value, so Krylov
      // exhaustion is never detected here (:204-205)
      QP_CHECK(qp_arnoldi_extend(op, q, m, 1.0, norm_min, reinterpret_cast<qp_c128*>(Hess.data()), ldh, &ext));
      if (diag(m) != QP_OK) return qp::fail(QP_E_INTERNAL, "Hessenberg QR did not converge");
      stats3(lo, hi, im);
      er_lo = std::fabs(1.0 - (lo / lo0));
      er_hi = std::fabs(1.0 - (hi / hi0));
      ei = std::fabs(1.0 - (im / im0));
      if (m == m_max) break;                                                     // :213-216
    }
  }
  *n_out = (int)ev.size();
  for (size_t i = 0; i < ev.size(); ++i) out[i] = qp_c128{ev[i].real(), ev[i].imag()};
  return QP_OK;
  QP_CATCH
}

int qp_specrange_arnoldi(qp_operator* op, const qp_state* state, int m_min, int m_max, double prec, double norm_min,
                         int enlarge, double* E_min, double* E_max) {
  QP_TRY
  if (!E_min || !E_max) return qp::fail(QP_E_BAD_ARG, "qp_specrange_arnoldi: NULL output");
  m_min = std::max(5, std::min(m_min, m_max - 1));                              // src/specrad.jl:97
  std::vector<qp_c128> R((size_t)std::max(m_max, 8));
  int n = 0;
  QP_CHECK(qp_ritzvals(op, state, m_min, m_max, prec, norm_min, R.data(), &n));
  double lo = R[0].re, hi = R[n - 1].re;                                        // :103-104
  if (enlarge && n > 1) {                                                        // :105-110
    lo = 2 * lo - R[1].re;
    hi = 2 * hi - R[n - 2].re;
  }
  *E_min = lo;
  *E_max = hi;
  return QP_OK;
  QP_CATCH
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Row-partitioned cheby! with the exchange inside the library (RCCL)
// ---------------------------------------------------------------------------
namespace {

// the entry points of librccl that this file uses, resolved at run time from the library
// the caller names (so that it is the same RCCL the rest of the process uses)
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl_load(const char* path, RcclApi* api) {
  if (!path || !*path) return qp::fail(QP_E_BAD_ARG, "path of librccl.so is empty");
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return qp::fail(QP_E_INTERNAL, "dlopen(%s) failed: %s", path, dlerror());
  api->handle = h;
  api->GetUniqueId = reinterpret_cast<decltype(api->GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api->CommInitRank = reinterpret_cast<decltype(api->CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api->CommDestroy = reinterpret_cast<decltype(api->CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api->AllGather = reinterpret_cast<decltype(api->AllGather)>(dlsym(h, "ncclAllGather"));
  api->Send = reinterpret_cast<decltype(api->Send)>(dlsym(h, "ncclSend"));
  api->Recv = reinterpret_cast<decltype(api->Recv)>(dlsym(h, "ncclRecv"));
  api->GroupStart = reinterpret_cast<decltype(api->GroupStart)>(dlsym(h, "ncclGroupStart"));
  api->GroupEnd = reinterpret_cast<decltype(api->GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  api->GetErrorString = reinterpret_cast<decltype(api->GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!api->GetUniqueId || !api->CommInitRank || !api->CommDestroy || !api->AllGather || !api->Send || !api->Recv ||
      !api->GroupStart || !api->GroupEnd || !api->GetErrorString)
    return qp::fail(QP_E_INTERNAL, "%s does not export the RCCL entry points", path);
  return QP_OK;
}

#define QP_RCCL(api, expr)                                                                             \
  do {                                                                                                 \
    ncclResult_t r__ = (expr);                                                                         \
    if (r__ != ncclSuccess) return qp::fail(QP_E_INTERNAL, "RCCL: %s failed: %s", #expr, (api).GetErrorString(r__)); \
  } while (0)

__global__ __launch_bounds__(qp::kThreads) void pack_rows_kernel(double2* __restrict__ slab,
                                                                 const double2* __restrict__ x,
                                                                 const int64_t* __restrict__ rows, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * qp::kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * qp::kThreads)
    slab[i] = x[rows[i]];
}

}  // namespace

struct qp_comm {
  qp_ctx* ctx = nullptr;
  RcclApi api;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

struct qp_sharded_cheby {
  qp_sharded_cheby_desc d;
  qp_ctx* ctx = nullptr;
  int64_t nloc = 0;
  int64_t* send_rows_dev = nullptr;   // M entries, padded with row 0
  hipStream_t side = nullptr;         // high priority: boundary blocks + collectives
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
  std::vector<int> send_to, recv_from;
  bool p2p = false;
};

extern "C" {

int qp_comm_unique_id(const char* rccl_lib_path, char id_out[128]) {
  QP_TRY
  if (!id_out) return qp::fail(QP_E_BAD_ARG, "qp_comm_unique_id: NULL output");
  RcclApi api;
  QP_CHECK(rccl_load(rccl_lib_path, &api));
  ncclUniqueId id;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  QP_RCCL(api, api.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return QP_OK;
  QP_CATCH
}

int qp_comm_create(qp_ctx* ctx, const char* rccl_lib_path, const char id[128], int rank, int world, qp_comm** out) {
  QP_TRY
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return qp::fail(QP_E_BAD_ARG, "qp_comm_create: bad arguments");
  QP_CHECK(use(ctx));
  auto c = std::make_unique<qp_comm>();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  QP_CHECK(rccl_load(rccl_lib_path, &c->api));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  QP_RCCL(c->api, c->api.CommInitRank(&c->comm, world, uid, rank));
  *out = c.release();
  return QP_OK;
  QP_CATCH
}

int qp_comm_destroy(qp_comm* comm) {
  QP_TRY
  if (!comm) return QP_OK;
  (void)hipSetDevice(comm->ctx->device);
  (void)hipDeviceSynchronize();
  if (comm->comm) (void)comm->api.CommDestroy(comm->comm);
  delete comm;
  return QP_OK;
  QP_CATCH
}

int qp_comm_allgather(qp_comm* comm, const qp_state* send, qp_state* recv, int64_t count, void* stream) {
  QP_TRY
  if (!comm || !send || !recv || count < 0 || send->n < count || recv->n < count * comm->world)
    return qp::fail(QP_E_BAD_ARG, "qp_comm_allgather: bad arguments");
  QP_CHECK(use(comm->ctx));
  hipStream_t s = stream ? (hipStream_t)stream : comm->ctx->stream;
  QP_RCCL(comm->api, comm->api.AllGather(send->d, recv->d, (size_t)(2 * count), ncclDouble, comm->comm, s));
  return QP_OK;
  QP_CATCH
}

int qp_sharded_cheby_create(const qp_sharded_cheby_desc* desc, qp_sharded_cheby** out) {
  QP_TRY
  if (!desc || !out || !desc->op || !desc->X0 || !desc->X1 || !desc->acc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: NULL argument");
  const qp_sharded_cheby_desc& d = *desc;
  const int64_t nloc = d.op->A.nrows, ncols = d.op->A.ncols;
  const int world = d.comm ? d.comm->world : 1;
  if (d.M < 0 || d.nsend < 0 || d.nsend > d.M || (d.nsend > 0 && !d.send_rows)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: bad send set");
  if (d.M > 0 && !d.comm) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: an exchange needs a communicator");
  if (ncols != nloc + (d.M > 0 ? (int64_t)world * d.M : 0)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: operator has %lld columns, expected nloc + world * M = %lld", (long long)ncols, (long long)(nloc + (int64_t)world * d.M));
  if (d.X0->n != ncols || d.X1->n != ncols || d.acc->n != nloc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: vector length mismatch");
  if (d.M > 0 && !d.direct_send && (!d.slab || d.slab->n < d.M)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: slab too small");
  if (d.direct_send && d.M != nloc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: direct_send needs M == nloc");
  if (d.split && (d.split->op != d.op || d.direct_send || d.M == 0)) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: split does not fit this exchange");
  for (int64_t i = 0; i < d.nsend; ++i)
    if (d.send_rows[i] < 0 || d.send_rows[i] >= nloc) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: send row out of range");
  const bool p2p = d.M > 0 && d.n_send_to >= 0;
  if (p2p) {
    if (d.n_recv_from < 0 || (d.n_send_to > 0 && !d.send_to) || (d.n_recv_from > 0 && !d.recv_from))
      return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: bad neighbour lists");
    for (int i = 0; i < d.n_send_to; ++i)
      if (d.send_to[i] < 0 || d.send_to[i] >= world) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: send_to rank out of range");
    for (int i = 0; i < d.n_recv_from; ++i)
      if (d.recv_from[i] < 0 || d.recv_from[i] >= world) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_create: recv_from rank out of range");
  }
  qp_ctx* ctx = d.op->ctx;
  QP_CHECK(use(ctx));
  auto s = std::make_unique<qp_sharded_cheby>();
  s->d = d;
  s->d.send_rows = nullptr;   // host arrays of the caller: not kept
  s->p2p = p2p;
  if (p2p) {
    s->send_to.assign(d.send_to, d.send_to + d.n_send_to);
    s->recv_from.assign(d.recv_from, d.recv_from + d.n_recv_from);
  }
  s->d.send_to = s->d.recv_from = nullptr;
  s->ctx = ctx;
  s->nloc = nloc;
  if (d.M > 0 && !d.direct_send) {
    std::vector<int64_t> rows((size_t)d.M, 0);   // padded with row 0 (never read by anyone)
    std::copy(d.send_rows, d.send_rows + d.nsend, rows.begin());
    QP_CHECK(dev_alloc(&s->send_rows_dev, (size_t)d.M));
    QP_HIP(hipMemcpy(s->send_rows_dev, rows.data(), rows.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  if (d.split) {
    int least = 0, greatest = 0;
    QP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    QP_HIP(hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, greatest));
    QP_HIP(hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming));
    QP_HIP(hipEventCreateWithFlags(&s->ev_side, hipEventDisableTiming));
  }
  *out = s.release();
  return QP_OK;
  QP_CATCH
}

int qp_sharded_cheby_destroy(qp_sharded_cheby* s) {
  QP_TRY
  if (!s) return QP_OK;
  (void)hipSetDevice(s->ctx->device);
  (void)hipDeviceSynchronize();
  if (s->send_rows_dev) (void)hipFree(s->send_rows_dev);
  if (s->side) (void)hipStreamDestroy(s->side);
  if (s->ev_main) (void)hipEventDestroy(s->ev_main);
  if (s->ev_side) (void)hipEventDestroy(s->ev_side);
  delete s;
  return QP_OK;
  QP_CATCH
}

int qp_sharded_cheby_step(qp_sharded_cheby* s, const double* a, int n_coeffs, double Delta, double E_min, double dt) {
  QP_TRY
  if (!s || !a) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_step: NULL argument");
  if (n_coeffs < 2) return qp::fail(QP_E_TOO_FEW_COEFFS, "Need at least 2 Chebychev coefficients");
  if (!(Delta > 0) || dt == 0.0) return qp::fail(QP_E_BAD_ARG, "qp_sharded_cheby_step: Delta must be positive, dt non-zero");
  const qp_sharded_cheby_desc& d = s->d;
  qp_ctx* ctx = s->ctx;
  QP_CHECK(use(ctx));
  const int64_t nloc = s->nloc;
  const int nterms = n_coeffs - 1;
  const double beta = (Delta / 2) + E_min;                               // src/cheby.jl:156
  cplx c = (dt > 0) ? cplx(0, -2.0) / Delta : cplx(0, 2.0) / Delta;      // :158-162
  const cplx phase = std::exp(cplx(0, -1) * beta * dt);                  // :211
  std::vector<qp_acc_defer> sched((size_t)nterms);
  acc_schedule(a, n_coeffs, qp::g_acc_defer != 0, sched.data());
  const bool exchanging = d.M > 0;
  const bool overlap = d.split != nullptr;
  hipStream_t S_c = ctx->stream;
  hipStream_t S_x = overlap ? s->side : S_c;   // the stream the collectives are ordered on
  qp_state* X[2] = {d.X0, d.X1};
  qp_state xloc[2] = {qp_state{ctx, d.X0->d, nloc, false}, qp_state{ctx, d.X1->d, nloc, false}};

  // fill the ghost slabs of X[k] with the other ranks' send rows of the same vector
  auto exchange = [&](int k, bool packed) -> int {
    if (!exchanging) return QP_OK;
    const double2* send = nullptr;
    if (d.direct_send) {
      send = X[k]->d;
    } else {
      send = d.slab->d;
      if (!packed) {
        const int grid = (int)std::min<int64_t>((d.M + qp::kThreads - 1) / qp::kThreads, 1024);
        hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(qp::kThreads), 0, S_x, d.slab->d, X[k]->d, s->send_rows_dev, d.M);
        QP_HIP(hipGetLastError());
        ctx->stats.n_launch++;
      }
    }
    const RcclApi& api = d.comm->api;
    if (s->p2p) {   // neighbour exchange: my slab to who reads it, their slabs into their ghost slots
      QP_RCCL(api, api.GroupStart());
      for (int o : s->recv_from)
        QP_RCCL(api, api.Recv(X[k]->d + nloc + (int64_t)o * d.M, (size_t)(2 * d.M), ncclDouble, o, d.comm->comm, S_x));
      for (int o : s->send_to) QP_RCCL(api, api.Send(send, (size_t)(2 * d.M), ncclDouble, o, d.comm->comm, S_x));
      QP_RCCL(api, api.GroupEnd());
    } else {
      QP_RCCL(api, api.AllGather(send, X[k]->d + nloc, (size_t)(2 * d.M), ncclDouble, d.comm->comm, S_x));
    }
    return QP_OK;
  };

  if (overlap) {   // join: the side stream starts after everything queued on the main stream
    QP_HIP(hipEventRecord(s->ev_main, S_c));
    QP_HIP(hipStreamWaitEvent(S_x, s->ev_main, 0));
  }
  QP_CHECK(exchange(0, false));
  bool updated = false;
  bool result_in_acc = true;
  for (int m = 1; m <= nterms; ++m) {
    const bool last = (m == nterms);
    const int xi = (m % 2 == 1) ? 0 : 1, oi = 1 - xi;
    const qp_acc_defer& df = sched[(size_t)m - 1];
    const qp_c128 cc{c.real(), c.imag()};
    const qp_c128 ph = last ? qp_c128{phase.real(), phase.imag()} : qp_c128{1.0, 0.0};
    const qp_state* v0 = (m == 1) ? nullptr : &xloc[oi];
    qp_state* vout = last ? nullptr : &xloc[oi];
    const qp_state* acc_in = (updated && !df.skip) ? d.acc : nullptr;
    // the state buffer X0 may be written only while it is not being gathered
    qp_state* out = (m > 1 && last && xi == 1) ? &xloc[0] : d.acc;
    if (!df.skip && m > 1) result_in_acc = (out == d.acc);
    const double a_prev = updated ? 0.0 : a[0];
    if (overlap) {
      QP_CHECK(qp_cheby_term_split(d.op, d.split, (void*)S_x, m == 1 ? 1 : 0, X[xi], 0, v0, vout, acc_in,
                                   df.skip ? nullptr : out, last ? nullptr : d.slab, cc, beta, a_prev, a[m], ph, &df));
    } else {
      QP_CHECK(qp_cheby_term(d.op, X[xi], 0, v0, vout, acc_in, df.skip ? nullptr : out, cc, beta, a_prev, a[m], ph, &df));
    }
    updated = updated || !df.skip;
    if (!last) QP_CHECK(exchange(oi, overlap));
    if (m == 1) c *= 2.0;                                                 // :184
  }
  if (overlap) {   // join back: later work on the main stream sees everything the side stream did
    QP_HIP(hipEventRecord(s->ev_side, S_x));
    QP_HIP(hipStreamWaitEvent(S_c, s->ev_side, 0));
  }
  if (result_in_acc)
    QP_HIP(hipMemcpyAsync(d.X0->d, d.acc->d, (size_t)nloc * sizeof(double2), hipMemcpyDeviceToDevice, S_c));
  ctx->stats.n_cheby_steps++;
  return QP_OK;
  QP_CATCH
}

}  // extern "C"
