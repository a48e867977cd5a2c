// C ABI of libqprop_hip.so: handles, the Operator lazy sum, BLAS-1, and the host-side
// drivers of cheby!, arnoldi!, newton!, ritzvals/specrange that enqueue the HIP kernels
// of kernels.hip.  Host logic follows the reference line by line (citations inline);
// device work is stream-ordered, with host synchronisation only where the reference
// algorithm needs a scalar on the host (once per Newton restart, once per Arnoldi call).
#include <mutex>
#include <thread>
#include <system_error>
#include <numeric>
#include <atomic>
#include <unordered_map>

#include "engine_host.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
namespace qp {
// the context knobs by name (qp_tuning_set / qp_ctx_tuning_set / _get)
int* tuning_field(Tuning& t, const char* key) {
  struct Entry {
    const char* name;
    int Tuning::*field;
  };
  static const Entry table[] = {
      {"rbcsr_variant", &Tuning::rbcsr_variant},
      {"arnoldi_mode", &Tuning::arnoldi_mode},   {"arnoldi_onepass", &Tuning::arnoldi_onepass},   {"split_mode", &Tuning::split_mode},
            {"arnoldi_fuse_dots", &Tuning::arnoldi_fuse_dots},   {"lattice_fill", &Tuning::lattice_fill},   {"sparse_controls", &Tuning::sparse_controls},
      {"liouville_fused_n", &Tuning::liouville_fused_n}, {"liouville_tile32_n", &Tuning::liouville_tile32_n}, {"liouville_tile32_min_n", &Tuning::liouville_tile32_min_n}, {"real_vals", &Tuning::real_vals},
      {"stencil", &Tuning::stencil}, {"block_map", &Tuning::block_map},             {"acc_defer", &Tuning::acc_defer},
      {"cheby_graph", &Tuning::cheby_graph},     {"small_nnz", &Tuning::small_nnz},
      {"roctx", &Tuning::roctx}, 
      {"colblock", &Tuning::colblock}, {"cb_log2w", &Tuning::cb_log2w},
      {"dense_auto", &Tuning::dense_auto},       {"dense_panel_mfma", &Tuning::dense_panel_mfma},
      {"newton_pipeline", &Tuning::newton_pipeline}, 
      {"spmm_nt", &Tuning::spmm_nt},             {"spmm_rows", &Tuning::spmm_rows},
      {"spmm_strip", &Tuning::spmm_strip},       {"spmm_rw", &Tuning::spmm_rw},
      {"hrb_walk", &Tuning::hrb_walk},           {"walk_waves", &Tuning::walk_waves},
      {"walk_min_blocks", &Tuning::walk_min_blocks}, {"walk_dbg", &Tuning::walk_dbg}, {"walk_nt", &Tuning::walk_nt}, {"value_dict", &Tuning::value_dict}, {"walk_pair", &Tuning::walk_pair}, {"split_spin_log2", &Tuning::split_spin_log2}, {"split_dbg", &Tuning::split_dbg},
  };
  for (const Entry& e : table)
    if (std::strcmp(e.name, key) == 0) return &(t.*(e.field));
  return nullptr;
}

int device_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();
    return 256;
  }
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    n = 256;
  }
  cached[dev].store(n, std::memory_order_relaxed);
  return n;
}

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
    case QP_E_RCCL: return "QP_E_RCCL";
    default: return "QP_E_UNKNOWN";
  }
}

int qp_version(void) { return 100; }

}  // extern "C"

// defaults that new contexts start from (qp_tuning_set); a context's own copy is qp_ctx::tun
static std::mutex g_tuning_mutex;
static qp::Tuning g_tuning_defaults;

// Settings that change RESULTS (walk_dbg bit 1: the edge blocks of the strip walk are skipped) or force a failure
// (split_dbg: the boundary launches of a split term do not signal, every waiting wavefront runs into the time-out) exist for
// measurements and for the time-out test only: a release build refuses them; `make dev` (-DQP_DEVELOPER) builds the flavour
// that takes them (lib/libqprop_hip_dev.so, loaded with QPROP_HIP_LIB).
static int tuning_value_allowed(const char* key, int value) {
#ifndef QP_DEVELOPER
  if ((std::strcmp(key, "walk_dbg") == 0 && (value & 2)) || (std::strcmp(key, "split_dbg") == 0 && value != 0))
    return qp::fail(QP_E_BAD_ARG, "%s = %d is a developer-build setting (csrc: make dev; it changes results or forces a time-out)", key, value);
#endif
  (void)key;
  (void)value;
  return QP_OK;
}

extern "C" {

/* 1 for the developer flavour of the library (-DQP_DEVELOPER), 0 for the release build */
int qp_developer_build(void) {
#ifdef QP_DEVELOPER
  return 1;
#else
  return 0;
#endif
}

int qp_tuning_set(const char* key, int value) {
  if (!key) return qp::fail(QP_E_BAD_ARG, "key is NULL");
  QP_CHECK(tuning_value_allowed(key, value));
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  int* f = qp::tuning_field(g_tuning_defaults, key);
  if (!f) return qp::fail(QP_E_BAD_ARG, "unknown tuning key %s", key);
  *f = value;
  return QP_OK;
}

int qp_ctx_tuning_set(qp_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return qp::fail(QP_E_BAD_ARG, "qp_ctx_tuning_set: NULL argument");
  QP_CHECK(tuning_value_allowed(key, value));
  int* f = qp::tuning_field(ctx->tun, key);
  if (!f) return qp::fail(QP_E_BAD_ARG, "unknown tuning key %s", key);
  *f = value;
  return QP_OK;
}

int qp_ctx_tuning_get(qp_ctx* ctx, const char* key, int* value_out) {
  if (!ctx || !key || !value_out) return qp::fail(QP_E_BAD_ARG, "qp_ctx_tuning_get: NULL argument");
  const int* f = qp::tuning_field(ctx->tun, key);
  if (!f) return qp::fail(QP_E_BAD_ARG, "unknown tuning key %s", key);
  *value_out = *f;
  return QP_OK;
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
  {
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    ctx->tun = g_tuning_defaults;
  }
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
  if (!ctx || ctx->closed) return QP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_part) (void)hipFree(ctx->d_part);
  if (ctx->h_part) (void)hipHostFree(ctx->h_part);
  for (int k = 0; k < 2; ++k) {
    if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
    if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    ctx->stage[k] = nullptr;
    ctx->stage_ev[k] = nullptr;
  }
  ctx->stage_bytes = 0;
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  // the record stays (a few dozen bytes): child handles destroyed later still find their device in it
  ctx->d_part = nullptr;
  ctx->h_part = nullptr;
  ctx->ev0 = ctx->ev1 = nullptr;
  ctx->stream = nullptr;      // what a late child destroy synchronises: the null stream
  ctx->own_stream = false;
  ctx->closed = true;
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
  if (nrows < 0 || ncols < 0 || nnz < 0 || ncols > INT32_MAX)   // columns are int32 on the device
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
  qp::HostVec<qp_c128> cv;
  const qp_c128* v128 = nullptr;
  if (val_dtype == QP_VAL_F64) {
    cv.resize(nnz);
    const double* r = static_cast<const double*>(vals);
    parallel_rows(nnz, [&](int64_t p0, int64_t p1) {
      for (int64_t p = p0; p < p1; ++p) cv[p] = qp_c128{r[p], 0.0};
    }, (int64_t)1 << 20);
    v128 = cv.data();
  } else {
    v128 = static_cast<const qp_c128*>(vals);
  }
  if (layout == QP_LAYOUT_CSC) {
    if (ptr[ncols] - index_base != nnz) return qp::fail(QP_E_BAD_ARG, "colptr[end] does not match nnz");
    int st = qp::csc_to_csr(nrows, ncols, ptr, idx, v128, index_base, m->rowptr.data(), m->col.data(),
                            reinterpret_cast<qp_c128*>(m->vals.data()));
    if (st != QP_OK)
      return qp::fail(st, "qp_matrix_create: colptr must start at the index base, be monotone and end at nnz, and every "
                          "row index must lie in [base, base + nrows)");
  } else if (layout == QP_LAYOUT_CSR) {
    if (ptr[nrows] - index_base != nnz) return qp::fail(QP_E_BAD_ARG, "rowptr[end] does not match nnz");
    if (ptr[0] != index_base) return qp::fail(QP_E_BAD_ARG, "rowptr[0] must equal the index base (%d)", index_base);
    // rows in chunks on the host threads (N = 2^24: 5 GB of index and value arrays; one thread took 1.5 s): the copy with its range
    // checks, then per row the canonical form -- columns ascending (stable: duplicates keep their order and are summed later)
    std::atomic<int64_t> bad_row{-1}, bad_col{-1};
    parallel_rows(nrows + 1, [&](int64_t r0, int64_t r1) {
      for (int64_t r = r0; r < r1; ++r) m->rowptr[r] = ptr[r] - index_base;
    });
    parallel_rows(nrows, [&](int64_t r0, int64_t r1) {
      for (int64_t r = r0; r < r1; ++r)
        if (m->rowptr[r + 1] < m->rowptr[r] || m->rowptr[r] < 0 || m->rowptr[r + 1] > nnz) {
          int64_t none = -1;
          bad_row.compare_exchange_strong(none, r);
          return;
        }
    });
    if (bad_row.load() >= 0) return qp::fail(QP_E_BAD_ARG, "rowptr not monotone at row %lld", (long long)bad_row.load());
    parallel_rows(nrows, [&](int64_t r0, int64_t r1) {
      std::vector<std::pair<int32_t, cplx>> tmp;
      for (int64_t r = r0; r < r1; ++r) {
        const int64_t a = m->rowptr[r], b = m->rowptr[r + 1];
        bool sorted = true;
        for (int64_t p = a; p < b; ++p) {
          const int64_t c = idx[p] - index_base;
          if (c < 0 || c >= ncols) {
            int64_t none = -1;
            bad_col.compare_exchange_strong(none, p);
            return;
          }
          m->col[p] = (int32_t)c;
          m->vals[p] = cplx(v128[p].re, v128[p].im);
          if (p > a && m->col[p] < m->col[p - 1]) sorted = false;
        }
        if (sorted) continue;
        tmp.resize((size_t)(b - a));
        for (int64_t p = a; p < b; ++p) tmp[(size_t)(p - a)] = {m->col[p], m->vals[p]};
        std::stable_sort(tmp.begin(), tmp.end(), [](auto& x, auto& y) { return x.first < y.first; });
        for (int64_t p = a; p < b; ++p) { m->col[p] = tmp[(size_t)(p - a)].first; m->vals[p] = tmp[(size_t)(p - a)].second; }
      }
    });
    if (bad_col.load() >= 0) return qp::fail(QP_E_BAD_ARG, "column index out of range at %lld", (long long)bad_col.load());
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
  if (op->support) (void)hipFree(op->support);
  if (op->support_vals) (void)hipFree(op->support_vals);
  if (op->base) (void)hipFree(op->base);
  op->support = nullptr;
  op->support_vals = nullptr;
  op->base = nullptr;
  op->sparse_from = -1;
  op->n_support = 0;
  op->base_valid = false;
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
  if (op->walk.edge_map) (void)hipFree(op->walk.edge_map);
  op->walk = qp::WalkPlan();
  if (op->walk2.edge_map) (void)hipFree(op->walk2.edge_map);
  op->walk2 = qp::WalkPlan();
  op->A.walk = nullptr;
  if (op->cb.segptr) (void)hipFree(op->cb.segptr);
  if (op->cb.rowoff) (void)hipFree(op->cb.rowoff);
  if (op->cb.cols) (void)hipFree(op->cb.cols);
  if (op->cb.map) (void)hipFree(op->cb.map);
  if (op->cb.vals) (void)hipFree(op->cb.vals);
  if (op->cb.vals_r) (void)hipFree(op->cb.vals_r);
  op->cb = qp::ColBlockPlan();
  op->A.cb = nullptr;
  for (auto p : op->cv_tplanes) (void)hipFree(p);
  op->cv_tplanes.clear();
  if (op->cv_tplanes_dev) (void)hipFree(op->cv_tplanes_dev);
  if (op->cv_tab_comb) (void)hipFree(op->cv_tab_comb);
  if (op->cv.codes) (void)hipFree(op->cv.codes);
  if (op->cv.tptr) (void)hipFree(op->cv.tptr);
  if (op->cv.tab_r) (void)hipFree(op->cv.tab_r);
  op->cv_tplanes_dev = nullptr;
  op->cv_tab_comb = nullptr;
  op->cv = qp::CodedVals();
  op->A.cv = nullptr;
  if (op->m_rowptr) (void)hipFree(op->m_rowptr);
  if (op->m_cols) (void)hipFree(op->m_cols);
  if (op->m_map) (void)hipFree(op->m_map);
  if (op->m_vals) (void)hipFree(op->m_vals);
  if (op->m_order) (void)hipFree(op->m_order);
  op->m_order = nullptr;
  op->m_order_valid = false;
  if (op->m_tiles.tiles) (void)hipFree(op->m_tiles.tiles);
  if (op->m_tiles.rest) (void)hipFree(op->m_tiles.rest);
  if (op->m_tiles.tab) (void)hipFree(op->m_tiles.tab);
  op->m_tiles = qp::SpmmTiles();
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
  if (op->mf_free) op->mf_free(op);
  operator_free_device(op);
  delete op;
  return QP_OK;
}

// is this canonical CSR exactly Hermitian (bitwise conj-symmetric values, symmetric
// pattern, real diagonal, strictly increasing columns)?
// Columns >= n (ghost columns of a row-partitioned operator in local numbering) are
// outside the square part and always carry their values.
// A term's values in union order: the term's own array (one canonical term: the union pattern IS the term's -- no 4 GB copy at
// N = 2^24) or an array of its own.
struct PlaneView {
  qp::HostVec<cplx> own;
  const cplx* p = nullptr;
  size_t n = 0;
  PlaneView() = default;
  PlaneView(PlaneView&&) = default;
  PlaneView& operator=(PlaneView&&) = default;
  PlaneView(const PlaneView&) = delete;
  PlaneView& operator=(const PlaneView&) = delete;
  const cplx& operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
  const cplx* begin() const { return p; }
  const cplx* end() const { return p + n; }
  void borrow(const qp::HostVec<cplx>& v) {
    qp::HostVec<cplx>().swap(own);
    p = v.data();
    n = v.size();
  }
  qp::HostVec<cplx>& make_own(size_t count) {      // zeros, written by the host threads
    own.resize(count);
    cplx* o = own.data();
    parallel_rows((int64_t)count, [o](int64_t a, int64_t b) { std::fill(o + a, o + b, cplx(0.0)); }, (int64_t)1 << 20);
    p = own.data();
    n = count;
    return own;
  }
  void clear() {
    qp::HostVec<cplx>().swap(own);
    p = nullptr;
    n = 0;
  }
};
using Planes = std::vector<PlaneView>;

static bool csr_is_hermitian(int64_t n, const qp::HostVec<int64_t>& rp, const qp::HostVec<int32_t>& col,
                             const PlaneView& vals) {
  // Every row on its own (rows in chunks on a few host threads): columns strictly ascending, a real diagonal, and for every
  // lower entry (r, c), c < r, the upper entry (c, r) with the conjugate value -- found by bisection in row c (rows are short);
  // as many lower entries as upper ones inside the square part then says that no upper entry lacks its partner.
  std::atomic<bool> ok{true};
  std::atomic<int64_t> nlower{0}, nupper{0};
  parallel_rows(n, [&](int64_t r_begin, int64_t r_end) {
    int64_t lo = 0, up = 0;
    for (int64_t r = r_begin; r < r_end && ok.load(std::memory_order_relaxed); ++r) {
      for (int64_t p = rp[r]; p < rp[r + 1]; ++p) {
        const int64_t c = col[p];
        bool good = !(p > rp[r] && col[p - 1] >= c);
        if (good && c == r) {
          good = vals[p].imag() == 0.0;
        } else if (good && c > r) {
          if (c < n) ++up;
        } else if (good) {
          ++lo;
          const int32_t* b = col.data() + rp[c];
          const int32_t* e = col.data() + rp[c + 1];
          const int32_t* q = std::lower_bound(b, e, (int32_t)r);
          good = q != e && *q == (int32_t)r;
          if (good) {
            const cplx t = vals[(size_t)(q - col.data())];
            good = t.real() == vals[p].real() && t.imag() == -vals[p].imag();
          }
        }
        if (!good) {
          ok.store(false, std::memory_order_relaxed);
          return;
        }
      }
    }
    nlower.fetch_add(lo, std::memory_order_relaxed);
    nupper.fetch_add(up, std::memory_order_relaxed);
  });
  return ok.load() && nlower.load() == nupper.load();
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
static_assert(sizeof(LowerStencilSlot) == 32, "layout shared with kernel_common.h");

// mode of a block's column section (low two bits of its meta word, the rest is the byte offset)
enum { kColInt32 = 0, kColInt16 = 1, kColStencil = 2, kColBlockMap = 3 };
constexpr size_t kBlockMapQuad = 16 + 4 * (size_t)kRB;   // bytes per quad of a block-map section: four column blocks + four lane bytes per row

// `special(b, w, out)`: a chance to emit a block in the stencil encoding (returns true and
// appends its bytes) before the per-entry encodings are tried.
// (blocks [b0, b1) into `bytes`, which starts empty: the offsets in meta[b] are relative to it)
template <class GetCol, class Special>
static void encode_col_sections_range(int64_t nrows, int64_t ncols, int64_t b0, int64_t b1, const std::vector<int64_t>& ptr, GetCol& get,
                                      Special& special, std::vector<char>& bytes, std::vector<int64_t>& meta, bool allow_block_map) {
  // (pad entries multiply a zero value with x[column]: the column must exist.  A TALL operator -- fewer columns than rows --
  // has rows beyond its last column: a pad takes min(row, ncols - 1), never the row itself.)
  const int64_t last_col = std::max<int64_t>(ncols - 1, 0);
  for (int64_t b = b0; b < b1; ++b) {
    const int64_t w = (ptr[b + 1] - ptr[b]) / kRB;
    while (bytes.size() % 32) bytes.push_back(0);
    const size_t start = bytes.size();
    if (w > 0 && special(b, w, bytes)) {
      meta[b] = ((int64_t)start << 2) | kColStencil;
      continue;
    }
    // Block map: every slot sends the 64 rows of the block into ONE 64-aligned block of columns (any lane to any lane of it)
    // -- the structure of qubit-register Hamiltonians, where a Pauli string couples row and row XOR mask: 64-row blocks map
    // onto 64-row blocks, but the distance is +2^i or -2^i by the row's own bit, so no block-wide distance exists.  Per quad of
    // slots: four column-block numbers for the whole block (a wave-uniform load) + one byte per row and slot (the lane inside
    // the column block): 1.06 bytes of index traffic per entry instead of 4 (transverse-field Ising chain of 20 spins:
    // 101 -> 27 MB of index bytes per term).
    if (allow_block_map && w > 0 && (w % 4) == 0) {
      std::vector<int64_t> cb((size_t)w, -1);
      bool okmap = true;
      for (int64_t l = 0; l < kRB && okmap; ++l) {
        const int64_t r = b * kRB + l;
        if (r >= nrows) break;
        for (int64_t k = 0; k < w; ++k) {
          bool pad = false;
          const int64_t c = get(r, k, &pad);
          if (pad) continue;
          if (cb[(size_t)k] < 0) cb[(size_t)k] = c >> 6;
          else if (cb[(size_t)k] != (c >> 6)) { okmap = false; break; }
        }
      }
      if (okmap) {
        const int64_t own = std::min(std::min(b, (nrows - 1) >> 6), last_col >> 6);
        for (int64_t k = 0; k < w; ++k)
          if (cb[(size_t)k] < 0) cb[(size_t)k] = own;          // a slot of pure padding: any valid column will do
        meta[b] = ((int64_t)bytes.size() << 2) | kColBlockMap;
        const size_t off = bytes.size();
        bytes.resize(off + (size_t)(w / 4) * kBlockMapQuad, 0);
        for (int64_t k = 0; k < w; ++k) {
          const int32_t c32 = (int32_t)cb[(size_t)k];
          std::memcpy(&bytes[off + (size_t)(k >> 2) * kBlockMapQuad + (size_t)(k & 3) * 4], &c32, 4);
        }
        for (int64_t l = 0; l < kRB; ++l) {
          const int64_t r = b * kRB + l;
          for (int64_t k = 0; k < w; ++k) {
            bool pad = (r >= nrows);
            const int64_t c = pad ? 0 : get(r, k, &pad);
            // pad entries (value 0) and the lanes beyond the last row: lane 0 of the slot's column block (a real column: the
            // block holds a real entry of this slot, or it is the row block itself)
            bytes[off + (size_t)(k >> 2) * kBlockMapQuad + 16 + (size_t)l * 4 + (size_t)(k & 3)] = pad ? (char)0 : (char)(c & 63);
          }
        }
        continue;
      }
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
      if (r - std::min(r, last_col) > 32767) ok16 = false;   // (a pad of this row could not be encoded as a distance)
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
        int64_t c = pad ? std::min(rc, last_col) : get(r, k, &pad);
        if (pad && ok16) c = std::min(rc, last_col);
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

// All blocks, in chunks on a few host threads (every block's bytes depend on that block alone; a block starts on a 32-byte boundary,
// so the chunks concatenate -- each padded to that boundary -- into exactly the bytes a single pass writes).
template <class GetCol, class Special>
static void encode_col_sections(int64_t nrows, int64_t ncols, int64_t nblocks, const std::vector<int64_t>& ptr, GetCol get,
                                Special special, std::vector<char>& bytes, std::vector<int64_t>& meta, bool allow_block_map = true) {
  meta.assign((size_t)nblocks, 0);
  bytes.clear();
  const unsigned T = (nblocks >= 4096) ? host_threads() : 1u;
  if (T <= 1) {
    encode_col_sections_range(nrows, ncols, 0, nblocks, ptr, get, special, bytes, meta, allow_block_map);
    return;
  }
  std::vector<std::vector<char>> part((size_t)T);
  const int64_t chunk = (nblocks + T - 1) / T;
  parallel_rows((int64_t)T, [&](int64_t t0, int64_t t1) {
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t c0 = std::min(nblocks, t * chunk), c1 = std::min(nblocks, (t + 1) * chunk);
      encode_col_sections_range(nrows, ncols, c0, c1, ptr, get, special, part[(size_t)t], meta, allow_block_map);
    }
  }, 0);
  for (unsigned t = 0; t < T; ++t) {
    while (bytes.size() % 32) bytes.push_back(0);
    const int64_t base = (int64_t)bytes.size();
    const int64_t c0 = std::min<int64_t>(nblocks, (int64_t)t * chunk), c1 = std::min<int64_t>(nblocks, (int64_t)(t + 1) * chunk);
    for (int64_t bb = c0; bb < c1; ++bb) meta[(size_t)bb] += base << 2;
    bytes.insert(bytes.end(), part[(size_t)t].begin(), part[(size_t)t].end());
    std::vector<char>().swap(part[(size_t)t]);
  }
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
  if (mode == kColBlockMap) {
    int32_t cb;
    std::memcpy(&cb, &bytes[off + (size_t)(k >> 2) * kBlockMapQuad + (size_t)(k & 3) * 4], 4);
    const unsigned char ln = (unsigned char)bytes[off + (size_t)(k >> 2) * kBlockMapQuad + 16 + (size_t)(r % kRB) * 4 + (size_t)(k & 3)];
    return ((int64_t)cb << 6) | (int64_t)ln;
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
static int operator_build_device_impl(qp_operator* op, int format, const Planes& planes_csr) {
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
  BuildTrace trace;

  if (qp::csr_layout(format)) {   // QP_FMT_DENSE: the same arrays with a complete pattern (vals[r ncols + c])
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
      parallel_rows(nrows, [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
          const int32_t* b = uc.data() + ur[r];
          const int32_t* e = uc.data() + ur[r + 1];
          Lh.nlow[r] = (int32_t)(std::lower_bound(b, e, (int32_t)r) - b);
        }
      });
    }
    // widths per block on a few threads (into the pointer arrays), then the running sums
    parallel_rows(A.nblocks, [&](int64_t b0, int64_t b1) {
      for (int64_t b = b0; b < b1; ++b) {
        int64_t wu = 0, wl = 0;
        for (int64_t r = b * kRB; r < std::min(nrows, (b + 1) * kRB); ++r) {
          const int64_t len = ur[r + 1] - ur[r];
          const int64_t nl = hrb ? Lh.nlow[r] : 0;
          wu = std::max(wu, len - nl);
          wl = std::max(wl, nl);
        }
        Lh.bptr[b + 1] = ((wu + 3) & ~(int64_t)3) * kRB;
        if (hrb) Lh.lptr[b + 1] = ((wl + 3) & ~(int64_t)3) * kRB;
      }
    }, 1024);
    for (int64_t b = 0; b < A.nblocks; ++b) {
      Lh.bptr[b + 1] += Lh.bptr[b];
      if (hrb) Lh.lptr[b + 1] += Lh.lptr[b];
    }
    Lh.stored = Lh.bptr[A.nblocks] + kRB;   // + one block of slack: padded lower entries read vals[0..63]
    Lh.lstored = hrb ? Lh.lptr[A.nblocks] : 0;
    A.stored = Lh.stored;
    A.lstored = Lh.lstored;
    trace.mark("  block widths and pointers");
    // upper (or full) column indices
    std::vector<char> cbytes;
    {
      auto get_upper = [&](int64_t r, int64_t k, bool* pad) -> int64_t {
        const int64_t nl = hrb ? Lh.nlow[r] : 0;
        const int64_t len = ur[r + 1] - ur[r] - nl;
        if (k < len) return uc[ur[r] + nl + k];
        *pad = true;
        return (ur[r + 1] > ur[r]) ? uc[ur[r]] : 0;
      };
      encode_col_sections(nrows, A.ncols, A.nblocks, Lh.bptr, get_upper,
                          [&](int64_t b, int64_t w, std::vector<char>& out) {
                            return ctx->tun.stencil != 0 && try_stencil_upper(nrows, A.ncols, b, w, get_upper, out);
                          },
                          cbytes, Lh.cmeta, ctx->tun.block_map != 0);
      A.colbytes = (int64_t)cbytes.size();
      QP_CHECK(dev_alloc(reinterpret_cast<char**>(&A.cols), cbytes.size()));
      QP_HIP(hipMemcpy(A.cols, cbytes.data(), cbytes.size(), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(&A.cmeta, Lh.cmeta.size()));
      QP_HIP(hipMemcpy(A.cmeta, Lh.cmeta.data(), Lh.cmeta.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    QP_CHECK(dev_alloc(&A.bptr, Lh.bptr.size()));
    QP_HIP(hipMemcpy(A.bptr, Lh.bptr.data(), Lh.bptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    trace.mark("  upper column sections + upload");
    if (hrb) {
      // lower section: (column, position of the conj-transposed value in the upper section)
      qp::HostVec<int32_t> lpos;      // (half a gigabyte at N = 2^24: filled on the host threads, not by one)
      lpos.resize((size_t)std::max<int64_t>(A.lstored, 1));
      parallel_rows((int64_t)lpos.size(), [&](int64_t a, int64_t b) { std::fill(lpos.begin() + a, lpos.begin() + b, (int32_t)-1); }, (int64_t)1 << 20);
      if (Lh.stored >= (int64_t)INT32_MAX) return qp::fail(QP_E_BAD_ARG, "Hermitian-packed format needs < 2^31 stored values per GPU");
      parallel_rows(nrows, [&](int64_t r_begin, int64_t r_end) {
        for (int64_t r = r_begin; r < r_end; ++r) {
          const int64_t nl = Lh.nlow[r];
          for (int64_t k = 0; k < nl; ++k) {
            const int64_t c = uc[ur[r] + k];
            const int32_t* b = uc.data() + ur[c];
            const int32_t* e = uc.data() + ur[c + 1];
            const int64_t kk = (std::lower_bound(b, e, (int32_t)r) - b) - Lh.nlow[c];  // index of (c,r) among row c's upper entries
            lpos[rb_quad_pos(Lh.lptr, r, k)] = (int32_t)rb_val_pos(Lh.bptr, c, kk);
          }
        }
      });
      std::vector<char> lbytes;
      // stencil lower block: every row has a real entry in every slot, at a block-wide
      // distance delta_k, and the conj-transposed values sit at one slot per column block
      // (at most two column blocks per slot): position = pb(column block) + column % 64
      auto try_stencil_lower = [&](int64_t b, int64_t w, std::vector<char>& out) -> bool {
        if (ctx->tun.stencil == 0) return false;
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
      encode_col_sections(nrows, A.ncols, A.nblocks, Lh.lptr,
                          [&](int64_t r, int64_t k, bool* pad) -> int64_t {
                            if (k < Lh.nlow[r]) return uc[ur[r] + k];
                            *pad = true;          // padded: any valid column, value masked by pos < 0
                            return r;
                          },
                          try_stencil_lower, lbytes, Lh.lcmeta, ctx->tun.block_map != 0);
      A.lcolbytes = (int64_t)lbytes.size();
      QP_CHECK(dev_alloc(&A.lptr, Lh.lptr.size()));
      QP_HIP(hipMemcpy(A.lptr, Lh.lptr.data(), Lh.lptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(reinterpret_cast<char**>(&A.lcols), std::max<size_t>(lbytes.size(), 16)));
      if (!lbytes.empty()) QP_HIP(hipMemcpy(A.lcols, lbytes.data(), lbytes.size(), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(&A.lcmeta, Lh.lcmeta.size()));
      QP_HIP(hipMemcpy(A.lcmeta, Lh.lcmeta.data(), Lh.lcmeta.size() * sizeof(int64_t), hipMemcpyHostToDevice));
      QP_CHECK(dev_alloc(&A.lpos, lpos.size()));
      QP_HIP(hipMemcpy(A.lpos, lpos.data(), lpos.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      trace.mark("  lower sections (positions, stencils) + upload");
      QP_CHECK(build_walk_plan(op));
      trace.mark("  strip-walk plan");
    }
  }

  // ---- value planes ----
  {
    std::atomic<bool> all_real{true};
    for (const auto& pv : planes_csr)
      parallel_rows((int64_t)pv.size(), [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1 && all_real.load(std::memory_order_relaxed); ++i)
          if (pv[(size_t)i].imag() != 0.0) all_real.store(false, std::memory_order_relaxed);
      }, (int64_t)1 << 20);
    op->planes_real = all_real.load();
  }
  const size_t hplane_n = (size_t)std::max<int64_t>(A.stored, 1);
  // positions a control term touches (kept while it may still belong to the sparse suffix, see qp_operator::sparse_from)
  const int drift_planes = nops - op->ncoeffs;
  std::vector<std::vector<std::pair<int32_t, cplx>>> touched((size_t)nops);
  std::vector<char> is_sparse((size_t)nops, 0);
  const bool sparse_candidates = nops >= 2 && op->ncoeffs >= 1 && A.stored < (int64_t)INT32_MAX;
  // position of the first stored value of 64-row unit u (row blocks of the two row-block formats, 64 rows of a CSR layout): the
  // values of the rows of units [u0, u1) fill the positions [unit_pos(u0), unit_pos(u1)) and nothing else
  const int64_t nunits = (nrows + kRB - 1) / kRB;
  auto unit_pos = [&](int64_t u) -> int64_t {
    if (qp::csr_layout(format)) return ur[std::min(nrows, u * kRB)];
    return u >= nunits ? op->layout.bptr[(size_t)A.nblocks] : op->layout.bptr[(size_t)u];
  };
  auto scatter_rows = [&](const PlaneView& pv, int64_t r_begin, int64_t r_end, cplx* dst, int64_t dst_pos0) {
    for (int64_t r = r_begin; r < r_end; ++r) {
      const int64_t nl = (format == QP_FMT_HRB) ? op->layout.nlow[r] : 0;
      for (int64_t k = nl; k < ur[r + 1] - ur[r]; ++k) {
        const int64_t pos = qp::csr_layout(format) ? ur[r] + k : rb_val_pos(op->layout.bptr, r, k - nl);
        dst[pos - dst_pos0] = pv[ur[r] + k];
      }
    }
  };
  constexpr size_t kStageBytes = (size_t)64 << 20;
  std::unique_ptr<cplx, void (*)(void*)> hplane_buf(nullptr, std::free);
  cplx* hplane = nullptr;
  for (int l = 0; l < nops; ++l) {
    const auto& pv = planes_csr[l];
    double2* dp = nullptr;
    QP_CHECK(dev_alloc(&dp, (size_t)A.stored));
    op->planes.push_back(dp);
    const bool candidate = sparse_candidates && l >= drift_planes && l >= 1;      // (its positions are inspected on the host below)
    if (!candidate && (size_t)A.stored * sizeof(cplx) >= 4 * kStageBytes) {
      // a large plane: units in chunks of 64 MiB, written in device order into one of two pinned buffers by the host threads while
      // the other buffer's chunk is on its way to the device (one pageable 2.4 GB copy after a 2.4 GB fill took 0.42 s at N = 2^24)
      if (ctx->stage_bytes < kStageBytes) {
        for (int k = 0; k < 2; ++k) {
          if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
          ctx->stage[k] = nullptr;
          QP_HIP(hipHostMalloc(&ctx->stage[k], kStageBytes, hipHostMallocDefault));
          if (!ctx->stage_ev[k]) QP_HIP(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
        }
        ctx->stage_bytes = kStageBytes;
      }
      const int64_t cap = (int64_t)(kStageBytes / sizeof(cplx));
      int which = 0;
      bool used[2] = {false, false};
      for (int64_t u0 = 0; u0 < nunits;) {
        int64_t u1 = u0 + 1;
        while (u1 < nunits && unit_pos(u1 + 1) - unit_pos(u0) <= cap) ++u1;
        const int64_t p0 = unit_pos(u0), p1 = unit_pos(u1);
        if (p1 - p0 > cap) return qp::fail(QP_E_BAD_ARG, "operator build: a 64-row block of %lld stored values exceeds the staging buffer", (long long)(p1 - p0));
        cplx* buf = static_cast<cplx*>(ctx->stage[which]);
        if (used[which]) QP_HIP(hipEventSynchronize(ctx->stage_ev[which]));
        parallel_rows(p1 - p0, [&](int64_t a, int64_t b) { std::fill(buf + a, buf + b, cplx(0.0)); }, (int64_t)1 << 18);
        parallel_rows(std::min(nrows, u1 * kRB) - u0 * kRB, [&](int64_t a, int64_t b) { scatter_rows(pv, u0 * kRB + a, u0 * kRB + b, buf, p0); }, 1024);
        if (p1 > p0) QP_HIP(hipMemcpyAsync(dp + p0, buf, (size_t)(p1 - p0) * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
        QP_HIP(hipEventRecord(ctx->stage_ev[which], ctx->stream));
        used[which] = true;
        which ^= 1;
        u0 = u1;
      }
      const int64_t pend = unit_pos(nunits);      // (the slack behind the last block: zeros)
      if (A.stored > pend) QP_HIP(hipMemsetAsync(dp + pend, 0, (size_t)(A.stored - pend) * sizeof(double2), ctx->stream));
      QP_HIP(hipStreamSynchronize(ctx->stream));
      continue;
    }
    if (!hplane) {
      // (raw storage: a std::vector would zero its gigabytes serially before the threaded fill below does it again)
      hplane_buf.reset(static_cast<cplx*>(std::malloc(hplane_n * sizeof(cplx))));
      if (!hplane_buf) return qp::fail(QP_E_ALLOC, "out of host memory building the operator (%zu bytes)", hplane_n * sizeof(cplx));
      hplane = hplane_buf.get();
    }
    parallel_rows((int64_t)hplane_n, [&](int64_t a, int64_t b) { std::fill(hplane + a, hplane + b, cplx(0.0)); }, (int64_t)1 << 20);
    parallel_rows(nrows, [&](int64_t r_begin, int64_t r_end) { scatter_rows(pv, r_begin, r_end, hplane, 0); });
    if (candidate) {
      auto& t = touched[(size_t)l];
      const size_t limit = (size_t)(A.stored / 4);
      bool few = true;
      for (int64_t p = 0; p < A.stored && few; ++p)
        if (hplane[(size_t)p] != cplx(0.0)) {
          t.emplace_back((int32_t)p, hplane[(size_t)p]);
          few = t.size() <= limit;
        }
      if (few) is_sparse[(size_t)l] = 1;
      else t.clear(), t.shrink_to_fit();
    }
    QP_HIP(hipMemcpy(dp, hplane, (size_t)A.stored * sizeof(double2), hipMemcpyHostToDevice));
  }
  {
    int sfrom = nops;
    while (sfrom - 1 >= std::max(drift_planes, 1) && is_sparse[(size_t)(sfrom - 1)]) --sfrom;
    const int nsp = nops - sfrom;
    if (nsp >= 1 && nsp <= qp::kCoefBlock) {
      std::vector<int32_t> sup;
      for (int l = sfrom; l < nops; ++l)
        for (const auto& e : touched[(size_t)l]) sup.push_back(e.first);
      std::sort(sup.begin(), sup.end());
      sup.erase(std::unique(sup.begin(), sup.end()), sup.end());
      if (!sup.empty() && (int64_t)sup.size() <= A.stored / 4) {
        std::vector<cplx> sv((size_t)nsp * sup.size(), cplx(0.0));
        for (int l = sfrom; l < nops; ++l)
          for (const auto& e : touched[(size_t)l]) {
            const size_t i = (size_t)(std::lower_bound(sup.begin(), sup.end(), e.first) - sup.begin());
            sv[(size_t)(l - sfrom) * sup.size() + i] = e.second;
          }
        QP_CHECK(dev_alloc(&op->support, sup.size()));
        QP_HIP(hipMemcpy(op->support, sup.data(), sup.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        QP_CHECK(dev_alloc(&op->support_vals, sv.size()));
        QP_HIP(hipMemcpy(op->support_vals, sv.data(), sv.size() * sizeof(double2), hipMemcpyHostToDevice));
        op->sparse_from = sfrom;
        op->n_support = (int64_t)sup.size();
      }
    }
  }
  QP_CHECK(dev_alloc(&op->planes_dev, (size_t)nops));
  QP_HIP(hipMemcpy(op->planes_dev, op->planes.data(), nops * sizeof(double2*), hipMemcpyHostToDevice));
  trace.mark("  value planes in device order + upload");
  A.vals = op->planes[0];
  (void)ctx;
  return QP_OK;
}

// Value-dictionary mirror (device.h: CodedVals) of a row-block operator: per 64-row block the distinct tuples
// (value in term 0, .., value in term L - 1) over its stored positions (pads: the all-zero tuple), sorted bytewise; one code byte
// per stored position; tables with the same content shared.  Built when every block has at most 256 tuples and codes + tables
// come to less than half of the value plane the mat-vec would stream instead.  Host index work (bit patterns, no arithmetic).
static int build_coded_values(qp_operator* op, const Planes& planes_csr) {
  qp_ctx* ctx = op->ctx;
  DevMatrix& A = op->A;
  op->cv_reason = 0;
  if (ctx->tun.value_dict == 0) return op->cv_reason = 4, QP_OK;
  if (A.format != QP_FMT_RBCSR || A.stored <= 0 || A.nblocks <= 0) return op->cv_reason = 1, QP_OK;
  const auto& ur = op->u_rowptr;
  const HostLayout& Lh = op->layout;
  const int L = (int)planes_csr.size();
  const int64_t nrows = A.nrows, nblocks = A.nblocks;
  const size_t tb = sizeof(cplx) * (size_t)L;       // bytes of one tuple
  std::vector<uint8_t> codes((size_t)A.stored, 0);
  std::vector<std::string> tables((size_t)nblocks);  // block b's sorted distinct tuples, tb bytes each
  std::atomic<bool> too_many{false};
  parallel_rows(nblocks, [&](int64_t b0, int64_t b1) {
    std::vector<char> ent;          // the block's tuples, position-major (slot, lane)
    std::vector<int32_t> order, code_of;
    for (int64_t b = b0; b < b1 && !too_many.load(std::memory_order_relaxed); ++b) {
      const int64_t w = (Lh.bptr[b + 1] - Lh.bptr[b]) / kRB;
      const int64_t npos = w * kRB;
      ent.assign((size_t)npos * tb, 0);
      for (int64_t l = 0; l < kRB; ++l) {
        const int64_t r = b * kRB + l;
        if (r >= nrows) break;
        const int64_t len = ur[r + 1] - ur[r];
        for (int64_t k = 0; k < len; ++k)
          for (int t = 0; t < L; ++t)
            std::memcpy(&ent[(size_t)(k * kRB + l) * tb + (size_t)t * sizeof(cplx)], &planes_csr[(size_t)t][(size_t)(ur[r] + k)], sizeof(cplx));
      }
      // distinct tuples: sort the positions by tuple bytes, walk the runs
      order.resize((size_t)npos);
      for (int64_t i = 0; i < npos; ++i) order[(size_t)i] = (int32_t)i;
      std::sort(order.begin(), order.end(), [&](int32_t a, int32_t c) {
        return std::memcmp(&ent[(size_t)a * tb], &ent[(size_t)c * tb], tb) < 0;
      });
      code_of.assign((size_t)npos, 0);
      std::string& T = tables[(size_t)b];
      T.clear();
      int n = 0;
      bool fits = true;
      for (int64_t i = 0; i < npos; ++i) {
        const int32_t p = order[(size_t)i];
        if (i == 0 || std::memcmp(&ent[(size_t)p * tb], &ent[(size_t)order[(size_t)i - 1] * tb], tb) != 0) {
          if (n == 256) {
            fits = false;
            break;
          }
          T.append(&ent[(size_t)p * tb], tb);
          ++n;
        }
        code_of[(size_t)p] = n - 1;
      }
      if (!fits) {
        too_many.store(true, std::memory_order_relaxed);
        break;
      }
      // codes in the quad-packed layout of the column sections (rb_quad_pos): byte (k & 3) of dword (k >> 2) * 64 + lane
      for (int64_t k = 0; k < w; ++k)
        for (int64_t l = 0; l < kRB; ++l)
          codes[(size_t)(Lh.bptr[b] + (k >> 2) * (4 * kRB) + l * 4 + (k & 3))] = (uint8_t)code_of[(size_t)(k * kRB + l)];
    }
  }, 64);
  if (too_many.load()) return op->cv_reason = 2, QP_OK;
  // shared tables: first block with a content owns it
  std::unordered_map<std::string, int64_t> where;
  std::vector<int64_t> tptr((size_t)nblocks);
  std::string all;
  for (int64_t b = 0; b < nblocks; ++b) {
    auto it = where.find(tables[(size_t)b]);
    if (it == where.end()) {
      it = where.emplace(tables[(size_t)b], (int64_t)(all.size() / tb)).first;
      all += tables[(size_t)b];
    }
    tptr[(size_t)b] = (it->second << 9) | (int64_t)(tables[(size_t)b].size() / tb);   // first entry << 9 | entries (<= 256)
    std::string().swap(tables[(size_t)b]);
  }
  const int64_t ntab = (int64_t)(all.size() / tb);
  // what a term streams: a byte per stored position + (a share of) the tables, against 16 (8: real) bytes per position
  const double coded_bytes = (double)A.stored + 16.0 * (double)ntab, plain_bytes = (op->planes_real ? 8.0 : 16.0) * (double)A.stored;
  if (coded_bytes > 0.5 * plain_bytes) return op->cv_reason = 3, QP_OK;
  qp::CodedVals& C = op->cv;
  C.ntab = ntab;
  C.ntables = (int64_t)where.size();
  QP_CHECK(dev_alloc(&C.codes, codes.size()));
  QP_HIP(hipMemcpy(C.codes, codes.data(), codes.size(), hipMemcpyHostToDevice));
  QP_CHECK(dev_alloc(&C.tptr, tptr.size()));
  QP_HIP(hipMemcpy(C.tptr, tptr.data(), tptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  std::vector<cplx> col((size_t)ntab);
  for (int t = 0; t < L; ++t) {
    for (int64_t e = 0; e < ntab; ++e) std::memcpy(&col[(size_t)e], &all[(size_t)e * tb + (size_t)t * sizeof(cplx)], sizeof(cplx));
    double2* dp = nullptr;
    QP_CHECK(dev_alloc(&dp, (size_t)ntab));
    op->cv_tplanes.push_back(dp);
    QP_HIP(hipMemcpy(dp, col.data(), (size_t)ntab * sizeof(double2), hipMemcpyHostToDevice));
  }
  QP_CHECK(dev_alloc(&op->cv_tplanes_dev, (size_t)L));
  QP_HIP(hipMemcpy(op->cv_tplanes_dev, op->cv_tplanes.data(), (size_t)L * sizeof(double2*), hipMemcpyHostToDevice));
  C.tab = op->cv_tplanes[0];
  C.valid = 1;
  A.cv = &op->cv;
  return QP_OK;
}

// ... timed: format conversion, encoding and upload are host work at qp_operator_create (and once more if a complex
// coefficient forces a Hermitian-packed operator back to plain row blocks); qp_operator_build_info reports it
static int operator_build_device(qp_operator* op, int format, const Planes& planes_csr) {
  const auto t0 = std::chrono::steady_clock::now();
  BuildTrace trace;
  int rc = operator_build_device_impl(op, format, planes_csr);
  trace.mark("  (layout and value planes, with their scratch released)");
  if (rc == QP_OK) rc = build_colblock(op);
  trace.mark("  column-block plan");
  if (rc == QP_OK && !(op->cb.valid && op->ctx->tun.colblock != 0)) rc = build_coded_values(op, planes_csr);
  trace.mark("  value dictionary");
  op->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  op->build_ms_total += op->build_ms;
  op->n_builds++;
  return rc;
}

// current per-term values (device planes) back in union-CSR order
static int operator_download_planes(qp_operator* op, Planes& planes_csr) {
  const DevMatrix& A = op->A;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  const int64_t nnz = ur[A.nrows];
  QP_HIP(hipStreamSynchronize(op->ctx->stream));
  std::vector<cplx> hv((size_t)std::max<int64_t>(A.stored, 1));
  planes_csr.clear();
  planes_csr.resize(op->planes.size());
  for (size_t l = 0; l < op->planes.size(); ++l) {
    QP_HIP(hipMemcpy(hv.data(), op->planes[l], (size_t)A.stored * sizeof(double2), hipMemcpyDeviceToHost));
    auto& out = planes_csr[l].make_own((size_t)nnz);
    for (int64_t r = 0; r < A.nrows; ++r) {
      const int64_t nl = (A.format == QP_FMT_HRB) ? op->layout.nlow[r] : 0;
      for (int64_t k = 0; k < ur[r + 1] - ur[r]; ++k) {
        if (qp::csr_layout(A.format)) {
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
  if (format < QP_FMT_AUTO || (format > QP_FMT_HRB && format != QP_FMT_DENSE)) return qp::fail(QP_E_BAD_ARG, "bad device format %d", format);
  QP_CHECK(use(ctx));
  std::unique_ptr<qp_operator, int (*)(qp_operator*)> op(new qp_operator(), operator_free);
  op->ctx = ctx;
  op->A.tun = &ctx->tun;
  op->nops = nops;
  op->ncoeffs = ncoeffs;
  op->coeffs.assign(ncoeffs, cplx(1.0));
  const int64_t nrows = ops[0]->nrows, ncols = ops[0]->ncols;
  op->A.nrows = nrows;
  op->A.ncols = ncols;

  const auto t_create = std::chrono::steady_clock::now();
  BuildTrace trace;
  // ---- union sparsity pattern (sorted merge per row) ----
  auto& ur = op->u_rowptr;
  auto& uc = op->u_col;
  // (one term whose rows are strictly ascending IS the union pattern; a term with repeated or unsorted columns goes through the
  // merge like several terms do, so that density, completeness and every layout decision below see each position once --
  // a duplicate could make the stored count reach nrows x ncols with positions missing)
  bool canonical = nops == 1;
  if (canonical) {
    std::atomic<bool> asc{true};
    parallel_rows(nrows, [&](int64_t r0, int64_t r1) {
      for (int64_t r = r0; r < r1 && asc.load(std::memory_order_relaxed); ++r)
        for (int64_t p = ops[0]->rowptr[r] + 1; p < ops[0]->rowptr[r + 1]; ++p)
          if (ops[0]->col[p] <= ops[0]->col[p - 1]) {
            asc.store(false, std::memory_order_relaxed);
            break;
          }
    });
    canonical = asc.load();
  }
  if (canonical) {
    ur.resize(ops[0]->rowptr.size());
    uc.resize(ops[0]->col.size());
    parallel_copy(ur.data(), ops[0]->rowptr.data(), ur.size());
    parallel_copy(uc.data(), ops[0]->col.data(), uc.size());
  } else {
    ur.assign(nrows + 1, 0);
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
  trace.mark("union pattern");
  // ---- a dense generator (QP_FMT_DENSE): requested, or AUTO with at least kDenseMinDensityPct % of the positions stored.
  // The pattern is made complete (the missing positions become explicit zeros, counted like the lattice completion's), so
  // that the CSR-ordered value array IS the row-major dense matrix and the dense kernels need no index at all.
  bool dense = false;
  {
    const double positions = (double)nrows * (double)ncols;
    if (format == QP_FMT_DENSE) {
      if (positions > (double)INT32_MAX) return qp::fail(QP_E_BAD_ARG, "QP_FMT_DENSE: %lld x %lld positions exceed the 2^31 limit", (long long)nrows, (long long)ncols);
      dense = nrows > 0 && ncols > 0;
    } else if (format == QP_FMT_AUTO && ctx->tun.dense_auto && nrows > 0 && ncols > 0 && positions <= 1073741824.0) {
      dense = 100.0 * (double)ur[nrows] >= (double)qp::kDenseMinDensityPct * positions;
    }
  }
  if (dense) {
    const int64_t before = ur[nrows];
    if (before != nrows * ncols) {
      uc.resize((size_t)(nrows * ncols));
      parallel_rows(nrows, [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r)
          for (int64_t c = 0; c < ncols; ++c) uc[(size_t)(r * ncols + c)] = (int32_t)c;
      });
      for (int64_t r = 0; r <= nrows; ++r) ur[r] = r * ncols;
    }
    op->n_lattice_fill = ur[nrows] - before;
  }
  // Lattice completion (explicit zeros) is for operators that END UP Hermitian-packed with a strip-walk plan; whether this
  // one does is known only after the Hermitian check and the format choice below, which need the values.  So: complete
  // tentatively, keep the original pattern, and take the completion back if the operator turns out non-Hermitian, is laid
  // out otherwise, or has no plan after all (a non-Hermitian lattice-shaped Liouvillian kept up to 12 % stored
  // zeros for nothing -- more bytes per mat-vec, and 0 * Inf = NaN where the reference has no entry).
  qp::HostVec<int64_t> ur_orig;
  qp::HostVec<int32_t> uc_orig;
  if (!dense && (format == QP_FMT_AUTO || format == QP_FMT_HRB)) {
    const int64_t before = ur[nrows];
    lattice_fill(ctx->tun, nrows, ncols, ur, uc, &ur_orig, &uc_orig);      // (ur_orig / uc_orig stay empty when nothing was completed)
    op->n_lattice_fill = ur[nrows] - before;
    if (op->n_lattice_fill == 0) {
      ur_orig.clear();
      uc_orig.clear();
    }
  }

  trace.mark("lattice completion");
  // ---- per-term values in union order (duplicates within a row are summed, as Julia's sparse() does) ----
  Planes planes_csr((size_t)nops);
  auto scatter_planes = [&]() {
  op->A.nnz = ur[nrows];
  for (int l = 0; l < nops; ++l) {
    const qp_matrix* M = ops[l];
    auto& pv = planes_csr[l];
    auto same = [](const auto& a, const auto& b) {      // a == b, on the host threads (a gigabyte of columns at N = 2^24)
      if (a.size() != b.size()) return false;
      std::atomic<bool> eq{true};
      parallel_rows((int64_t)a.size(), [&](int64_t i0, int64_t i1) {
        if (i1 > i0 && std::memcmp(a.data() + i0, b.data() + i0, (size_t)(i1 - i0) * sizeof(a[0])) != 0) eq.store(false, std::memory_order_relaxed);
      }, (int64_t)1 << 20);
      return eq.load();
    };
    if (nops == 1 && canonical && (int64_t)M->vals.size() == op->A.nnz && same(ur, M->rowptr) && same(uc, M->col)) {
      // one canonical term and no completion: the union pattern IS the term's own (same columns, not merely as many) -- a plain copy
      pv.borrow(M->vals);      // (the term outlives this call: the operator build reads it, nothing keeps the pointer)
      continue;
    }
    auto& o = pv.make_own((size_t)op->A.nnz);
    parallel_rows(nrows, [&](int64_t r0, int64_t r1) {
      for (int64_t r = r0; r < r1; ++r) {
        int64_t k = 0;
        for (int64_t p = M->rowptr[r]; p < M->rowptr[r + 1]; ++p) {
          while (uc[ur[r] + k] != M->col[p]) ++k;
          o[ur[r] + k] += M->vals[p];
        }
      }
    });
  }
  };
  scatter_planes();
  trace.mark("value planes in union order");
  auto take_completion_back = [&]() {
    ur.swap(ur_orig);
    uc.swap(uc_orig);
    ur_orig.clear();
    uc_orig.clear();
    op->n_lattice_fill = 0;
    for (auto& pv : planes_csr) pv.clear();
    scatter_planes();
  };
  bool hermitian = !dense && (ncols >= nrows) && (format == QP_FMT_AUTO || format == QP_FMT_HRB);
  for (int l = 0; hermitian && l < nops; ++l) hermitian = csr_is_hermitian(nrows, ur, uc, planes_csr[l]);
  op->hermitian_planes = hermitian;
  trace.mark("Hermitian check");
  int fmt = dense ? (int)QP_FMT_DENSE : choose_format(op.get(), format, hermitian);
  trace.mark("format choice");
  if (fmt < 0) return qp::fail(QP_E_BAD_ARG, "QP_FMT_HRB requested but the operator terms are not exactly Hermitian");
  if (!ur_orig.empty() && fmt != QP_FMT_HRB) {   // (a lattice completion happened;) not Hermitian, or not packed: the zeros would buy nothing
    take_completion_back();
    fmt = choose_format(op.get(), format, hermitian);
    if (fmt < 0) return qp::fail(QP_E_BAD_ARG, "QP_FMT_HRB requested but the operator terms are not exactly Hermitian");
  }
  QP_CHECK(operator_build_device(op.get(), fmt, planes_csr));
  trace.mark("device build (layout, encodings, uploads, plans)");
  if (!ur_orig.empty() && !op->walk.valid) {     // packed, completed, and still no plan: build once more without the zeros
    QP_CHECK(operator_free_device(op.get()));
    take_completion_back();
    fmt = choose_format(op.get(), format, hermitian);
    if (fmt < 0) return qp::fail(QP_E_BAD_ARG, "QP_FMT_HRB requested but the operator terms are not exactly Hermitian");
    QP_CHECK(operator_build_device(op.get(), fmt, planes_csr));
  }
  planes_csr.clear();
  // the whole host side of the creation (union pattern, lattice completion, value planes, Hermitian check, format choice)
  // belongs to what qp_operator_build_info reports for the first build
  op->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
  op->build_ms_total = op->build_ms;
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
  if (op->A.format == QP_FMT_MATFREE) return op->mf_refresh(op);
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
    Planes planes_csr;
    QP_CHECK(operator_download_planes(op, planes_csr));
    operator_free_device(op);
    const int fmt = choose_format(op, QP_FMT_AUTO, false);
    QP_CHECK(operator_build_device(op, fmt, planes_csr));
    op->n_relayouts++;
  }
  // real terms with real coefficients: the mat-vec kernels stream a real copy (8 instead of 16
  // bytes per value); everything else keeps reading the complex array
  const bool want_real = op->ctx->tun.real_vals && op->planes_real && all_real && op->A.stored > 0;
  if (want_real && !op->real_vals) QP_CHECK(dev_alloc(&op->real_vals, (size_t)op->A.stored));
  if (op->sparse_from > 0 && ctx->tun.sparse_controls && !(op->nops == 1 && all_one)) {
    // only the sparse trailing control terms' positions are rewritten (qp_operator::sparse_from)
    const int sf = op->sparse_from;
    bool same = op->base_valid && op->base_real == want_real && (int)op->base_eff.size() == sf;
    for (int l = 0; same && l < sf; ++l) same = (eff[l].x == op->base_eff[(size_t)l].x && eff[l].y == op->base_eff[(size_t)l].y);
    if (!same) {
      if (!op->base) QP_CHECK(dev_alloc(&op->base, (size_t)op->A.stored));
      if (!op->combined) QP_CHECK(dev_alloc(&op->combined, (size_t)op->A.stored));
      QP_CHECK(qp::launch_combine_planes(ctx->stream, op->base, op->planes_dev, eff.data(), sf, op->A.stored, nullptr, &ctx->stats));
      QP_HIP(hipMemcpyAsync(op->combined, op->base, (size_t)op->A.stored * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
      op->base_eff.assign(eff.begin(), eff.begin() + sf);
      op->base_valid = true;
      op->base_real = want_real;
    }
    QP_CHECK(qp::launch_sparse_planes_update(ctx->stream, op->combined, op->base, op->support, op->n_support, op->support_vals,
                                             op->nops - sf, eff.data() + sf, (same && want_real) ? op->real_vals : nullptr,
                                             &ctx->stats));
    if (!same && want_real) QP_CHECK(qp::launch_real_part(ctx->stream, op->real_vals, op->combined, op->A.stored, &ctx->stats));
    op->A.vals = op->combined;
    op->real_of = nullptr;
    op->last_refresh_sparse = true;
  } else if (op->nops == 1 && all_one) {
    op->last_refresh_sparse = false;
    op->base_valid = false;   // (`combined` is not what the sparse update left: the next sparse update rebuilds its base)
    op->A.vals = op->planes[0];
    if (want_real && op->real_of != op->A.vals) {   // a plane never changes: extract once
      QP_CHECK(qp::launch_real_part(ctx->stream, op->real_vals, op->A.vals, op->A.stored, &ctx->stats));
      op->real_of = op->A.vals;
    }
  } else {
    // the full combination rewrites `combined` with THESE coefficients everywhere: whatever the sparse update's base was
    // built for no longer describes it (knob sparse_controls switched off and on again on a live operator)
    op->base_valid = false;
    op->last_refresh_sparse = false;
    if (!op->combined) QP_CHECK(dev_alloc(&op->combined, (size_t)op->A.stored));
    QP_CHECK(qp::launch_combine_planes(ctx->stream, op->combined, op->planes_dev, eff.data(), op->nops, op->A.stored,
                                       want_real ? op->real_vals : nullptr, &ctx->stats));
    op->A.vals = op->combined;
    op->real_of = nullptr;
  }
  op->A.vals_r = want_real ? op->real_vals : nullptr;
  if (op->cv.valid) {   // the value-dictionary mirror: the same combination on the table entries instead of the stored positions
    qp::CodedVals& C = op->cv;
    if (want_real && !C.tab_r) QP_CHECK(dev_alloc(&C.tab_r, (size_t)C.ntab));
    if (op->nops == 1 && all_one) {
      C.tab = op->cv_tplanes[0];
      if (want_real) QP_CHECK(qp::launch_real_part(ctx->stream, C.tab_r, C.tab, C.ntab, &ctx->stats));
    } else {
      if (!op->cv_tab_comb) QP_CHECK(dev_alloc(&op->cv_tab_comb, (size_t)C.ntab));
      QP_CHECK(qp::launch_combine_planes(ctx->stream, op->cv_tab_comb, op->cv_tplanes_dev, eff.data(), op->nops, C.ntab,
                                         want_real ? C.tab_r : nullptr, &ctx->stats));
      C.tab = op->cv_tab_comb;
    }
    C.use_real = want_real ? 1 : 0;
  }
  if (op->cb.valid) {   // the column-blocked mirror follows the values (one gather pass per evaluate!)
    if (want_real && !op->cb.vals_r) QP_CHECK(dev_alloc(&op->cb.vals_r, (size_t)op->cb.nnz));
    op->cb.use_real = want_real ? 1 : 0;
    QP_CHECK(qp::launch_colblock_gather(ctx->stream, op->cb, op->A.vals, &ctx->stats));
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
  if (qp::csr_layout(A.format) || A.format == QP_FMT_MATFREE) return QP_OK;
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

/* how the column sections of the row blocks are encoded: out[0..3] = upper (or only) sections as int32 columns / int16 distances
   / stencil (one distance per slot) / block map (one column block per slot + a byte per entry); out[4..7] = the same for the lower
   sections of a Hermitian-packed operator */
int qp_operator_encoding_info(const qp_operator* op, int64_t out[8]) {
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_encoding_info: NULL argument");
  for (int i = 0; i < 8; ++i) out[i] = 0;
  const DevMatrix& A = op->A;
  if (qp::csr_layout(A.format) || A.format == QP_FMT_MATFREE) return QP_OK;
  const HostLayout& Lh = op->layout;
  for (int64_t b = 0; b < A.nblocks; ++b) {
    out[(int)(Lh.cmeta[b] & 3)]++;
    if (A.format == QP_FMT_HRB) out[4 + (int)(Lh.lcmeta[b] & 3)]++;
  }
  return QP_OK;
}

/* the value-dictionary mirror (device.h: CodedVals): out[0] = 1 when the mat-vec reads it, out[1] = table entries, out[2] = distinct
   tables, out[3] = bytes a term streams for the values through it (codes + tables), out[4] = bytes of the value plane it replaces,
   out[5] = why there is none (0: there is one, 1: not a plain row-block operator, 2: a block with more than 256 distinct values,
   3: no saving, 4: knob value_dict off, 5: column-blocked mirror in use) */
int qp_operator_value_encoding_info(const qp_operator* op, int64_t out[6]) {
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_value_encoding_info: NULL argument");
  for (int i = 0; i < 6; ++i) out[i] = 0;
  const qp::CodedVals& C = op->cv;
  const bool on = C.valid && op->ctx->tun.value_dict != 0;
  out[0] = on ? 1 : 0;
  out[1] = C.ntab;
  out[2] = C.ntables;
  out[3] = C.valid ? op->A.stored + (C.use_real ? 8 : 16) * C.ntab : 0;
  out[4] = (op->A.vals_r ? 8 : 16) * op->A.stored;
  out[5] = on ? 0 : (C.valid ? 4 : (op->cb.valid ? 5 : (op->cv_reason ? op->cv_reason : 1)));
  return QP_OK;
}

int qp_operator_build_info(const qp_operator* op, double out[4]) {
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_build_info: NULL argument");
  out[0] = op->build_ms;
  out[1] = op->build_ms_total;
  out[2] = (double)op->n_relayouts;
  out[3] = (double)op->A.format;
  return QP_OK;
}

int qp_lattice_fill_host(int64_t nrows, int64_t ncols, const int64_t* rowptr, const int32_t* col, int min_blocks,
                         int64_t* rowptr_out, int32_t* col_out, int64_t cap, int64_t* nnz_out) {
  QP_TRY
  if (!rowptr || !col || !rowptr_out || !col_out || !nnz_out || nrows < 0 || ncols < 0 || rowptr[0] != 0)
    return qp::fail(QP_E_BAD_ARG, "qp_lattice_fill_host: bad arguments");
  qp::HostVec<int64_t> ur(rowptr, rowptr + nrows + 1);
  qp::HostVec<int32_t> uc(col, col + rowptr[nrows]);
  qp::Tuning tun;
  tun.lattice_fill = 1;
  tun.walk_min_blocks = min_blocks;
  lattice_fill(tun, nrows, ncols, ur, uc);
  *nnz_out = ur[nrows];
  if (ur[nrows] > cap) return qp::fail(QP_E_BAD_ARG, "qp_lattice_fill_host: col_out holds %lld entries, %lld needed", (long long)cap, (long long)ur[nrows]);
  std::memcpy(rowptr_out, ur.data(), ur.size() * sizeof(int64_t));
  std::memcpy(col_out, uc.data(), uc.size() * sizeof(int32_t));
  return QP_OK;
  QP_CATCH
}

int qp_operator_fill_info(const qp_operator* op, int64_t* n_filled) {
  if (!op || !n_filled) return qp::fail(QP_E_BAD_ARG, "qp_operator_fill_info: NULL argument");
  *n_filled = op->n_lattice_fill;
  return QP_OK;
}

int qp_operator_walk_info(const qp_operator* op, int64_t out[8]) {
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_walk_info: NULL argument");
  const qp::WalkPlan& P = op->walk;
  const qp::Tuning& tun = op->ctx->tun;
  // "has one" = the fused term of a whole-operator cheby! takes the walk under the context's current knobs
  const bool on = P.valid && op->A.walk == &op->walk && tun.hrb_walk && (tun.rbcsr_variant & 31) == 15 &&
                  P.R1 - P.W0 >= tun.walk_min_blocks;
  out[0] = on ? 1 : 0;
  out[1] = on ? P.nn : 0;
  out[2] = on ? P.K : 0;
  out[3] = on ? P.z0 : 0;
  out[4] = on ? P.g : 0;
  out[5] = on ? P.W0 : 0;
  out[6] = on ? P.R1 : 0;
  out[7] = on ? P.n_edge : 0;
  return QP_OK;
}

int qp_operator_evaluate_info(const qp_operator* op, int64_t out[3]) {
  QP_TRY
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_evaluate_info: NULL argument");
  out[0] = op->sparse_from;
  out[1] = op->sparse_from > 0 ? op->n_support : 0;
  out[2] = op->last_refresh_sparse ? 1 : 0;
  return QP_OK;
  QP_CATCH
}

int qp_operator_walk_reason(const qp_operator* op, int* code, char* text, size_t text_len) {
  QP_TRY
  if (!op || !code) return qp::fail(QP_E_BAD_ARG, "qp_operator_walk_reason: NULL argument");
  int c = QP_WALK_OK;
  std::string why;
  const qp::WalkPlan& P = op->walk;
  const qp::Tuning& tun = op->ctx->tun;
  if (op->A.format != QP_FMT_HRB) {
    if (op->A.format == QP_FMT_MATFREE || op->A.format == QP_FMT_DENSE) {
      c = QP_WALK_NOT_PACKED;
      why = "a dense / matrix-free operator has no sparse lattice to walk";
    } else if (!op->hermitian_planes) {
      c = QP_WALK_NOT_HERMITIAN;
      why = "a term of the operator is not exactly Hermitian (or a format other than AUTO / HRB was requested): no Hermitian-packed layout";
    } else if (op->n_relayouts > 0) {
      c = QP_WALK_COMPLEX_COEFF;
      why = "a complex coefficient took the Hermitian-packed operator back to plain row blocks (qp_operator_build_info: re-layouts)";
    } else {
      c = QP_WALK_NOT_PACKED;
      why = "Hermitian, but not laid out Hermitian-packed (transposed entries too far apart for the L2, irregular row blocks, or padding)";
    }
  } else if (!P.valid) {
    c = op->walk_reason != QP_WALK_OK ? op->walk_reason : (int)QP_WALK_NO_UNIFORM_RUN;
    why = op->walk_reason_text;
  } else if (!tun.hrb_walk) {
    c = QP_WALK_DISABLED;
    why = "knob hrb_walk is 0";
  } else if (P.R1 - P.W0 < tun.walk_min_blocks || P.R1 - P.W0 < P.S) {
    c = QP_WALK_TOO_FEW_BLOCKS;
    char buf[160];
    std::snprintf(buf, sizeof buf, "%lld walkable row blocks, knob walk_min_blocks is %d: the per-block kernel runs (as fast at this size)",
                  (long long)(P.R1 - P.W0), tun.walk_min_blocks);
    why = buf;
  }
  *code = c;
  if (text && text_len > 0) {
    std::strncpy(text, why.c_str(), text_len - 1);
    text[text_len - 1] = '\0';
  }
  return QP_OK;
  QP_CATCH
}

/* column-blocked mirror (device.h: ColBlockPlan): out = {1 if the operator has one, column blocks, log2 of the columns per
   block, rows per tile, entries of the longest segment, tiles}; *line_share (nullable) = the sampled share of gathers that
   pull a 128-byte line of their own -- what the decision was taken on (0 when it never came to sampling) */
int qp_operator_colblock_info(const qp_operator* op, int64_t out[6], double* line_share) {
  QP_TRY
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_colblock_info: NULL argument");
  out[0] = op->cb.valid;
  out[1] = op->cb.P;
  out[2] = op->cb.log2w;
  out[3] = 64 * op->cb.rpt;
  out[4] = op->cb.max_seg;
  out[5] = op->cb.ntiles;
  if (line_share) *line_share = op->cb_line_share;
  return QP_OK;
  QP_CATCH
}

int qp_operator_walk_long(const qp_operator* op, int64_t* glong) {
  if (!op || !glong) return qp::fail(QP_E_BAD_ARG, "qp_operator_walk_long: NULL argument");
  *glong = (op->walk.valid && op->A.walk == &op->walk && op->walk.xl) ? op->walk.glong : 0;
  return QP_OK;
}

/* the whole stencil shape of the walk plan: {near distances, far reach K, diagonal entry, long pairs, diagonal far neighbours
   (1: the far distances of step m are m g - 1, m g, m g + 1), strip step g, shorter long distance, longest distance}; zeros
   without a plan */
int qp_operator_walk_shape(const qp_operator* op, int64_t out[8]) {
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_walk_shape: NULL argument");
  for (int i = 0; i < 8; ++i) out[i] = 0;
  if (!(op->walk.valid && op->A.walk == &op->walk)) return QP_OK;
  const qp::WalkPlan& P = op->walk;
  out[0] = P.nn;
  out[1] = P.K;
  out[2] = P.z0;
  out[3] = P.xl;
  out[4] = P.fd;
  out[5] = P.g;
  out[6] = P.xl == 2 ? P.glong1 : 0;
  out[7] = P.xl ? P.glong : 0;
  return QP_OK;
}

int qp_operator_walk_long_pairs(const qp_operator* op, int64_t out[2]) {
  if (!op || !out) return qp::fail(QP_E_BAD_ARG, "qp_operator_walk_long_pairs: NULL argument");
  const bool on = op->walk.valid && op->A.walk == &op->walk;
  out[0] = (on && op->walk.xl == 2) ? op->walk.glong1 : (on && op->walk.xl == 1) ? op->walk.glong : 0;
  out[1] = (on && op->walk.xl == 2) ? op->walk.glong : 0;
  return QP_OK;
}

int qp_operator_spmm_tiles(qp_operator* op, int batch, int64_t out[6]) {
  QP_TRY
  if (!op || !out || batch < 1) return qp::fail(QP_E_BAD_ARG, "qp_operator_spmm_tiles: bad arguments");
  if (op->A.format == QP_FMT_MATFREE) return qp::fail(QP_E_BAD_ARG, "a matrix-free operator has no stored entries");
  QP_CHECK(use(op->ctx));
  for (int k = 0; k < 6; ++k) out[k] = 0;
  if (qp::spmm_uses_rows_kernel(op->ctx->tun, batch) && op->ctx->tun.spmm_rw < 0) {
    const qp::SpmmTiles* P = nullptr;
    QP_CHECK(operator_spmm_tiles(op, &P));
    if (P) {
      out[0] = 1;
      out[1] = P->ntiles;
      out[2] = P->nrest;
      out[3] = P->g;
      out[4] = P->shape.K;
      out[5] = P->shape.NN;
    }
  }
  return QP_OK;
  QP_CATCH
}

int qp_operator_spmm_walk(qp_operator* op, int batch, int64_t out[2]) {
  QP_TRY
  if (!op || !out || batch < 1) return qp::fail(QP_E_BAD_ARG, "qp_operator_spmm_walk: bad arguments");
  if (op->A.format == QP_FMT_MATFREE) return qp::fail(QP_E_BAD_ARG, "a matrix-free operator has no stored entries");
  QP_CHECK(use(op->ctx));
  out[0] = out[1] = 0;
  if (qp::spmm_uses_rows_kernel(op->ctx->tun, batch)) {
    const int32_t* order = nullptr;
    QP_CHECK(operator_spmm_order(op, batch, &order));
    out[0] = order ? op->m_order_g : 0;
    out[1] = order ? op->m_order_sw : 0;
  }
  return QP_OK;
  QP_CATCH
}

// download the *device* copy (current combined values and indices) back as canonical CSR
int qp_operator_get_csr(qp_operator* op, int64_t* rowptr, int32_t* col, qp_c128* vals) {
  QP_TRY
  if (!op || !rowptr || !col || !vals) return qp::fail(QP_E_BAD_ARG, "qp_operator_get_csr: NULL argument");
  if (op->A.format == QP_FMT_MATFREE) return qp::fail(QP_E_BAD_ARG, "a matrix-free operator has no stored entries");
  QP_CHECK(use(op->ctx));
  const DevMatrix& A = op->A;
  const auto& ur = op->u_rowptr;
  QP_HIP(hipStreamSynchronize(op->ctx->stream));
  std::vector<cplx> hv((size_t)std::max<int64_t>(A.stored, 1));
  QP_HIP(hipMemcpy(hv.data(), A.vals, (size_t)A.stored * sizeof(double2), hipMemcpyDeviceToHost));
  if (op->cv.valid && op->ctx->tun.value_dict != 0) {
    // the values the mat-vec really reads: table[tptr[block] + code], decoded back into the plane's positions (the
    // reconstruction is exact: tests compare it with what was passed in)
    const qp::CodedVals& C = op->cv;
    std::vector<uint8_t> codes((size_t)A.stored);
    std::vector<int64_t> tptr((size_t)A.nblocks);
    std::vector<cplx> tab((size_t)C.ntab);
    QP_HIP(hipMemcpy(codes.data(), C.codes, codes.size(), hipMemcpyDeviceToHost));
    QP_HIP(hipMemcpy(tptr.data(), C.tptr, tptr.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    QP_HIP(hipMemcpy(tab.data(), C.tab, tab.size() * sizeof(double2), hipMemcpyDeviceToHost));
    const HostLayout& Lc = op->layout;
    for (int64_t b = 0; b < A.nblocks; ++b) {
      const int64_t w = (Lc.bptr[b + 1] - Lc.bptr[b]) / kRB;
      for (int64_t k = 0; k < w; ++k)
        for (int64_t l = 0; l < kRB; ++l)
          hv[(size_t)(Lc.bptr[b] + k * kRB + l)] = tab[(size_t)((tptr[(size_t)b] >> 9) + codes[(size_t)(Lc.bptr[b] + (k >> 2) * (4 * kRB) + l * 4 + (k & 3))])];
    }
  }
  if (qp::csr_layout(A.format)) {
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

int qp_host_register(void* host, size_t bytes) {
  QP_TRY
  if (!host || bytes == 0) return qp::fail(QP_E_BAD_ARG, "qp_host_register: NULL / empty array");
  // portable: the registration holds for every device / context of the process, so it may be undone from
  // any thread (a garbage collector's finalizer thread included), whichever device is current there
  QP_HIP(hipHostRegister(host, bytes, hipHostRegisterPortable));
  return QP_OK;
  QP_CATCH
}

int qp_host_unregister(void* host) {
  QP_TRY
  if (!host) return qp::fail(QP_E_BAD_ARG, "qp_host_unregister: NULL");
  QP_HIP(hipHostUnregister(host));
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

}  // extern "C"

// CSR-ordered mirror of the operator for the batched (SpMM) path and the persistent
// small-system kernels, built lazily
int operator_csr_mirror(qp_operator* op, bool gather) {
  if (op->A.format == QP_FMT_MATFREE) return qp::fail(QP_E_BAD_ARG, "a matrix-free operator has no stored entries");
  qp_ctx* ctx = op->ctx;
  const DevMatrix& A = op->A;
  const auto& ur = op->u_rowptr;
  const auto& uc = op->u_col;
  const int64_t nnz = A.nnz;
  if (!op->m_rowptr) {
    std::vector<int64_t> map;
    csr_value_map(op, map);
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

