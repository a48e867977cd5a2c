// arnoldi!, newton!, ritzvals / specrange and the building blocks of their row-partitioned
// variants.
#include <thread>

#include "engine.h"

// spin-wait hint of the host's polling loops (not only x86)
#if defined(__x86_64__) || defined(__i386__)
#define QP_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define QP_CPU_RELAX() asm volatile("yield" ::: "memory")
#else
#define QP_CPU_RELAX() std::this_thread::yield()
#endif

#include <functional>

extern "C" {

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
  QP_CHECK(dev_alloc(&q->mgs_coef, (size_t)nvec));
  QP_HIP(hipMalloc((void**)&q->ticket, sizeof(unsigned)));
  QP_HIP(hipMemsetAsync(q->ticket, 0, sizeof(unsigned), ctx->stream));
  // coherent (fine-grained) pinned memory: what a kernel stores there is visible to the host while the sweep is
  // still running, which the flag hand-off of the folded sweep relies on
  const unsigned hflags = hipHostMallocMapped | hipHostMallocCoherent;
  QP_HIP(hipHostMalloc((void**)&q->h_hess, sizeof(double2) * (size_t)nvec * nvec, hflags));
  QP_HIP(hipHostMalloc((void**)&q->h_norms, sizeof(double) * (size_t)nvec, hflags));
  // [0, nvec): column complete (with its norm); [nvec, 2 nvec): the column's MGS coefficients are written (early flag)
  QP_HIP(hipHostMalloc((void**)&q->col_flags, sizeof(unsigned) * (size_t)nvec * 2, hflags));
  std::memset(q->col_flags, 0, sizeof(unsigned) * (size_t)nvec * 2);
  QP_HIP(hipHostGetDevicePointer((void**)&q->hess_map, q->h_hess, 0));
  QP_HIP(hipHostGetDevicePointer((void**)&q->norms_map, q->h_norms, 0));
  QP_HIP(hipHostGetDevicePointer((void**)&q->col_flags_map, q->col_flags, 0));
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
  if (q->raw[0]) (void)hipFree(q->raw[0]);
  if (q->raw[1]) (void)hipFree(q->raw[1]);
  if (q->hess_dev) (void)hipFree(q->hess_dev);
  if (q->norms_dev) (void)hipFree(q->norms_dev);
  if (q->part) (void)hipFree(q->part);
  if (q->md_part) (void)hipFree(q->md_part);
  if (q->gram) (void)hipFree(q->gram);
  if (q->h_hess) (void)hipHostFree(q->h_hess);
  if (q->h_norms) (void)hipHostFree(q->h_norms);
  if (q->col_flags) (void)hipHostFree(q->col_flags);
  if (q->hcoef) (void)hipFree(q->hcoef);
  if (q->mgs_coef) (void)hipFree(q->mgs_coef);
  if (q->ticket) (void)hipFree(q->ticket);
  if (q->op_gram) (void)hipFree(q->op_gram);
  if (q->op_hhat) (void)hipFree(q->op_hhat);
  if (q->op_part[0]) (void)hipFree(q->op_part[0]);
  if (q->op_part[1]) (void)hipFree(q->op_part[1]);
  if (q->op_svals) (void)hipFree(q->op_svals);
  if (q->op_nu_dev) (void)hipFree(q->op_nu_dev);
  if (q->h_nu) (void)hipHostFree(q->h_nu);
  for (hipEvent_t e : q->col_events) (void)hipEventDestroy(e);
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

// One Arnoldi column: w = H x (x = q[j], or the UNNORMALISED q[j] with the previous column's norm + scale
// folded into this mat-vec, see PlainEpi), then modified Gram-Schmidt of w against q[0..j]; leaves |w|^2
// partials in part[(j+1)&1].  hess column `hcol` (device, length >= j+1) receives dt*<q_i|w>.
struct FoldArgs {
  const double2* norm_part;   // |x|^2 partials of the previous column's projection
  double2* hess_slot;         // Hess[j, j-1]
  double* norm_slot;          // norms[j-1]
  double norm_min;
  unsigned* flag;             // col_flags[j-1] (device address) or null
  unsigned flag_value;
  unsigned* early_flag = nullptr;   // this column's early flag (set by the projection kernel once Hess[0..j, j] is written)
  bool* early_armed = nullptr;
};

int arnoldi_column(qp_operator* op, qp_krylov* q, int j, double dt, double2* hcol, const double2* xin = nullptr,
                   double2* w = nullptr, const FoldArgs* fold = nullptr) {
  qp_ctx* ctx = op->ctx;
  if (!xin) xin = q->q(j);
  if (!w) w = q->q(j + 1);
  qp::PlainEpi pe;
  pe.y = w;
  pe.alpha = make_double2(1.0, 0.0);
  pe.beta = make_double2(0.0, 0.0);
  pe.beta_zero = 1;
  if (fold) {
    pe.norm_part = fold->norm_part;
    pe.xloc = xin;
    pe.qn_out = q->q(j);
    pe.hess_slot = fold->hess_slot;
    pe.norm_slot = fold->norm_slot;
    pe.dt = dt;
    pe.norm_min = fold->norm_min;
    pe.flag = fold->flag;
    pe.flag_value = fold->flag_value;
  }
  const bool lowsync = ctx->tun.arnoldi_mode == 1 && q->gram_rows >= j && qp::mgs_lowsync_fits(j);
  bool dots_done = false;
  {
    const qp::ScopedRange mv_range(ctx->tun.roctx != 0 || qp::ranges_enabled_by_env(), "matrix-vector product");   // src/arnoldi.jl:81
    if (lowsync && ctx->tun.arnoldi_fuse_dots)   // knob: the multidot in the mat-vec's epilogue (kernels_arnoldi.hip)
      QP_CHECK(qp::launch_arnoldi_matvec_dots(ctx->stream, op->A, xin, pe, q->Q, q->n, j, q->md_part, &dots_done, &ctx->stats));
    if (!dots_done) QP_CHECK(qp::launch_spmv_plain(ctx->stream, op->A, xin, pe, &ctx->stats));  // src/arnoldi.jl:82
  }
  if (lowsync) {
    // low-synchronisation MGS: same coefficients (to rounding), 3 launches per column;
    // leaves |w|^2 partials in part[(j+1)&1] like the sequential path.  Needs the Gram
    // rows of the earlier basis vectors, which only this path maintains (a basis built by
    // the persistent small-system kernel or by sequential passes continues sequentially).
    q->gram_rows = j + 1;
    return qp::launch_mgs_lowsync(ctx->stream, q->Q, q->n, j, w, q->md_part, q->gram, q->nvec, hcol,
                                  q->hcoef, q->mgs_coef, q->ticket, q->part + (size_t)((j + 1) & 1) * kRedBlocks, dt,
                                  q->n, &ctx->stats, /* reduction + solve in the projection's prologue */ true, fold ? fold->early_flag : nullptr,
                                  fold ? fold->flag_value : 0u, fold ? fold->early_armed : nullptr, dots_done,
                                  /* rows per XCD, basis back to front: L2 reuse of what the dots pass read last */ true);
  }
  q->gram_rows = std::min(q->gram_rows, j);
  for (int i = 0; i <= j + 1; ++i) {                                              // :84-87
    qp::MgsArgs a;
    a.w = w;
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

__global__ void norm_guard_scale_kernel(double2* w, const double2* __restrict__ part_in, double2* hess_slot,
                                        double* norm_slot, double dt, double norm_min, int64_t n, const double2* w_in,
                                        unsigned* flag, unsigned flag_value);

// workgroups of norm_guard_scale_kernel: every one re-reduces the 256 partials, so no more of them than
// two elements per lane need (the partial order, hence the norm, does not depend on the grid)
static inline int guard_grid(int64_t n) {
  return (int)std::max<int64_t>(64, std::min<int64_t>(2048, (n + 2 * qp::kThreads - 1) / (2 * qp::kThreads)));
}

// arnoldi! with an optional normalisation of the start vector: beta_out != NULL means `psi` is
// not normalised; q_0 = psi / |psi| and *beta_out = |psi| (newton! :268-272 folded in, so that
// the persistent small-system kernel does it in the same launch).
//
// on_column != NULL (multi-launch path only): the sweep is enqueued whole, every column followed by
// an event; the host then takes the columns as they arrive and calls on_column(j) once column j is
// in Hess -- the caller's work on the leading (j+1) x (j+1) block (newton!: its eigenvalues,
// src/newton.jl:297) overlaps the device's work on the later columns.
using ColumnHook = std::function<int(int)>;
constexpr double kOnepassNormDrift = 1e-4;   // |nu_i - 1| of a one-pass sweep's stored basis vectors beyond which the sweep is redone

// The sweep that reads the basis once per column (knob arnoldi_onepass; kernels_onepass.hip says how): m + 1 column kernels, a
// single-workgroup solve after each; column j of the Hessenberg matrix reaches the pinned host buffer with the solve after
// column kernel j + 1 and announces itself through col_flags[j], like the folded sweep's.  The stored basis vectors have norm
// q->h_nu[i] (1 to rounding unless a projection cancelled nearly everything): the caller divides its combination coefficients by it.
static int arnoldi_onepass(qp_operator* op, qp_krylov* q, int m, const qp_state* psi, double dt, double norm_min, qp_c128* Hess,
                           int ldh, int* m_out, double* beta_out, const ColumnHook* on_column) {
  qp_ctx* ctx = op->ctx;
  const int ldd = q->nvec;
  if (!q->op_gram) {
    QP_CHECK(dev_alloc(&q->op_gram, (size_t)ldd * ldd));
    QP_CHECK(dev_alloc(&q->op_hhat, (size_t)ldd * ldd));
    QP_CHECK(dev_alloc(&q->op_part[0], (size_t)qp::op_part_slots(ldd) * kRedBlocks));
    QP_CHECK(dev_alloc(&q->op_part[1], (size_t)qp::op_part_slots(ldd) * kRedBlocks));
    QP_CHECK(dev_alloc(&q->op_svals, (size_t)ldd + 1));
    QP_CHECK(dev_alloc(&q->op_nu_dev, (size_t)ldd + 1));
    QP_HIP(hipHostMalloc((void**)&q->h_nu, sizeof(double) * (size_t)(ldd + 1), hipHostMallocMapped));
    QP_HIP(hipHostGetDevicePointer((void**)&q->nu_map, q->h_nu, 0));
  }
  if (!q->raw[0]) {
    QP_CHECK(dev_alloc(&q->raw[0], (size_t)q->n));
    QP_CHECK(dev_alloc(&q->raw[1], (size_t)q->n));
  }
  std::memset(q->h_hess, 0, sizeof(double2) * (size_t)ldd * ldd);
  std::memset(q->h_norms, 0, sizeof(double) * (size_t)ldd);
  for (int i = 0; i <= ldd; ++i) q->h_nu[i] = 1.0;
  double s0 = 1.0;
  if (beta_out) {   // newton! :268-272: beta = |Psi|, q_0 = Psi / beta -- the first column kernel scales by 1 / beta
    cplx n2;
    QP_CHECK(dot_sync(ctx, psi->d, psi->d, q->n, &n2));
    *beta_out = std::sqrt(n2.real());
    s0 = 1.0 / *beta_out;
  }
  q->seq = q->seq + 1 == 0 ? 1 : q->seq + 1;
  q->gram_rows = 0;            // (the low-synchronisation sweep's Gram rows do not describe this basis)
  q->nu_valid = true;
  {
    const qp::ScopedRange mv_range(ctx->tun.roctx != 0 || qp::ranges_enabled_by_env(), "matrix-vector product");
    QP_CHECK(qp::launch_arnoldi_onepass_sweep(ctx->stream, op->A, psi->d, s0, q->Q, q->n, q->raw, m, ldd, q->op_part, q->op_gram,
                                              q->op_hhat, q->op_svals, q->op_nu_dev, dt, q->hess_map, q->norms_map, q->nu_map,
                                              q->col_flags_map, q->seq, &ctx->stats));
  }
  const cplx* hh = reinterpret_cast<const cplx*>(q->h_hess);
  const double* hn = q->h_norms;
  int m_eff = m, hook_rc = QP_OK;
  for (int j = 0; j < m; ++j) {
    const auto t_begin = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&q->col_flags[j], __ATOMIC_ACQUIRE) != q->seq) {
      QP_CPU_RELAX();
      if ((++spins & 0xfffffu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 5.0) {
        QP_HIP(hipStreamSynchronize(ctx->stream));      // never spin for good: fall back to the stream and look once more
        if (__atomic_load_n(&q->col_flags[j], __ATOMIC_ACQUIRE) != q->seq)
          return qp::fail(QP_E_INTERNAL, "Arnoldi column %d never announced itself to the host", j);
      }
    }
    if (j + 1 == m) q->t_last_column = std::chrono::steady_clock::now();
    for (int i = 0; i < std::min(j + 2, m + 1); ++i) {
      const cplx v = hh[(size_t)j * ldd + i];
      Hess[(size_t)j * ldh + i] = qp_c128{v.real(), v.imag()};
    }
    if (on_column && (hook_rc = (*on_column)(j)) != QP_OK) break;
    if (hn[j] < norm_min) {      // dimensionality exhausted  src/arnoldi.jl:91-95
      m_eff = j + 1;
      break;
    }
  }
  // a complete sweep needs no wait (its last column announced itself; the stream is consumed in order); after a breakdown or
  // a failed hook the later columns are discarded: wait for them
  if (m_eff != m || hook_rc != QP_OK) QP_HIP(hipStreamSynchronize(ctx->stream));
  if (hook_rc != QP_OK) return hook_rc;
  if (m_eff < m) {
    // the reference leaves the vector of a breakdown UNNORMALISED (src/arnoldi.jl:91-95 breaks before :96); the stored one was
    // scaled by s: its coefficient is to be divided by nu / h instead of nu
    const double h = hn[m_eff - 1];
    q->h_nu[m_eff] = h > 0.0 ? q->h_nu[m_eff] / h : 0.0;
  }
  *m_out = m_eff;
  return QP_OK;
}
static int arnoldi_impl(qp_operator* op, qp_krylov* q, int m, const qp_state* psi, double dt, int extended,
                        double norm_min, qp_c128* Hess, int ldh, int* m_out, double* beta_out,
                        const ColumnHook* on_column = nullptr, bool scaled_basis_ok = false) {
  QP_TRY
  if (!op || !q || !psi || !Hess || !m_out) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi: NULL argument");
  const int dim = extended ? m + 1 : m;
  if (m < 1 || ldh < dim || q->nvec < m + 1) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi: Hess/q too small for m=%d", m);
  if (op->A.nrows != op->A.ncols || psi->n != op->A.nrows || q->n != psi->n) return qp::fail(QP_E_BAD_ARG, "qp_arnoldi: shape mismatch");
  qp_ctx* ctx = op->ctx;
  QP_CHECK(use(ctx));
  const int ldd = q->nvec;
  std::memset(Hess, 0, sizeof(qp_c128) * (size_t)ldh * ldh);                                      // :78
  qp::SmallArgs plan;
  bool small = false;
  bool piped = false;
  bool fold = false;
  bool flags = false;   // folded + pipelined sweep: columns are announced through col_flags, not events
  bool early_last = false;   // the last column also raises its early flag
  if (ctx->tun.small_nnz > 0 && op->A.nnz <= (int64_t)ctx->tun.small_nnz * (qp::kSmallEptArnoldi / qp::kSmallEpt) &&
      qp::small_arnoldi_fits(q->n, m)) {
    int64_t maxrow = 0;
    for (int64_t r = 0; r < q->n; ++r) maxrow = std::max<int64_t>(maxrow, op->u_rowptr[r + 1] - op->u_rowptr[r]);
    // the plan of the 16-slot kernels where it exists (same lanes per row, hence the same rounding, as
    // before the 32-slot variants were added), the larger one only for systems that need it
    small = qp::small_plan(q->n, maxrow, &plan, qp::kSmallEpt) || qp::small_plan(q->n, maxrow, &plan, qp::kSmallEptArnoldi);
  }
  q->nu_valid = false;
  // (only for a caller that divides its combination coefficients by the stored vectors' norms: newton!)
  const double sweep_bytes = 16.0 * (double)q->n * (m + 3) + (op->A.vals_r ? 12.0 : 20.0) * (double)op->A.stored;
  const bool onepass_wanted = ctx->tun.arnoldi_onepass >= 2 || (ctx->tun.arnoldi_onepass == 1 && sweep_bytes > 224.0 * 1024 * 1024);
  if (!small && extended && scaled_basis_ok && ctx->tun.arnoldi_mode == 1 && onepass_wanted && qp::arnoldi_onepass_fits(op->A, m, q->nvec)) {
    QP_CHECK(arnoldi_onepass(op, q, m, psi, dt, norm_min, Hess, ldh, m_out, beta_out, on_column));
    ++q->n_onepass;
    // The one-pass sweep never applies H to the new basis vector itself: it carries a_{t+1} = s (H a_t - sum gamma_i q_i) forward by
    // linearity and takes the scale from |a|^2 - sum |h|^2, which cancels when h_{t+1,t} << |a_t| -- the known error growth of
    // pipelined Krylov sweeps.  The stored vectors' measured norms nu_i say when that happened: all conversions are exact in nu,
    // but a nu far from 1 means the recurrence lost digits.  Then the sweep is done again in the two-pass form (the column
    // hook is idempotent: it recomputes the Ritz values of the leading blocks from the new columns).
    double drift = 0.0;
    for (int i = 0; i <= *m_out && i <= m; ++i)
      if (!(i == *m_out && *m_out < m)) drift = std::max(drift, std::fabs(q->h_nu[i] - 1.0));    // (the vector of a breakdown is unnormalised by design)
    if (!(drift > kOnepassNormDrift) && ctx->tun.arnoldi_onepass != 3) return QP_OK;
    ++q->n_onepass_redone;
    q->nu_valid = false;
    std::memset(Hess, 0, sizeof(qp_c128) * (size_t)ldh * ldh);
  }
  if (small) {
    // all m columns in one persistent single-workgroup launch (kernels_small.hip: arnoldi_small_kernel)
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
    a.normalize_start = beta_out ? 1 : 0;     // the kernel also zero-fills hess / norms
    QP_CHECK(qp::launch_arnoldi_small(ctx->stream, a, &ctx->stats));
    q->gram_rows = 0;
  } else {
    piped = on_column != nullptr;
    // Hessenberg entries and norms go straight into the pinned host buffers (nothing of an earlier
    // sweep is in flight: every sweep ends with a synchronisation)
    std::memset(q->h_hess, 0, sizeof(double2) * (size_t)ldd * ldd);
    std::memset(q->h_norms, 0, sizeof(double) * (size_t)ldd);
    QP_HIP(hipMemsetAsync(q->ticket, 0, sizeof(unsigned), ctx->stream));
    QP_HIP(hipMemcpyAsync(q->q(0), psi->d, (size_t)q->n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));  // :79
    if (beta_out) {
      cplx n2;
      QP_CHECK(dot_sync(ctx, q->q(0), q->q(0), q->n, &n2));
      *beta_out = std::sqrt(n2.real());
      QP_CHECK(qp::launch_scal(ctx->stream, q->q(0), make_double2(1.0 / *beta_out, 0.0), q->n, &ctx->stats));
    }
    if (piped) {
      while ((int)q->col_events.size() < m) {
        hipEvent_t e;
        QP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        q->col_events.push_back(e);
      }
    }
    // "norm + scale" of column j is done by the mat-vec of column j + 1 (it scales its row
    // sums by 1 / |q_j| and stores the normalised q_j as it goes; src/arnoldi.jl:89-96 applied on the fly), so a
    // column is mat-vec + projection only.  The unnormalised vectors ping-pong between two scratch vectors.
    fold = op->A.format != QP_FMT_MATFREE && m > 1;
    if (fold && !q->raw[0]) {
      QP_CHECK(dev_alloc(&q->raw[0], (size_t)q->n));
      QP_CHECK(dev_alloc(&q->raw[1], (size_t)q->n));
    }
    flags = fold && piped;
    if (flags) {      // a new sequence number per sweep: the columns announce themselves with it
      q->seq = q->seq + 1 == 0 ? 1 : q->seq + 1;
    }
    auto enqueue_columns = [&]() -> int {
    for (int j = 0; j < m; ++j) {
      double2* hcol = q->hess_map + (size_t)j * ldd;
      if (fold) {
        FoldArgs fa{q->part + (size_t)(j & 1) * kRedBlocks, j > 0 ? q->hess_map + (size_t)(j - 1) * ldd + j : nullptr,
                    j > 0 ? q->norms_map + (j - 1) : nullptr, norm_min, (flags && j > 0) ? q->col_flags_map + (j - 1) : nullptr,
                    q->seq};
        // The hook of the LAST column is the one the device waits for (the others run while it works on later columns):
        // its Hessenberg block does not contain the column's norm, so it may start as soon as the projection kernel has
        // solved for the coefficients -- while that kernel still streams the basis.
        if (flags && j + 1 == m && j > 0) {
          fa.early_flag = q->col_flags_map + q->nvec + j;
          fa.early_armed = &early_last;
        }
        QP_CHECK(arnoldi_column(op, q, j, dt, hcol, j == 0 ? q->q(0) : q->raw[j & 1], q->raw[(j + 1) & 1], j > 0 ? &fa : nullptr));
        if (j + 1 == m) {   // the last vector: normalised into the basis (extended), or handed over as it is
          if (extended) {
            hipLaunchKernelGGL(norm_guard_scale_kernel, dim3(guard_grid(q->n)), dim3(qp::kThreads), 0, ctx->stream, q->q(j + 1),
                               q->part + (size_t)((j + 1) & 1) * kRedBlocks, hcol + (j + 1), q->norms_map + j, dt, norm_min,
                               q->n, (const double2*)q->raw[(j + 1) & 1], flags ? q->col_flags_map + j : (unsigned*)nullptr, q->seq);
            QP_HIP(hipGetLastError());
            ctx->stats.n_launch++;
          } else {
            QP_HIP(hipMemcpyAsync(q->q(j + 1), q->raw[(j + 1) & 1], (size_t)q->n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
          }
        }
      } else {
        QP_CHECK(arnoldi_column(op, q, j, dt, hcol));
        if ((j + 1 < m) || extended) {                                                               // :88-97
          hipLaunchKernelGGL(norm_guard_scale_kernel, dim3(guard_grid(q->n)), dim3(qp::kThreads), 0, ctx->stream, q->q(j + 1),
                             q->part + (size_t)((j + 1) & 1) * kRedBlocks, hcol + (j + 1), q->norms_map + j, dt, norm_min,
                             q->n, (const double2*)q->q(j + 1), (unsigned*)nullptr, 0u);
          QP_HIP(hipGetLastError());
          ctx->stats.n_launch++;
        }
      }
      if (piped && !flags) QP_HIP(hipEventRecord(q->col_events[j], ctx->stream));
    }
    return QP_OK;
    };
    QP_CHECK(enqueue_columns());
  }
  const cplx* hh = reinterpret_cast<const cplx*>(q->h_hess);
  const double* hn = q->h_norms;
  if (small) {
    // one download of the Hessenberg matrix and the norms, into pinned memory
    QP_HIP(hipMemcpyAsync(q->h_hess, q->hess_dev, (size_t)ldd * ldd * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipMemcpyAsync(q->h_norms, q->norms_dev, (size_t)ldd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (!piped) QP_HIP(hipStreamSynchronize(ctx->stream));
  if (small && beta_out) *beta_out = hn[ldd - 1];   // written by the kernel (slot ldd - 1 is never a column's)
  int m_eff = m;
  int hook_rc = QP_OK;
  // wait until column j is complete on the host: an event, or (folded sweep) the flag that the mat-vec of
  // column j + 1 sets after the column's last entries -- the last column has no successor: stream end
  auto wait_column = [&](int j) -> int {
    if (!piped) return QP_OK;
    if (!flags) {
      QP_HIP(hipEventSynchronize(q->col_events[j]));
      return QP_OK;
    }
    if (j + 1 == m && !extended) {   // no kernel after the last column that could announce it
      QP_HIP(hipStreamSynchronize(ctx->stream));
      return QP_OK;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&q->col_flags[j], __ATOMIC_ACQUIRE) != q->seq) {
      QP_CPU_RELAX();
      if ((++spins & 0xfffffu) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 5.0) {
        // never spin for good: fall back to the stream and look once more
        QP_HIP(hipStreamSynchronize(ctx->stream));
        if (__atomic_load_n(&q->col_flags[j], __ATOMIC_ACQUIRE) != q->seq)
          return qp::fail(QP_E_INTERNAL, "Arnoldi column %d never announced itself to the host", j);
      }
    }
    return QP_OK;
  };
  auto copy_column = [&](int j, int rows) {
    for (int i = 0; i < rows; ++i) {
      cplx v = hh[(size_t)j * ldd + i];
      Hess[(size_t)j * ldh + i] = qp_c128{v.real(), v.imag()};
    }
  };
  for (int j = 0; j < m; ++j) {
    const int rows = std::min(j + 2, dim);
    bool hooked = false;
    if (early_last && j + 1 == m) {   // coefficients first, hook, then the norm
      unsigned spins = 0;
      const unsigned* ef = q->col_flags + q->nvec + j;
      while (__atomic_load_n(ef, __ATOMIC_ACQUIRE) != q->seq && ++spins < (1u << 24)) QP_CPU_RELAX();
      if (__atomic_load_n(ef, __ATOMIC_ACQUIRE) == q->seq) {
        copy_column(j, j + 1);
        if ((hook_rc = (*on_column)(j)) != QP_OK) break;
        hooked = true;
      }
    }
    QP_CHECK(wait_column(j));
    if (j + 1 == m) q->t_last_column = std::chrono::steady_clock::now();
    copy_column(j, rows);
    if (piped && !hooked && (hook_rc = (*on_column)(j)) != QP_OK) break;
    if (((j + 1 < m) || extended) && hn[j] < norm_min) {  // dimensionality exhausted  :91-95
      m_eff = j + 1;
      break;
    }
  }
  // the columns after a breakdown are discarded: wait for them.  (A complete folded sweep needs no wait: its last
  // column announced itself, and what the device still does -- normalising the last vector -- is consumed in stream order.)
  if (piped && !(flags && extended && m_eff == m && hook_rc == QP_OK)) QP_HIP(hipStreamSynchronize(ctx->stream));
  if (hook_rc != QP_OK) return hook_rc;
  *m_out = m_eff;
  return QP_OK;
  QP_CATCH
}

int qp_arnoldi(qp_operator* op, qp_krylov* q, int m, const qp_state* psi, double dt, int extended, double norm_min,
               qp_c128* Hess, int ldh, int* m_out) {
  return arnoldi_impl(op, q, m, psi, dt, extended, norm_min, Hess, ldh, m_out, nullptr);
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
                                q->mgs_coef, norm_partials->d, dt, q->n, &q->ctx->stats);
  QP_CATCH
}

int qp_krylov_normalize(qp_krylov* q, int j, double dt, double norm_min, const qp_state* norm_partials,
                        qp_state* hess_norm) {
  QP_TRY
  if (!q || !norm_partials || !hess_norm || j < 0 || j + 1 >= q->nvec || norm_partials->n < kRedBlocks || hess_norm->n < 2)
    return qp::fail(QP_E_BAD_ARG, "qp_krylov_normalize: bad arguments");
  QP_CHECK(use(q->ctx));
  hipLaunchKernelGGL(norm_guard_scale_kernel, dim3(guard_grid(q->n)), dim3(qp::kThreads), 0, q->ctx->stream, q->q(j + 1),
                     norm_partials->d, hess_norm->d, reinterpret_cast<double*>(hess_norm->d + 1), dt, norm_min, q->n,
                     (const double2*)q->q(j + 1), (unsigned*)nullptr, 0u);
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
// (w_in != w: the scaled -- or, past a breakdown, the unscaled -- vector goes to w, w_in is left alone)
// (flag != NULL: host-visible announcement that the slots are written, see PlainEpi::flag)
__global__ __launch_bounds__(qp::kThreads) void norm_guard_scale_kernel(double2* w, const double2* __restrict__ part_in,
                                                                        double2* hess_slot, double* norm_slot, double dt,
                                                                        double norm_min, int64_t n, const double2* w_in,
                                                                        unsigned* flag, unsigned flag_value) {
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
    if (flag) __hip_atomic_store(flag, flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (h < norm_min && w_in == w) return;
  const double inv = (h < norm_min) ? 1.0 : 1.0 / h;
  for (int64_t i = (int64_t)blockIdx.x * qp::kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * qp::kThreads) {
    double2 t = w_in[i];
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
  const bool ranges = ctx->tun.roctx != 0 || qp::ranges_enabled_by_env();
  const qp::ScopedRange step_range(ranges, "prop_step!");               // src/newton_propagator.jl:121
  const int m_max = w->m_max;
  int m = m_max;                                                        // :253
  std::fill(w->a.begin(), w->a.end(), cplx(0));                         // :254-255
  std::fill(w->leja.begin(), w->leja.end(), cplx(0));
  const int ldh = m_max + 1;
  std::vector<cplx>&Hess = w->Hess, &R = w->R, &P = w->P, &Rn = w->Rn, &ritz = w->ritz;
  Hess.assign((size_t)ldh * ldh, cplx(0));
  int n_a = 0, n_leja = 0, s = 0, n_matvec = 0;
  double last_relerr = 0, norm_psi = 0;
  qp_state vstate{ctx, w->v, w->n, false};
  // v = Psi / beta, beta = |Psi| (:268-272) is done by the first Arnoldi sweep itself (q_0)
  double beta = 0.0;
  double ms_arnoldi = 0, ms_eig = 0, ms_leja = 0, ms_coeffs = 0, ms_poly = 0, ms_update = 0, ms_exposed = 0;
  const int onepass0 = w->q->n_onepass, redone0 = w->q->n_onepass_redone;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  while (true) {                                                                     // :274
    int m_req = m;
    auto t0 = now();
    // Ritz values of every leading block (:297), block j+1 as soon as column j has arrived -- on
    // the multi-launch path while the device is still orthogonalising the later columns
    ritz.assign((size_t)m_req * (m_req + 1) / 2, cplx(0));
    // ... and, from the second restart on, the head of every candidate's Leja product (the factors of
    // the Leja points of the earlier restarts, src/newton.jl:127-136)
    std::vector<qp::ScaledProd>& lprod = w->leja_prod;
    lprod.assign(ritz.size(), qp::ScaledProd{1.0, 0});
    double ms_eig_sweep = 0, ms_fold_sweep = 0;
    int blocks_done = 0;
    const ColumnHook eig_block = [&](int j) -> int {
      auto t1 = now();
      const qp::ScopedRange eig_range(ranges, "diagonalize_hessenberg_matrix");   // src/newton.jl:296
      const size_t off = (size_t)j * (j + 1) / 2;
      const int st = qp::diagonalize_hessenberg_block(Hess.data(), ldh, j + 1, ritz.data() + off);
      ms_eig_sweep += ms_since(t1);
      if (st == QP_OK && n_leja > 0) {
        t1 = now();
        for (int i = 0; i <= j; ++i) lprod[off + i] = qp::leja_fold_candidate(w->leja.data(), n_leja, ritz[off + i]);
        ms_fold_sweep += ms_since(t1);
      }
      blocks_done = j + 1;
      return st == QP_OK ? QP_OK : qp::fail(QP_E_INTERNAL, "Hessenberg QR did not converge");
    };
    {
      const qp::ScopedRange arnoldi_range(ranges, "arnoldi!");                       // src/newton.jl:276
      QP_CHECK(arnoldi_impl(op, w->q, m_req, s == 0 ? psi : &vstate, dt, 1, norm_min,
                            reinterpret_cast<qp_c128*>(Hess.data()), ldh, &m, s == 0 ? &beta : nullptr,
                            ctx->tun.newton_pipeline ? &eig_block : nullptr, true));
    }
    ms_arnoldi += ms_since(t0) - ms_eig_sweep - ms_fold_sweep;
    ms_eig += ms_eig_sweep;
    ms_leja += ms_fold_sweep;
    n_matvec += m_req;
    if (m == 1 && s == 0) {                                                          // :289-295
      const cplx lam = beta * Hess[0];
      const cplx f = qp::eval_func(func_id, cb, user, lam);
      QP_CHECK(qp::launch_scal(ctx->stream, psi->d, d2(f), psi->n, &ctx->stats));
      break;
    }
    t0 = now();
    for (int j = blocks_done; j < m; ++j)   // persistent-kernel sweep, or the pipeline switched off
      QP_CHECK(eig_block(j));
    ritz.resize((size_t)m * (m + 1) / 2);
    const bool folded = n_leja > 0;   // (the products do not depend on m: valid also after a breakdown)
    ms_eig += ms_since(t0);
    if (s == 0) {                                                                    // :301-303, :67-70
      double rmax = 0;
      for (auto& z : ritz) rmax = std::max(rmax, std::abs(z));
      w->radius = 1.2 * rmax;
    }
    const int n_s = n_leja;                                                          // :307
    if ((int)w->leja.size() < n_leja + m) w->leja.resize((size_t)2 * (n_leja + m), cplx(0));  // :105-110
    t0 = now();
    {
      const qp::ScopedRange leja_range(ranges, "get Leja points");                   // src/newton.jl:306
      qp::extend_leja(w->leja.data(), n_leja, ritz.data(), (int)ritz.size(), m, folded ? lprod.data() : nullptr);
    }
    ms_leja += ms_since(t0);
    n_leja += m;
    if ((int)w->a.size() < n_leja) w->a.resize((size_t)2 * n_leja, cplx(0));         // :187-192
    {
      t0 = now();
      int st;
      {
        const qp::ScopedRange coeff_range(ranges, "get Newton coeffs");              // src/newton.jl:313
        st = qp::extend_newton_coeffs(w->a.data(), n_a, w->leja.data(), func_id, cb, user, n_leja, w->radius);  // :314
      }
      ms_coeffs += ms_since(t0);
      if (st == QP_E_DIVDIFF_UNDERFLOW) return qp::fail(st, "Divided differences too small");
      if (st != QP_OK) return qp::fail(st, "extend_newton_coeffs failed (radius=%g)", w->radius);
      n_a = n_leja;
    }
    // Newton polynomial in the extended Hessenberg matrix                           :328-343
    if (ranges) qp::range_push("evaluate polynomial");                               // src/newton.jl:328 (closed after the update below)
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
    // starting vector of the next restart (host part)                                :356-367
    apply(w->leja[n_s + m - 1]);
    double b2 = 0;
    for (int i = 0; i < mp; ++i) {
      const double ab = std::abs(R[i]);
      b2 += ab * ab;
    }
    beta = std::sqrt(b2);
    for (int i = 0; i < mp; ++i) R[i] *= (1.0 / beta);
    // Psi = (s == 0 ? 0 : Psi) + sum_{i<m} P_i q_i  (:346-352)  and  v = sum_{i<=m} R_i q_i  (q_0 is the
    // start vector of this sweep): one pass over the basis; fixed-size coefficient blocks, else one by one
    if (w->q->nu_valid) {
      // one-pass sweep: the stored basis vectors have norm nu_i (not exactly one): coefficients in the orthonormal basis -> stored basis
      for (int i = 0; i < mp; ++i) {
        const double nu = w->q->h_nu[i];
        const double inv = nu > 0.0 ? 1.0 / nu : 0.0;
        if (i < m) P[i] *= inv;
        R[i] *= inv;
      }
    }
    ms_exposed += ms_since(w->q->t_last_column);
    if (!qp::launch_combine2_vecs(ctx->stream, psi->d, s == 0 ? 0 : 1, m, reinterpret_cast<const double2*>(P.data()), w->v,
                                  m + 1, reinterpret_cast<const double2*>(R.data()), w->q->q(0), w->n, w->npart, w->n,
                                  &ctx->stats)) {
      QP_CHECK(qp::launch_combine_vecs(ctx->stream, psi->d, s == 0 ? 0 : 1, make_double2(1.0, 0.0), w->q->q(0), w->n, m,
                                       reinterpret_cast<const double2*>(P.data()), w->npart, w->n, &ctx->stats));
      QP_CHECK(qp::launch_combine_vecs(ctx->stream, w->v, 0, make_double2(1.0, 0.0), w->q->q(0), w->n, m + 1,
                                       reinterpret_cast<const double2*>(R.data()), nullptr, w->n, &ctx->stats));
    }
    QP_HIP(hipMemcpyAsync(w->h_npart, w->npart, kRedBlocks * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QP_HIP(hipStreamSynchronize(ctx->stream));
    norm_psi = std::sqrt(sum_partials(w->h_npart).real());
    if (ranges) qp::range_pop();
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
    stats->ms_exposed = ms_exposed;
    stats->sweeps_onepass = w->q->n_onepass - onepass0;
    stats->sweeps_onepass_redone = w->q->n_onepass_redone - redone0;
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
