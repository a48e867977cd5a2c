// Internal header of the engine translation units (engine_*.hip): handle types behind the
// opaque pointers of include/qprop.h and the small helpers they share.
#pragma once

#include <vector>
#include <string>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>

#include "device.h"

using qp::cplx;
using qp::DevMatrix;
using qp::kRB;
using qp::kRedBlocks;
using qp::Stats;

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
  // qp_ctx_destroy releases the resources and marks the record, which itself stays allocated: handles
  // created from the context may be destroyed in any order, also after it (finalizers of a garbage
  // collector -- Python's, Julia's -- run in no particular order)
  bool closed = false;
  qp::Tuning tun;              // this context's developer knobs (qp_ctx_tuning_set)
  // two pinned staging buffers for the operator build's value upload (host threads fill one while the other is on its way to
  // the device), allocated by the first large build and kept: pinning is the slow part
  void* stage[2] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
};

struct qp_state {
  qp_ctx* ctx;
  double2* d;
  int64_t n;
  bool own;
};

// Host arrays of the size of a matrix (gigabytes at N = 2^24): resize() leaves the new elements unwritten -- whoever resizes fills them,
// on several threads -- instead of one thread zeroing them first (std::vector<T>: a serial pass over memory before the real one)
namespace qp {
template <class T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
  void deallocate(T* p, size_t) { ::operator delete(p); }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) > 0) ::new ((void*)p) U(std::forward<A>(a)...);      // (no arguments: nothing written)
  }
  template <class U>
  bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using HostVec = std::vector<T, NoInitAlloc<T>>;
}  // namespace qp

struct qp_matrix {  // canonical host CSR (the result of the boundary's index work)
  qp_ctx* ctx;
  int64_t nrows, ncols, nnz;
  qp::HostVec<int64_t> rowptr;
  qp::HostVec<int32_t> col;
  qp::HostVec<cplx> vals;
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
  int walk_reason = 0;            // QP_WALK_*: why the operator has no strip-walk plan (build_walk_plan)
  std::string walk_reason_text;
  qp::WalkPlan walk;              // strip-walk plan of a Hermitian-packed lattice operator (A.walk points here when valid)
  qp::WalkPlan walk2;             // the same run for the two-term walk (kernels_walk2.hip): [W0, R1) shrunk by K strip steps at either end,
                                  // edge list = every block outside that region; valid = 0 when the operator has none
  qp::ColBlockPlan cb;            // column-blocked mirror of an operator with irregular columns (A.cb points here when valid)
  qp::CodedVals cv;               // value-dictionary mirror (A.cv points here when valid): codes, per-block tables
  std::vector<double2*> cv_tplanes;   // device: per term the tuple component of every table entry [cv.ntab] (static)
  double2** cv_tplanes_dev = nullptr;
  double2* cv_tab_comb = nullptr;     // device: combined table, allocated on the first non-trivial coefficient set
  int cv_reason = 0;                  // why there is no mirror: 0 built / not tried, 1 format, 2 a block with > 256 tuples, 3 no saving, 4 knob off
  double cb_line_share = 0.0;     // what decided: share of a row block's gathers that pull a line of their own (sampled)
  double build_ms = 0, build_ms_total = 0;   // host time of the latest / of all device layout builds
  int64_t n_lattice_fill = 0;               // explicit zeros that complete a lattice operator's rows (engine_plans.hip: lattice_fill)
  int n_builds = 0, n_relayouts = 0;         // re-layouts: builds forced after creation (complex coefficient on a packed operator)
  bool hermitian_planes = false;
  // CSR-ordered mirror of the current values for the batched (SpMM) path, built lazily
  int64_t* m_rowptr = nullptr;
  int32_t* m_cols = nullptr;
  int64_t* m_map = nullptr;     // position in A.vals (>= 0) or -(position)-1 for a conj-transposed value
  double2* m_vals = nullptr;
  uint64_t vals_epoch = 1, m_epoch = 0;
  // row walk of the batched path (operator_spmm_order): device permutation of the rows, or null for
  // the natural order; rebuilt when the panel width or the knob it was built for changes
  int32_t* m_order = nullptr;
  int m_order_batch = 0, m_order_knob = 0;
  bool m_order_valid = false;
  int64_t m_order_g = 0, m_order_sw = 0;   // what was detected: inner dimension and strip width (0: none)
  qp::SpmmTiles m_tiles;                   // LDS-staged tiles of the batched path (operator_spmm_tiles), built lazily
  int nops = 0, ncoeffs = 0;
  qp::HostVec<int64_t> u_rowptr;  // union pattern (host), for get_csr and plane scatter
  qp::HostVec<int32_t> u_col;
  std::vector<double2*> planes;   // device value planes, one per term, layout of A.vals
  double2** planes_dev = nullptr;
  double2* combined = nullptr;    // device, allocated on first non-trivial coefficient set
  // Sparse control planes (knob sparse_controls): the trailing control terms whose entries cover at most a quarter of the
  // stored values -- a dipole operator on a grid is a diagonal.  evaluate! then rewrites only those positions:
  //   combined[p] = base[p] + sum_{l >= sparse_from} c_l plane_l[p],  p in the union of their supports,
  // base = the sum over the planes before them with the coefficients it was last built with (rebuilt when those change).
  int sparse_from = -1;
  int64_t n_support = 0;
  int32_t* support = nullptr;        // device: positions in the value array, ascending
  double2* support_vals = nullptr;   // device: [nops - sparse_from][n_support]
  double2* base = nullptr;
  std::vector<double2> base_eff;
  bool base_valid = false, base_real = false;
  bool last_refresh_sparse = false;   // the latest evaluate! rewrote only the sparse control terms' positions
  bool planes_real = false;       // every value of every term has a zero imaginary part
  double* real_vals = nullptr;    // device copy of the real parts of the current values (see DevMatrix::vals_r)
  const double2* real_of = nullptr;  // the complex array real_vals was extracted from, when still valid
  std::vector<cplx> coeffs;
  cplx scale = 1.0;
  // QP_FMT_MATFREE operators (engine_liouville.hip): owner data and its hooks
  void* mf = nullptr;
  int (*mf_refresh)(qp_operator* op) = nullptr;   // coefficients or scale changed
  void (*mf_free)(qp_operator* op) = nullptr;
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
  unsigned* counter = nullptr;        // [0] signal counter (device memory)
  unsigned* timeout_host = nullptr;   // spin-timeout flag: pinned host memory mapped into the device, so the host can
  unsigned* timeout_dev = nullptr;    //   look at it without synchronising (its device address)
  unsigned signals_issued = 0;
  unsigned wait_from_wg = 0;          // interior workgroups at or beyond this position poll
  unsigned n_waiting_wg = 0;          // how many of them there are
  // the interior set as a strip walk (lattice operators): a run inside the interior that never reads a boundary row, the
  // rest of the interior as its edge blocks (only they wait for the boundary launch)
  qp::WalkPlan walk;
};
// QP_E_INTERNAL once an in-launch wait of this split has ever timed out (no synchronisation)
int split_timed_out(const qp_split* sp);

struct qp_krylov {
  qp_ctx* ctx;
  int64_t n;
  int nvec;
  double2* Q = nullptr;         // nvec vectors of length n, contiguous
  double2* raw[2] = {nullptr, nullptr};   // unnormalised vectors of the folded sweep, on demand
  double2* hess_dev = nullptr;  // nvec x nvec column major
  double* norms_dev = nullptr;  // nvec
  double2* part = nullptr;      // 2 x kRedBlocks ping-pong partials
  double2* md_part = nullptr;   // kRedBlocks x 2 nvec multidot partials (low-sync MGS)
  double2* gram = nullptr;      // nvec x nvec Gram rows <q_i|q_k>, k < i
  int gram_rows = 0;            // rows 0 .. gram_rows-1 of `gram` describe the current basis
  double2* hcoef = nullptr;     // 2 nvec reduced inner products of the current column
  double2* mgs_coef = nullptr;  // nvec axpy coefficients of the current column (low-sync MGS)
  unsigned* ticket = nullptr;   // finishing-workgroup counter of the multidot launch (zero between launches)
  double2* h_hess = nullptr;    // pinned host mirrors of hess_dev / norms_dev
  double* h_norms = nullptr;
  double2* hess_map = nullptr;  // the same pinned buffers as the device sees them: the multi-launch
  double* norms_map = nullptr;  // Arnoldi sweep writes its Hessenberg entries straight to the host
  std::vector<hipEvent_t> col_events;  // one per column: "column j is on the host" (pipelined restarts)
  // folded sweep: the mat-vec of column j + 1 announces "column j is complete on the host" by storing the sweep's
  // sequence number into col_flags[j] (coherent pinned memory, system-scope release) after it has written
  // Hess[j+1, j] and the norm -- the host polls instead of waiting for an event (no event record between columns)
  unsigned* col_flags = nullptr;
  unsigned* col_flags_map = nullptr;
  unsigned seq = 0;
  std::chrono::steady_clock::time_point t_last_column;   // when the host saw the last column of the latest sweep
  // one-pass sweep (knob arnoldi_onepass, kernels_onepass.hip), buffers on first use: per-column coefficient records, Gram
  // matrix and Hessenberg matrix in the stored (not exactly normalised) basis, scales, partials; nu = norms of the stored
  // basis vectors of the latest sweep (pinned, written by the device; all one after any other kind of sweep)
  double2* op_gram = nullptr;
  double2* op_hhat = nullptr;
  double2* op_part[2] = {nullptr, nullptr};
  double* op_svals = nullptr;
  double* op_nu_dev = nullptr;
  double* h_nu = nullptr;
  double* nu_map = nullptr;
  int n_onepass = 0, n_onepass_redone = 0;   // one-pass sweeps so far / of those, done again with the two-pass sweep (norm drift)
  bool nu_valid = false;        // the latest sweep left nu != 1: combination coefficients are divided by it
  double2* q(int i) const { return Q + (size_t)i * n; }
};

struct qp_cheby {
  qp_ctx* ctx;
  int64_t n;
  double2* bufA = nullptr;
  double2* acc = nullptr;
  double2* bufC = nullptr;     // two more term vectors for the two-term walk (allocated on first use): a pair writes v_m and v_{m+1}
  double2* bufD = nullptr;     //   to vectors that nobody reads during the launch
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
  std::vector<cplx> Hess, R, P, Rn, ritz;   // host work arrays of a step, kept between calls
  std::vector<qp::ScaledProd> leja_prod;            // head of each Leja candidate's product chain (built while the columns arrive)
  double radius = 0;
  int n_a = 0, n_leja = 0, restarts = 0;
};


inline double2 d2(cplx z) { return make_double2(z.real(), z.imag()); }
inline double2 d2(qp_c128 z) { return make_double2(z.re, z.im); }
inline cplx cx(qp_c128 z) { return cplx(z.re, z.im); }

inline int use(qp_ctx* ctx) {
  if (ctx->closed) return qp::fail(QP_E_BAD_ARG, "the context of this handle has been destroyed");
  QP_HIP(hipSetDevice(ctx->device));
  return QP_OK;
}

template <class T>
inline int dev_alloc(T** p, size_t count) {
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
inline cplx sum_partials(const double2* h) {
  double re = 0, im = 0;
  for (int i = 0; i < kRedBlocks; ++i) {
    re += h[i].x;
    im += h[i].y;
  }
  return cplx(re, im);
}

inline int dot_sync(qp_ctx* ctx, const double2* x, const double2* y, int64_t n, cplx* out) {
  QP_CHECK(qp::launch_dot_partials(ctx->stream, x, y, ctx->d_part, n, &ctx->stats));
  QP_HIP(hipMemcpyAsync(ctx->h_part, ctx->d_part, kRedBlocks * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
  QP_HIP(hipStreamSynchronize(ctx->stream));
  *out = sum_partials(ctx->h_part);
  return QP_OK;
}


// ---- shared between the engine translation units -------------------------------------
// CSR-ordered mirror of the operator (index arrays; `gather`: also the current values)
int operator_csr_mirror(qp_operator* op, bool gather = true);
// row order in which the batched (SpMM) kernel visits the rows for a panel of `batch` states
int operator_spmm_order(qp_operator* op, int batch, const int32_t** order_out);
int operator_spmm_tiles(qp_operator* op, const qp::SpmmTiles** out);
// which terms of a cheby! touch the Psi accumulator (include/qprop.h, qp_acc_defer)
void acc_schedule(const double* a, int n_coeffs, bool defer, qp_acc_defer* out);
void set_defer(qp::ChebyEpi& e, const qp_acc_defer* d);
