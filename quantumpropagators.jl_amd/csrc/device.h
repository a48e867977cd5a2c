// Device-side data structures and kernel launch declarations (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include "qprop_internal.h"

namespace qp {

#define QP_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      (void)hipGetLastError(); /* reported here: do not leave it for a later hipGetLastError() */ \
      return qp::fail(QP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                      __LINE__);                                                             \
    }                                                                                        \
  } while (0)

constexpr int kRB = 64;           // rows per row block = one wavefront
constexpr int kRedBlocks = 256;   // fixed grid of the reduction kernels (deterministic order)
constexpr int kThreads = 256;

struct Stats;
struct Tuning;

// ---- device view of one sparse operator -------------------------------------------
struct DevMatrix {
  int format = QP_FMT_RBCSR;
  const Tuning* tun = nullptr;   // the owning context's knobs (never null once the operator exists)
  int64_t nrows = 0, ncols = 0, nnz = 0, stored = 0;
  // RBCSR: 64-row blocks, element (r,k) of block b at bptr[b] + k*64 + r;
  // cols packed 4 per lane: cols4[(bptr[b]>>2) + (k>>2)*64 + r].{x,y,z,w}
  int64_t nblocks = 0;
  int64_t* bptr = nullptr;    // nblocks+1
  int32_t* cols = nullptr;    // CSR: plain int32.  Row-block formats: a byte stream; block b's
                              // column section starts at byte cmeta[b] >> 2; cmeta[b] & 3 = 0: quad-packed
                              // int32 columns, 1: quad-packed int16 deltas to the lane's row, 2: stencil
                              // block, one int32 delta per slot for all 64 rows, 3: block map, one column
                              // block per slot for all 64 rows + the lane inside it, a byte per entry
  int64_t* cmeta = nullptr;   // nblocks (row-block formats)
  int64_t colbytes = 0;
  double2* vals = nullptr;    // stored, the current values (always valid)
  const double* vals_r = nullptr;  // non-null: every value is real and this copy of the real parts is
                                   // what the SpMV kernels stream (half the matrix bytes)
  // CSR
  int64_t* rowptr = nullptr;  // nrows+1
  int lanes_per_row = 16;     // CSR kernel: sub-wave width
  // HRB (Hermitian-packed row blocks): bptr/cols/vals hold the entries with col >= row;
  // the entries with col < row are (lcols, lpos): column and the position in `vals` of
  // the transposed entry, whose complex conjugate is the value.  Same block layout.
  int64_t* lptr = nullptr;    // nblocks+1
  int32_t* lcols = nullptr;   // byte stream like `cols`, indexed through lcmeta
  int64_t* lcmeta = nullptr;  // nblocks
  int64_t lcolbytes = 0;
  int32_t* lpos = nullptr;    // lstored, quad packed; -1 = padding
  int64_t lstored = 0;
  // HRB strip walk (kernels_walk.hip): valid when the operator is a lattice whose row blocks repeat one stencil
  const struct WalkPlan* walk = nullptr;
  // column-blocked mirror of an operator with irregular columns (kernels_colblock.hip); used for whole-operator launches
  const struct ColBlockPlan* cb = nullptr;
  // value-dictionary mirror of a row-block operator whose blocks hold few distinct values (kernels_coded.hip)
  const struct CodedVals* cv = nullptr;
  // QP_FMT_MATFREE: no stored entries; y = beta y + alpha A x is delegated to the owner
  // (engine_liouville.hip), and the Chebyshev term runs it followed by an unfused epilogue
  void* matfree = nullptr;
  int (*matfree_apply)(hipStream_t s, void* self, const double2* x, double2* y, double2 alpha, double2 beta,
                       Stats* st) = nullptr;
  double2* (*matfree_scratch)(void* self) = nullptr;   // n entries of workspace for the Chebyshev term
  // optional: the fused Chebyshev term of the owner's own kernel (engine_pauli.hip); without it the term is apply + epilogue
  int (*matfree_cheby)(hipStream_t s, void* self, const double2* x, const struct ChebyEpi& e, Stats* st) = nullptr;
};

// ---- strip walk over a lattice operator (Hermitian-packed format) -----------------------------------------------
// A run of row blocks [R0, R1) that all carry the same stencil: upper section
//   [z0 slots at distance 0 (the diagonal)] [nn near distances 0 < d_1 < ... < d_nn <= 16] [K far distances m g, m = 1..K]
//   [pads], lower section its mirror image [-K g ... -g] [-d_nn ... -d_1]; S = ceil(g / 64) column chunks per strip step.  Inside the run the position of
// every value is a formula (U0 + (b - R0) ustride + 64 slot + lane), and a wavefront that WALKS down one strip column --
// row blocks b, b + S, b + 2 S, ... -- finds everything a block needs beyond its own streams in what it loaded for the
// blocks before: the gathered elements x[r + m g] are the row-local elements of the blocks m steps ahead / behind (a ring
// of 2 K + 1 registers, one new load per step), the conj-transposed values of the far lower entries are the far upper
// values it streamed m steps ago (a FIFO in LDS), the near gathers and the near conj-transposed values are lane shifts
// of the block's own element / values, staged through a per-wavefront LDS window with a halo of the neighbouring block.
// Blocks outside [W0, R1) (W0 = R0 + K S: the first blocks whose history lies inside the run; the periodic wrap-around,
// a ragged end) are listed in edge_map and take the per-block code path in the same launch.
constexpr int kWalkMaxNear = 8;
constexpr int kWalkHalo = 16;      // largest near distance
struct WalkPlan {
  int valid = 0;
  int nn = 0, K = 0, z0 = 0;  // shape of the stencil (see above)
  int S = 0;                  // 64-row column chunks per strip step: ceil(g / 64)
  int xl = 0;                 // 1: one more pair of distances +- glong beyond the ring's reach (loaded directly); 2: two, +- glong1 and +- glong
  int64_t glong = 0;          // the longest distance of the stencil
  int64_t glong1 = 0;         // xl = 2: the shorter long distance, K g < glong1 < glong
  int fd = 0;                 // 1: diagonal far neighbours -- the far distances of strip step m are m g - 1, m g, m g + 1 (three slots per step)
  int64_t g = 0;              // rows per strip step (the far distances are g, 2 g, .., K g); need not be a multiple of 64
  int near[kWalkMaxNear] = {0};
  int64_t R0 = 0, R1 = 0, W0 = 0;
  int64_t U0 = 0;             // bptr[R0]
  int ustride = 0;            // stored upper values per row block (64 x padded width)
  int32_t* edge_map = nullptr;   // device: the blocks outside [W0, R1)
  int64_t n_edge = 0;
};

// ---- column-blocked mirror (kernels_colblock.hip) ----------------------------------------------------------------------
// An operator whose columns are irregular (no lattice, no band: src/generators.jl:634-645 allows any sparse H_l) gathers
// x[col] from all over the vector; at N = 2^20 the vector is 16 MB against the 4 MB of L2 an XCD has, so nearly every
// gathered element is its own line fetched from the Infinity Cache (233 us per term).  The mirror holds
// the same entries grouped by (row tile, column block): tiles of 64 rpt rows, blocks of 2^log2w columns; the segment of
// (tile t, block c) lists the entries of the tile's rows whose column lies in block c, row by row, columns ascending.  The
// kernel walks the column blocks in the OUTER loop -- every wavefront is resident from the start and owns its tiles for the
// whole launch -- so at any time the whole chip gathers from one 2^log2w-element window of x that every XCD's L2 holds.
struct ColBlockPlan {
  int valid = 0;
  int log2w = 17;               // columns per block = 1 << log2w  (2 MB of x)
  int P = 0;                    // column blocks
  int rpt = 2;                  // 64-row groups per tile
  int max_seg = 0;              // entries of the longest segment (the per-wavefront LDS buffer holds one segment)
  int64_t ntiles = 0, nnz = 0;
  int32_t* segptr = nullptr;    // device [ntiles P + 1]: first entry of segment t P + c
  uint16_t* rowoff = nullptr;   // device [ntiles P (64 rpt + 1)]: first entry of every row inside its segment
  uint32_t* cols = nullptr;     // device [nnz]
  int64_t* map = nullptr;       // device [nnz]: position in the operator's value array (-(position) - 1: its complex conjugate)
  double2* vals = nullptr;      // device [nnz]: the current values in mirror order (refreshed when the operator's values change)
  double* vals_r = nullptr;     // device [nnz]: their real parts, streamed instead when every value is real
  int use_real = 0;
};
// ---- value-dictionary mirror (kernels_coded.hip) ------------------------------------------------------------------------
// The reference's typical generator is a sum of a few structured terms (src/generators.jl:634-645): a spin chain's couplings, a
// grid's hopping amplitudes -- a handful of distinct numbers, stored 16 (or 8) bytes per entry.  When every 64-row block of a
// row-block operator holds at most 256 distinct TUPLES (value in term 1, .., value in term L) over its stored positions, the
// operator gets a mirror: one byte per stored position (`codes`, quad-packed like the column sections: byte j of dword
// (bptr[b] >> 2) + 64 q + lane = slot 4 q + j of that lane's row) and per block a table of its tuples -- tables with the same
// content shared between blocks.  The mat-vec kernels read the COMBINED table tab[(tptr[b] >> 9) + code] = sum_l c_l tuple_l: the
// same arithmetic on the same numbers as the value plane they replace (bit-identical results), 1 B + a cached table line per
// entry instead of 16 B.  evaluate! (src/generators.jl:757-766) recombines the tables IN ADDITION to the value plane (operator_refresh,
// engine_core.hip): the plane stays the source of truth for the consumers that read it (Arnoldi mat-vecs without a dictionary path,
// the panel kernels, qp_operator_get_csr, the split), so the mirror saves bytes per mat-vec, not work per evaluate!.
struct CodedVals {
  int valid = 0;
  int64_t ntab = 0;               // table entries (all blocks, shared tables counted once)
  int64_t ntables = 0;            // distinct tables
  uint8_t* codes = nullptr;       // device [stored]
  int64_t* tptr = nullptr;        // device [nblocks]: (first table entry of the block << 9) | its number of entries (1 .. 256)
  double2* tab = nullptr;         // device [ntab]: the current combined table (the only term's own table when there is one term, scale 1)
  double* tab_r = nullptr;        // device [ntab]: its real parts, read instead when every value is real
  int use_real = 0;
};
constexpr int kCbMaxTilesPerWave = 8;
constexpr int kCbMaxSeg = 1024;

// epilogue of the fused Chebyshev term (see qp_cheby_term in qprop.h)
struct ChebyEpi {
  const double2* xloc;   // x + xoff  (row-local element of the gathered vector)
  const double2* v0;     // nullable
  double2* vout;         // nullable, may alias v0
  const double2* acc_in; // nullable
  double2* acc_out;
  double2 c;
  double beta, a_prev, a;
  double2 phase;
  int apply_phase;
  double* check_partials;  // nullable: per-workgroup {Re<v1,t>, Im<v1,t>, |v1|^2}
  // fused pack for the multi-GPU exchange: rows with mirror[slot] >= 0 also store the new
  // term vector into slab[mirror[slot]] (slot = 64 * position-in-row-set + lane)
  const int32_t* mirror = nullptr;
  double2* slab = nullptr;
  // deferred accumulation of Psi: the row epilogue of term m has v_{m-2} (v0), v_{m-1}
  // (xloc) and v_m at hand, so Psi += a_{m-2} v_{m-2} + a_{m-1} v_{m-1} + a_m v_m can be one
  // read + one write of the accumulator every third term -- the same FMA sequence
  int acc_skip = 0;      // 1: this term does not touch the accumulator
  int n_defer = 0;       // earlier terms folded into this update (0, 1 or 2)
  double a_d1 = 0.0;     // coefficient of v_{m-1}
  double a_d2 = 0.0;     // coefficient of v_{m-2}
};

// a subset of the 64-row blocks of an operator (device list), optionally with a mirror map
// In-launch dependency between the boundary launch of term m (side stream) and the
// interior launch of term m+1 (main stream), replacing a cross-queue event wait that costs
// ~11 us of idle queue per term on gfx950 (profiles/r01/overlap_timeline.txt).
//   signal: every workgroup of the boundary launch adds 1 after an agent-scope release;
//   wait:   workgroups at position >= wait_from_wg of the interior launch (the row blocks
//           adjacent to boundary rows, listed last) poll until *wait >= wait_target, then
//           acquire at agent scope (cdna_hip_programming.md Guideline 16 recipe).
struct SyncArgs {
  unsigned* signal = nullptr;
  const unsigned* wait = nullptr;
  unsigned wait_target = 0;
  unsigned wait_from_wg = 0;
  unsigned* timeout_flag = nullptr;
  unsigned spin_limit = 1u << 28;   // polls (with s_sleep 8) before a wait gives up: ~1 min (knob split_spin_log2)
};

struct RowSet {
  const int32_t* block_map = nullptr;  // nullptr = all blocks
  int64_t nmap = 0;
  bool count = true;                   // count this launch as a mat-vec in the stats
  SyncArgs sync;
  // the same set of blocks as a strip walk (interior launch of a split term; engine_cheby.hip: qp_split_create): a plan
  // whose walkable run lies inside the set and never reads a row of the complementary set, its edge list = the rest of the
  // set.  Only the edge blocks can depend on the other launch: they wait (sync.wait), the walk itself does not.
  const struct WalkPlan* walk = nullptr;
  int reserve_cu = 0;                  // compute units the walk leaves free for what runs beside it (the other launch, the collective)
};

struct PlainEpi {
  double2* y;
  double2 alpha, beta;
  int beta_zero;
  // Arnoldi column with the previous column's "norm + scale" folded in (src/arnoldi.jl:89-96 applied on the
  // fly): x is the UNNORMALISED q_j; h = |x| comes from the kRedBlocks partials of |x|^2 that the projection
  // kernel left (every workgroup re-reduces them in its prologue, fixed order), the row sum is scaled by 1 / h
  // (unless h < norm_min: dimensionality exhausted, nothing is scaled), the row's own element x_i / h goes to
  // the basis vector qn_out, and workgroup 0 records Hess[j, j-1] = dt h and the norm for the host.
  const double2* norm_part = nullptr;
  const double2* xloc = nullptr;   // row-local element of x (x + xoff)
  double2* qn_out = nullptr;
  double2* hess_slot = nullptr;
  double* norm_slot = nullptr;
  double dt = 1.0, norm_min = 0.0;
  unsigned* flag = nullptr;        // host-visible: set to flag_value once hess_slot / norm_slot are written
  unsigned flag_value = 0;
};

struct Stats {
  uint64_t n_matvec = 0, n_cheby_steps = 0, n_newton_steps = 0, n_restarts = 0, n_launch = 0, n_graph_launch = 0;
  double spmv_bytes = 0;
};

int launch_spmv_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, Stats* st,
                      const RowSet* rs = nullptr);
int launch_spmv_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, Stats* st);
// kernels_arnoldi.hip: the Arnoldi column's mat-vec with the multidot (c_k = <q_k|w>, Gram row <q_k|q_j>, k <= j) in its
// epilogue; partials in the multidot's layout.  *launched = false: no instance (format, column index): nothing was done
int launch_arnoldi_matvec_dots(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, const double2* Q,
                               int64_t ldq, int j, double2* partials, bool* launched, Stats* st);
// kernels_onepass.hip: the sweep that reads the basis once per column (knob arnoldi_mode = 2)
bool arnoldi_onepass_fits(const DevMatrix& A, int m, int nvec);
int op_part_slots(int nvec);
int launch_arnoldi_onepass_sweep(hipStream_t s, const DevMatrix& A, const double2* start, double s0, double2* Q, int64_t ldq,
                                 double2* const a_buf[2], int m, int nvec, double2* const part[2], double2* gram, double2* hhat,
                                 double* svals, double* nu_dev, double dt, double2* hess_map, double* norms_map, double* nu_map,
                                 unsigned* flags_map, unsigned flag_value, Stats* st);
// the strip walk for the fused Chebyshev term of a whole Hermitian-packed lattice operator; *launched = false when the
// plan's shape has no kernel instance (the caller then takes the per-block kernel)
// is there a strip-walk kernel instance for this stencil shape?  (The dispatch of kernels_walk_impl.h: launch_shape instantiates
// exactly these; inline here so that the host planners -- and their sanitizer build, tests/sanitize_host_index.cpp -- see the same list.)
// near distances 1..4 of at most 16 rows, far reach 1..4 strip steps, with or without a diagonal
// ... and, with one or two long pairs beyond the ring (xl = 1, 2), near 1..2 and one or two far distances
// ... and, with diagonal far neighbours (fd = 1: m g - 1, m g, m g + 1), near 1..2, one strip step and at most one long pair
inline bool walk_shape_supported(int nn, int K, int z0, int xl = 0, int fd = 0) {
  if (fd) return fd == 1 && (xl == 0 || xl == 1) && K == 1 && nn >= 1 && nn <= 2 && (z0 == 0 || z0 == 1);
  if (xl) return (xl == 1 || xl == 2) && nn >= 1 && nn <= 2 && (K == 1 || K == 2) && (z0 == 0 || z0 == 1);
  return nn >= 1 && nn <= 4 && K >= 1 && K <= 4 && (z0 == 0 || z0 == 1);
}
// the two-term strip walk (kernels_walk2.hip): both terms of a pair (m, m + 1) on the two-term region of plan `P2`, term m of its edge list
inline bool walk2_shape_supported(int nn, int K, int z0) {
  return (z0 == 0 || z0 == 1) && nn >= 1 && nn <= 4 && K >= 1 && K <= 4;
}
int launch_hrb_walk2_cheby(hipStream_t s, const DevMatrix& A, const WalkPlan& P2, const double2* x, const ChebyEpi& e1,
                           const ChebyEpi& e2, const Tuning& tun, bool* launched);
int launch_hrb_walk_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, const Tuning& tun,
                          bool* launched, const RowSet* rs = nullptr);
// kernels_dense.hip (QP_FMT_DENSE: CSR arrays with a complete pattern, i.e. vals / vals_r is the row-major dense matrix)
int launch_dense_gemv_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, Stats* st);
int launch_dense_gemv_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, Stats* st);
int launch_dense_zgemm_cheby(hipStream_t s, const DevMatrix& A, const double2* X, int batch, const ChebyEpi& e, Stats* st);
// kernels_coded.hip: the row-block mat-vec through the value-dictionary mirror (A.cv valid); grid / row-set arguments as launch_spmv
int launch_rbcsr_coded_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, int64_t nblk,
                             const int32_t* bmap, const SyncArgs& sy, bool wide);
int launch_rbcsr_coded_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, int64_t nblk,
                             const int32_t* bmap, const SyncArgs& sy);
// kernels_colblock.hip: *launched = false when the mirror does not apply to this launch (the caller then takes the format's kernel)
int launch_colblock_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, const Tuning& tun, bool* launched);
int launch_colblock_plain(hipStream_t s, const DevMatrix& A, const double2* x, const PlainEpi& e, const Tuning& tun, bool* launched);
// vals[p] = src[map[p]] (conjugated for a negative map entry); real != NULL: also real[p] = Re vals[p]
int launch_colblock_gather(hipStream_t s, const ColBlockPlan& P, const double2* src, Stats* st);
// the formats whose value array is in CSR order (rowptr / cols / vals[p])
inline bool csr_layout(int format) { return format == QP_FMT_CSR || format == QP_FMT_DENSE; }
int spmv_grid_size(const DevMatrix& A);
// dense row-sum kernel (kernels_dense.hip): rows per wavefront and the grid that follows from it -- shared with spmv_grid_size,
// which sizes the per-workgroup partials of check_normalization
inline int dense_gemv_rows_per_wave(int64_t nrows) { return nrows >= 16384 ? 4 : 1; }
inline int dense_gemv_grid(int64_t nrows) {
  const int64_t per_wg = (int64_t)(kThreads / 64) * dense_gemv_rows_per_wave(nrows);
  return (int)((nrows + per_wg - 1) / per_wg);
}
// Developer knobs for A/B measurements.  Every context carries its own copy (qp_ctx::tun, set with
// qp_ctx_tuning_set); qp_tuning_set only changes the defaults that contexts created afterwards start
// from, so handles driven from different threads never observe each other's switches.
// fixed choices that were knobs while they were being measured (docs/history/): the density from which AUTO lays an operator out dense,
// the resident wavefronts per CU the column-blocked mirror is sized for (24 and 32 measured slower), the compute units an interior strip
// walk leaves to the boundary launch and the collective's kernel
constexpr int kDenseMinDensityPct = 75;
constexpr int kCbWavesPerCu = 16;
constexpr int kCbMinLog2N = 20;     // column-blocked mirror: smallest number of columns (log2) it is built for (measured: 2^18 columns 0.9 x, 2^19 1.04 x, 2^20 1.37 x, 2^21 1.62 x, 2^22 1.52 x)
constexpr int kWalkReserveCu = 8;
constexpr int kWalkEdgeSteps = 4;   // strip walk: a wavefront that also takes an edge block walks this many steps less (a block on the per-block path is three dependent rounds of loads; a step of the walk takes about one)

struct Tuning {
  int rbcsr_variant = 15;     // bit 0 nt matrix loads, bit 1 early row-local loads, bit 2 deeper unroll, bit 3 (Hermitian-packed kernel) all loads of an all-stencil block up front (A/B in profiles/)
  int arnoldi_mode = 1;       // 0 = sequential fused MGS passes, 1 = low-synchronisation MGS
  int arnoldi_onepass = 1;    // newton!'s sweeps read the basis ONCE per column (kernels_onepass.hip): 0 never, 1 when basis + matrix exceed the Infinity Cache (bytes, not latency, then bound the sweep: profiles/r05/newton_onepass.txt), 2 wherever an instance exists, 3 = 2 with every sweep done again in the two-pass form (exercises the norm-drift fall-back of engine_krylov.hip)
  int sparse_controls = 1;    // 1 = evaluate! rewrites only the positions of sparse trailing control terms (see qp_operator::sparse_from)
  int lattice_fill = 1;       // 1 = rows of a lattice operator that lack a few of its distances (open boundaries of a grid) are completed with explicit zeros
  int arnoldi_fuse_dots = 1;  // 1 = the multidot of a column runs in its mat-vec's epilogue where an instance exists (row-block format, j <= 19): 2 launches per column
  int split_spin_log2 = 28;   // in-launch hand-off: a polling workgroup gives up after 2^this polls (~1 min) and raises the split's time-out flag
  int split_dbg = 0;          // tests only: 1 = the boundary launches do not signal (every polling workgroup runs into the time-out)
  int split_mode = 2;         // boundary -> interior hand-off: 0 = cross-stream events, 1 = in-launch counter, 2 = the counter when at most 256 workgroups poll
  int liouville_tile32_min_n = 260;  // matrix-free Liouvillian: n in [this, liouville_tile32_n] takes the 32 x 32 matrix-core
  int liouville_tile32_n = 2048;     //   kernel; other n <= liouville_fused_n the 16 x 16 one; the rest library GEMMs
  int liouville_fused_n = 320;  // matrix-free Liouvillian: largest n that takes the fused matrix-core kernel (else library GEMMs)
  int real_vals = 1;          // operator refresh: stream a real copy of the values when they are all real
  int block_map = 1;          // operator build: encode blocks whose slots each map the 64 rows into one 64-aligned column block (row XOR mask: qubit-register Hamiltonians) with one byte of index per entry
  int stencil = 1;            // operator build: encode blocks with block-wide column distances as stencil blocks
  int acc_defer = 1;          // qp_cheby_step: touch the Psi accumulator every third term only (1) or every term (0)
  int cheby_graph = 0;        // qp_cheby_step: replay a repeated step as a hipGraph when the mat-vec grid has at most this many workgroups (0: off; measured: no gain)
  int roctx = 0;              // 1 = named profiler ranges around the steps' phases (qprop_internal.h: ScopedRange); also QP_ROCTX=1
  int dense_auto = 1;         // 1 = AUTO lays an operator out dense (QP_FMT_DENSE) when at least 75 % (kDenseMinDensityPct) of its positions are stored
  int dense_panel_mfma = 1;   // 1 = the batched step of a dense operator runs H X on the fp64 matrix cores (kernels_dense.hip); 0 = the sparse panel kernels (A/B)
  int colblock = 1;           // 1 = an operator with irregular columns whose vector outgrows the L2 gets a column-blocked mirror (kernels_colblock.hip); 2 = any row-block / CSR operator that fits the mirror's limits (tests); 0 = off
  int cb_log2w = 0;           // ... columns per block (log2); 0 = about sixteen blocks, 2^16 ... 2^18 columns each (1 - 4 MB of the vector; an XCD's L2 holds 4 MB)
  int small_nnz = 8192;       // qp_propagate: register-resident Cheby systems up to this nnz run as ONE persistent launch (0: off)
  int newton_pipeline = 1;    // newton!: Hessenberg eigenvalues overlap the Arnoldi sweep
  int spmm_rows = 1;          // batched SpMM: wave-per-row kernel (lane = state) for panels of more than 32 states (0: always the state-tiled kernel)
  int spmm_rw = -1;           // batched SpMM, wave-per-row kernel: -1 = 4 x 4 tiles of a lattice operator's rows staged in LDS where the pattern allows (SpmmTiles), else as 0; 0 = matrix entries through the scalar unit (one row per wavefront), 1 / 2 / 4 / 8 = entries one per lane + readlane broadcast, that many rows per wavefront
  int spmm_strip = 0;         // batched SpMM row walk: inner-index strip width (0 = chosen from the L2 size; -1 = natural row order)
  int hrb_walk = 1;           // Hermitian-packed fused term of a whole lattice operator: the strip-walk kernel (kernels_walk.hip) when the operator has a walk plan
  int walk_waves = 0;         // strip walk: wavefronts the walk is cut into (0: 768 for an operator that fits the Infinity Cache, else 8 per CU on every CU the edge workgroups leave free -- 1856 for the headline lattice --, or 2048 with the edge blocks inside the walk; the two-term walk: one per SIMD = 4 per compute unit)
  int walk_pair = -1;         // strip walk, two Chebyshev terms per pass over the values (kernels_walk2.hip): -1 = for operators beyond the Infinity Cache whose strip columns are long enough (>= 24 steps per wavefront), 0 never, 1 wherever a plan exists (its cut: walk_waves)
  int value_dict = 1;         // value-dictionary mirror of row-block operators with few distinct values per block (0: never built / used)
  int walk_nt = -1;           // strip walk: nontemporal accesses (-1: the matrix values when the operator does not fit the Infinity Cache; bit 0 matrix values; bits 1, 2: vector loads, stores -- measurement variants of the headline shape)
  int walk_dbg = 0;           // strip walk, measurements only: 1 = in-wave edge block after the walk instead of before it, 2 = edge blocks skipped (WRONG results), 4 = never as workgroups of their own
  int walk_min_blocks = 3072; // strip walk: smallest number of walkable row blocks for which the plan is used
  int spmm_nt = 1;            // nontemporal matrix / row-local streams in the batched SpMM kernel: 0 never, 2 always, 1 for large panels
};
// compute units of the CURRENT device (hipGetDevice; asked once per device and cached; 256 if the runtime will not say)
int device_cu_count();
// address of the knob called `key` inside `t`, or nullptr
int* tuning_field(Tuning& t, const char* key);

// small coefficient vectors are passed by value in the kernel-argument segment
constexpr int kCoefBlock = 32;
struct CoefBlock {
  double2 c[kCoefBlock];
};

// planes: vals[p] = sum_l coef[l] * plane_l[p]   (coefs: host array)
int launch_real_part(hipStream_t s, double* out, const double2* v, int64_t n, Stats* st);
int launch_combine_planes(hipStream_t s, double2* vals, const double2* const* planes_dev, const double2* coefs,
                          int nplanes, int64_t n, double* vals_r, Stats* st);
// vals[support[i]] = base[support[i]] + sum_l coefs[l] * support_vals[l * n_support + i]  (and its real part into vals_r)
int launch_sparse_planes_update(hipStream_t s, double2* vals, const double2* base, const int32_t* support, int64_t n_support,
                                const double2* support_vals, int nplanes, const double2* coefs, double* vals_r, Stats* st);


// batched states: CSR SpMM with the fused Chebyshev epilogue, panel X[i*b + s]
int launch_spmm_cheby(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals,
                      const double2* X, int64_t nrows, int64_t nnz, int b, const ChebyEpi& e, const Tuning& tun,
                      bool rows_kernel, const int32_t* order, Stats* st);
// Tiles of the batched path.  A lattice operator's interior rows all carry the same entries (row + d for the nd distances d of
// the pattern, each either near, |d| <= NN <= 4, or far, d = m g with |m| <= K <= 4).  A 4 x 4 patch of such rows -- rows
// r0 + i g + j, i, j < 4 -- reads 17 x 16 panel rows of which only 16 + 8 K + 8 NN are distinct: the workgroup stages those once
// in LDS (one KiB per row: 64 states) and its sixteen wavefronts, one per row as in the row kernel, take their operands from
// there.  What decides the speed of the row kernel is the bytes it pulls out of L2 (17-21 KiB per row at the L2's ~18 TB/s:
// tools/probe/panel_tile_probe.hip, profiles/r06/panel_tile_probe.txt); the tile pulls 5 + the row-local streams.
constexpr int kSpmmTileMaxEntries = 24;   // entries per row (2 K + 2 NN + 1 = 17 at most; a multiple of 8: the sums go in groups of eight)
constexpr int kSpmmTileSlots = 80;        // staged rows at most (K = NN = 4): 80 KiB of LDS, two workgroups per compute unit
struct SpmmTileShape {
  int nd = 0, K = 0, NN = 0;
  int dfar[kSpmmTileMaxEntries] = {0};    // entry k: strip steps m (d = m g) when dnear[k] == 0 ...
  int dnear[kSpmmTileMaxEntries] = {0};   // ... else the near distance d
};
// the kernel's table (device, int32): [0, 80) row of staged slot s relative to the tile's first row r0; then for wavefront w = 4 i + j
// and entry k the LDS byte offset of that entry's operand (24 per wavefront); then the wavefront's own row relative to r0 (16) and
// the LDS byte offset of its own element (16)
constexpr int kSpmmTileTab = kSpmmTileSlots + 16 * kSpmmTileMaxEntries + 32;
struct SpmmTiles {
  bool built = false;
  int valid = 0;
  int knob = 0;
  SpmmTileShape shape;
  int64_t g = 0, sw = 0;
  int32_t* tiles = nullptr;   // device: first row r0 of every tile, in the order of the row walk (strips of sw columns, tiles of a strip step side by side)
  int64_t ntiles = 0;
  int32_t* rest = nullptr;    // device: the rows outside the tiles (edges of the lattice, ragged ends): row kernel
  int64_t nrest = 0;
  int32_t* tab = nullptr;     // device: kSpmmTileTab entries (above)
  int T = 0;                  // staged rows: 16 + 8 K + 8 NN
};
int launch_spmm_tile_cheby(hipStream_t s, const int64_t* rowptr, const int32_t* cols, const double2* vals, const double2* X,
                           int64_t nrows, int64_t nnz, int b, const ChebyEpi& e, const Tuning& tun, const SpmmTiles& P, Stats* st);
// panels of more than 32 states take the wave-per-row kernel (lane = state) unless knob spmm_rows is 0
inline bool spmm_uses_rows_kernel(const Tuning& tun, int b) { return tun.spmm_rows != 0 && b > 32; }
// ---- small systems: the whole Cheby time grid in one single-workgroup launch -------------
struct SmallObs {
  const int64_t* rowptr;
  const int32_t* cols;
  const double2* vals;
};
struct SmallArgs {
  int64_t n = 0, nnz = 0;
  int lanes = 1;                       // lanes per row (power of two <= 64)
  int ent = 1;                         // entries per lane per row
  int rows_per_group = 1;              // rows per lane group;  rows_per_group * ent <= kSmallEpt
  int obs_lanes = 1;                   // lanes per row for the observables
  const int64_t* rowptr = nullptr;     // CSR mirror of the operator
  const int32_t* cols = nullptr;
  const int64_t* map = nullptr;        // position in a plane; negative: conj of plane[-m-1]
  const double2* const* planes = nullptr;
  int nops = 0, ncoeffs = 0;
  double2 scale = {1.0, 0.0};
  const double2* table = nullptr;      // [nsteps * ncoeffs]
  int nsteps = 0;
  const double* a = nullptr;           // Chebychev coefficients
  int n_coeffs = 0;
  double2 c = {0.0, 0.0};
  double beta = 0.0;
  double2 phase = {1.0, 0.0};
  double2* psi = nullptr;              // in / out
  int nobs = 0;
  const SmallObs* obs = nullptr;
  double2* expvals = nullptr;          // [(nsteps + 1) * nobs]
  double2* states = nullptr;           // [(nsteps + 1) * n] or null
  int check = 0;
  double limit = 0.0;
  int* fail = nullptr;                 // {flag, step, term}
};
constexpr int kSmallThreads = 512;
constexpr int kSmallEpt = 16;          // register slots per lane of the persistent Chebychev kernel
constexpr int kSmallEptArnoldi = 32;   // ... of the persistent Arnoldi kernel (fewer live values per slot)
constexpr int64_t kSmallLdsRows = 2048;
// arnoldi! (src/arnoldi.jl:74-100) for a register-resident operator: all m columns in one
// single-workgroup launch, Krylov basis in LDS (and written to Q for the caller)
struct SmallArnoldiArgs {
  int64_t n = 0;
  int lanes = 1, ent = 1, rows_per_group = 1;
  const int64_t* rowptr = nullptr;
  const int32_t* cols = nullptr;
  const int64_t* map = nullptr;
  const double2* vals = nullptr;       // current values of the operator, device layout
  const double2* start = nullptr;      // q_0
  double2* Q = nullptr;                // [m + 1][n]
  double2* hess = nullptr;             // column major, leading dimension ldd (zeroed by the caller)
  double* norms = nullptr;             // [ldd]
  int ldd = 0, m = 0, extended = 0;
  double dt = 1.0, norm_min = 0.0;
  int normalize_start = 0;             // q_0 = start / |start|, |start| -> norms[ldd - 1]
};
constexpr size_t kSmallLdsBytes = 152 * 1024;
inline bool small_arnoldi_fits(int64_t n, int m) {
  return sizeof(double2) * ((size_t)kSmallThreads / 64 + (size_t)(m + 2) * (size_t)n) <= kSmallLdsBytes;
}
int launch_arnoldi_small(hipStream_t s, const SmallArnoldiArgs& a, Stats* st);
bool small_plan(int64_t n, int64_t maxrow, SmallArgs* a, int max_slots = kSmallEpt);
int launch_cheby_propagate_small(hipStream_t s, const SmallArgs& a, Stats* st);

int launch_gather_csr_vals(hipStream_t s, double2* out, const double2* vals, const int64_t* map, int64_t nnz,
                           Stats* st);

// BLAS-1
int launch_fill(hipStream_t s, double2* x, double2 a, int64_t n, Stats* st);
int launch_scal(hipStream_t s, double2* x, double2 a, int64_t n, Stats* st);
int launch_axpy(hipStream_t s, double2 a, const double2* x, double2* y, int64_t n, Stats* st);
// partials[kRedBlocks] (double2): sum conj(x) y   (x == y gives |x|^2 in .x)
int launch_dot_partials(hipStream_t s, const double2* x, const double2* y, double2* partials, int64_t n,
                        Stats* st);

// Arnoldi building blocks.  `h_in` partials belong to the projection on q_prev that is
// applied before the new inner product is accumulated (fused axpy -> dot pass).
struct MgsArgs {
  double2* w;              // q_{j+1}, updated in place
  const double2* q_prev;   // nullable: projection to subtract first
  const double2* q_cur;    // nullable: next inner product; null => accumulate |w|^2
  const double2* part_in;  // partials of <q_prev, w> (kRedBlocks)
  double2* part_out;       // partials of <q_cur, w'> or |w'|^2
  double2* hess_prev;      // device Hess slot for dt*<q_prev,w> (nullable)
  double dt;
  int64_t n;
};
int launch_mgs_pass(hipStream_t s, const MgsArgs& a, Stats* st);
// low-synchronisation MGS of column j (see kernels_blas.hip): c = Q^H w and the Gram row in one
// pass, a small reduction whose last workgroup (ticket counter) solves for the MGS
// coefficients, then w -= Q h with |w|^2 partials: three launches
int launch_mgs_lowsync(hipStream_t s, const double2* Q, int64_t ldq, int j, double2* w, double2* md_partials,
                       double2* G, int ldg, double2* hess_col, double2* reduced, double2* coef, unsigned* ticket,
                       double2* norm_partials, double dt, int64_t n, Stats* st, bool solve_in_update = false,
                       unsigned* early_flag = nullptr, unsigned flag_value = 0, bool* early_armed = nullptr,
                       bool dots_done = false, bool l2_order = false);
// the same in pieces for row-partitioned runs: local sums -> (all-reduce by the caller) ->
// solve (one workgroup) + update
int launch_mgs_multidot(hipStream_t s, const double2* Q, int64_t ldq, int j, const double2* w, double2* md_partials,
                        double2* reduced, int64_t n, Stats* st);
int launch_mgs_project(hipStream_t s, const double2* Q, int64_t ldq, int j, double2* w, const double2* reduced,
                       double2* G, int ldg, double2* hess_col, double2* coef, double2* norm_partials, double dt,
                       int64_t n, Stats* st);
// the solve keeps the packed Gram triangle of columns 0..j in LDS (3 (j+1) + j (j+1)/2 complex numbers):
// up to j = 87 within the 64 KB a launch gets without opting in to more; longer bases continue with
// the sequential passes
inline bool mgs_lowsync_fits(int j) { return sizeof(double2) * (size_t)(3 * (j + 1) + j * (j + 1) / 2) <= 64 * 1024; }
// w *= 1/sqrt(sum part_in.x);  hess_slot = dt * norm
int launch_norm_scale(hipStream_t s, double2* w, const double2* part_in, double2* hess_slot, double dt,
                      int64_t n, Stats* st);
bool launch_combine2_vecs(hipStream_t s, double2* out1, int use_out1, int m1, const double2* coefs1, double2* out2, int m2,
                          const double2* coefs2, const double2* Q, int64_t ldq, double2* norm_partials, int64_t n,
                          Stats* st);
// out = (use_out ? s0*out : 0) + sum_{i<m} coef[i] * Q[i*ldq + k];  optional |out|^2 partials
int launch_combine_vecs(hipStream_t s, double2* out, int use_out, double2 s0, const double2* Q, int64_t ldq,
                        int m, const double2* coefs /* host */, double2* norm_partials, int64_t n, Stats* st);
// check_normalization finalize: reduce per-workgroup triples to one triple
int launch_reduce_triples(hipStream_t s, const double* partials, int nwg, double* out3, Stats* st);

}  // namespace qp
