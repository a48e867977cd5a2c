/* qprop.h -- C ABI of libqprop_hip.so, the MI355X (gfx950) engine behind
 * QuantumPropagators.jl's `init_prop / prop_step! / propagate` for the Chebyshev and
 * Newton/Arnoldi propagators.
 *
 * The reference is pure Julia: there is no existing FFI for this path.  The boundary it
 * sits behind is Julia dispatch on `init_prop(state, generator, tlist, ::Val{:Method})`
 * and `prop_step!(propagator)` (src/propagator.jl:208-264, :315-329); a backend package
 * extension (the pattern of ext/QuantumPropagatorsExponentialUtilitiesExt.jl:74-210)
 * would `ccall` the entry points below.  Each entry point cites the reference code it
 * replaces (paths relative to the reference checkout, v0.8.5+dev).  INTEGRATION.md shows
 * the Julia-side `ccall` stubs.
 *
 * Conventions
 *  - every function returns an int status (QP_OK == 0); no C++ exception crosses;
 *    qp_last_error() gives the thread-local message of the last failure.
 *  - qp_c128 is layout- and ABI-compatible with C `double _Complex` and Julia
 *    `ComplexF64` (two doubles, passed in two SSE registers on x86-64 SysV).
 *  - host arrays belong to the caller; the library copies at *_create / upload and
 *    never retains host pointers.  Device buffers belong to the handle unless wrapped
 *    with qp_state_wrap().
 *  - all device work of a context is enqueued on that context's HIP stream; calls that
 *    return host scalars synchronise that stream, all others are asynchronous.
 *  - handles are not thread-safe; distinct contexts may be used from distinct threads.
 *  - handles may be destroyed in ANY order, a context also before the handles created from it
 *    (a garbage collector's finalizers -- Python's, Julia's -- run in no particular order); a
 *    compute call on a handle whose context is gone returns QP_E_BAD_ARG.
 */
#ifndef QPROP_H
#define QPROP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } qp_c128;

typedef struct qp_ctx qp_ctx;           /* device + stream */
typedef struct qp_matrix qp_matrix;     /* one sparse matrix resident in HBM */
typedef struct qp_operator qp_operator; /* lazy sum  sum_l c_l H_l  (Generators.Operator) */
typedef struct qp_state qp_state;       /* one state vector, ComplexF64[n] in HBM */
typedef struct qp_krylov qp_krylov;     /* (m_max+1) Arnoldi vectors, contiguous */
typedef struct qp_cheby qp_cheby;       /* Cheby.ChebyWrk device workspace */
typedef struct qp_newton qp_newton;     /* Newton.NewtonWrk device workspace */

/* ---- status codes; the Julia glue maps them back to the reference's exceptions ---- */
enum {
  QP_OK = 0,
  QP_E_BAD_ARG = 1,            /* ArgumentError */
  QP_E_HIP = 2,                /* HIP runtime failure */
  QP_E_DT_MISMATCH = 3,        /* @assert abs(dt) ~ abs(wrk.dt)        src/cheby.jl:157 */
  QP_E_TOO_FEW_COEFFS = 4,     /* @assert length(a) > 1                src/cheby.jl:165 */
  QP_E_NORMALIZATION = 5,      /* "Incorrect normalization"            src/cheby.jl:196-199 */
  QP_E_MAX_RESTARTS = 6,       /* @assert s <= max_restarts            src/newton.jl:375 */
  QP_E_DIVDIFF_UNDERFLOW = 7,  /* "Divided differences too small"      src/newton.jl:209 */
  QP_E_NO_DEVICE = 8,          /* no MI355X visible: the product path has no CPU fallback */
  QP_E_ALLOC = 9,
  QP_E_INTERNAL = 10,
  QP_E_M_MAX = 11,             /* "Newton propagation requires m_max > 2" src/newton.jl:38-45 */
  QP_E_RCCL = 12               /* a collective of the library's own communicator failed (qp_comm_*) */
};

enum { QP_LAYOUT_CSR = 0, QP_LAYOUT_CSC = 1 };
enum { QP_VAL_C128 = 0, QP_VAL_F64 = 1 };
/* device storage format.  RBCSR = row-block CSR: 64-row blocks stored lane-interleaved so
 * that one wavefront streams a block with 1-KiB coalesced loads.  HRB = Hermitian-packed
 * row blocks: only entries with col >= row carry values; an entry with col < row stores
 * the position of its transpose and uses the complex conjugate (cheby! requires a
 * Hermitian H, src/cheby.jl:135), which removes ~40 % of the matrix bytes.  CSR = plain
 * CSR with a sub-wave per row.  AUTO: HRB when every term is exactly (bitwise) Hermitian,
 * else RBCSR; CSR when row-block padding would exceed 50 %.  An HRB operator that is given
 * complex coefficients re-lays itself out as RBCSR. */
enum { QP_FMT_AUTO = 0, QP_FMT_CSR = 1, QP_FMT_RBCSR = 2, QP_FMT_HRB = 3,
       QP_FMT_MATFREE = 4, /* reported by qp_operator_info for qp_liouvillian_create operators */
       /* DENSE: the union pattern of the terms is (made) complete and the values are stored row-major, 16 B per entry (8 B
        * for an all-real operator), no index bytes: the reference's own dense generators (test/test_cheby.jl:24-47,
        * test/test_newton.jl:53-65: N = 1000 Hermitian(rand(ComplexF64, N, N)); BASELINE configs[0]).  One state: a row-sum
        * kernel with the fused Chebyshev / plain epilogues; a panel of states (qp_cheby_step_batched): H [psi_1 .. psi_b] on
        * the fp64 matrix cores.  AUTO takes it when at least 3/4 of the nrows x ncols positions are stored (the missing ones
        * become explicit zeros, which qp_operator_get_csr then shows, as with qp_operator_fill_info). */
       QP_FMT_DENSE = 5 };
enum { QP_CONV_TDSE = 0, QP_CONV_LVN = 1 };   /* `convention` of liouvillian(), src/generators.jl:473-631 */
enum { QP_FUNC_EXPMI = 0,    /* z -> exp(-i z)   default of newton!, src/newton.jl:247 */
       QP_FUNC_EXP = 1,      /* z -> exp(z)      test/test_newton.jl:171 */
       QP_FUNC_CALLBACK = 2 };
enum { QP_SPECRANGE_ARNOLDI = 0, QP_SPECRANGE_DIAG = 1 };

const char* qp_last_error(void);
const char* qp_status_name(int status);
int qp_version(void);
/* 1 for the developer flavour of the library (csrc: `make dev`, -DQP_DEVELOPER), which accepts the measurement-only settings
 * that change results or force a time-out (knobs walk_dbg bit 1, split_dbg); the release build refuses them. */
int qp_developer_build(void);
int qp_device_count(int* n_out);
/* Developer knobs for A/B kernel experiments (e.g. "rbcsr_variant": bit0 nt matrix loads, bit1 early
 * row-local loads, bit2 deeper unroll; the full list is `struct Tuning` in csrc/device.h); not part of
 * the reference-facing API.  Every context owns a copy of the knobs: qp_ctx_tuning_set changes the
 * copy of ONE context (and of the handles created from it), so contexts driven from different threads
 * never see each other's switches (SURVEY 8b "no global mutable state").  qp_tuning_set only changes
 * the defaults that contexts created AFTERWARDS start from; existing contexts are not touched. */
int qp_tuning_set(const char* key, int value);
int qp_ctx_tuning_set(qp_ctx* ctx, const char* key, int value);
int qp_ctx_tuning_get(qp_ctx* ctx, const char* key, int* value_out);

/* ---- context -------------------------------------------------------------------- */
/* `stream` may be NULL (the library creates a non-blocking stream of its own), an existing
 * hipStream_t, or QP_STREAM_NULL for HIP's null (legacy default) stream -- whose handle is 0
 * and therefore cannot be passed literally.  Pass the caller's stream (e.g.
 * torch.cuda.current_stream().cuda_stream, which IS the null stream unless the caller set
 * another one) whenever the caller's own copies / collectives touch the same buffers. */
#define QP_STREAM_NULL ((void*)(intptr_t)-1)
int qp_ctx_create(int device, void* stream, qp_ctx** out);
int qp_ctx_destroy(qp_ctx* ctx);
int qp_sync(qp_ctx* ctx);

typedef struct {
  uint64_t n_matvec;       /* mat-vecs enqueued  (timing section "matrix-vector product") */
  uint64_t n_cheby_steps;
  uint64_t n_newton_steps;
  uint64_t n_restarts;
  uint64_t n_kernel_launches;
  double spmv_bytes;       /* algorithmic bytes of fused mat-vec kernels */
  uint64_t n_graph_launches; /* hipGraph replays of a whole cheby! step (their kernels are counted above) */
} qp_stats;
int qp_stats_get(qp_ctx* ctx, qp_stats* out);
int qp_stats_reset(qp_ctx* ctx);
/* HIP-event timing of everything enqueued between begin and end on the ctx stream */
int qp_timer_begin(qp_ctx* ctx);
int qp_timer_end(qp_ctx* ctx, double* elapsed_ms_out);

/* ---- index work at the boundary (host only; bit-exact; no GPU needed) ------------- */
/* Julia SparseMatrixCSC{ComplexF64,Int64} (src/generators.jl:473-486) -> 0-based CSR,
 * int64 rowptr, int32 col (ascending within a row), stable in the value order. */
int qp_csc_to_csr_host(int64_t nrows, int64_t ncols, const int64_t* colptr,
                       const int64_t* rowval, const qp_c128* nzval, int index_base,
                       int64_t* rowptr_out, int32_t* col_out, qp_c128* vals_out);
/* contiguous row blocks for `nparts` ranks; balance 0 = rows, 1 = nnz */
int qp_partition_rows_host(const int64_t* rowptr, int64_t nrows, int nparts, int balance,
                           int64_t* bounds_out /* nparts+1 */);

/* The index work of the lattice completion (see qp_operator_fill_info) on a host CSR pattern, without a device:
 * rowptr / col of an nrows x ncols pattern with sorted, unique columns; `min_blocks` = the smallest operator (in 64-row
 * blocks) that is completed (the walk_min_blocks knob of a context).  Writes the completed pattern -- the input itself
 * when nothing is to be completed -- into rowptr_out (nrows + 1) / col_out (capacity `cap` entries) and its entry count
 * into *nnz_out; QP_E_BAD_ARG if `cap` is too small (*nnz_out then holds what is needed). */
int qp_lattice_fill_host(int64_t nrows, int64_t ncols, const int64_t* rowptr, const int32_t* col, int min_blocks,
                         int64_t* rowptr_out, int32_t* col_out, int64_t cap, int64_t* nnz_out);

/* ---- matrices --------------------------------------------------------------------- */
/* ptr/idx: rowptr/col (CSR) or colptr/rowval (CSC), int64, `index_base` 0 or 1.
 * nrows may be a row block of a larger operator (local rows, global ncols).
 * The matrix handle holds the canonical host CSR (the result of the index work);
 * qp_operator_create() lays it out in HBM.  `format` is reserved (pass QP_FMT_AUTO). */
int qp_matrix_create(qp_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nnz,
                     const int64_t* ptr, const int64_t* idx, const void* vals, int val_dtype,
                     int layout, int index_base, int format, qp_matrix** out);
int qp_matrix_destroy(qp_matrix* m);
int qp_matrix_info(const qp_matrix* m, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                   int* format, int64_t* stored_nnz);
/* the canonical CSR held by the handle (tests: index work must be bit-exact) */
int qp_matrix_get_csr(const qp_matrix* m, int64_t* rowptr, int32_t* col, qp_c128* vals);

/* ---- Operator: lazy sum with drift terms (src/generators.jl:111-125) -------------- */
/* ops[0..nops), the last `ncoeffs` of them carry coefficients (drift_offset =
 * nops - ncoeffs, src/generators.jl:635).  All ops must share shape.  The library
 * keeps one union sparsity pattern with a value plane per term; set_coeffs() is the
 * device-side `evaluate!` (src/generators.jl:757-766): vals = sum_l c_l plane_l. */
int qp_operator_create(qp_ctx* ctx, qp_matrix* const* ops, int nops, int ncoeffs,
                       int format, qp_operator** out);
int qp_operator_set_coeffs(qp_operator* op, const qp_c128* coeffs, int ncoeffs);
int qp_operator_set_scale(qp_operator* op, qp_c128 scale); /* ScaledOperator :238-249 */
int qp_operator_destroy(qp_operator* op);
int qp_operator_info(const qp_operator* op, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                     int* format);
/* How the operator is laid out on the device (row-block formats; zeros for CSR):
 * out[0] = 64-row blocks, out[1] / out[2] = blocks whose upper (or full) / lower column
 * section is a stencil section (one column distance per slot for the whole block),
 * out[3] = bytes of index data a mat-vec streams (column sections + transpose positions
 * of the non-stencil lower sections), out[4] = stored values. */
int qp_operator_layout_info(const qp_operator* op, int64_t out[5]);
/* How the row blocks' column sections are encoded (index bytes per entry: 4 / 2 / 0 / 1): out[0..3] = blocks whose upper (or
 * only) section holds int32 columns / int16 distances to the row / one distance per slot for all 64 rows (stencil: lattices) /
 * one 64-aligned column block per slot + the lane inside it as a byte per entry (block map: a slot sends the 64 rows into one
 * block of columns -- row XOR mask, the structure of qubit-register Hamiltonians); out[4..7] = the same for the lower sections
 * of a Hermitian-packed operator.  All zero for the CSR-ordered formats. */
int qp_operator_encoding_info(const qp_operator* op, int64_t out[8]);
/* The value-dictionary mirror of a row-block operator whose 64-row blocks hold at most 256 distinct values each (the reference's
 * generators are sums of a few structured terms, src/generators.jl:634-645: spin-chain couplings, constant hopping amplitudes):
 * one byte per stored entry + per block a table of its distinct values (shared between blocks with the same values); with
 * several terms the table entry is the tuple of the terms' values and evaluate! (src/generators.jl:757-766) recombines the
 * tables, not a value plane.  The reconstruction is exact (qp_operator_get_csr returns the decoded values) and the mat-vec
 * is bit-identical to the uncoded one.  out[0] = 1 when the mat-vec kernels read the mirror, out[1] = table entries,
 * out[2] = distinct tables, out[3] = bytes a term streams for the values through it (codes + tables), out[4] = bytes of the
 * value plane it replaces, out[5] = why there is none (0: there is one; 1: not a plain row-block operator; 2: a block with more
 * than 256 distinct values; 3: no saving; 4: knob value_dict = 0; 5: the column-blocked mirror is in use). */
int qp_operator_value_encoding_info(const qp_operator* op, int64_t out[6]);
/* What creating the operator cost on the host (union pattern, lattice completion, value planes, Hermitian check, format
 * choice, encoding, upload): out[0] = ms of the latest build (the creation itself, or a later re-layout), out[1] = ms of all builds, out[2] = re-layouts after creation -- evaluate! (src/generators.jl:757-766)
 * only rewrites coefficients, but a complex coefficient on a Hermitian-packed operator forces ONE rebuild as plain
 * row blocks (a slower mat-vec from then on) --, out[3] = the current device format (QP_FMT_*). */
int qp_operator_build_info(const qp_operator* op, double out[4]);
/* Lattice completion at qp_operator_create (knob lattice_fill): explicit zeros added to the union pattern so that every
 * interior row of a lattice operator with open boundaries (a finite-difference Hamiltonian on a grid: the rows at the
 * grid's x-edges lack one neighbour) carries the same list of column distances -- what the strip walk and the stencil
 * encoding need.  *n_filled = 0 for every other operator.  The zeros show up in qp_operator_get_csr and in the stored
 * count of qp_operator_layout_info; the reference's sparse matrices (src/generators.jl:634-645 mul!) keep explicit
 * zeros in the same way. */
int qp_operator_fill_info(const qp_operator* op, int64_t* n_filled);
/* Strip-walk plan of a Hermitian-packed lattice operator (the fused Chebyshev term then walks down strip columns and
 * keeps the re-read data in registers / LDS): out[0] = 1 if the operator has one, out[1] = near distances, out[2] = far
 * reach K (far distances m g, m = 1..K), out[3] = 1 if the stencil has a diagonal, out[4] = rows per strip step g
 * (any g >= 64), out[5], out[6] = the walkable row blocks [W0, R1), out[7] = row blocks on the per-block path. */
int qp_operator_walk_info(const qp_operator* op, int64_t out[8]);
/* The two-term strip walk: whether a whole-operator qp_cheby_step of this operator forms TWO Chebyshev terms per pass over the
 * matrix values under the context's knobs (lattice operators beyond the Infinity Cache whose strip columns are long enough; knob
 * walk_pair).  out[0] = 1 if it does, out[1], out[2] = the row blocks [W0, R1) that take both terms in one launch, out[3] = row
 * blocks outside them (two per-block launches per pair of terms), out[4] = rows of a 64-row column chunk that form the second
 * term (64 - 2 x the largest near distance: the chunks overlap), out[5] = chunks per strip step, out[6] = steps of the second term per
 * wavefront (each wavefront runs in 2 K steps before its first), out[7] = wavefronts per strip column.  Same results bit for bit
 * either way. */
int qp_operator_walk2_info(const qp_operator* op, int64_t out[8]);
/* *glong = the long distance L (rows) of a walk plan with one further pair of distances +-L beyond its far reach
 * (a three-dimensional grid's plane distance; its operands are loaded directly), 0 if the plan has none / there is no plan */
int qp_operator_walk_long(const qp_operator* op, int64_t* glong);
/* A plan may carry TWO such pairs (the fourth-order Laplacian of a three-dimensional grid: +-nx ny and +-2 nx ny; a
 * four-dimensional grid): out = {L_0, L_1}, the shorter first, 0 for a pair the plan does not have.  qp_operator_walk_long
 * reports the longest. */
int qp_operator_walk_long_pairs(const qp_operator* op, int64_t out[2]);
/* The whole stencil shape of the plan: out = {near distances, far reach K, 1 if there is a diagonal entry, long pairs (0-2),
 * 1 if the far distances come with their DIAGONAL neighbours (strip step m: m g - 1, m g, m g + 1 -- the nine-point stencil of a
 * two-dimensional grid with next-nearest hopping), strip step g, shorter long distance, longest distance}; zeros without a plan. */
int qp_operator_walk_shape(const qp_operator* op, int64_t out[8]);
/* Column-blocked mirror of an operator with IRREGULAR columns (src/generators.jl:634-645 accepts any sparse H_l; a random
 * graph's gathers are each their own cache line and the vector outgrows the L2): the entries are also kept grouped by
 * (row tile, column block) and whole-operator mat-vecs (cheby!, arnoldi!, mul!) walk the column blocks in their outer loop, so
 * that the whole chip gathers from one L2-resident window of the vector at a time (csrc/kernels_colblock.hip).  Built at
 * qp_operator_create when the operator is laid out as plain row blocks / CSR, has at least 2^20 columns and its sampled
 * gathers are irregular (knobs colblock, cb_log2w, cb_min_log2n).  out = {1 if the operator has the mirror, column blocks,
 * log2 of the columns per block, rows per tile, entries of the longest (tile, block) segment, tiles}; *line_share
 * (nullable) = sampled share of gathers that pull a 128-byte line of their own, 0 if the decision never got that far. */
int qp_operator_colblock_info(const qp_operator* op, int64_t out[6], double* line_share);
/* How evaluate! (qp_operator_set_coeffs / _set_scale) updates the stored values: out[0] = index of the first of the trailing
 * control terms that are kept as (position, value) lists because they touch at most a quarter of the stored values (a dipole
 * operator on a grid is a diagonal), -1 if there are none; out[1] = positions such an update rewrites; out[2] = 1 if the latest
 * update rewrote only those positions (knob sparse_controls; the full combination otherwise). */
int qp_operator_evaluate_info(const qp_operator* op, int64_t out[3]);
/* WHY an operator does not take the strip walk (the fast path of the fused Chebyshev term is a cliff: the same banded
 * operator costs 1.8 x as much per term on the per-block kernels): *code = QP_WALK_OK when it has a plan that its launches
 * use, else what broke it; `text` (may be NULL) receives a sentence with the offending numbers ("near column distance 17
 * exceeds the 16-row halo ..."), truncated to text_len.  The Python / Julia wrappers warn once per operator. */
enum { QP_WALK_OK = 0,
       QP_WALK_NOT_HERMITIAN = 1,   /* a term is not exactly Hermitian: no packed format, no walk (newton! does not need it) */
       QP_WALK_COMPLEX_COEFF = 2,   /* a complex coefficient un-packed the operator (qp_operator_build_info counts the re-layout) */
       QP_WALK_NOT_PACKED = 3,      /* Hermitian, but laid out otherwise (format forced, or AUTO found no L2 locality for the packing) */
       QP_WALK_TOO_FEW_BLOCKS = 4,  /* fewer walkable row blocks than knob walk_min_blocks (3072): the per-block kernel is as fast there */
       QP_WALK_NO_UNIFORM_RUN = 5,  /* rows do not repeat one list of column distances: not a lattice */
       QP_WALK_ROW_LENGTH = 6,      /* fewer than 3 or more than 19 entries per row */
       QP_WALK_NOT_MIRRORED = 7,    /* a column distance without its mirror image */
       QP_WALK_NO_FAR = 8,          /* a plain band (no distance >= 64 rows): nothing to walk along */
       QP_WALK_NO_NEAR = 9,
       QP_WALK_NEAR_TOO_FAR = 10,   /* a near distance beyond the 16-row halo of the walk's window */
       QP_WALK_TOO_MANY_NEAR = 11,  /* more than 4 near distances per side */
       QP_WALK_TOO_MANY_FAR = 12,   /* more than 4 far distances per side (or more than 2 beside long pairs) */
       QP_WALK_INCOMMENSURATE = 13, /* far distances that are not multiples of one strip step (two strides) */
       QP_WALK_NO_KERNEL = 14,      /* a shape without a kernel instance (long pairs with more than 2 near / 2 far distances) */
       QP_WALK_LAYOUT = 15,         /* the run's upper sections are not at equal strides / positions beyond 2^31 */
       QP_WALK_DISABLED = 16 };     /* knob hrb_walk 0 */
int qp_operator_walk_reason(const qp_operator* op, int* code, char* text, size_t text_len);
/* How qp_cheby_step_batched will visit the rows for a panel of `batch` states (wave-per-row kernel,
 * more than 32 states): out[0] = inner dimension g detected in the pattern (far offsets are multiples of
 * g: H = H_a (x) 1 + 1 (x) H_c), out[1] = strip width (rows are visited strip by strip so that the gather
 * window of the panel stays inside an XCD's L2); both 0 when the rows are visited in natural order. */
int qp_operator_spmm_walk(qp_operator* op, int batch, int64_t out[2]);
/* Whether qp_cheby_step_batched will take the LDS-staged tiles for a panel of `batch` states (a lattice operator whose
 * interior rows all carry the same entries, near distances <= 4 and far distances m g with |m| <= 4: 4 x 4 patches of rows
 * whose operands are staged once per workgroup): out = {taken (0 / 1), tiles, rows left to the row kernel, g, K, NN}. */
int qp_operator_spmm_tiles(qp_operator* op, int batch, int64_t out[6]);
/* read the DEVICE copy (union pattern, currently combined values) back as canonical
 * CSR: the device-format round trip must be bit-exact. */
int qp_operator_get_csr(qp_operator* op, int64_t* rowptr, int32_t* col, qp_c128* vals);

/* ---- states and BLAS-1 (the six primitives of src/cheby.jl:146-148) ---------------- */
int qp_state_create(qp_ctx* ctx, int64_t n, qp_state** out);
int qp_state_wrap(qp_ctx* ctx, void* device_ptr, int64_t n, qp_state** out);
int qp_state_destroy(qp_state* s);
int qp_state_upload(qp_state* s, const qp_c128* host);
int qp_state_download(const qp_state* s, qp_c128* host);
/* Pin a caller-owned host array (page-lock it for the device) so that qp_state_upload / _download from
 * and to it run at PCIe speed instead of through the runtime's staging buffers: a host-resident
 * caller -- the propagator types of the reference keep `state` in host memory (src/propagator.jl:119-126) --
 * registers its state vector once and unregisters it before the array is freed.  The registration is
 * portable (valid for every device of the process): qp_host_unregister may run on any thread, e.g. from a
 * finalizer.  Only transfers from / to the registered array itself benefit: a caller that allocates a fresh
 * array per step (the reference's inplace = false mode) downloads through pageable memory. */
int qp_host_register(void* host, size_t bytes);
int qp_host_unregister(void* host);
void* qp_state_ptr(const qp_state* s);
int64_t qp_state_len(const qp_state* s);
int qp_copy(qp_state* dst, const qp_state* src);              /* copyto! */
int qp_scal(qp_state* x, qp_c128 alpha);                      /* lmul!   */
int qp_axpy(qp_c128 alpha, const qp_state* x, qp_state* y);   /* axpy!   */
int qp_fill(qp_state* x, qp_c128 alpha);                      /* fill!   */
int qp_dot(const qp_state* x, const qp_state* y, qp_c128* out);  /* dot: conj(x).y */
int qp_norm(const qp_state* x, double* out);                  /* norm    */

/* mul!(y, A, x, alpha, beta): y <- beta y + alpha A x    src/generators.jl:634-645 */
int qp_mul(qp_operator* op, const qp_state* x, qp_state* y, qp_c128 alpha, qp_c128 beta);
/* dot(x, A, y)                                            src/generators.jl:648-660 */
int qp_dot_op(const qp_state* x, qp_operator* op, const qp_state* y, qp_state* tmp,
              qp_c128* out);

/* ---- Chebyshev (src/cheby.jl) ------------------------------------------------------ */
/* cheby_coeffs / cheby_coeffs!  src/cheby.jl:25-39, :54-72.  Returns QP_E_BAD_ARG with
 * *n_out = required length when cap is too small. */
int qp_cheby_coeffs(double Delta, double dt, double limit, double* out, int cap, int* n_out);
/* ChebyWrk  src/cheby.jl:87-124 (device workspace: two vectors instead of three) */
int qp_cheby_create(qp_ctx* ctx, int64_t n, qp_cheby** out);
int qp_cheby_destroy(qp_cheby* w);
/* cheby!(psi, H, dt, wrk; E_min, check_normalization)  src/cheby.jl:150-213.
 * `a[0..n_coeffs)`, Delta, E_min, wrk_dt, limit are the ChebyWrk fields; dt is signed.
 * Whole step is enqueued without host synchronisation unless check_normalization. */
int qp_cheby_step(qp_cheby* w, qp_operator* op, qp_state* psi, const double* a, int n_coeffs,
                  double Delta, double E_min, double dt, double wrk_dt, double limit,
                  int check_normalization);
/* Batched states (BASELINE configs[4]; the reference propagates one state per propagator,
 * optimal-control callers step many): `psi` is a panel of `batch` states stored with the
 * state index contiguous, psi[i*batch + s], length n*batch; `w` was created for n*batch.
 * Same arithmetic as qp_cheby_step for every state, the matrix is streamed once for all. */
int qp_cheby_step_batched(qp_cheby* w, qp_operator* op, qp_state* psi, int batch, const double* a,
                          int n_coeffs, double Delta, double E_min, double dt, double wrk_dt);
/* One fused term, for row-partitioned multi-GPU drivers that exchange x between terms:
 *   s = (H x)[i];  t = c (s - beta x[xoff+i]) (+ v0[i] if v0);  if vout: vout[i] = t;
 *   r = (acc_in ? acc_in[i] : a_prev x[xoff+i]) + a t;  acc_out[i] = phase r.
 * x has op.ncols entries, the other vectors op.nrows; vout may alias v0.
 * `defer` (nullable) moves accumulator traffic out of two of every three terms: the row
 * epilogue of term m sees v_{m-2} = v0[i] and v_{m-1} = x[xoff+i], so the axpys of src/cheby.jl
 * :182/:205 for those two terms can be applied here, in the same order and with the same FMAs:
 *   skip != 0 : the term does not touch acc_in / acc_out (both may be NULL);
 *   otherwise : r = (acc_in ? acc_in[i] : a_prev * (n_defer == 1 ? v0[i] : x[xoff+i]));
 *               if n_defer == 2: r += a_d2 v0[i];  if n_defer >= 1: r += a_d1 x[xoff+i];
 *               r += a t;  acc_out[i] = phase r.
 * qp_acc_schedule_host fills one entry per term for a cheby! of n_coeffs coefficients. */
typedef struct {
  int skip, n_defer;
  double a_d1, a_d2;
} qp_acc_defer;
int qp_acc_schedule_host(const double* a, int n_coeffs, qp_acc_defer* out /* n_coeffs-1 */);
int qp_cheby_term(qp_operator* op, const qp_state* x, int64_t xoff, const qp_state* v0,
                  qp_state* vout, const qp_state* acc_in, qp_state* acc_out, qp_c128 c,
                  double beta, double a_prev, double a, qp_c128 phase, const qp_acc_defer* defer);

/* Overlap of the multi-GPU exchange with compute (no reference counterpart: the reference
 * is single-process).  qp_split_create partitions the operator's 64-row blocks into
 * *boundary* blocks (contain a row of `send_rows`, i.e. a row another rank reads, or a row
 * that itself reads a ghost column >= nrows) and *interior* blocks.  qp_cheby_term_split
 * runs one fused term as two launches: the boundary blocks on `boundary_stream` (they also
 * pack the new term vector into slab[i] for send_rows[i]), the interior blocks on the
 * operator's context stream (`first` != 0 for the first call after the two streams were
 * joined by the caller).  Consecutive calls are ordered by a HIP event on the side stream and,
 * on the main stream, by a completion counter that only the interior workgroups adjacent to
 * boundary rows poll (tuning key "split_mode" = 0 falls back to events on both streams).  The caller enqueues the
 * RCCL all-gather of `slab` on `boundary_stream` after the call. */
typedef struct qp_split qp_split;
int qp_split_create(qp_operator* op, const int64_t* send_rows, int64_t nsend, qp_split** out);
int qp_split_destroy(qp_split* sp);
int qp_split_info(const qp_split* sp, int64_t* n_boundary_blocks, int64_t* n_interior_blocks);
/* The interior launch as a strip walk (lattice operators, see qp_operator_walk_info): out[0] = 1 if the split has a walk
 * plan, out[1], out[2] = the walked row blocks [first, end) -- inside the interior, no walked block reads a boundary row --,
 * out[3] = interior blocks on the per-block path (the only ones that wait for the boundary launch). */
int qp_split_walk_info(const qp_split* sp, int64_t out[4]);
/* device sync + check that no in-launch wait of the split schedule ever timed out */
int qp_split_check(qp_split* sp);
int qp_cheby_term_split(qp_operator* op, qp_split* sp, void* boundary_stream, int first,
                        const qp_state* x, int64_t xoff, const qp_state* v0, qp_state* vout,
                        const qp_state* acc_in, qp_state* acc_out, qp_state* slab, qp_c128 c,
                        double beta, double a_prev, double a, qp_c128 phase, const qp_acc_defer* defer);

/* ---- matrix-free Liouvillian ------------------------------------------
 * liouvillian(H, c_ops; convention) (src/generators.jl:473-631) builds the n^2 x n^2 sparse
 * superoperator  L = s_h (1 (x) H - H^T (x) 1) + s_d sum_k (A_k^{+T} (x) A_k - (1 (x) G_k + G_k^T (x) 1) / 2),
 * G_k = A_k^+ A_k, (s_h, s_d) = (1, i) for :TDSE and (i, 1) for :LvN, acting on the
 * column-major vec(rho).  For DENSE H and A_k that matrix has 2 n^3 entries; this operator
 * applies the same map to rho as complex n x n products instead (hand-written fp64 matrix-core kernels up to
 * n = 2048, rocBLAS zgemm above):
 *   L rho = M_L rho - rho M_R + s_d sum_k A_k rho A_k^+,   M_L/R = s_h H -/+ (s_d / 2) sum_k G_k,
 * with H = sum_l c_l H_l the usual lazy sum (qp_operator_set_coeffs; the first
 * nterms - ncoeffs terms are the drift).  Matrices are dense, column-major (Julia layout).
 * The result is a qp_operator of size n^2: qp_mul, qp_dot_op, the Arnoldi / Newton / specrange
 * entry points and qp_cheby_step accept it (qp_cheby_step with an unfused epilogue); the entry
 * points that need stored entries (get_csr, split, batched, persistent kernels) do not. */
/* ---- qubit registers: H = sum_l c_l H_l with every H_l a sum of Pauli strings, applied from the strings (no stored matrix) ----
 * The reference's generator for a register of n qubits is the usual lazy sum of sparse matrices (src/generators.jl:634-645 mul!);
 * when those matrices are sums of Pauli strings  a P,  P = sigma_{n-1} (x) ... (x) sigma_0  (qubit i = bit i of the basis index),
 * this operator applies them from the bit masks:  xmask = the qubits with X or Y, zmask = those with Z or Y (a Y is in both),
 *     (P psi)[r] = i^{#Y} (-1)^{popcount((r xor xmask) and zmask)} psi[r xor xmask]
 * -- zero matrix bytes per term.  `op` = which H_l of the lazy sum a string belongs to (the first nops - ncoeffs terms are the drift,
 * the others carry the coefficients of qp_operator_set_coeffs: evaluate! rewrites a few hundred numbers).  6 <= nqubits <= 30.
 * The result is a qp_operator of size 2^nqubits (QP_FMT_MATFREE): qp_cheby_step (fused term), qp_mul, qp_dot_op, the Arnoldi /
 * Newton / specrange entry points accept it; the entry points that need stored entries do not. */
typedef struct {
  uint64_t xmask, zmask;
  qp_c128 coef;      /* the string's amplitude a (real for a Hermitian H; the factor i^{#Y} is applied by the library) */
  int op;            /* 0 .. nops - 1 */
} qp_pauli_string;
int qp_pauli_operator_create(qp_ctx* ctx, int nqubits, const qp_pauli_string* strings, int nstrings, int nops, int ncoeffs,
                             qp_operator** out);
int qp_liouvillian_create(qp_ctx* ctx, int64_t n, const qp_c128* const* H_terms, int nterms,
                          int ncoeffs, const qp_c128* const* c_ops, int nc, int convention,
                          qp_operator** out);

/* ---- row-partitioned cheby! with the exchange inside the library (RCCL over xGMI) ----
 * No reference counterpart (the reference is single-process).  One process per GPU; the
 * caller partitions H by rows, renumbers columns locally (own rows [0, nloc), then `world`
 * ghost slabs of M rows each: nloc + owner*M + position) and creates the operator on those
 * local rows.  qp_comm wraps an RCCL communicator created from a ncclUniqueId that rank 0
 * obtains (qp_comm_unique_id) and the caller distributes by any means; `rccl_lib_path` is
 * the librccl.so to use (the one PyTorch-ROCm ships when the caller also uses torch).
 * qp_sharded_cheby_step runs one cheby! (src/cheby.jl:150-213) on the partitioned state with
 * ONE host call: per term the boundary row blocks go to a high-priority side stream, pack the
 * rows other ranks read into `slab`, and ncclAllGather follows on that stream while the
 * interior row blocks run on the context stream (see qp_cheby_term_split). */
typedef struct qp_comm qp_comm;
int qp_comm_unique_id(const char* rccl_lib_path, char id_out[128]);
int qp_comm_create(qp_ctx* ctx, const char* rccl_lib_path, const char id[128], int rank, int world,
                   qp_comm** out);
/* The same in two phases: qp_comm_prepare is LOCAL (dlopen of librccl, symbol resolution; no communication),
 * qp_comm_connect is the COLLECTIVE ncclCommInitRank.  A caller with several ranks makes sure every rank
 * came through qp_comm_prepare (agreeing over whatever channel carried the id) before any rank connects: a rank
 * that failed locally would otherwise leave the others blocked in the collective.  qp_comm_create = both. */
int qp_comm_prepare(qp_ctx* ctx, const char* rccl_lib_path, int rank, int world, qp_comm** out);
int qp_comm_connect(qp_comm* comm, const char id[128]);
/* A communicator whose exchange is performed by the caller: qp_sharded_cheby_step calls `cb`
 * where it would enqueue the RCCL collective.  The callback must make `send[0..count)` of every
 * rank r arrive at `recv_base + r * count` on the ranks that read it (all of them when
 * n_send_to < 0; otherwise this rank sends to send_to[] and receives the slabs of recv_from[]),
 * ordered after the work already queued on `stream` and before anything queued on it later.
 * For drivers that own their transport (MPI.jl ...) and for testing the step with several ranks
 * on one GPU, where RCCL will not form a communicator. */
typedef int (*qp_exchange_cb)(void* user, const qp_c128* send_dev, int64_t count, qp_c128* recv_base_dev,
                              const int* send_to, int n_send_to, const int* recv_from, int n_recv_from,
                              void* stream);
int qp_comm_create_callback(qp_ctx* ctx, int rank, int world, qp_exchange_cb cb, void* user, qp_comm** out);
/* What the communicator is: the `world` / `rank` it was created with; `rccl_ranks` / `rccl_rank` = ncclCommCount /
 * ncclCommUserRank of the CONNECTED RCCL communicator (0 / -1 for a callback communicator or before qp_comm_connect);
 * `is_callback` = 1 for qp_comm_create_callback.  Any output may be NULL.  (bench.py --gpus N prints rccl_ranks and
 * refuses to report a multi-GPU line whose exchange did not run on an N-rank RCCL communicator.) */
int qp_comm_info(const qp_comm* comm, int* world, int* rank, int* rccl_ranks, int* rccl_rank, int* is_callback);
int qp_comm_destroy(qp_comm* comm);
/* recv[r*count .. (r+1)*count) = rank r's send[0..count);  stream NULL = the ctx stream */
int qp_comm_allgather(qp_comm* comm, const qp_state* send, qp_state* recv, int64_t count,
                      void* stream);
typedef struct qp_sharded_cheby qp_sharded_cheby;
typedef struct {
  qp_operator* op;          /* local rows, local column numbering */
  qp_split* split;          /* NULL: no overlap, everything on the context stream */
  qp_comm* comm;            /* NULL: no exchange (single rank, M = 0) */
  qp_state* X0;             /* op.ncols entries; X0[0..nloc) holds Psi before and after a step */
  qp_state* X1;             /* op.ncols entries, work */
  qp_state* acc;            /* nloc entries, work */
  qp_state* slab;           /* M entries: send buffer (NULL when direct_send) */
  const int64_t* send_rows; /* host, nsend <= M local rows that other ranks read */
  int64_t nsend, M;
  int direct_send;          /* every local row is sent and nloc == M: X[0..nloc) is the send buffer */
  /* neighbour exchange instead of the all-gather (banded H: each rank reads from few others):
   * the slab goes to the ranks in send_to, the slabs of the ranks in recv_from land in their
   * ghost slots, as one ncclGroup of ncclSend / ncclRecv.  n_send_to < 0: all-gather. */
  const int* send_to;
  int n_send_to;
  const int* recv_from;
  int n_recv_from;
} qp_sharded_cheby_desc;
int qp_sharded_cheby_create(const qp_sharded_cheby_desc* desc, qp_sharded_cheby** out);
int qp_sharded_cheby_destroy(qp_sharded_cheby* s);
int qp_sharded_cheby_step(qp_sharded_cheby* s, const double* a, int n_coeffs, double Delta,
                          double E_min, double dt);

/* ---- Arnoldi (src/arnoldi.jl) ------------------------------------------------------ */
int qp_krylov_create(qp_ctx* ctx, int64_t n, int nvec, qp_krylov** out);
int qp_krylov_destroy(qp_krylov* q);
int qp_krylov_download(const qp_krylov* q, int i, qp_c128* host);
/* arnoldi!(Hess, q, m, psi, H, dt; extended, norm_min)  src/arnoldi.jl:60-100.
 * Hess: caller-allocated column-major ldh x ldh, zero-filled by the call (:78). */
int qp_arnoldi(qp_operator* op, qp_krylov* q, int m, const qp_state* psi, double dt,
               int extended, double norm_min, qp_c128* Hess, int ldh, int* m_out);
/* extend_arnoldi!(Hess, q, m, H, dt; norm_min)  src/arnoldi.jl:115-129 (1-based m) */
int qp_arnoldi_extend(qp_operator* op, qp_krylov* q, int m, double dt, double norm_min,
                      qp_c128* Hess, int ldh, int* extended_out);
/* diagonalize_hessenberg_matrix(Hess, m; accumulate)  src/arnoldi.jl:143-170 (host) */
int qp_hessenberg_eigvals(const qp_c128* Hess, int ldh, int m, int accumulate, qp_c128* out);

/* Building blocks of a row-partitioned Arnoldi / Newton (the inner products of
 * src/arnoldi.jl:85,89 become partial sums over the local rows; the caller all-reduces them).
 * Column j (0-based): q_{j+1} = H q_j by qp_mul into qp_krylov_vec(q, j+1), then
 *   qp_krylov_multidot  -> reduced[0..j] = <q_i|q_{j+1}>, reduced[j+1..2j+1] = <q_i|q_j>   (local sums)
 *   [all-reduce reduced]
 *   qp_krylov_project   -> MGS coefficients from the global sums, q_{j+1} -= Q h, hess_col[i] = dt h_i,
 *                          256 partial sums of |q_{j+1}|^2 over the local rows
 *   [all-reduce norm_partials]
 *   qp_krylov_normalize -> h = sqrt(sum), hess_norm[0] = (dt h, 0), hess_norm[1] = (h, 0),
 *                          q_{j+1} *= 1/h unless h < norm_min. */
int qp_krylov_vec(qp_krylov* q, int i, qp_state** out);   /* non-owning view of Arnoldi vector i */
int qp_krylov_multidot(qp_krylov* q, int j, qp_state* reduced);
int qp_krylov_project(qp_krylov* q, int j, double dt, const qp_state* reduced, qp_state* hess_col,
                      qp_state* norm_partials);
int qp_krylov_normalize(qp_krylov* q, int j, double dt, double norm_min, const qp_state* norm_partials,
                        qp_state* hess_norm);
/* out = (use_out ? s0 out : 0) + sum_{i<m} coefs[i] q[first+i]  (src/newton.jl:346-367);
 * optional 256 partial sums of |out|^2 */
int qp_combine(qp_state* out, int use_out, qp_c128 s0, qp_krylov* q, int first, int m, const qp_c128* coefs,
               qp_state* norm_partials);

/* ---- Newton (src/newton.jl) -------------------------------------------------------- */
typedef void (*qp_func_cb)(const qp_c128* z, qp_c128* out, void* user);
/* extend_leja!  src/newton.jl:97-148 (zero-based arrays; newpoints is clobbered) */
int qp_extend_leja(qp_c128* leja, int n, qp_c128* newpoints, int n_newpoints, int n_use);
/* extend_newton_coeffs!  src/newton.jl:176-214 */
int qp_extend_newton_coeffs(qp_c128* a, int n_a, const qp_c128* leja, int func_id,
                            qp_func_cb cb, void* user, int n_leja, double radius);
typedef struct {
  int restarts, n_a, n_leja, m_last, n_matvec;
  double radius, last_relerr, norm_psi;
  /* host wall-clock breakdown of the call, milliseconds (replaces the TimerOutputs sections
   * "arnoldi!", "diagonalize_hessenberg_matrix", "get Leja points", "get Newton coeffs",
   * "evaluate polynomial" of src/newton.jl:276-343) */
  double ms_arnoldi, ms_eig, ms_leja, ms_coeffs, ms_poly, ms_update;
  /* host time the DEVICE waited for: from the arrival of a sweep's last Hessenberg column to the launch of
   * the kernel that applies the Newton polynomial (last eigenvalue block, Leja ordering, coefficients,
   * polynomial), summed over the restarts of the step */
  double ms_exposed;
  /* Arnoldi sweeps of the step that took the one-pass form (basis read once per column, lagged normalisation: taken when
   * basis + matrix exceed the Infinity Cache), and how many of THOSE were done again with the two-pass sweep because the
   * norm of a stored basis vector had drifted from 1 by more than 1e-4 (pipelined-Krylov error growth when h_{t+1,t} << |H q_t|) */
  int sweeps_onepass, sweeps_onepass_redone;
} qp_newton_stats;
/* NewtonWrk(v0; m_max)  src/newton.jl:23-60 */
int qp_newton_create(qp_ctx* ctx, int64_t n, int m_max, qp_newton** out);
int qp_newton_destroy(qp_newton* w);
/* newton!(psi, H, dt, wrk; func, norm_min, relerr, max_restarts)  src/newton.jl:246-385 */
int qp_newton_step(qp_newton* w, qp_operator* op, qp_state* psi, double dt, int func_id,
                   qp_func_cb cb, void* user, double norm_min, double relerr,
                   int max_restarts, qp_newton_stats* stats);
/* inspection of wrk.a / wrk.leja after a step (src/newton.jl:23-36) */
int qp_newton_get_coeffs(const qp_newton* w, qp_c128* a, qp_c128* leja, int cap);

/* ---- propagate step loop (src/propagate.jl:283-344), rows N1/N2 of SURVEY 8f --------- */
/* Runs `nsteps` prop_step!s in ONE call: per step the coefficients of the operator are taken
 * from coeff_table[step*ncoeffs .. ] (the values `parameters[control][n]` that
 * _pwc_set_genop! would look up, src/pwc_utils.jl:86-92) and the time step from dts[step]
 * (signed; Cheby needs |dts[step]| ~ wrk_dt).  Optional outputs, filled without any host
 * synchronisation inside the loop and downloaded once at the end:
 *   expvals_out[(step+1)*nobs + o] = <psi|O_o|psi> after the step (row 0: initial state)
 *                                    -- map_observable for matrices, src/storage.jl:121-123
 *   states_out[(step+1)*n ..]      = the state after the step (row 0: initial state)
 * method 0 = Cheby (spec fields of qp_cheby_step), 1 = Newton (fields of qp_newton_step). */
typedef struct {
  int method;
  qp_cheby* cheby;  const double* a;  int n_coeffs;  double Delta, E_min, wrk_dt, limit;
  int check_normalization;
  qp_newton* newton;  int func_id;  qp_func_cb cb;  void* user;
  double norm_min, relerr;  int max_restarts;
} qp_prop_spec;
int qp_propagate(qp_operator* op, qp_state* psi, const qp_prop_spec* spec, const double* dts,
                 const qp_c128* coeff_table, int ncoeffs, int nsteps, qp_operator* const* observables,
                 int nobs, qp_c128* expvals_out, qp_c128* states_out);

/* ---- SpectralRange (src/specrad.jl) ------------------------------------------------ */
/* ritzvals(G, state, m_min, m_max; prec, norm_min)  src/specrad.jl:170-220 */
int qp_ritzvals(qp_operator* op, const qp_state* state, int m_min, int m_max, double prec,
                double norm_min, qp_c128* out, int* n_out);
/* specrange(H, :arnoldi; state, m_min, m_max, prec, norm_min, enlarge)  :88-112 */
int qp_specrange_arnoldi(qp_operator* op, const qp_state* state, int m_min, int m_max,
                         double prec, double norm_min, int enlarge, double* E_min,
                         double* E_max);

#ifdef __cplusplus
}
#endif
#endif /* QPROP_H */
