// Device helpers shared by the kernel translation units (kernels*.hip): complex FMA forms,
// wavefront / workgroup sums, the XCD-aware workgroup remap, the in-launch hand-off, the row epilogues of the
// fused Chebyshev term and of the plain mat-vec, and the loaders of the row-block formats.  gfx950 only.
#pragma once

#include "device.h"

namespace qp {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
// grid of an elementwise kernel: 8 workgroups per CU at most, grid-stride the rest
inline int ew_grid(int64_t n) {
  int64_t g = (n + kThreads - 1) / kThreads;
  if (g > 256 * 8) g = 256 * 8;  // 8 workgroups per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ void cfma(double2& s, const double2 a, const double2 b) {
  s.x = fma(a.x, b.x, s.x);
  s.x = fma(-a.y, b.y, s.x);
  s.y = fma(a.x, b.y, s.y);
  s.y = fma(a.y, b.x, s.y);
}
// One rounded product and one fused multiply-add per component, spelled out: left to -ffp-contract=fast, WHICH of the two
// products of `a.x * b.x - a.y * b.y` gets fused is the compiler's choice per call site, and two kernels that must agree
// bit for bit (the strip walk and the per-block kernel: the step's final phase exp(-i beta dt) is a complex scalar as soon
// as the spectral window is not centred on zero) rounded differently.
__device__ __forceinline__ double2 cmul(const double2 a, const double2 b) {
  return make_double2(fma(a.x, b.x, -__dmul_rn(a.y, b.y)), fma(a.x, b.y, __dmul_rn(a.y, b.x)));
}
__device__ __forceinline__ double2 cconj_mul(const double2 a, const double2 b) {  // conj(a)*b
  return make_double2(fma(a.x, b.x, __dmul_rn(a.y, b.y)), fma(a.x, b.y, -__dmul_rn(a.y, b.x)));
}

// Sum over the 64 lanes of a wavefront, returned in EVERY lane, through the data-parallel primitives of the vector ALU
// (row shifts inside the rows of 16 lanes, then the two row broadcasts of gfx9): about twenty instructions and no
// trip through the LDS crossbar -- the ds_bpermute form that __shfl_down compiles to is six dependent LDS round trips,
// ~500 clocks per sum, which dominated the prologues and epilogues of the Arnoldi kernels (dozens of sums each).
// Fixed order: prefix sums inside each row of 16, rows 0+1 and 2+3, then the halves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {   // lanes without a source (or outside the row mask) get 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_take<0x111, 0xf>(v);   // row_shr:1
  v += dpp_take<0x112, 0xf>(v);   // row_shr:2
  v += dpp_take<0x114, 0xf>(v);   // row_shr:4
  v += dpp_take<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row's sum
  v += dpp_take<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3  -> lane 63 holds the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// the same sum through the LDS crossbar (__shfl_down = ds_bpermute; total in lane 0): for kernels that are short of
// vector-ALU issue slots rather than waiting on the chain
__device__ __forceinline__ double wave_sum_lds(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the 256-thread workgroup, result broadcast to every thread; fixed order
__device__ __forceinline__ double2 block_sum(double2 v, double2* lds4) {
  v.x = wave_sum(v.x);
  v.y = wave_sum(v.y);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) lds4[w] = v;
  __syncthreads();
  double2 r = lds4[0];
#pragma unroll
  for (int i = 1; i < kThreads / 64; ++i) {
    r.x += lds4[i].x;
    r.y += lds4[i].y;
  }
  __syncthreads();
  return r;
}

// XCD-aware workgroup remap: hardware deals workgroups round-robin over the 8 XCDs
// (MI355X_MICROARCH "Workgroup dispatch"), so ids congruent mod 8 share an L2.  Give
// each XCD one contiguous range of row blocks so the gather window of x stays in its
// L2.  Bijective for any grid size; a different placement only changes speed.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u;
  const unsigned xcd = bid & 7u, j = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + j;
}

// ---------------------------------------------------------------------------
// in-launch producer/consumer hand-off between two launches on different streams
// ---------------------------------------------------------------------------
__device__ __forceinline__ void sync_wait(const SyncArgs& sy, unsigned wg) {
  if (sy.wait && wg >= sy.wait_from_wg) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(sy.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sy.wait_target) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > sy.spin_limit) {  // bounded (~1 min): never hang the queue for good; the host checks the flag
          if (sy.timeout_flag) __hip_atomic_store(sy.timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-visible
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void sync_signal(const SyncArgs& sy) {
  if (sy.signal) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(sy.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------
// NT: the row-local streams (v0, the accumulator, the new term) are nontemporal -- for the
// batched panel, where they would evict the gather window of X from L2 and are not touched
// again before the next launch
typedef double d2nt __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld_stream(const double2* p) {
  if (NT) {
    const d2nt t = __builtin_nontemporal_load(reinterpret_cast<const d2nt*>(p));
    return make_double2(t.x, t.y);
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st_stream(double2* p, const double2 v) {
  if (NT) {
    d2nt t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<d2nt*>(p));
  } else {
    *p = v;
  }
}

template <bool NT>
struct ChebyOpT {
  static constexpr bool kStream = NT;
  ChebyEpi e;
  struct Pre {
    double2 xi, v0, acc;
  };
  __device__ __forceinline__ void begin_issue() {}
  __device__ __forceinline__ void begin(double2*) {}
  // the row's own element of the gathered vector, when the epilogue has it in `pre` (x_i with x + xoff == xloc)
  static constexpr bool kHasXi = true;
  __device__ __forceinline__ const double2* xloc() const { return e.xloc; }
  static __device__ __forceinline__ double2 xi_of(const Pre& p) { return p.xi; }
  // row-local operands, issued ahead of the mat-vec loop so their latency overlaps it
  __device__ __forceinline__ Pre pre(int64_t i) const {
    Pre p;
    p.xi = e.xloc[i];
    p.v0 = e.v0 ? ld_stream<NT>(e.v0 + i) : make_double2(0.0, 0.0);
    p.acc = e.acc_in ? ld_stream<NT>(e.acc_in + i) : make_double2(0.0, 0.0);
    return p;
  }
  // pre() without x_i, for a kernel that holds the row's own element already (the LDS tile of the batched path sets p.xi itself)
  __device__ __forceinline__ Pre pre_streams(int64_t i) const {
    Pre p;
    p.xi = make_double2(0.0, 0.0);
    p.v0 = e.v0 ? ld_stream<NT>(e.v0 + i) : make_double2(0.0, 0.0);
    p.acc = e.acc_in ? ld_stream<NT>(e.acc_in + i) : make_double2(0.0, 0.0);
    return p;
  }
  __device__ __forceinline__ void row(int64_t i, double2 s, const Pre& p, double2& chk, double& nrm,
                                      int64_t slot) const {
    const double2 xi = p.xi;
    // t = c * (s - beta * x_i) [+ v0_i]        src/cheby.jl:178-179, :192-193, :202
    double2 t = make_double2(fma(-e.beta, xi.x, s.x), fma(-e.beta, xi.y, s.y));
    t = cmul(e.c, t);
    if (e.check_partials) {  // src/cheby.jl:194-200: measured before "+ v0"
      const double2 d = cconj_mul(xi, t);
      chk.x += d.x;
      chk.y += d.y;
      nrm += xi.x * xi.x + xi.y * xi.y;
    }
    if (e.v0) {
      t.x += p.v0.x;
      t.y += p.v0.y;
    }
    if (e.vout) st_stream<NT>(e.vout + i, t);
    if (e.mirror) {
      const int sp = e.mirror[slot];
      if (sp >= 0) e.slab[sp] = t;
    }
    if (e.acc_skip) return;  // folded into a later term's update
    double2 r;
    if (e.acc_in) {
      r = p.acc;
    } else {
      const double2 ps = (e.n_defer == 1) ? p.v0 : xi;     // Psi itself: v_0
      r = make_double2(e.a_prev * ps.x, e.a_prev * ps.y);  // lmul!(a[1], Psi)  :172
    }
    if (e.n_defer == 2) {
      r.x = fma(e.a_d2, p.v0.x, r.x);
      r.y = fma(e.a_d2, p.v0.y, r.y);
    }
    if (e.n_defer >= 1) {
      r.x = fma(e.a_d1, xi.x, r.x);
      r.y = fma(e.a_d1, xi.y, r.y);
    }
    r.x = fma(e.a, t.x, r.x);  // axpy!(a[i], v, Psi)  :182, :205
    r.y = fma(e.a, t.y, r.y);
    if (e.apply_phase) r = cmul(e.phase, r);  // lmul!(exp(-i beta dt), Psi)  :211
    st_stream<NT>(e.acc_out + i, r);
  }
  // row() in two halves for a kernel that needs the new term's VALUE back (the two-term strip walk, kernels_walk2.hip, feeds
  // v_m of a row into the row sums of v_{m+1}): term() = the value row() would store -- the same operations in the same order,
  // hence the same bits; no normalisation check, no mirror map (that kernel is not taken with either) --, finish() = its stores.
  __device__ __forceinline__ double2 term(double2 s, const Pre& p) const {
    double2 t = make_double2(fma(-e.beta, p.xi.x, s.x), fma(-e.beta, p.xi.y, s.y));
    t = cmul(e.c, t);
    if (e.v0) {
      t.x += p.v0.x;
      t.y += p.v0.y;
    }
    return t;
  }
  __device__ __forceinline__ void finish(int64_t i, double2 t, const Pre& p) const {
    const double2 xi = p.xi;
    if (e.vout) st_stream<NT>(e.vout + i, t);
    if (e.acc_skip) return;
    double2 r;
    if (e.acc_in) {
      r = p.acc;
    } else {
      const double2 ps = (e.n_defer == 1) ? p.v0 : xi;
      r = make_double2(e.a_prev * ps.x, e.a_prev * ps.y);
    }
    if (e.n_defer == 2) {
      r.x = fma(e.a_d2, p.v0.x, r.x);
      r.y = fma(e.a_d2, p.v0.y, r.y);
    }
    if (e.n_defer >= 1) {
      r.x = fma(e.a_d1, xi.x, r.x);
      r.y = fma(e.a_d1, xi.y, r.y);
    }
    r.x = fma(e.a, t.x, r.x);
    r.y = fma(e.a, t.y, r.y);
    if (e.apply_phase) r = cmul(e.phase, r);
    st_stream<NT>(e.acc_out + i, r);
  }
};
using ChebyOp = ChebyOpT<false>;

struct PlainOp {
  PlainEpi e;
  double inv = 1.0;   // 1 / |x| of the folded normalisation (begin())
  double2 np;         // this thread's partial of |x|^2 (begin_issue())
  struct Pre {
    double2 y;
  };
  // The scale of the folded "norm + scale", in two halves: the load of the partials is issued first thing in
  // the kernel, the reduction (a barrier) runs after the row sums, just before the first row() -- so the
  // workgroup's matrix loads do not queue behind it.  All threads of the workgroup call both.
  __device__ __forceinline__ void begin_issue() {
    static_assert(kRedBlocks == kThreads, "one partial per thread");
    if (e.norm_part) np = e.norm_part[threadIdx.x];
  }
  __device__ __forceinline__ void begin(double2* lds4) {
    if (!e.norm_part) return;
    const double2 s2 = block_sum(np, lds4);
    const double h = sqrt(s2.x);                       // h = norm(q[j])              src/arnoldi.jl:89
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (e.hess_slot) *e.hess_slot = make_double2(e.dt * h, 0.0);   // :90
      if (e.norm_slot) *e.norm_slot = h;
      // everything the earlier kernels of the column wrote for the host is complete (kernel boundary); the two
      // stores above are ordered before the flag by the release
      if (e.flag) __hip_atomic_store(e.flag, e.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    inv = (h < e.norm_min) ? 1.0 : 1.0 / h;            // lmul!(1 / h, q[j])          :96  (not past a breakdown :91-95)
  }
  static constexpr bool kHasXi = false;
  __device__ __forceinline__ const double2* xloc() const { return nullptr; }
  static __device__ __forceinline__ double2 xi_of(const Pre&) { return make_double2(0.0, 0.0); }
  __device__ __forceinline__ Pre pre(int64_t i) const {
    Pre p;
    p.y = e.beta_zero ? make_double2(0.0, 0.0) : e.y[i];
    return p;
  }
  __device__ __forceinline__ void row(int64_t i, double2 s, const Pre& p, double2&, double&, int64_t) const {
    if (e.norm_part) {
      s.x *= inv;
      s.y *= inv;
      if (e.qn_out) {
        const double2 xi = e.xloc[i];
        e.qn_out[i] = make_double2(xi.x * inv, xi.y * inv);
      }
    }
    double2 r = cmul(e.alpha, s);
    if (!e.beta_zero) {
      const double2 by = cmul(e.beta, p.y);
      r.x += by.x;
      r.y += by.y;
    }
    e.y[i] = r;
  }
};

template <class Op>
__device__ __forceinline__ void finish_check(const Op&, double2, double, double2*) {}
template <>
__device__ __forceinline__ void finish_check<ChebyOp>(const ChebyOp& op, double2 chk, double nrm,
                                                      double2* lds) {
  if (op.e.check_partials) {
    const double2 a = block_sum(chk, lds);
    const double2 b = block_sum(make_double2(nrm, 0.0), lds);
    if (threadIdx.x == 0) {
      double* p = op.e.check_partials + 3 * (size_t)blockIdx.x;
      p[0] = a.x;
      p[1] = a.y;
      p[2] = b.x;
    }
  }
}

// ---------------------------------------------------------------------------
// RBCSR SpMV: one wavefront streams one 64-row block, lane r owns row 64 b + r.
// Per k: one 1-KiB coalesced load of 64 values, one 16-B gather of x per lane; column
// indices arrive four k at a time in one 1-KiB load.  No cross-lane reduction, no LDS.
// ---------------------------------------------------------------------------
typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));

// streaming (read-once) loads of the matrix: the `nt` policy keeps the matrix from
// displacing the vectors in L2 / Infinity Cache
template <bool NT>
__device__ __forceinline__ double2 ld_val(const double2* p) {
  if (NT) {
    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
    return make_double2(t.x, t.y);
  }
  return *p;
}
// real-valued operators (all terms and coefficients real: half the matrix bytes); the value
// enters the same complex FMA sequence with a zero imaginary part, so results are identical
template <bool NT>
__device__ __forceinline__ double2 ld_val(const double* p) {
  if (NT) return make_double2(__builtin_nontemporal_load(p), 0.0);
  return make_double2(*p, 0.0);
}
template <bool NT>
__device__ __forceinline__ int4 ld_col(const int4* p) {
  if (NT) {
    const i4v t = __builtin_nontemporal_load(reinterpret_cast<const i4v*>(p));
    return make_int4(t.x, t.y, t.z, t.w);
  }
  return *p;
}

typedef int i2v __attribute__((ext_vector_type(2)));

// Column indices of quad q for this lane.  A block stores either int32 columns or, when
// every column of the block is within +-32767 of its row (banded H), int16 deltas to the
// lane's own row: 2 instead of 4 bytes per entry of index traffic; a stencil block one distance per
// slot (mode 2); a block-map block (mode 3: engine_core.hip, encode_col_sections) one column block per
// slot + one byte per entry.
template <bool NT>
__device__ __forceinline__ int4 ld_cols(const char* __restrict__ colbytes, int64_t meta, int q, int lane, int rowc) {
  const char* p = colbytes + (meta >> 2);
  const int mode = (int)(meta & 3);
  if (mode == 2) {  // stencil block: one delta per slot for the whole block (wave-uniform load)
    const int4 d = *(reinterpret_cast<const int4*>(p) + q);
    return make_int4(rowc + d.x, rowc + d.y, rowc + d.z, rowc + d.w);
  }
  if (mode == 3) {  // block map: one column block per slot for the whole block (wave-uniform) + the lane inside it, a byte per row and slot
    const char* pq = p + (size_t)q * (16 + 4 * 64);
    const int4 cb = *reinterpret_cast<const int4*>(pq);
    const unsigned* lp = reinterpret_cast<const unsigned*>(pq + 16) + lane;
    const unsigned lb = NT ? __builtin_nontemporal_load(lp) : *lp;
    return make_int4((cb.x << 6) | (int)(lb & 63u), (cb.y << 6) | (int)((lb >> 8) & 63u), (cb.z << 6) | (int)((lb >> 16) & 63u),
                     (cb.w << 6) | (int)((lb >> 24) & 63u));
  }
  if (mode == 1) {
    const i2v* q8 = reinterpret_cast<const i2v*>(p) + (size_t)q * 64 + lane;
    i2v t;
    if (NT) t = __builtin_nontemporal_load(q8);
    else t = *q8;
    return make_int4(rowc + (short)(t.x & 0xffff), rowc + (short)(t.x >> 16), rowc + (short)(t.y & 0xffff),
                     rowc + (short)(t.y >> 16));
  }
  return ld_col<NT>(reinterpret_cast<const int4*>(p) + (size_t)q * 64 + lane);
}

// one slot of a stencil lower section (engine_core.hip: LowerStencilSlot): column = row + delta,
// the conj-transposed value sits at pb(column block) + column % 64
struct LowerStencilSlot {
  int delta, cb0;
  int64_t pb0, pb1, pad;
};

// Hermitian-packed format: a lower entry multiplies by the complex conjugate of the stored transposed value
__device__ __forceinline__ void cfma_conj(double2& s, const double2 a, const double2 b) {  // s += conj(a) * b
  s.x = fma(a.x, b.x, s.x);
  s.x = fma(a.y, b.y, s.x);
  s.y = fma(a.x, b.y, s.y);
  s.y = fma(-a.y, b.x, s.y);
}
template <class VT>
__device__ __forceinline__ double2 ld_tr(const VT* __restrict__ vals, int pos) {
  const double2 a = ld_val<false>(vals + (pos < 0 ? 0 : pos));
  return pos < 0 ? make_double2(0.0, 0.0) : a;
}


}  // namespace qp
