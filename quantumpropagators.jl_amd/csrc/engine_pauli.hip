// Matrix-free operator for QUBIT-REGISTER generators: H = sum_l c_l H_l with every H_l a sum of Pauli strings.
//
// The reference's typical generator is a lazy sum of a few sparse matrices (src/generators.jl:634-645); for a register of n qubits
// those matrices are sums of Pauli strings, and a string P = i^{nY} X^x Z^z (bit masks x, z over the qubits; a Y contributes to
// both and one factor i) acts as                       (P psi)[r] = i^{nY} (-1)^{popcount((r xor x) and z)} psi[r xor x].
// Stored as a sparse matrix a 20-spin Ising chain is 21 entries per row -- 108 B per row and term even with the value dictionary
// and the block-map columns (kernels_coded.hip: 37 us per fused Chebyshev term); applied from the masks it is ZERO matrix bytes:
// the row's partners r xor x are, for every x, one permuted 1-KiB line of the vector (bits below 6 permute the lanes inside the
// wavefront's own line), the signs are a popcount, and what the term moves is the vectors alone.
//
// One wavefront per 64 rows.  Strings are grouped by their x mask (a group shares the gathered element); the group's weight for
// row r is  w_g(r) = sum_{t in g} coef_t (-1)^{popcount((r xor x_g) and z_t)},  coef_t = scale c_{l(t)} a_t i^{nY_t}  kept
// current on the device by evaluate! (qp_operator_set_coeffs -> pauli_refresh: a few hundred numbers).  Row sums in group order
// (groups ascending in x, strings in input order inside a group), one accumulator: run-to-run identical bits.
// The fused Chebyshev term (ChebyOp epilogue in the same launch) and the plain y = beta y + alpha A x both live here; everything
// else a matrix-free operator can do (Arnoldi / Newton / specrange through qp_mul-style applications) comes with QP_FMT_MATFREE.
#include <algorithm>

#include "engine.h"
#include "kernel_common.h"

namespace {

using qp::cplx;

struct Pauli {
  qp_ctx* ctx = nullptr;
  int nq = 0;
  int64_t n = 0;
  int nterms = 0, ngroups = 0;
  // host: strings sorted by (x mask, input order)
  std::vector<uint32_t> xmask, zmask;
  std::vector<cplx> amp;            // a_t i^{nY_t}
  std::vector<int> term_op;         // l(t): which H_l of the lazy sum
  std::vector<uint32_t> gx;         // [ngroups] x mask of the group
  std::vector<int> gfirst;          // [ngroups + 1]
  // device
  uint32_t* d_gx = nullptr;
  int* d_gfirst = nullptr;
  uint32_t* d_tz = nullptr;
  double2* d_tc = nullptr;          // [nterms] current coefficients
  double2* scratch = nullptr;       // n entries (the unfused paths of a matrix-free operator)
  // the diagonal group (x mask 0: the ZZ.. strings) as a vector of per-row weights, rebuilt by evaluate! (pauli_refresh)
  int ndiag = 0;                    // strings of the diagonal group (group 0), 0: none
  double2* d_diag = nullptr;        // n entries of storage; read as double[n] while every current coefficient of the group is real
  bool diag_real = false;
  bool walked_real = false;         // every current coefficient of the walked strings (those outside the diagonal group) is real (pauli_refresh)
  bool walked_nosign = false;       // none of them has a Z mask (X strings only): no sign to compute
};

// epilogue of the plain application
struct PauliPlain {
  double2* y;
  double2 alpha, beta;
  int beta_zero;
};

constexpr int kPauliMaxGroups = 1024;   // tables staged in LDS up to these sizes (above: read through the scalar cache)
constexpr int kPauliMaxStrings = 2048;

struct PauliTables {
  const uint32_t* gx;      // [ngroups] x mask of the group (ascending; a diagonal group, x = 0, is listed first)
  const int* gfirst;       // [ngroups + 1]
  const uint32_t* tz;      // [nstrings]
  const double2* tc;       // [nstrings] current coefficients
  int ngroups, nstrings;
  int g_begin;             // first group the term kernel walks (1 when the diagonal group is applied from the diagonal vector)
  int g_hi;                // first walked group whose x mask reaches another 64-row block (masks ascending)
  int gs;                  // > 0: every walked group has exactly gs strings (an Ising chain's X_i: 1; XX + YY pairs: 2) -- the strings of
                           // group g are then gfirst[g_begin] + (g - g_begin) gs .. without a look-up of gfirst per group
  const double* diag_r;    // [n] the diagonal group's weights when they are all real, else
  const double2* diag_c;   //     complex (both NULL: no diagonal vector)
};

// the element of lane (lane xor m), m wave-uniform, 0 < m < 64
__device__ __forceinline__ double2 lane_xor2(double2 v, int lane, int m) {
  const int idx = (lane ^ m) << 2;
  return make_double2(__hiloint2double(__builtin_amdgcn_ds_bpermute(idx, __double2hiint(v.x)), __builtin_amdgcn_ds_bpermute(idx, __double2loint(v.x))),
                      __hiloint2double(__builtin_amdgcn_ds_bpermute(idx, __double2hiint(v.y)), __builtin_amdgcn_ds_bpermute(idx, __double2loint(v.y))));
}

// One wavefront per 64-row block (the grid-stride loop only matters beyond 2^31 workgroups).  A group's
// partner elements x[r xor xm] are the 1-KiB line of block (b xor (xm >> 6)) with its lanes permuted by xm & 63: one coalesced,
// line-aligned load per group with high bits (none for a group inside the block: the wavefront's own line) + a lane permutation
// through the LDS crossbar -- no address ever leaves the wavefront's line set.
// FAST (with GS = 1): X strings with real coefficients only (a transverse field) -- no sign, two FMAs per string instead of four
template <class Epi, bool STAGE, int GS = 0, bool FAST = false>
__global__ __launch_bounds__(256) void pauli_spmv_kernel(PauliTables T, const double2* __restrict__ x, int64_t nblocks, Epi ep) {
  // dynamic LDS sized by the launcher to the tables it stages: [tc: nstrings double2][tz: nstrings][gx: ngroups][first: ngroups + 1]
  extern __shared__ double2 pauli_lds[];
  double2* s_tc = pauli_lds;
  uint32_t* s_tz = reinterpret_cast<uint32_t*>(s_tc + (STAGE ? T.nstrings : 0));
  uint32_t* s_gx = s_tz + (STAGE ? T.nstrings : 0);
  int* s_first = reinterpret_cast<int*>(s_gx + (STAGE ? T.ngroups : 0));
  if constexpr (STAGE) {
    for (int i = threadIdx.x; i < T.ngroups; i += 256) s_gx[i] = T.gx[i];
    for (int i = threadIdx.x; i <= T.ngroups; i += 256) s_first[i] = T.gfirst[i];
    for (int i = threadIdx.x; i < T.nstrings; i += 256) {
      s_tz[i] = T.tz[i];
      s_tc[i] = T.tc[i];
    }
    __syncthreads();
  }
  const uint32_t* __restrict__ gx = STAGE ? s_gx : T.gx;
  const int* __restrict__ gfirst = STAGE ? s_first : T.gfirst;
  const uint32_t* __restrict__ tz = STAGE ? s_tz : T.tz;
  const double2* __restrict__ tc = STAGE ? s_tc : T.tc;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned wg = qp::xcd_remap(blockIdx.x, gridDim.x);
  for (int64_t b = (int64_t)wg * 4 + wave; b < nblocks; b += (int64_t)gridDim.x * 4) {
    const int64_t r = b * qp::kRB + lane;
    typename std::conditional<std::is_same<Epi, PauliPlain>::value, int, typename qp::ChebyOp::Pre>::type pre{};
    double2 yold = make_double2(0.0, 0.0), own;
    if constexpr (std::is_same<Epi, PauliPlain>::value) {
      if (!ep.beta_zero) yold = ep.y[r];
      own = x[r];
    } else {
      pre = ep.pre(r);
      own = pre.xi;      // (the launcher takes this kernel only with xloc == x)
    }
    double2 s = make_double2(0.0, 0.0);
    if (T.diag_r) {
      const double d = T.diag_r[r];
      s = make_double2(d * own.x, d * own.y);
    } else if (T.diag_c) {
      s = qp::cmul(T.diag_c[r], own);
    }
    constexpr int GB = 8;      // groups whose lines are in flight together
    const int tbase = GS > 0 ? __builtin_amdgcn_readfirstlane(gfirst[T.g_begin]) : 0;
    // the strings of group g applied to its (permuted) line pv; signs as an XOR of the sign bit (bit 31 of the high dword): no
    // compare / select per string
    auto apply = [&](int g, uint32_t xmg, const double2& pv) {
      const uint32_t partner = (uint32_t)r ^ xmg;
      int t0, t1;
      if constexpr (GS > 0) {      // groups of GS strings each: no look-up (a dependent LDS round trip + readfirstlane per group otherwise)
        t0 = tbase + (g - T.g_begin) * GS;
        t1 = t0 + GS;
      } else {
        t0 = __builtin_amdgcn_readfirstlane(gfirst[g]);
        t1 = __builtin_amdgcn_readfirstlane(gfirst[g + 1]);
      }
      if constexpr (FAST) {
        const double c = tc[t0].x;
        s.x = fma(c, pv.x, s.x);
        s.y = fma(c, pv.y, s.y);
      } else if (GS == 1 || (GS == 0 && t1 - t0 == 1)) {      // one string (every X_i of an Ising chain): its sign goes onto the gathered element
        const int flip = (int)((__popc(partner & tz[t0]) & 1u) << 31);
        const double2 c = tc[t0];
        const double2 sv = make_double2(__hiloint2double(__double2hiint(pv.x) ^ flip, __double2loint(pv.x)),
                                        __hiloint2double(__double2hiint(pv.y) ^ flip, __double2loint(pv.y)));
        qp::cfma(s, c, sv);
      } else {
        double2 w = make_double2(0.0, 0.0);
#pragma unroll
        for (int t = t0; t < (GS > 0 ? t0 + GS : t1); ++t) {
          const double2 c = tc[t];
          const int flip = (int)((__popc(partner & tz[t]) & 1u) << 31);
          w.x += __hiloint2double(__double2hiint(c.x) ^ flip, __double2loint(c.x));
          w.y += __hiloint2double(__double2hiint(c.y) ^ flip, __double2loint(c.y));
        }
        qp::cfma(s, w, pv);
      }
    };
    // groups inside the block (x mask < 64; the masks are listed in ascending order): lane permutations of the wavefront's own line
    for (int g = T.g_begin; g < T.g_hi; ++g) {
      const uint32_t xmg = (uint32_t)__builtin_amdgcn_readfirstlane((int)gx[g]);
      apply(g, xmg, lane_xor2(own, lane, (int)xmg));
    }
    // groups that reach another block: GB lines in flight, then a lane permutation where the mask has low bits too
    for (int g0 = T.g_hi; g0 < T.ngroups; g0 += GB) {
      double2 xv[GB];
      uint32_t xm[GB];
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int g = min(g0 + u, T.ngroups - 1);
        xm[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)gx[g]);      // (wave-uniform: the branches below are scalar)
        xv[u] = x[(b ^ (int64_t)(xm[u] >> 6)) * qp::kRB + lane];
      }
#pragma unroll
      for (int u = 0; u < GB; ++u) {
        const int g = g0 + u;
        if (g >= T.ngroups) break;
        const int lo = (int)(xm[u] & 63u);
        apply(g, xm[u], lo ? lane_xor2(xv[u], lane, lo) : xv[u]);
      }
    }
    if constexpr (std::is_same<Epi, PauliPlain>::value) {
      double2 out = qp::cmul(ep.alpha, s);
      if (!ep.beta_zero) {
        const double2 by = qp::cmul(ep.beta, yold);
        out.x += by.x;
        out.y += by.y;
      }
      ep.y[r] = out;
    } else {
      double2 chk = make_double2(0.0, 0.0);
      double nrm = 0.0;
      ep.row(r, s, pre, chk, nrm, 0);
    }
  }
}

// the diagonal group's weights as a vector: d[r] = sum_{t in group 0} c_t (-1)^{popcount(r and z_t)}  (once per evaluate!, not per term)
__global__ __launch_bounds__(256) void pauli_diag_kernel(const uint32_t* __restrict__ tz, const double2* __restrict__ tc, int count,
                                                         double* __restrict__ dr, double2* __restrict__ dc, int64_t n) {
  __shared__ uint32_t s_tz[kPauliMaxStrings];
  __shared__ double2 s_tc[kPauliMaxStrings];
  for (int i = threadIdx.x; i < count; i += 256) {
    s_tz[i] = tz[i];
    s_tc[i] = tc[i];
  }
  __syncthreads();
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
    double2 w = make_double2(0.0, 0.0);
    for (int t = 0; t < count; ++t) {
      const double2 c = s_tc[t];
      const bool neg = (__popc((uint32_t)r & s_tz[t]) & 1) != 0;
      w.x += neg ? -c.x : c.x;
      w.y += neg ? -c.y : c.y;
    }
    if (dr) dr[r] = w.x;
    else dc[r] = w;
  }
}

int pauli_refresh(qp_operator* op) {
  Pauli* P = static_cast<Pauli*>(op->mf);
  const int drift = op->nops - op->ncoeffs;
  std::vector<double2> tc((size_t)P->nterms);
  for (int t = 0; t < P->nterms; ++t) {
    cplx c = op->scale * P->amp[(size_t)t];
    const int l = P->term_op[(size_t)t];
    if (l >= drift) c *= op->coeffs[(size_t)(l - drift)];
    tc[(size_t)t] = make_double2(c.real(), c.imag());
  }
  // stream-ordered (a term launched before this call keeps the coefficients it was launched with); from pageable memory the call
  // returns once the bytes have left `tc` for the runtime's staging buffer
  QP_HIP(hipMemcpyAsync(P->d_tc, tc.data(), tc.size() * sizeof(double2), hipMemcpyHostToDevice, op->ctx->stream));
  P->walked_real = true;
  for (int t = P->ndiag; t < P->nterms; ++t) P->walked_real = P->walked_real && tc[(size_t)t].y == 0.0;
  if (P->ndiag > 0) {
    P->diag_real = true;
    for (int t = 0; t < P->ndiag; ++t) P->diag_real = P->diag_real && tc[(size_t)t].y == 0.0;
    const int grid = (int)std::min<int64_t>((P->n + 255) / 256, 4096);
    hipLaunchKernelGGL(pauli_diag_kernel, dim3(grid), dim3(256), 0, op->ctx->stream, P->d_tz, P->d_tc, P->ndiag,
                       P->diag_real ? reinterpret_cast<double*>(P->d_diag) : nullptr, P->diag_real ? nullptr : P->d_diag, P->n);
    QP_HIP(hipGetLastError());
    op->ctx->stats.n_launch++;
  }
  return QP_OK;
}

static PauliTables pauli_tables(const Pauli* P) {
  PauliTables T;
  T.gx = P->d_gx;
  T.gfirst = P->d_gfirst;
  T.tz = P->d_tz;
  T.tc = P->d_tc;
  T.ngroups = P->ngroups;
  T.nstrings = P->nterms;
  T.g_begin = P->ndiag > 0 ? 1 : 0;
  T.g_hi = T.g_begin;
  while (T.g_hi < P->ngroups && P->gx[(size_t)T.g_hi] < 64u) ++T.g_hi;
  T.gs = 0;
  if (T.g_begin < P->ngroups) {
    const int gs0 = P->gfirst[(size_t)T.g_begin + 1] - P->gfirst[(size_t)T.g_begin];
    bool same = gs0 >= 1 && gs0 <= 2;
    for (int g = T.g_begin; same && g < P->ngroups; ++g) same = P->gfirst[(size_t)g + 1] - P->gfirst[(size_t)g] == gs0;
    if (same) T.gs = gs0;
  }
  T.diag_r = (P->ndiag > 0 && P->diag_real) ? reinterpret_cast<const double*>(P->d_diag) : nullptr;
  T.diag_c = (P->ndiag > 0 && !P->diag_real) ? P->d_diag : nullptr;
  return T;
}

template <class Epi>
static void pauli_launch(hipStream_t s, const Pauli* P, const double2* x, const Epi& ep) {
  const int64_t nblocks = P->n / qp::kRB;
  const PauliTables T = pauli_tables(P);
  // one 64-row block per wavefront (measured: 22 spins 109 -> 96 us per term, 24 spins 509 -> 480 against a few blocks per wavefront
  // in turn; the tables a workgroup stages are a few hundred bytes)
  const size_t lds_tables = (size_t)P->nterms * (sizeof(double2) + sizeof(uint32_t)) + (size_t)(2 * P->ngroups + 1) * sizeof(int);
  const unsigned grid = (unsigned)std::min<int64_t>((nblocks + 3) / 4, (int64_t)INT32_MAX);
  if (P->ngroups <= kPauliMaxGroups && P->nterms <= kPauliMaxStrings) {
    const size_t lds = lds_tables;
    if (T.gs == 1 && P->walked_real && P->walked_nosign) hipLaunchKernelGGL((pauli_spmv_kernel<Epi, true, 1, true>), dim3(grid), dim3(256), lds, s, T, x, nblocks, ep);
    else if (T.gs == 1) hipLaunchKernelGGL((pauli_spmv_kernel<Epi, true, 1>), dim3(grid), dim3(256), lds, s, T, x, nblocks, ep);
    else if (T.gs == 2) hipLaunchKernelGGL((pauli_spmv_kernel<Epi, true, 2>), dim3(grid), dim3(256), lds, s, T, x, nblocks, ep);
    else hipLaunchKernelGGL((pauli_spmv_kernel<Epi, true>), dim3(grid), dim3(256), lds, s, T, x, nblocks, ep);
  } else {
    hipLaunchKernelGGL((pauli_spmv_kernel<Epi, false>), dim3(grid), dim3(256), 0, s, T, x, nblocks, ep);
  }
}

int pauli_apply(hipStream_t s, void* self, const double2* x, double2* y, double2 alpha, double2 beta, qp::Stats* st) {
  Pauli* P = static_cast<Pauli*>(self);
  PauliPlain ep{y, alpha, beta, (beta.x == 0.0 && beta.y == 0.0) ? 1 : 0};
  pauli_launch(s, P, x, ep);
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    st->n_matvec++;
  }
  return QP_OK;
}

int pauli_cheby(hipStream_t s, void* self, const double2* x, const qp::ChebyEpi& e, qp::Stats* st) {
  Pauli* P = static_cast<Pauli*>(self);
  qp::ChebyOp op{e};
  pauli_launch(s, P, x, op);
  QP_HIP(hipGetLastError());
  if (st) {
    st->n_launch++;
    st->n_matvec++;
    st->spmv_bytes += 80.0 * (double)P->n;
  }
  return QP_OK;
}

double2* pauli_scratch(void* self) { return static_cast<Pauli*>(self)->scratch; }

void pauli_free(qp_operator* op) {
  Pauli* P = static_cast<Pauli*>(op->mf);
  if (!P) return;
  if (P->d_gx) (void)hipFree(P->d_gx);
  if (P->d_gfirst) (void)hipFree(P->d_gfirst);
  if (P->d_tz) (void)hipFree(P->d_tz);
  if (P->d_tc) (void)hipFree(P->d_tc);
  if (P->scratch) (void)hipFree(P->scratch);
  if (P->d_diag) (void)hipFree(P->d_diag);
  delete P;
  op->mf = nullptr;
}

}  // namespace

extern "C" {

int qp_pauli_operator_create(qp_ctx* ctx, int nqubits, const qp_pauli_string* strings, int nstrings, int nops, int ncoeffs,
                             qp_operator** out) {
  QP_TRY
  if (!ctx || !out || !strings || nstrings < 1 || nops < 1 || ncoeffs < 0 || ncoeffs > nops)
    return qp::fail(QP_E_BAD_ARG, "qp_pauli_operator_create: bad arguments");
  if (nqubits < 6 || nqubits > 30)
    return qp::fail(QP_E_BAD_ARG, "qp_pauli_operator_create: %d qubits (6 .. 30: a wavefront owns 64 rows; 2^30 rows x 16 B is the largest state)", nqubits);
  const uint64_t all = (1ull << nqubits) - 1;
  for (int t = 0; t < nstrings; ++t) {
    if ((strings[t].xmask | strings[t].zmask) & ~all) return qp::fail(QP_E_BAD_ARG, "Pauli string %d acts on a qubit beyond the register", t);
    if (strings[t].op < 0 || strings[t].op >= nops) return qp::fail(QP_E_BAD_ARG, "Pauli string %d belongs to term %d of %d", t, strings[t].op, nops);
  }
  QP_CHECK(use(ctx));
  auto op = std::make_unique<qp_operator>();
  auto P = std::make_unique<Pauli>();
  op->ctx = ctx;
  op->A.tun = &ctx->tun;
  P->ctx = ctx;
  P->nq = nqubits;
  P->n = (int64_t)1 << nqubits;
  P->nterms = nstrings;
  // strings sorted by x mask, input order kept inside a group (stable): the summation order of a row
  std::vector<int> order((size_t)nstrings);
  for (int t = 0; t < nstrings; ++t) order[(size_t)t] = t;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return strings[a].xmask < strings[b].xmask; });
  static const cplx ipow[4] = {cplx(1, 0), cplx(0, 1), cplx(-1, 0), cplx(0, -1)};
  for (int k = 0; k < nstrings; ++k) {
    const qp_pauli_string& sgl = strings[order[(size_t)k]];
    const uint32_t xm = (uint32_t)sgl.xmask, zm = (uint32_t)sgl.zmask;
    if (P->gx.empty() || P->gx.back() != xm) {
      P->gx.push_back(xm);
      P->gfirst.push_back(k);
    }
    P->xmask.push_back(xm);
    P->zmask.push_back(zm);
    P->amp.push_back(cplx(sgl.coef.re, sgl.coef.im) * ipow[__builtin_popcount(xm & zm) & 3]);      // one factor i per Y
    P->term_op.push_back(sgl.op);
  }
  P->gfirst.push_back(nstrings);
  P->ngroups = (int)P->gx.size();
  op->mf_free = pauli_free;
  op->mf_refresh = pauli_refresh;
  op->mf = P.get();
  Pauli* Pp = P.release();
  struct Guard {
    qp_operator* op;
    bool armed = true;
    ~Guard() {
      if (armed && op->mf_free) op->mf_free(op);
    }
  } guard{op.get()};
  QP_CHECK(dev_alloc(&Pp->d_gx, (size_t)Pp->ngroups));
  QP_CHECK(dev_alloc(&Pp->d_gfirst, (size_t)Pp->ngroups + 1));
  QP_CHECK(dev_alloc(&Pp->d_tz, (size_t)nstrings));
  QP_CHECK(dev_alloc(&Pp->d_tc, (size_t)nstrings));
  QP_CHECK(dev_alloc(&Pp->scratch, (size_t)Pp->n));
  // the diagonal group as a vector when it is worth a 16 N byte array: at least two strings (one string is a sign flip)
  if (Pp->gx[0] == 0 && Pp->gfirst[1] >= 2 && Pp->gfirst[1] <= kPauliMaxStrings) {
    Pp->ndiag = Pp->gfirst[1];
    QP_CHECK(dev_alloc(&Pp->d_diag, (size_t)Pp->n));
  }
  Pp->walked_nosign = true;
  for (int t = Pp->ndiag; t < nstrings; ++t) Pp->walked_nosign = Pp->walked_nosign && Pp->zmask[(size_t)t] == 0;
  QP_HIP(hipMemcpy(Pp->d_gx, Pp->gx.data(), Pp->gx.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  QP_HIP(hipMemcpy(Pp->d_gfirst, Pp->gfirst.data(), Pp->gfirst.size() * sizeof(int), hipMemcpyHostToDevice));
  QP_HIP(hipMemcpy(Pp->d_tz, Pp->zmask.data(), Pp->zmask.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  DevMatrix& A = op->A;
  A.format = QP_FMT_MATFREE;
  A.nrows = A.ncols = Pp->n;
  A.nnz = INT64_MAX / 4;   // never "small": the persistent kernels need stored entries
  A.stored = 0;
  A.matfree = Pp;
  A.matfree_apply = pauli_apply;
  A.matfree_scratch = pauli_scratch;
  A.matfree_cheby = pauli_cheby;
  op->nops = nops;
  op->ncoeffs = ncoeffs;
  op->coeffs.assign((size_t)ncoeffs, cplx(1.0));
  QP_CHECK(pauli_refresh(op.get()));
  guard.armed = false;
  *out = op.release();
  return QP_OK;
  QP_CATCH
}

}  // extern "C"
