// Matrix-free Liouvillian: the superoperator of liouvillian(H, c_ops;
// convention) (src/generators.jl:473-631) applied to the column-major vec(rho) as n x n
// complex products on the fp64 matrix cores (two hand-written kernels by size; rocBLAS zgemm, a plain
// library GEMM, for the sizes outside them), instead of an n^2 x n^2 sparse matrix with 2 n^3 entries for dense H:
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
  double2* MRn = nullptr;      // -M_R                      } operands of the 32 x 32 matrix-core kernel, whose
  std::vector<double2*> Ah;    // A_k^+ (conj-transposed)   } products all have coefficient one and no conjugation
  double2* T = nullptr;        // n x n workspace: A_k rho (library path)
  double2* Tk = nullptr;       // nc x (n x n) workspace: all A_k rho (fused path)
  double2* scratch = nullptr;  // n^2 workspace for the unfused Chebyshev term
  cplx scale = 1.0;            // the operator's scale, applied to the dissipator GEMMs
};

// ML / MR = sum_l c_l H_l  -/+  g G   (c_l already carry s_h and the scale, g = scale s_d / 2)
__global__ __launch_bounds__(qp::kThreads) void liouville_combine_kernel(double2* __restrict__ ML, double2* __restrict__ MR,
                                                                         double2* __restrict__ MRn,
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
    if (MRn) MRn[p] = make_double2(-(h.x + gg.x), -(h.y + gg.y));
  }
}

// ---------------------------------------------------------------------------
// Hand-written fp64 matrix-core kernel for the sizes where a chain of library GEMMs is bound
// by its launches (default: n < 260):  Y = beta Y + sum_j alpha_j P_j op_j(Q_j)  in ONE launch, all
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

template <int BM, int D>   // D = k-steps of prefetch
__global__ __launch_bounds__(256) void zgemm_sum_kernel(double2* __restrict__ Y, int n, double2 beta, GemmTerms terms,
                                                        int batched) {
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

// ---------------------------------------------------------------------------
// The sum of products for the sizes above (default 260 <= n <= 2048: knobs liouville_tile32_min_n, liouville_tile32_n):
//   Y = beta Y + alpha sum_j P_j Q_j      (`batched`: Y_z = alpha P_z Q_z for every z)
// On MI355X the fp64 MFMA runs at the rate of the fp64 vector unit and shares its issue: every vector-ALU
// instruction between two MFMAs is time the matrix pipe stands still, from the same or from another wavefront
// (tools/probe/mfma_f64_rate.hip: 27.2 ns per v_mfma_f64_16x16x4_f64 and SIMD with none, 31.8 with two, 38.0 with six).
// So the k loop of this kernel has next to no vector-ALU work in it:
//   * one workgroup = one 32 x 32 tile of Y; each of its four wavefronts owns a quarter of the inner dimension of
//     every product and keeps the whole tile -- 2 x 2 MFMA tiles, real and imaginary part, 64 accumulator
//     registers (VGPR form: -mllvm -amdgpu-mfma-vgpr-form, no AGPR copies) -- so a k-step of 4 is four 1-KiB
//     loads for 16 MFMAs (16 B per clock and CU from L2; 16 x 16 tiles need twice that and run into the L2 -> CU rate);
//   * the operands go from the loaded registers into the MFMAs as they are: every product has coefficient one and
//     no conjugation -- the caller stores -M_R and the conjugate-transposed Lindblad operators once, the factor of
//     the dissipator is applied to T_k = A_k rho by the first launch's epilogue, alpha and beta by the second's --
//     and the only vector-ALU instructions per k-step are the sign flips of the two imaginary A fragments;
//   * addresses are a scalar base per operand, advanced by the scalar unit, plus a per-lane offset that never
//     changes (buffer loads); the cursors over the flattened (product, k) range advance by scalar selects,
//     not branches, so that a k-step stays one basic block and its four refills and scalar work are spread between
//     the MFMAs (sched_group_barrier);
//   * software pipeline of D slots: step g runs on slot g mod D while the slot of step g - 1 is refilled for step
//     g - 1 + D; no conditions between the first load and the last refill of the steady state (the compiler's
//     vmcnt values are exact only along an unconditional path).
// The four partial tiles are summed through LDS in wave order, as in the 16 x 16 kernel: deterministic.
// ---------------------------------------------------------------------------
struct Gemm32Terms {
  const double2* P[kMaxTerms];
  const double2* Q[kMaxTerms];
  int n_terms;
};

// 16 bytes at (wave-uniform base) + (32-bit lane offset): a buffer load, whose descriptor the scalar unit builds from
// the base -- no vector-ALU address arithmetic
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 ld_off(const double2* base, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(base), (short)0, -1, 0x00020000);
  const u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  double2 d;
  __builtin_memcpy(&d, &v, 16);
  return d;
}

template <int D>
__global__ __launch_bounds__(256) void zgemm_sum32_kernel(double2* __restrict__ Y, int n, double2 alpha, double2 beta,
                                                          Gemm32Terms terms, int batched) {
  __shared__ double red[4][4][2][4][64];   // the partial tiles of the four wavefronts (64 KB)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  const int first = batched ? blockIdx.z : 0;
  const int nt = batched ? 1 : terms.n_terms;
  const int tlast = first + nt - 1;
  double2* __restrict__ Yz = Y + (batched ? (size_t)blockIdx.z * n * n : 0);
  const int ksteps = n >> 2;   // whole k-steps; the n & 3 inner indices left over: one masked step after the loop
  const int per = (ksteps + 3) / 4;
  const int sbeg = wave * per;
  const int nsteps = max(min(ksteps, sbeg + per) - sbeg, 0);
  const int total = nt * nsteps;
  const int li = lane & 15, lk = lane >> 4;
  const int ra0 = min(row0 + li, n - 1), ra1 = min(row0 + 16 + li, n - 1);
  const int cb0 = min(col0 + li, n - 1), cb1 = min(col0 + 16 + li, n - 1);
  // A fragment: P[r + k n], k = 4 step + lk;  B fragment: Q[k + c n]
  const unsigned oa0 = (unsigned)(lk * n + ra0) * 16u, oa1 = (unsigned)(lk * n + ra1) * 16u;
  const unsigned ob0 = (unsigned)(cb0 * n + lk) * 16u, ob1 = (unsigned)(cb1 * n + lk) * 16u;
  const size_t startA = (size_t)sbeg * 4 * n, startB = (size_t)sbeg * 4;   // this wave's first k, in elements
  const size_t strideA = (size_t)4 * n, strideB = 4;

  v4d cr[2][2], ci[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) cr[a][b] = ci[a][b] = v4d{0.0, 0.0, 0.0, 0.0};

  double2 fa[D][2], fb[D][2];
  int lt = first, ls = 0;   // refill cursor: product, k-step
  const double2* baseA = terms.P[first] + startA;
  const double2* baseB = terms.Q[first] + startB;
  // the bases of the product after the current one, read from the argument block a step before they can be needed
  const double2* nextA = terms.P[min(first + 1, tlast)] + startA;
  const double2* nextB = terms.Q[min(first + 1, tlast)] + startB;
  auto load = [&](int slot) {
    fa[slot][0] = ld_off(baseA, oa0);
    fa[slot][1] = ld_off(baseA, oa1);
    fb[slot][0] = ld_off(baseB, ob0);
    fb[slot][1] = ld_off(baseB, ob1);
    asm volatile("" : "+s"(nextA), "+s"(nextB));   // both in registers here: the selects below stay selects (else: branches around the loads)
    const bool wrap = ls + 1 == nsteps;
    ls = wrap ? 0 : ls + 1;
    const int ltn = min(lt + 1, tlast);
    lt = wrap ? ltn : lt;
    baseA = wrap ? nextA : baseA + strideA;
    baseB = wrap ? nextB : baseB + strideB;
    const int lt2 = min(lt + 1, tlast);
    nextA = terms.P[lt2] + startA;
    nextB = terms.Q[lt2] + startB;
  };
  auto mfma = [&](int slot) {   // 16 MFMAs; consecutive ones never share an accumulator
    double nai[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) nai[a] = -fa[slot][a].y;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a].x, fb[slot][b].x, cr[a][b], 0, 0, 0);
        ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a].x, fb[slot][b].y, ci[a][b], 0, 0, 0);
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai[a], fb[slot][b].y, cr[a][b], 0, 0, 0);
        ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a].y, fb[slot][b].x, ci[a][b], 0, 0, 0);
      }
  };
  int s = 0;
  if (total >= 2 * D - 1) {
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
      load(d);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; s + 2 * D - 1 <= total; s += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        load((j + D - 1) % D);
        mfma(j);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                 // at most one vector-ALU instruction
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);                 // scalar work of the cursor
          if (g % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one of the four refills
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // here steps s .. s + D - 2 are loaded or in flight, in slots 0 .. D - 2
    for (; s < total; s += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        if (s + j < total) {
          if (s + j + D - 1 < total) load((j + D - 1) % D);
          mfma(j);
        }
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < total) load(d);
    for (; s < total; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (s + d < total) {
          mfma(d);
          if (s + d + D < total) load(d);
        }
      }
    }
  }
  // n not a multiple of 4: one more k-step per product for the n & 3 inner indices left over, the lanes past the end
  // masked out of the A fragment (products dealt round-robin to the wavefronts; outside the pipelined loop)
  if (n & 3) {
    const int k = ksteps * 4 + lk;
    const bool kin = k < n;
    const int kc = min(k, n - 1);
    for (int t = first + wave; t <= tlast; t += 4) {
      const double2 zero = make_double2(0.0, 0.0);
      const double2 p0 = terms.P[t][(size_t)kc * n + ra0], p1 = terms.P[t][(size_t)kc * n + ra1];
      fa[0][0] = kin ? p0 : zero;
      fa[0][1] = kin ? p1 : zero;
      fb[0][0] = terms.Q[t][(size_t)cb0 * n + kc];
      fb[0][1] = terms.Q[t][(size_t)cb1 * n + kc];
      mfma(0);
    }
  }
  // sum the four k-quarters in wave order; every wavefront finishes one of the four 16 x 16 tiles (all four partial
  // tiles of everybody go through LDS, so that no register is indexed by the wave number)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wave][a * 2 + b][0][r][lane] = cr[a][b][r];
        red[wave][a * 2 + b][1][r][lane] = ci[a][b][r];
      }
  __syncthreads();
  {
    const bool bz = (beta.x == 0.0 && beta.y == 0.0) || batched;
    const int a = wave >> 1, b = wave & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double sr = red[0][wave][0][r][lane], si = red[0][wave][1][r][lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        sr += red[w][wave][0][r][lane];
        si += red[w][wave][1][r][lane];
      }
      const int row = row0 + a * 16 + lk + 4 * r, col = col0 + b * 16 + li;
      if (row < n && col < n) {
        double vr = alpha.x * sr - alpha.y * si, vi = alpha.x * si + alpha.y * sr;
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

int launch_zgemm_sum32(hipStream_t s, double2* Y, int n, double2 alpha, double2 beta, const Gemm32Terms& terms, int batched,
                       Stats* st) {
  if (terms.n_terms == 0) return QP_OK;
  const int tiles = (n + 31) / 32;
  const dim3 grid(tiles, tiles, batched ? terms.n_terms : 1);
  // six slots: 172 VGPRs, two wavefronts per SIMD; four or five slots (three per SIMD) measured 1-2 % either way up to
  // n = 768 and 15 % slower at n = 1024 (profiles/r02/liouville_paths.txt)
  hipLaunchKernelGGL((zgemm_sum32_kernel<6>), grid, dim3(256), 0, s, Y, n, alpha, beta, terms, batched);
  QP_HIP(hipGetLastError());
  if (st) st->n_launch++;
  return QP_OK;
}

// B = A^+ (n x n, column-major), once per Lindblad operator
__global__ __launch_bounds__(256) void conj_transpose_kernel(double2* __restrict__ B, const double2* __restrict__ A, int n) {
  __shared__ double2 tile[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r = blockIdx.x * 16 + tx, c = blockIdx.y * 16 + ty;
  if (r < n && c < n) tile[ty][tx] = A[(size_t)c * n + r];
  __syncthreads();
  const int r2 = blockIdx.y * 16 + tx, c2 = blockIdx.x * 16 + ty;   // B[r2][c2] = conj(A[c2][r2])
  if (r2 < n && c2 < n) {
    const double2 v = tile[tx][ty];
    B[(size_t)c2 * n + r2] = make_double2(v.x, -v.y);
  }
}

int launch_zgemm_sum(hipStream_t s, double2* Y, int n, double2 beta, const GemmTerms& terms, int batched, Stats* st) {
  if (terms.n_terms == 0) return QP_OK;
  // 16 x 16 tiles: as many workgroups as the matrix offers (small n: the launch count decides)
  const int tiles = (n + 15) / 16;
  const dim3 grid(tiles, tiles, batched ? terms.n_terms : 1);
  hipLaunchKernelGGL((zgemm_sum_kernel<1, 8>), grid, dim3(256), 0, s, Y, n, beta, terms, batched);
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
  hipLaunchKernelGGL(liouville_combine_kernel, dim3(grid), dim3(qp::kThreads), 0, ctx->stream, L->ML, L->MR, L->MRn, L->H_dev, cb,
                     op->nops, L->nc > 0 ? L->G : nullptr, d2(op->scale * L->s_d * 0.5), n2);
  QP_HIP(hipGetLastError());
  ctx->stats.n_launch++;
  return QP_OK;
}

// y = beta y + alpha L x,  x and y the column-major n x n density matrices
int liouville_apply(hipStream_t s, void* self, const double2* x, double2* y, double2 alpha, double2 beta, Stats* st) {
  Liouville* L = static_cast<Liouville*>(self);
  const rocblas_int n = (rocblas_int)L->n;
  const qp::Tuning& tun = L->ctx->tun;
  if (L->MRn && L->n >= tun.liouville_tile32_min_n && L->n <= tun.liouville_tile32_n && 2 + L->nc <= kMaxTerms) {
    // two launches of the 32 x 32 matrix-core kernel: T_k = scale s_d A_k X (batched), then
    // Y = beta Y + alpha (M_L X + X (-M_R) + sum_k T_k A_k^+)
    if (L->nc > 0) {
      Gemm32Terms t1;
      t1.n_terms = L->nc;
      for (int k = 0; k < L->nc; ++k) {
        t1.P[k] = L->A[k];
        t1.Q[k] = x;
      }
      QP_CHECK(launch_zgemm_sum32(s, L->Tk, n, d2(L->scale * L->s_d), make_double2(0.0, 0.0), t1, 1, st));
    }
    Gemm32Terms t2;
    t2.n_terms = 2 + L->nc;
    t2.P[0] = L->ML;
    t2.Q[0] = x;
    t2.P[1] = x;
    t2.Q[1] = L->MRn;
    for (int k = 0; k < L->nc; ++k) {
      t2.P[2 + k] = L->Tk + (size_t)k * L->n * L->n;
      t2.Q[2 + k] = L->Ah[k];
    }
    QP_CHECK(launch_zgemm_sum32(s, y, n, alpha, beta, t2, 0, st));
    if (st) st->n_matvec++;
    return QP_OK;
  }
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
  if (L->MRn) (void)hipFree(L->MRn);
  for (auto p : L->Ah) (void)hipFree(p);
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
  const bool tile32 = n <= 2048;   // sizes the 32 x 32 kernel can take (knobs liouville_tile32_min_n / _n decide per call)
  if (tile32) QP_CHECK(dev_alloc(&Lp->MRn, n2));
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
      if (tile32) {
        double2* h = nullptr;
        QP_CHECK(dev_alloc(&h, n2));
        Lp->Ah.push_back(h);
        const int tiles = (int)((n + 15) / 16);
        hipLaunchKernelGGL(conj_transpose_kernel, dim3(tiles, tiles), dim3(256), 0, ctx->stream, h, d, (int)n);
        QP_HIP(hipGetLastError());
      }
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
