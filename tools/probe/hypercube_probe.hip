// Probe (round 6): the fused Chebyshev term of a transverse-field Ising chain applied from its Pauli strings (csrc/engine_pauli.hip;
// the `tfim20` / `tfim20_pauli` points of bench.py: 36-39 us per term, 0.37 of 8 TB/s by the stored form's bytes, where the four
// vector streams alone would take about 10 us).  y[r] = d[r] x[r] - h sum_i x[r xor 2^i], i < n: a hypercube stencil.  What bounds it,
// and what would a workgroup tile buy?  Bare kernels on synthetic data, n = 20 by default:
//   gather     one wavefront per 64 rows (today): bits 0-5 by lane permutes, bits 6 .. n-1 by one gathered 1-KiB line each
//   nogather   the same without the gathers (x_i stands in): the floor of the structure
//   permonly   the lane permutes only / gatheronly: the gathers only
//   tile10     a workgroup of 16 wavefronts owns 2^10 consecutive rows staged in LDS: bits 6-9 from LDS, bits 10 .. n-1 gathered
//   tile12     a workgroup of 16 wavefronts owns 2^12 rows (four per lane): bits 6-11 from LDS, bits 12 .. n-1 gathered
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe/hypercube_probe.hip -o tools/probe/hypercube_probe ; hypercube_probe [n = 20]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s -> %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

struct Args {
  const d2* x;
  const d2* p;
  const d2* acc;
  const double* diag;
  d2* y;
  d2* acc_out;
  int n, with_acc;
};

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u;
  const unsigned xcd = bid & 7u, j = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + j;
}
__device__ __forceinline__ d2 lane_xor(d2 v, int m, int lane) {
  const int a = ((lane ^ m) << 2);
  d2 r;
  r.x = __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v.x)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v.x)));
  r.y = __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v.y)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v.y)));
  return r;
}
__device__ __forceinline__ void finish(const Args& A, long long e, d2 s, d2 xi, d2 v0, d2 av) {
  d2 t;
  t.x = 2.0 * s.x - 0.1 * xi.x + v0.x;
  t.y = 2.0 * s.y - 0.1 * xi.y + v0.y;
  A.y[e] = t;
  if (A.with_acc) {
    av.x = fma(0.3, t.x, av.x);
    av.y = fma(0.3, t.y, av.y);
    A.acc_out[e] = av;
  }
}

// MODE bit 0: lane permutes, bit 1: gathers
template <int MODE>
__global__ __launch_bounds__(256) void wave_kernel(Args A) {
  const int lane = threadIdx.x & 63;
  const long long blk = (long long)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  const long long r = blk * 64 + lane;
  const d2 xi = A.x[r];
  const d2 v0 = A.p[r];
  d2 av = {0.0, 0.0};
  if (A.with_acc) av = A.acc[r];
  const double dg = A.diag[r];
  d2 s = {dg * xi.x, dg * xi.y};
  d2 g[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) g[i] = (MODE & 2) ? ((6 + i < A.n) ? A.x[r ^ (1ll << (6 + i))] : d2{0.0, 0.0}) : xi;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const d2 q = (MODE & 1) ? lane_xor(xi, 1 << i, lane) : xi;
    s.x -= 0.7 * q.x;
    s.y -= 0.7 * q.y;
  }
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    s.x -= 0.7 * g[i].x;
    s.y -= 0.7 * g[i].y;
  }
  finish(A, r, s, xi, v0, av);
}

// a workgroup of 16 wavefronts owns 2^LT consecutive rows (RPT = 2^LT / 1024 per lane), staged in LDS
template <int LT>
__global__ __launch_bounds__(1024) void tile_kernel(Args A) {
  extern __shared__ d2 lds[];
  constexpr int RPT = (1 << LT) / 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const long long base = (long long)xcd_remap(blockIdx.x, gridDim.x) << LT;
  d2 xi[RPT], v0[RPT], av[RPT], g[RPT][20 - LT > 0 ? 20 - LT : 1];
  double dg[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const long long r = base + q * 1024 + tid;
    xi[q] = A.x[r];
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const long long r = base + q * 1024 + tid;
#pragma unroll
    for (int i = 0; i < 20 - LT; ++i) g[q][i] = (LT + i < A.n) ? A.x[r ^ (1ll << (LT + i))] : d2{0.0, 0.0};
    v0[q] = A.p[r];
    av[q] = A.with_acc ? A.acc[r] : d2{0.0, 0.0};
    dg[q] = A.diag[r];
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) lds[q * 1024 + tid] = xi[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int l = q * 1024 + tid;
    d2 s = {dg[q] * xi[q].x, dg[q] * xi[q].y};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const d2 t = lane_xor(xi[q], 1 << i, lane);
      s.x -= 0.7 * t.x;
      s.y -= 0.7 * t.y;
    }
#pragma unroll
    for (int i = 6; i < LT; ++i) {
      const d2 t = lds[l ^ (1 << i)];
      s.x -= 0.7 * t.x;
      s.y -= 0.7 * t.y;
    }
#pragma unroll
    for (int i = 0; i < 20 - LT; ++i) {
      s.x -= 0.7 * g[q][i].x;
      s.y -= 0.7 * g[q][i].y;
    }
    finish(A, base + l, s, xi[q], v0[q], av[q]);
  }
}

static double rnd(unsigned long long& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return ((double)(s >> 11) / 9007199254740992.0) - 0.5;
}

template <class F>
static int time_it(const char* name, Args A, d2* bufs[3], F launch, int reps, double bytes, std::vector<d2>* out, size_t pe) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float sum = 0.0f, best = 1e30f;
  for (int it = 0; it < reps + 3; ++it) {
    A.x = bufs[it % 2];
    A.p = bufs[(it + 1) % 2];
    A.y = bufs[(it + 1) % 2];
    A.acc = A.acc_out = bufs[2];
    A.with_acc = (it % 3 == 2);
    CK(hipEventRecord(e0, 0));
    launch(A);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3) {
      sum += ms;
      best = std::min(best, ms);
    }
  }
  CK(hipGetLastError());
  printf("{\"variant\": \"%s\", \"us_avg\": %.2f, \"us_best\": %.2f, \"stream_tbs\": %.2f}\n", name, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12);
  if (out) {
    out->resize(pe);
    CK(hipMemcpy(out->data(), bufs[2], pe * sizeof(d2), hipMemcpyDeviceToHost));
  }
  return 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20;
  const size_t N = (size_t)1 << n;
  d2* bufs[3];
  double* diag;
  for (int i = 0; i < 3; ++i) CK(hipMalloc(&bufs[i], N * sizeof(d2)));
  CK(hipMalloc(&diag, N * sizeof(double)));
  std::vector<d2> hx(N);
  std::vector<double> hd(N);
  unsigned long long s = 5;
  for (auto& v : hx) v = d2{1e-3 * rnd(s), 1e-3 * rnd(s)};
  for (auto& v : hd) v = rnd(s);
  CK(hipMemcpy(diag, hd.data(), N * sizeof(double), hipMemcpyHostToDevice));
  auto reset = [&]() -> int {
    for (int i = 0; i < 3; ++i) CK(hipMemcpy(bufs[i], hx.data(), N * sizeof(d2), hipMemcpyHostToDevice));
    return 0;
  };
  Args A{};
  A.n = n;
  A.diag = diag;
  const double bytes = (double)N * (8.0 + 16.0 * (3.0 + 2.0 / 3.0));      // the four vector streams + the diagonal weights
  const int reps = 60;
  std::vector<d2> ref, got;
#define WAVE(MODE, NAME)                                                                                                          \
  if (reset()) return 1;                                                                                                          \
  if (time_it(NAME, A, bufs, [&](const Args& a) { hipLaunchKernelGGL((wave_kernel<MODE>), dim3((unsigned)(N / 256)), dim3(256), 0, 0, a); }, reps, bytes, MODE == 3 ? &ref : nullptr, N)) return 1;
  WAVE(3, "gather (today: one wavefront per 64 rows, 6 lane permutes + 14 gathered lines)")
  WAVE(0, "nogather (the structure with x_i standing in for every neighbour)")
  WAVE(1, "permonly (the six lane permutes, no gathers)")
  WAVE(2, "gatheronly (the fourteen gathers, no lane permutes)")
#define TILE(LT, NAME)                                                                                                            \
  {                                                                                                                               \
    if (reset()) return 1;                                                                                                        \
    const size_t ldsb = ((size_t)1 << LT) * sizeof(d2);                                                                            \
    CK(hipFuncSetAttribute((const void*)tile_kernel<LT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));                  \
    if (time_it(NAME, A, bufs, [&](const Args& a) { hipLaunchKernelGGL((tile_kernel<LT>), dim3((unsigned)(N >> LT)), dim3(1024), ldsb, 0, a); }, reps, bytes, &got, N)) return 1; \
    double md = 0.0;                                                                                                              \
    for (size_t i = 0; i < N; ++i) md = std::max(md, std::max(std::abs(ref[i].x - got[i].x), std::abs(ref[i].y - got[i].y)));    \
    printf("{\"tile\": %d, \"max_abs_difference_to_gather\": %.3e}\n", LT, md);                                                   \
  }
  TILE(10, "tile10 (2^10 rows per workgroup in LDS: bits 6-9 from LDS, 10 gathers)")
  TILE(11, "tile11 (2^11 rows, two per lane: bits 6-10 from LDS, 9 gathers)")
  TILE(12, "tile12 (2^12 rows, four per lane: bits 6-11 from LDS, 8 gathers)")
  return 0;
}
