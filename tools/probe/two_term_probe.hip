// Probe (round 6): what is the CEILING of a strip walk that forms TWO Chebyshev terms per pass over the matrix
// values (temporal blocking of the three-term recurrence, src/cheby.jl:186-209)?  Before building that kernel, its access pattern in
// time is issued here as a bare loop -- loads, stores, LDS retention traffic and the FMAs, placeholder sums -- next to the same bare
// loop of today's one-term walk (tools/probe/placement_probe.hip measured the real walk within 3-8 % of ITS bare loop).
//
// Pattern of the two-term walk (headline lattice: near +-1..4, far +-g..4g, g = 1024 rows, K = 4):
//   a wavefront walks down a column chunk; at step j it forms  y = v_{t+1}  for block j (phase Y: exactly today's step: 8 value lines,
//   one new ring element of x = v_t, p = v_{t-1}, two packed halo loads) and  z = v_{t+2}  for block j - K (phase Z: needs y of blocks
//   j - 2K .. j, all its own; the values of block j - K a second time: RETAINED in LDS for K steps (8 writes + 8 reads of 1 KiB per
//   step) or loaded again (`reload`: an L2 / Infinity-Cache hit at best)).  Lateral neighbours: y of the rows just outside the chunk
//   belongs to another wavefront, so a chunk computes y on 64 rows and z only on the W = 56 in the middle (near reach 4 on each side):
//   ceil(g / W) chunks per strip step instead of g / 64.  Along the walk a segment of L steps needs 2K steps of run-in (y only).
//   Stores: y and z (two fresh buffers: in-place would race with the neighbours' halo reads); the accumulator every third term.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe/two_term_probe.hip -o tools/probe/two_term_probe
//   two_term_probe [log2n = 22] [waves = 2048]      -> one JSON line
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s -> %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int K = 4;

struct Args {
  const d2* vals;   // [block][8][64]
  const d2* x;
  const d2* p;
  const d2* acc;
  d2* y;
  d2* z;
  d2* acc_out;
  long long n;
  int g, L, W;      // rows per strip step, z-steps per wavefront, useful rows per chunk (64: no lateral redundancy, 56: the real thing)
  int mode;         // 0: one-term walk (today), 1: two-term walk, values retained in LDS, 2: two-term walk, values loaded again,
                    // 3: two-term walk, values retained in REGISTERS (walk unrolled by K + 1 so that the rotation is static; 1 wavefront per SIMD)
  int with_acc;
  int waves_per_wg;
};

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u;
  const unsigned xcd = bid & 7u, j = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + j;
}

template <int MODE>
__global__ __launch_bounds__(MODE == 3 ? 256 : 512) void walk(Args A) {
  extern __shared__ d2 lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const long long task = (long long)wg * A.waves_per_wg + wave;
  const int S = (A.g + A.W - 1) / A.W;
  const long long seg = task / S, col = task - seg * S;
  const long long J = A.n / A.g;
  const long long j0 = seg * A.L, j1 = std::min<long long>(j0 + A.L, J - 2 * K);
  if (j0 >= j1) return;
  constexpr int RET = (MODE == 1) ? (K + 1) * 8 * 64 : 0;      // retained values: K + 1 blocks of 8 lines
  d2* win = lds + (size_t)wave * (RET + 6 * 96);
  d2* ret = win + 6 * 96;
  const int pad = (64 - A.W) / 2;                            // lanes [pad, pad + W) own z rows; all 64 lanes form y
  const long long coff = col * A.W - pad;                    // first row of the chunk's lanes inside the strip step (linear row index: may reach into the step before)
  const long long nlast = A.n - 1;
  auto vline = [&](long long row, int u) -> const d2* {      // value of slot u of `row`: blocks of 64 rows, 8 slots of 64
    row = std::min(std::max(row, 0LL), nlast);
    return A.vals + ((row >> 6) * 8 + u) * 64 + (row & 63);
  };
  d2 va[8], vz[8], xn, pp, ac = {0.0, 0.0}, h0, h1, ring = {0.0, 0.0}, yring = {0.0, 0.0};
  d2 hist[MODE == 3 ? K + 1 : 1][8];
  if (MODE == 3) {
#pragma unroll
    for (int a = 0; a < K + 1; ++a)
#pragma unroll
      for (int u = 0; u < 8; ++u) hist[a][u] = d2{0.0, 0.0};
  }
  const long long jbeg = (MODE == 0) ? j0 : j0 - K;      // first Y step (the two-term walk runs in K steps before its first z)
  // run-in of a segment: ring (2K loads) and history values (K (K + 1) / 2 lines), as the real walk
  {
    const long long r = jbeg * A.g + coff + lane;
    for (int m = 0; m < 2 * K; ++m) {
      const d2 t = A.x[std::min(std::max(r + (m - K) * (long long)A.g, 0LL), nlast)];
      ring.x += 1e-30 * t.x;
    }
    for (int m = 0; m < K * (K + 1) / 2; ++m) {
      const d2 t = *vline(r - (1 + m % K) * (long long)A.g, 4 + m % K);
      ring.y += 1e-30 * t.y;
    }
  }
  auto load = [&](long long j) {
    const long long r = j * A.g + coff + lane;
#pragma unroll
    for (int u = 0; u < 8; ++u) va[u] = __builtin_nontemporal_load(vline(r, u));
    xn = A.x[std::min(r + (long long)K * A.g, nlast)];
    pp = A.p[std::min(r, nlast)];
    const long long r0 = r - lane;
    h0 = A.x[std::min(std::max(r0 + (((lane >> 4) & 1) ? 64 + (lane & 3) : -4 + (lane & 3)), 0LL), nlast)];
    h1 = *vline(r0 - 4 + (lane & 3), lane >> 4 & 3);
    if (MODE == 2 && j - K >= j0) {      // phase Z's values, loaded again (they were streamed K steps ago)
      const long long rz = (j - K) * A.g + coff + lane;
#pragma unroll
      for (int u = 0; u < 8; ++u) vz[u] = *vline(rz, u);
    }
    if (A.with_acc && (MODE == 0 || j - K >= j0)) ac = A.acc[std::min((MODE == 0 ? r : r - (long long)K * A.g), nlast)];
  };
  const long long ybeg = (MODE == 0) ? j0 : j0 - K, yend = (MODE == 0) ? j1 : j1 + K;      // Y steps: K of run-in, K of run-out
  load(ybeg);
  auto body = [&](long long j, auto phase) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase)::value;
    // ---- phase Y (today's step): near windows through LDS, 128 FMAs
#pragma unroll
    for (int k = 0; k < 5; ++k) win[k * 96 + 16 + lane] = k < 4 ? va[k] : ring;
    win[5 * 96 + (lane & 31)] = h0 + h1;
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) ret[((j - ybeg) % (K + 1)) * 512 + u * 64 + lane] = va[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    d2 s = ring;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const d2 t = win[(k % 5) * 96 + 16 + lane + (k < 6 ? -1 - k % 4 : 1 + k % 4)];
      s.x = __builtin_fma(va[k % 8].x, t.x, s.x);
      s.y = __builtin_fma(va[k % 8].y, t.y, s.y);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s.x = __builtin_fma(va[u].y, s.y, s.x);
        s.y = __builtin_fma(va[u].x, s.x, s.y);
      }
    d2 yv = {0.1 * s.x + pp.x, 0.1 * s.y + pp.y};
    d2 zv = yv;
    const bool zstep = MODE != 0 && j - K >= j0 && j - K < j1;
    if (MODE != 0) {
      // ---- phase Z for block j - K: y windows through LDS again, the retained (or reloaded) values, 128 FMAs
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) vz[u] = ret[((j - ybeg + 1) % (K + 1)) * 512 + u * 64 + lane];
      }
      if (MODE == 3) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          vz[u] = hist[(PH + 1) % (K + 1)][u];
          hist[PH % (K + 1)][u] = va[u];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 5; ++k) win[k * 96 + 16 + lane] = k < 4 ? vz[k] : yring;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      d2 s2 = yring;
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const d2 t = win[(k % 5) * 96 + 16 + lane + (k < 6 ? -1 - k % 4 : 1 + k % 4)];
        s2.x = __builtin_fma(vz[k % 8].x, t.x, s2.x);
        s2.y = __builtin_fma(vz[k % 8].y, t.y, s2.y);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s2.x = __builtin_fma(vz[u].y, s2.y, s2.x);
          s2.y = __builtin_fma(vz[u].x, s2.x, s2.y);
        }
      zv = d2{0.1 * s2.x + ring.x + ac.x, 0.1 * s2.y + ring.y + ac.y};
      yring = yv;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    ring = xn;
    const long long r = j * A.g + coff + lane;
    if (j + 1 < yend) load(j + 1);
    const bool mine = lane >= pad && lane < pad + A.W && coff + lane < A.g;
    if (MODE == 0) {
      A.y[std::min(r, nlast)] = d2{yv.x + ac.x, yv.y + ac.y};
    } else {
      if (mine && j >= j0 && j < j1) A.y[std::min(r, nlast)] = yv;
      if (mine && zstep) {
        A.z[std::min(r - (long long)K * A.g, nlast)] = zv;
        if (A.with_acc) A.acc_out[std::min(r - (long long)K * A.g, nlast)] = d2{zv.x + yv.x, zv.y};
      }
    }
  };
  if (MODE == 3) {      // (a few steps beyond the segment's end: addresses clamp, a probe)
    for (long long j = ybeg; j < yend; j += K + 1) {
      body(j, std::integral_constant<int, 0>());
      body(j + 1, std::integral_constant<int, 1>());
      body(j + 2, std::integral_constant<int, 2>());
      body(j + 3, std::integral_constant<int, 3>());
      body(j + 4, std::integral_constant<int, 4>());
    }
  } else {
    for (long long j = ybeg; j < yend; ++j) body(j, std::integral_constant<int, 0>());
  }
}

__global__ void fill_random(d2* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long h = (i + seed) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    p[i] = d2{1e-3 * ((double)(h & 0xffffff) / 16777216.0 - 0.5), 1e-3 * ((double)((h >> 24) & 0xffffff) / 16777216.0 - 0.5)};
  }
}

int main(int argc, char** argv) {
  const int log2n = argc > 1 ? atoi(argv[1]) : 22;
  const long long waves = argc > 2 ? atoll(argv[2]) : 2048;
  const long long n = 1LL << log2n;
  const size_t slack = 16u << 20, margin = 2u << 20;
  char *vals, *vec[7];
  CK(hipMalloc(&vals, (size_t)n * 8 * 16 + slack));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<d2*>(vals), ((size_t)n * 8 * 16 + slack) / 16, 1u);
  for (int k = 0; k < 7; ++k) {
    CK(hipMalloc(&vec[k], (size_t)n * 16 + slack));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<d2*>(vec[k]), ((size_t)n * 16 + slack) / 16, 7u + k);
  }
  CK(hipDeviceSynchronize());
  Args A{};
  A.n = n;
  A.g = 1024;
  A.vals = reinterpret_cast<const d2*>(vals + margin);
  A.x = reinterpret_cast<const d2*>(vec[0] + margin);
  A.p = reinterpret_cast<const d2*>(vec[1] + margin);
  A.acc = reinterpret_cast<const d2*>(vec[2] + margin);
  A.y = reinterpret_cast<d2*>(vec[3] + margin);
  A.z = reinterpret_cast<d2*>(vec[4] + margin);
  A.acc_out = reinterpret_cast<d2*>(vec[5] + margin);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long long J = n / A.g;
  // configurations: {mode, W, waves per workgroup}
  struct Cfg { int mode, W, wpw; const char* name; };
  // (values retained in LDS: 5 blocks x 8 KiB + windows = 50 KB per wavefront: at most THREE wavefronts per compute unit)
  const Cfg cfgs[] = {{0, 64, 8, "one_term_x2"},           {1, 64, 3, "two_term_lds_W64_3w"}, {1, 56, 3, "two_term_lds_W56_3w"},
                      {3, 64, 4, "two_term_regs_W64_4w"},  {3, 56, 4, "two_term_regs_W56_4w"}, {2, 56, 8, "two_term_reload_W56_8w"}};
  const int nc = (int)(sizeof(cfgs) / sizeof(cfgs[0]));
  std::vector<std::vector<double>> us((size_t)nc);
  int Ls[8] = {0};
  for (int round = 0; round < 4; ++round)
    for (int c = 0; c < nc; ++c) {
      const Cfg& cf = cfgs[c];
      A.mode = cf.mode;
      A.W = cf.W;
      A.waves_per_wg = cf.wpw;
      const long long S = (A.g + A.W - 1) / A.W;
      const long long segs = std::max<long long>(1, waves / S);
      A.L = (int)((J + segs - 1) / segs);
      Ls[c] = A.L;
      const long long ntask = ((J + A.L - 1) / A.L) * S;
      const int nwg = (int)((ntask + cf.wpw - 1) / cf.wpw);
      const size_t ldsb = sizeof(d2) * (size_t)cf.wpw * ((cf.mode == 1 ? (K + 1) * 512 : 0) + 6 * 96);
      auto kern = cf.mode == 0 ? &walk<0> : (cf.mode == 1 ? &walk<1> : (cf.mode == 2 ? &walk<2> : &walk<3>));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      CK(hipEventRecord(e0));
      // six terms' worth: six one-term launches (accumulator on two of them) or three two-term launches (accumulator on one)
      const int launches = cf.mode == 0 ? 6 : 3;
      for (int t = 0; t < launches; ++t) {
        A.with_acc = (t % 3 == 0);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * cf.wpw), ldsb, 0, A);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (round > 0) us[(size_t)c].push_back(1e3 * ms / 6);      // per TERM
    }
  printf("{\"log2n\": %d, \"waves\": %lld, \"unit\": \"us per Chebyshev term (median of 3 x 6 terms)\"", log2n, waves);
  for (int c = 0; c < nc; ++c) {
    std::sort(us[(size_t)c].begin(), us[(size_t)c].end());
    printf(", \"%s\": {\"us\": %.2f, \"min\": %.2f, \"max\": %.2f, \"L\": %d}", cfgs[c].name, us[(size_t)c][1], us[(size_t)c][0], us[(size_t)c][2], Ls[c]);
  }
  printf("}\n");
  return 0;
}
