// Probe (round 5): does the 570-630 us spread of the strip walk at N = 2^24 between PROCESSES (profiles/r05/walk_hbm_resident.txt, 4.)
// come from the relative placement of the five arrays a term streams in lockstep?  The walk's access pattern in time -- every
// wavefront walks down a strip column: per 64-row block 8 value lines, one new ring element 4 strip steps ahead, v0, every third
// term the accumulator, the store -- with placeholder sums, the arrays offset against their allocations by a set of skews; the
// same process times every skew set (interleaved rounds), and the binary is run several times.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/placement_probe.hip -o tools/probe/placement_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s -> %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

struct Args {
  const d2* vals;
  const d2* x;
  const d2* v0;
  const d2* acc;
  d2* out;
  long long n;
  int g;       // rows per strip step
  int L;       // steps per wavefront
  int with_acc;
  int feat;    // bit 0: XCD-contiguous workgroup remap, 1: two halo loads per step, 2: dummy accumulator load when absent, 4: LDS window traffic,
               // 8: ring (8) + history (10) loads at the start of a segment, 16: 128 FMAs per step, 32: in-place (out = v0)
};

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u;
  const unsigned xcd = bid & 7u, j = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + j;
}

__global__ __launch_bounds__(512) void walk_mix(Args A) {
  __shared__ d2 win[8][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned wg = (A.feat & 1) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const long long task = (long long)wg * 8 + wave;
  const int S = A.g / 64;
  const long long seg = task / S, col = task - seg * S;
  const long long J = A.n / A.g;
  const long long j0 = seg * A.L, j1 = std::min<long long>(j0 + A.L, J - 4);
  if (j0 >= j1) return;
  long long r = j0 * A.g + col * 64 + lane;
  d2 ring = A.x[r];
  d2 va[8], xn, w0, ac = {0.0, 0.0}, h0 = {0.0, 0.0}, h1 = {0.0, 0.0};
  if (A.feat & 8) {   // set-up of a segment: the ring's other elements and the history values of the blocks before
    for (int m = 1; m < 8; ++m) {
      const long long rx = r + (m - 4LL) * A.g;
      const d2 t = A.x[rx < 0 ? r : rx];
      ring.x += 1e-30 * t.x;
    }
    for (int m = 0; m < 10; ++m) {
      const long long rb = r - (1 + m % 4) * (long long)A.g;
      const d2 t = A.vals[((rb < 0 ? r : rb) >> 6) * 512 + (4 + m % 4) * 64 + lane];
      ring.y += 1e-30 * t.y;
    }
  }
  auto load = [&](long long rr) {
    const long long b = rr >> 6;
#pragma unroll
    for (int u = 0; u < 8; ++u) va[u] = __builtin_nontemporal_load(A.vals + (b * 8 + u) * 64 + lane);
    xn = A.x[rr + 4LL * A.g];
    w0 = A.v0[rr];
    if (A.with_acc) ac = A.acc[rr];
    else if (A.feat & 4) ac = A.x[lane];
    if (A.feat & 2) {
      const long long r0 = rr - lane;
      h0 = A.x[r0 + ((lane >> 4) & 1 ? 64 + (lane & 3) : -4 + (lane & 3))];                                   // x halo: 2 x 4 elements
      h1 = A.vals[((r0 >> 6) - (r0 >= 64 ? 1 : 0)) * 512 + (lane >> 4) * 64 + 60 + (lane & 3)];             // value halo: 4 x 4 elements of the block before
    }
  };
  load(r);
  for (long long j = j0; j < j1; ++j) {
    d2 s = ring;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s.x = __builtin_fma(va[u].x, xn.x, s.x);
      s.y = __builtin_fma(va[u].y, xn.y, s.y);
    }
    if (A.feat & 16) {
#pragma unroll
      for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s.x = __builtin_fma(va[u].y, s.y, s.x);
          s.y = __builtin_fma(va[u].x, s.x, s.y);
        }
    }
    if (A.feat & 2) {
      s.x += 1e-30 * (h0.x + h1.x);
    }
    if (A.feat & 8 + 0 && false) {}
    if (A.feat & 64) {   // LDS windows: 6 writes, a wave barrier, 12 reads
#pragma unroll
      for (int k = 0; k < 6; ++k) win[wave][k * 96 + 16 + lane] = va[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const d2 t = win[wave][(k % 6) * 96 + 16 + lane + (k < 6 ? -1 - k % 4 : 1 + k % 4)];
        s.y += 1e-30 * t.x;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const d2 o = {0.1 * s.x + w0.x + ac.x, 0.1 * s.y + w0.y + ac.y};
    ring = xn;
    const long long rr = r;
    r += A.g;
    if (j + 1 < j1) load(r);
    A.out[rr] = o;
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
  const int log2n = argc > 1 ? atoi(argv[1]) : 24;
  const long long n = 1LL << log2n;
  const size_t slack = 16u << 20, margin = 2u << 20;   // bytes; every base sits `margin` into its allocation (the halo / history loads reach a few rows before row 0)
  char *vals, *vec[4];
  CK(hipMalloc(&vals, (size_t)n * 8 * 16 + slack));
  CK(hipMemset(vals, 0, (size_t)n * 8 * 16 + slack));
  for (int k = 0; k < 4; ++k) {
    CK(hipMalloc(&vec[k], (size_t)n * 16 + slack));
    CK(hipMemset(vec[k], 0, (size_t)n * 16 + slack));
  }
  if (argc > 3) {   // random data instead of zeros
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<d2*>(vals), ((size_t)n * 8 * 16 + slack) / 16, 1u);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<d2*>(vec[k]), ((size_t)n * 16 + slack) / 16, 7u + k);
    CK(hipDeviceSynchronize());
  }
  // skew sets (bytes, multiples of 1 KiB so that every wavefront access stays one aligned line run): vals, x, v0, acc, out
  const size_t K = 1024;
  const size_t sets[][5] = {
      {0, 0, 0, 0, 0},
      {0, 4 * K, 8 * K, 12 * K, 16 * K},
      {0, 16 * K, 32 * K, 48 * K, 64 * K},
      {0, 65 * K, 130 * K, 195 * K, 260 * K},
      {0, 256 * K, 512 * K, 768 * K, 1024 * K},
      {0, 1029 * K, 2058 * K, 3087 * K, 4116 * K},
      {0, 2048 * K, 4096 * K, 6144 * K, 2048 * K},
      {0, 1 * K, 2 * K, 3 * K, 5 * K},
  };
  const int nsets = (int)(sizeof(sets) / sizeof(sets[0]));
  Args A{};
  A.n = n;
  A.g = 1024;
  A.feat = 0;
  const long long waves = 2048, S = A.g / 64, J = n / A.g;
  A.L = (int)((J + waves / S - 1) / (waves / S));
  const int nwg = (int)(((J + A.L - 1) / A.L) * S / 8);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<std::vector<double>> us((size_t)nsets);
  for (int round = 0; round < 5; ++round)
    for (int s = 0; s < nsets; ++s) {
      A.vals = reinterpret_cast<const d2*>(vals + margin + sets[s][0]);
      A.x = reinterpret_cast<const d2*>(vec[0] + margin + sets[s][1]);
      A.v0 = reinterpret_cast<const d2*>(vec[1] + margin + sets[s][2]);
      A.acc = reinterpret_cast<const d2*>(vec[2] + margin + sets[s][3]);
      A.out = reinterpret_cast<d2*>(vec[3] + margin + sets[s][4]);
      CK(hipEventRecord(e0));
      for (int t = 0; t < 6; ++t) {
        A.with_acc = (t % 3 == 0);
        hipLaunchKernelGGL(walk_mix, dim3(nwg), dim3(512), 0, 0, A);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (round > 0) us[(size_t)s].push_back(1e3 * ms / 6);
    }
  if (argc > 2) {   // feature sweep (skew set 0): what each piece of the walk's machinery costs on top of the bare loop
    const int feats[] = {0, 1, 2, 4, 8, 16, 32, 64, 1 | 2 | 4 | 8 | 16 | 32 | 64, 2 | 4 | 8 | 16 | 32 | 64};
    const int nf = (int)(sizeof(feats) / sizeof(feats[0]));
    std::vector<std::vector<double>> uf((size_t)nf);
    for (int round = 0; round < 5; ++round)
      for (int f = 0; f < nf; ++f) {
        A.feat = feats[f];
        A.vals = reinterpret_cast<const d2*>(vals + margin);
        A.x = reinterpret_cast<const d2*>(vec[0] + margin);
        A.v0 = reinterpret_cast<const d2*>(vec[1] + margin);
        A.acc = reinterpret_cast<const d2*>(vec[2] + margin);
        A.out = (feats[f] & 32) ? reinterpret_cast<d2*>(vec[1] + margin) : reinterpret_cast<d2*>(vec[3] + margin);
        CK(hipEventRecord(e0));
        for (int t = 0; t < 6; ++t) {
          A.with_acc = (t % 3 == 0);
          hipLaunchKernelGGL(walk_mix, dim3(nwg), dim3(512), 0, 0, A);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (round > 0) uf[(size_t)f].push_back(1e3 * ms / 6);
      }
    printf("{\"n\": %lld, \"feature_sweep_us_per_term\": {", n);
    for (int f = 0; f < nf; ++f) {
      std::sort(uf[(size_t)f].begin(), uf[(size_t)f].end());
      printf("%s\"%d\": %.1f", f ? ", " : "", feats[f], uf[(size_t)f][uf[(size_t)f].size() / 2]);
    }
    printf("}}\n");
    return 0;
  }
  printf("{\"n\": %lld, \"workgroups\": %d, \"steps_per_wavefront\": %d, \"us_per_term_by_skew_set\": [", n, nwg, A.L);
  for (int s = 0; s < nsets; ++s) {
    std::sort(us[(size_t)s].begin(), us[(size_t)s].end());
    printf("%s%.1f", s ? ", " : "", us[(size_t)s][us[(size_t)s].size() / 2]);
  }
  printf("], \"ptr_low_bits\": [");
  printf("%llu", (unsigned long long)((size_t)vals & 0xffffff));
  for (int k = 0; k < 4; ++k) printf(", %llu", (unsigned long long)((size_t)vec[k] & 0xffffff));
  printf("]}\n");
  return 0;
}
