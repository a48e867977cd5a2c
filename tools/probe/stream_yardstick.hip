// Yardstick for the HBM-resident sizes of the fused Chebyshev term: what this box streams from HBM
//   (a) read-only (one 2 GiB array, every byte read once), and
//   (b) in the strip walk's own mix: 8 read streams to 1 write stream (per row the walk reads 128 B of values, v1, v0 and
//       every third term Psi -- 165 B -- and writes v2 and every third term Psi -- 21 B: 7.75 : 1),
// each as the BEST of a few plain streaming shapes (elements per thread 1 / 2 / 4, nontemporal or not), median of 5 timed
// regions of 10 launches.  Working sets are 2 GiB and 2.25 GiB: far beyond the 256 MiB Infinity Cache.
// bench.py runs this binary as a child process and puts the two figures into `roofline` (stream_read_gbs,
// stream_walk_mix_gbs); it replaces the y += a x triad (one third writes) the earlier lines used as a "ceiling".
//   hipcc -O3 --offload-arch=gfx950 tools/probe/stream_yardstick.hip -o tools/probe/stream_yardstick
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s -> %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// read only: n elements of one array
template <bool NT, int PER>
__global__ __launch_bounds__(256) void read_only(const d2* __restrict__ x, d2* __restrict__ sink, size_t n) {
  const size_t base = ((size_t)blockIdx.x * 256) * PER + threadIdx.x;
  d2 s = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i < n) s += ld<NT>(x + i);
  }
  if (s.x == 1.2345e300) sink[0] = s;
}

// the walk's mix: element i of 8 read streams (x + k * stride), one write stream
template <bool NT, int PER>
__global__ __launch_bounds__(256) void mix8to1(const d2* __restrict__ x, d2* __restrict__ y, size_t n, size_t stride) {
  const size_t base = ((size_t)blockIdx.x * 256) * PER + threadIdx.x;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i >= n) break;
    d2 s = ld<NT>(x + i);
#pragma unroll
    for (int j = 1; j < 8; ++j) s += ld<NT>(x + i + (size_t)j * stride);
    if (NT) __builtin_nontemporal_store(s, y + i); else y[i] = s;
  }
}

// the arrays hold pseudo-random numbers, not zeros: a zero-filled array streams ~8 % FASTER on this chip (profiles/r05/
// placement_probe_zero_vs_random.txt), and the walk it is compared with streams real data
__global__ void fill_random(d2* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long h = (i + seed) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    p[i] = d2{1e-3 * ((double)(h & 0xffffff) / 16777216.0 - 0.5), 1e-3 * ((double)((h >> 24) & 0xffffff) / 16777216.0 - 0.5)};
  }
}

template <class F> static int timed(F launch, double bytes, double* best_gbs) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) launch();
  std::vector<double> us;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    for (int k = 0; k < 10; ++k) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    us.push_back(1e3 * ms / 10);
  }
  std::sort(us.begin(), us.end());
  const double gbs = bytes / us[2] / 1e3;
  if (gbs > *best_gbs) *best_gbs = gbs;
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return 0;
}

#define GRID(n, per) dim3((unsigned)(((n) + 256 * (size_t)(per) - 1) / (256 * (size_t)(per))))

int main(int argc, char** argv) {
  const int log2n = argc > 1 ? atoi(argv[1]) : 24;      // elements per stream: 2^24 x 16 B = 256 MiB
  const size_t n = (size_t)1 << log2n;
  d2 *x, *y;
  CK(hipMalloc(&x, 8 * n * 16));
  CK(hipMalloc(&y, n * 16));
  const bool zeros = argc > 2;   // (second argument: zero-filled arrays, the r05 first-session yardstick, for comparison)
  if (zeros) {
    CK(hipMemset(x, 0, 8 * n * 16));
    CK(hipMemset(y, 0, n * 16));
  } else {
    hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, x, 8 * n, 1u);
    hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, y, n, 2u);
    CK(hipDeviceSynchronize());
  }
  double rd = 0.0, mix = 0.0;
  const size_t nr = 8 * n;
  const double rbytes = 16.0 * nr, mbytes = 16.0 * 9 * n;
#define RD(NT, PER) if (timed([&] { hipLaunchKernelGGL((read_only<NT, PER>), GRID(nr, PER), dim3(256), 0, 0, x, y, nr); }, rbytes, &rd)) return 1
#define MX(NT, PER) if (timed([&] { hipLaunchKernelGGL((mix8to1<NT, PER>), GRID(n, PER), dim3(256), 0, 0, x, y, n, n); }, mbytes, &mix)) return 1
  RD(false, 1); RD(false, 2); RD(false, 4); RD(true, 1); RD(true, 2); RD(true, 4);
  MX(false, 1); MX(false, 2); MX(true, 1); MX(true, 2);
  printf("{\"stream_read_gbs\": %.1f, \"stream_walk_mix_gbs\": %.1f, \"read_bytes\": %.0f, \"mix_bytes\": %.0f, \"data\": \"%s\"}\n", rd, mix, rbytes, mbytes,
         zeros ? "zeros" : "random");
  CK(hipFree(x));
  CK(hipFree(y));
  return 0;
}
