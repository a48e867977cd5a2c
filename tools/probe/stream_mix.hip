// Probe: what rate does the chip reach on the stream MIX of a fused Chebyshev term of a panel
// (config C5: read X, read v0, write v2 in place, read + write the accumulator every third term),
// as opposed to a plain copy?  Sets the floor the batched kernel can be priced against.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/stream_mix.hip -o tools/probe/stream_mix && tools/probe/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0: copy (1R 1W); 1: y = y + x (2R 1W, in place); 2: y = y + x, z = z + y (3R 2W); 3: read only (sum)
// PER = elements per thread handled one after another (one-shot grid covers n / PER threads)
template <int MODE, bool NT, int PER>
__global__ __launch_bounds__(256) void mix(const d2* __restrict__ x, d2* __restrict__ y, d2* __restrict__ z, size_t n) {
  const size_t base = ((size_t)blockIdx.x * 256) * PER + threadIdx.x;
  d2 s = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i >= n) break;
    if (MODE == 0) {
      st<NT>(y + i, ld<NT>(x + i));
    } else if (MODE == 1) {
      const d2 a = x[i];
      const d2 b = ld<NT>(y + i);
      st<NT>(y + i, a + b);
    } else if (MODE == 2) {
      const d2 a = x[i];
      const d2 b = ld<NT>(y + i);
      const d2 c = ld<NT>(z + i);
      const d2 r = a + b;
      st<NT>(y + i, r);
      st<NT>(z + i, c + r);
    } else {
      s += ld<NT>(x + i);
    }
  }
  if (MODE == 3 && s.x == 1.2345e300) y[0] = s;
}

template <int MODE, bool NT, int PER>
static int run(const char* name, d2* x, d2* y, d2* z, size_t n, double bytes_per_elem) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((n + 256 * PER - 1) / (256 * (size_t)PER));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((mix<MODE, NT, PER>), dim3(grid), dim3(256), 0, 0, x, y, z, n);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mix<MODE, NT, PER>), dim3(grid), dim3(256), 0, 0, x, y, z, n);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps;
  printf("%-34s nt=%d per=%d  %8.1f us  %7.0f GB/s\n", name, (int)NT, PER, us, bytes_per_elem * n / us / 1e3);
  return 0;
}

int main(int argc, char** argv) {
  const int log2n = argc > 1 ? atoi(argv[1]) : 24;   // default 2^24 elements = 2^18 rows x 64 states: 268 MB per vector
  const size_t n = (size_t)1 << log2n;
  printf("%zu elements = %.0f MB per vector\n", n, n * 16.0 / 1e6);
  d2 *x, *y, *z;
  CK(hipMalloc(&x, n * 16));
  CK(hipMalloc(&y, n * 16));
  CK(hipMalloc(&z, n * 16));
  CK(hipMemset(x, 0, n * 16));
  CK(hipMemset(y, 0, n * 16));
  CK(hipMemset(z, 0, n * 16));
  run<3, false, 1>("read only", x, y, z, n, 16);
  run<3, false, 4>("read only", x, y, z, n, 16);
  run<3, true, 4>("read only", x, y, z, n, 16);
  run<0, false, 1>("copy 1R 1W", x, y, z, n, 32);
  run<0, false, 4>("copy 1R 1W", x, y, z, n, 32);
  run<0, true, 1>("copy 1R 1W", x, y, z, n, 32);
  run<0, true, 4>("copy 1R 1W", x, y, z, n, 32);
  run<1, false, 1>("y += x  2R 1W (in place)", x, y, z, n, 48);
  run<1, false, 4>("y += x  2R 1W (in place)", x, y, z, n, 48);
  run<1, true, 1>("y += x  2R 1W (in place)", x, y, z, n, 48);
  run<1, true, 4>("y += x  2R 1W (in place)", x, y, z, n, 48);
  run<2, false, 1>("y += x, z += y  3R 2W", x, y, z, n, 80);
  run<2, true, 1>("y += x, z += y  3R 2W", x, y, z, n, 80);
  run<2, true, 4>("y += x, z += y  3R 2W", x, y, z, n, 80);
  return 0;
}
