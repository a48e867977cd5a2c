// Probe: issue rate of v_mfma_f64_16x16x4_f64 for one to four wavefronts per SIMD, NACC independent accumulators
// per wavefront (the complex 32 x 32 tile of engine_liouville.hip cycles through 8), optionally with VALU work between.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/probe/mfma_f64_rate.hip -o tools/probe/mfma_f64_rate && tools/probe/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC, int VALU>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  double t = a0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VALU; ++v) t = t * 1.0000001 + b;
      }
  }
  double s = t;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345e300) out[0] = s;
}

template <int NACC, int VALU>
static int run(int wgs_per_cu, double* out) {
  const int iters = 1024;   // x 16 MFMAs
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL((rate<NACC, VALU>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((rate<NACC, VALU>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfmas_per_simd = (double)iters * 16 * wgs_per_cu;   // one wave of each WG per SIMD
  const double flops = (double)grid * 4 * iters * 16 * 2048.0;
  printf("acc %d  valu/mfma %d  waves/SIMD %d : %8.1f us  %6.1f TFLOP/s  %6.1f ns per MFMA and SIMD\n", NACC, VALU, wgs_per_cu,
         ms * 1e3, flops / ms / 1e9, ms * 1e6 / mfmas_per_simd);
  return 0;
}

int main() {
  double* out;
  CK(hipMalloc(&out, 8));
  for (int w = 1; w <= 4; ++w) run<8, 0>(w, out);
  run<4, 0>(1, out);
  run<2, 0>(1, out);
  run<1, 0>(1, out);
  run<16, 0>(1, out);
  for (int w = 1; w <= 2; ++w) run<8, 2>(w, out);
  for (int w = 1; w <= 2; ++w) run<8, 6>(w, out);
  return 0;
}
