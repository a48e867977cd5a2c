// Probe: can a kernel on stream A release a stream-level wait (hipStreamWaitValue32) on stream B,
// and how long does the hand-off take?  (Used to decide the multi-GPU schedule)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void producer(unsigned* flag, unsigned value, int spin) {
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void consumer(unsigned* out, unsigned v) { if (threadIdx.x == 0) *out = v; }

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("CanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  unsigned *flag = nullptr, *out = nullptr, *flag2 = nullptr;
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipExtMallocWithFlags((void**)&flag2, 8, hipMallocSignalMemory));
  CK(hipMalloc((void**)&out, 4));
  CK(hipMemset(out, 0, 4));
  *flag = 0;
  *flag2 = 0;
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  // B waits for flag >= 1, then runs the consumer and writes flag2 = 7
  CK(hipStreamWaitValue32(b, flag, 1, hipStreamWaitValueGte, 0xffffffffu));
  hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, out, 42u);
  CK(hipStreamWriteValue32(b, flag2, 7, 0));
  unsigned h = 1;
  CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
  printf("before producer: out = %u (expect 0), flag2 = %u\n", h, *flag2);
  hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, flag, 1u, 1000);
  CK(hipStreamSynchronize(b));
  CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
  printf("after producer : out = %u (expect 42), flag = %u, flag2 = %u (expect 7)\n", h, *flag, *flag2);
  // latency of 200 hand-offs A(kernel) -> B(wait, kernel, write) -> host poll
  auto t0 = std::chrono::steady_clock::now();
  const int n = 200;
  for (int i = 0; i < n; ++i) {
    CK(hipStreamWaitValue32(b, flag, 2 + i, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, out, (unsigned)i);
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, flag, (unsigned)(2 + i), 0);
  }
  CK(hipStreamSynchronize(b));
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
  printf("%d chained hand-offs: %.2f us each\n", n, us);
  return 0;
}
