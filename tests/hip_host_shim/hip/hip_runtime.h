// TEST INFRASTRUCTURE ONLY (tests/test_cabi_host.py::test_host_index_work_under_sanitizers): a host stand-in for the handful of
// HIP runtime calls that the library's HOST index work makes (csrc/engine_core.hip, engine_plans.hip: format choice, lattice
// completion, strip-walk plan, column encodings, column-blocked mirror, value dictionary, the get_csr round trip), so that
// exactly those sources compile with g++ -fsanitize=address,undefined on a box without a GPU.  "Device" memory is heap memory:
// AddressSanitizer then checks every copy into and out of a device array against the size it was allocated with -- the
// bounds a GPU never checks.  No kernel runs here and nothing of the product is built against this header.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct double2 { double x, y; };
struct int4 { int x, y, z, w; };
struct dim3 { unsigned x = 1, y = 1, z = 1; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef int hipError_t;
typedef struct shim_stream* hipStream_t;
typedef struct shim_event* hipEvent_t;
typedef struct shim_graph* hipGraph_t;
typedef struct shim_graph_exec* hipGraphExec_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostRegisterPortable = 1, hipEventDisableTiming = 2 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };

static inline const char* hipGetErrorString(hipError_t) { return "shim"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
