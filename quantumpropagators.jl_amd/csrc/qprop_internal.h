// Internal declarations shared by the translation units of libqprop_hip.so.
#pragma once

#include <complex>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <new>
#include <string>
#include <vector>

// the library is built with -fvisibility=hidden: only the C ABI of qprop.h is exported
#pragma GCC visibility push(default)
#include "../../include/qprop.h"
#pragma GCC visibility pop

namespace qp {

int fail(int status, const char* fmt, ...);
void set_error(const char* msg);

using cplx = std::complex<double>;

// host numerics (host_numerics.cpp)
std::vector<double> cheby_coeffs(double Delta, double dt, double limit);
bool hessenberg_eigvals_inplace(int n, cplx* A, cplx* w);
int diagonalize_hessenberg(const cplx* Hess, int ldh, int m, bool accumulate, cplx* out);
int diagonalize_hessenberg_block(const cplx* Hess, int ldh, int j, cplx* out);
struct ScaledProd {   // m * 2^e with m in [0.5, 1), or m = 0
  double m;
  int e;
};
void extend_leja(cplx* leja, int n, cplx* newpoints, int n_new, int n_use, const ScaledProd* prod_folded = nullptr);
ScaledProd leja_fold_candidate(const cplx* leja, int n, cplx z);
cplx eval_func(int func_id, qp_func_cb cb, void* user, cplx z);
int extend_newton_coeffs(cplx* a, int n_a, const cplx* leja, int func_id, qp_func_cb cb, void* user,
                         int n_leja, double radius);
int csc_to_csr(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowval,
               const qp_c128* nzval, int base, int64_t* rowptr, int32_t* col, qp_c128* vals);
void partition_rows(const int64_t* rowptr, int64_t nrows, int nparts, int balance, int64_t* bounds);

// Named ranges for profiler timelines (rocprofv3 --marker-trace), with the NAMES of the reference's TimerOutputs sections
// -- timing_data["prop_step!"]["matrix-vector product"], "arnoldi!", "diagonalize_hessenberg_matrix", "get Leja points",
// "get Newton coeffs", "evaluate polynomial" (src/cheby_propagator.jl:349, src/cheby.jl:175, src/newton.jl:276-328,
// test/test_timings.jl:28-30) -- so that a trace of a step reads like the reference's timer table instead of 98 anonymous
// launches.  Off unless knob `roctx` is 1 or QP_ROCTX=1 is in the environment; the marker library (rocprofiler-sdk-roctx,
// else roctx64) is resolved with dlopen on first use and its absence is not an error.
void range_push(const char* name);
void range_pop();
bool ranges_enabled_by_env();
struct ScopedRange {
  bool on;
  ScopedRange(bool enabled, const char* name) : on(enabled) {
    if (on) range_push(name);
  }
  ~ScopedRange() {
    if (on) range_pop();
  }
  ScopedRange(const ScopedRange&) = delete;
  ScopedRange& operator=(const ScopedRange&) = delete;
};

}  // namespace qp

#define QP_TRY try {
#define QP_CATCH                                                    \
  }                                                                 \
  catch (const std::bad_alloc&) {                                   \
    return qp::fail(QP_E_ALLOC, "out of host memory");              \
  }                                                                 \
  catch (const std::exception& e) {                                 \
    return qp::fail(QP_E_INTERNAL, "internal error: %s", e.what()); \
  }                                                                 \
  catch (...) {                                                     \
    return qp::fail(QP_E_INTERNAL, "unknown internal error");       \
  }
