// Internal header shared by engine_core.hip (handles, layouts, the operator API) and engine_plans.hip (host index work that
// decides HOW an operator is run: format choice, lattice completion, strip-walk plan, column-blocked mirror).
#pragma once

#include <thread>
#include <system_error>

#include "engine.h"

// host threads the library may keep busy at once: at most 8, and never more than the container's CPU quota leaves (cgroup
// cpu.max: a control group that exceeds its quota is frozen for the rest of the scheduler period -- up to 100 ms in which the
// caller's enqueueing thread does not run either)
inline unsigned host_threads() {
  static const unsigned n = [] {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      long period = 0;
      if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
        const long quota = std::atol(q) / period;
        if (quota >= 1) hw = std::min<unsigned>(hw, (unsigned)quota);
      }
      std::fclose(f);
    }
    // up to 16 (the passes are memory-bound: more threads than that buy little; eight ranks of a node each take their share);
    // QP_HOST_THREADS overrides
    unsigned cap = 16;
    if (const char* e = std::getenv("QP_HOST_THREADS")) {
      const long v = std::atol(e);
      if (v >= 1) cap = (unsigned)std::min<long>(v, 256);
    }
    return std::max(1u, std::min(cap, hw > 2 ? hw - 1 : hw));
  }();
  return n;
}

// Phase timer of the host-side operator build (QP_BUILD_TRACE=1: one line per phase on stderr; otherwise two clock reads per phase)
struct BuildTrace {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  const bool on = std::getenv("QP_BUILD_TRACE") != nullptr;
  void mark(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[qp build] %-44s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};


// rows [0, n) in contiguous chunks on a few host threads (index work whose iterations write disjoint positions)
template <class F>
inline void parallel_rows(int64_t n, F&& fn, int64_t serial_below = (int64_t)1 << 16) {
  const unsigned hw = host_threads();
  if (n < serial_below || hw == 1) {
    fn((int64_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  const int64_t chunk = (n + hw - 1) / hw;
  int64_t done = 0;   // rows [0, done) have been handed to a thread
  try {
    for (unsigned t = 0; t < hw; ++t) {
      const int64_t a = (int64_t)t * chunk, b = std::min(n, a + chunk);
      if (a >= b) break;
      th.emplace_back([&fn, a, b] { fn(a, b); });
      done = b;
    }
  } catch (const std::system_error&) {
    // no more threads to be had (resource limits): the started ones are joined below -- a joinable std::thread destroyed
    // means std::terminate -- and this thread takes the rest
  }
  for (auto& x : th) x.join();
  if (done < n) fn(done, n);
}


// dst[0, n) = src[0, n) on the host threads (gigabyte arrays: one thread's memcpy is a third of the machine's rate)
template <class T>
inline void parallel_copy(T* dst, const T* src, size_t n) {
  if (n == 0) return;      // (memcpy's pointers must not be null, even for no bytes)
  parallel_rows((int64_t)n, [&](int64_t a, int64_t b) { std::memcpy(static_cast<void*>(dst + a), static_cast<const void*>(src + a), (size_t)(b - a) * sizeof(T)); },
                (int64_t)1 << 20);
}

// ---- host-side layout of the two row-block formats --------------------------------
// Within a 64-row block, entry k of row r sits at  base + 64 k + (r % 64); column
// indices (and the lower section's positions) are packed four k per lane.
inline int64_t rb_val_pos(const std::vector<int64_t>& bptr, int64_t r, int64_t k) {
  return bptr[r / kRB] + k * kRB + (r % kRB);
}
inline int64_t rb_quad_pos(const std::vector<int64_t>& bptr, int64_t r, int64_t k) {
  return bptr[r / kRB] + (k >> 2) * (4 * kRB) + (r % kRB) * 4 + (k & 3);
}

using HostLayout = HostLayoutData;

// engine_plans.hip
struct WalkShape;
int choose_format(qp_operator* op, int requested, bool hermitian);
void lattice_fill(const qp::Tuning& tun, int64_t n, int64_t ncols, qp::HostVec<int64_t>& ur, qp::HostVec<int32_t>& uc,
                  qp::HostVec<int64_t>* ur_before = nullptr, qp::HostVec<int32_t>* uc_before = nullptr);
int build_walk_plan(qp_operator* op);
int build_colblock(qp_operator* op);
// position of every union-CSR entry in the operator's value array (-(position) - 1: its complex conjugate)
void csr_value_map(const qp_operator* op, std::vector<int64_t>& map);
