// Two-term strip walk (kernels_walk2_impl.h): host side (launch geometry, dispatch over the kernel shapes) and the translation unit of the
// shapes with four near distances -- the headline's (4, 4).
#include "kernels_walk2_impl.h"

namespace qp {

QP_WALK2_DEFINE(walk2_launch_nn4, 4)

// Both terms of a pair on the two-term region of `P2` (device.h: walk2_plan_supported; the plan's edge list = every block
// outside the region) + term m of the edge list.  *launched = false: not taken (the caller issues two one-term launches).
// The caller follows up with term m + 1 of the edge list (per-block kernel, RowSet::block_map = P2.edge_map).
int launch_hrb_walk2_cheby(hipStream_t s, const DevMatrix& A, const WalkPlan& P2, const double2* x, const ChebyEpi& e1,
                           const ChebyEpi& e2, const Tuning& tun, bool* launched) {
  *launched = false;
  if (!P2.valid || A.format != QP_FMT_HRB || !walk2_shape_supported(P2.nn, P2.K, P2.z0) || P2.xl || P2.fd || P2.g % kRB) return QP_OK;
  if (!e1.v0 || e1.check_partials || e2.check_partials || e1.mirror || e2.mirror || e1.xloc != x) return QP_OK;
  Walk2Geom G;
  const int dmax = P2.near[P2.nn - 1];
  G.W = kRB - 2 * dmax;
  if (G.W < 16) return QP_OK;
  G.S2 = (int)((P2.g + G.W - 1) / G.W);
  const int64_t Jz = ((P2.R1 - P2.W0) * (int64_t)kRB + P2.g - 1) / P2.g;
  const int64_t waves = tun.walk_waves > 0 ? tun.walk_waves : (int64_t)kWalk2Waves * device_cu_count();
  const int64_t nseg_target = std::max<int64_t>(1, waves / G.S2);
  G.L = (int)((Jz + nseg_target - 1) / nseg_target);
  G.nseg = (int)((Jz + G.L - 1) / G.L);
  const int64_t ntask = std::max<int64_t>((int64_t)G.nseg * G.S2, 1);
  G.n_walk_wg = (int)((ntask + kWalk2Waves - 1) / kWalk2Waves);
  G.ntask = G.n_walk_wg * kWalk2Waves;
  G.xlast = A.ncols - 1;
  G.vend = (P2.R1 + (int64_t)P2.K * P2.S) * (int64_t)kRB;      // the one-term plan's run end
  HrbArrays H{A.bptr, A.cmeta, reinterpret_cast<const char*>(A.cols), A.lptr, A.lcmeta,
              reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos)};
  ChebyOp op1{e1}, op2{e2};
  // value loads with the default cache policy once the values are well beyond the Infinity Cache: the chunks overlap by 2 d_max rows and
  // a chunk's packed value halo is its neighbour's stream -- streamed nontemporally, each of those lines comes from memory twice
  // (N = 2^22: 103.6 -> 99.8 us per term, 2^24: 399 -> 363); while most of the values still fit the cache the nontemporal stream
  // leaves it to the vectors (2^21: 56.8 -> 54.6)
  const double value_bytes = (double)(P2.z0 + P2.nn + P2.K) * kRB * (double)A.nblocks * (A.vals_r ? 8.0 : 16.0);
  const int ntm = tun.walk_nt >= 0 ? tun.walk_nt : (value_bytes <= 300e6 ? 1 : 0);
  bool ok = false;
  switch (P2.nn) {
#define QP_WALK2_CASE(NN_)                                                                                   \
  case NN_:                                                                                                  \
    ok = A.vals_r ? walk2_launch_nn##NN_##_f64(s, A.vals_r, x, P2, G, H, A.nrows, op1, op2, ntm)                \
                  : walk2_launch_nn##NN_##_c128(s, A.vals, x, P2, G, H, A.nrows, op1, op2, ntm);                \
    break;
    QP_WALK2_CASE(1)
    QP_WALK2_CASE(2)
    QP_WALK2_CASE(3)
    QP_WALK2_CASE(4)
#undef QP_WALK2_CASE
    default: break;
  }
  if (!ok) return QP_OK;
  QP_HIP(hipGetLastError());
  *launched = true;
  return QP_OK;
}

}  // namespace qp
