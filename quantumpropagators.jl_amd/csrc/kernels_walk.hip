// Strip walk of the fused Chebyshev term: host side (launch geometry, dispatch over the kernel shapes) and the translation
// unit of the complex-valued shapes with 3-4 near distances -- the headline's (4, 4).  The kernel itself: kernels_walk_impl.h.
#include "kernels_walk_impl.h"

namespace qp {

bool walk_launch_c128_hi(hipStream_t s, dim3 grid, const double2* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy) {
  return launch_shape<double2, 0>(s, grid, uvals, x, P, G, H, nrows, op, ntm, sy);
}

int launch_hrb_walk_cheby(hipStream_t s, const DevMatrix& A, const double2* x, const ChebyEpi& e, const Tuning& tun,
                          bool* launched, const RowSet* rs) {
  *launched = false;
  const WalkPlan* P = (rs && rs->walk) ? rs->walk : A.walk;
  const SyncArgs sy = rs ? rs->sync : SyncArgs();
  const int reserve = rs ? rs->reserve_cu : 0;
  if (!P || !P->valid || A.format != QP_FMT_HRB) return QP_OK;
  const int64_t nW = P->R1 - P->W0;
  if (nW < tun.walk_min_blocks || nW < P->S) return QP_OK;
  WalkGeom G;
  const int64_t J = (nW * kRB + P->g - 1) / P->g;                 // steps of the longest strip column
  // wavefronts.  While the operator and the vectors sit in the Infinity Cache, 768 wavefronts as 192 workgroups of four
  // (one per CU on three quarters of the chip) draw what it delivers: the set-up of a walk (8 + 10 loads) is paid less
  // often than with the 2048 that fill every SIMD twice, and the edge blocks run beside the walk on the free compute
  // units (profiles/r03/kbench_walk_development.txt: N = 2^20 31.9 us per term; 1280 as workgroups of eight 33.4, 2048 36.5)
  // ... and beyond it the matrix values are streamed nontemporally: they are read once per term, and what the
  // Infinity Cache then keeps from one term to the next is the vectors
  // (the slots the walk streams: the pad slots of the quad-padded upper sections are never read)
  const double footprint = (double)(P->z0 + P->nn + P->K * (1 + 2 * P->fd) + P->xl) * kRB * (double)A.nblocks * (A.vals_r ? 8.0 : 16.0) + 64.0 * (double)A.nrows;
  const bool resident = footprint <= 230e6;
  // beyond it: every CU but the few the edge workgroups take (8 x (256 - 24) = 1856 for the headline lattice), so that
  // the edge blocks run BESIDE the walk there too; 2048 with the edge blocks inside the walk's wavefronts when that would
  // leave more than an eighth of the chip to them (profiles/r03/kbench_walk_development.txt: 2^21 rows 71.4 -> 68.5 us,
  // 2^22 126.0 -> 121.8, 2^23 275 -> 278)
  const int ws = resident ? 4 : kWalkWaves;            // wavefronts per workgroup (two 4-wavefront workgroups fit a CU)
  const int64_t wg_slots = (int64_t)std::max(device_cu_count() - reserve, 8) * (kWalkWaves / ws);  // workgroups the walk may hold at once
  const int64_t edge_wgs_all = (P->n_edge + ws - 1) / ws;
  const int waves_beside = (int)(ws * std::max<int64_t>(0, wg_slots - edge_wgs_all)) / P->S * P->S;
  const int waves = tun.walk_waves > 0 ? tun.walk_waves
                    : resident ? (A.vals_r ? 1024 : 768)   // (real copy, half the value bytes per step: 1024; N = 2^20: 26.6 -> 24.3 us)
                    : ((rs || waves_beside >= 7 * kWalkWaves * device_cu_count() / 8) ? std::max(waves_beside, P->S) : kWalkWaves * device_cu_count());
  const int ntm = tun.walk_nt >= 0 ? tun.walk_nt : (resident ? 0 : 1);
  const int64_t nseg_target = std::max<int64_t>(1, waves / P->S);
#ifdef QP_DEVELOPER
  const bool no_edges = (tun.walk_dbg & 2) != 0;   // measurement only, developer builds only: the edge blocks are skipped, results are WRONG
#else
  const bool no_edges = false;
#endif
  const int64_t edge_wgs = edge_wgs_all;
  // edge blocks as workgroups of their own while every workgroup of the launch still finds room on the chip at once
  const bool edge_beside = !no_edges && (tun.walk_dbg & 4) == 0 &&
                           (nseg_target * P->S + ws - 1) / ws + edge_wgs <= wg_slots;
  G.n_edge_wg = edge_beside ? (int)edge_wgs : 0;
  G.edge_steps = (no_edges || edge_beside) ? 0 : kWalkEdgeSteps;
  G.edge_last = (tun.walk_dbg & 1) ? 1 : 0;
  G.edge_segs = (no_edges || edge_beside) ? 0 : (int)std::min<int64_t>(nseg_target, (P->n_edge + P->S - 1) / P->S);
  G.xlast = A.ncols - 1;
  G.L = (int)std::max<int64_t>(G.edge_steps + 1, (J + (int64_t)G.edge_segs * G.edge_steps + nseg_target - 1) / nseg_target);
  G.nseg = (int)((J + (int64_t)G.edge_segs * G.edge_steps + G.L - 1) / G.L);
  while ((int64_t)G.nseg * G.L - (int64_t)std::min(G.edge_segs, G.nseg) * G.edge_steps < J) ++G.nseg;   // (tiny operators)
  G.edge_segs = std::min(G.edge_segs, G.nseg);
  const int64_t ntask = (int64_t)G.nseg * P->S;
  G.n_walk_wg = (int)((ntask + ws - 1) / ws);
  G.ntask = G.n_walk_wg * ws;
  HrbArrays H{A.bptr, A.cmeta, reinterpret_cast<const char*>(A.cols), A.lptr, A.lcmeta,
              reinterpret_cast<const char*>(A.lcols), reinterpret_cast<const int4*>(A.lpos)};
  ChebyOp op{e};
  const dim3 grid((unsigned)(G.n_edge_wg + G.n_walk_wg));
  WalkPlan Pl = *P;
  if (no_edges) Pl.n_edge = 0;
  const bool hi = Pl.nn >= 3 && !Pl.xl;   // which translation unit holds the shape
  const bool xlu = Pl.xl && !(Pl.xl == 1 && Pl.K == 1);
  const bool ok = Pl.fd ? (A.vals_r ? walk_launch_f64_fd(s, grid, A.vals_r, x, Pl, G, H, A.nrows, op, ntm, sy)
                                    : walk_launch_c128_fd(s, grid, A.vals, x, Pl, G, H, A.nrows, op, ntm, sy))
                  : A.vals_r ? (xlu  ? walk_launch_f64_xl(s, grid, A.vals_r, x, Pl, G, H, A.nrows, op, ntm, sy)
                              : hi ? walk_launch_f64_hi(s, grid, A.vals_r, x, Pl, G, H, A.nrows, op, ntm, sy)
                                   : walk_launch_f64_lo(s, grid, A.vals_r, x, Pl, G, H, A.nrows, op, ntm, sy))
                           : (xlu  ? walk_launch_c128_xl(s, grid, A.vals, x, Pl, G, H, A.nrows, op, ntm, sy)
                              : hi ? walk_launch_c128_hi(s, grid, A.vals, x, Pl, G, H, A.nrows, op, ntm, sy)
                                   : walk_launch_c128_lo(s, grid, A.vals, x, Pl, G, H, A.nrows, op, ntm, sy));
  if (!ok) return QP_OK;
  QP_HIP(hipGetLastError());
  *launched = true;
  return QP_OK;
}

}  // namespace qp
