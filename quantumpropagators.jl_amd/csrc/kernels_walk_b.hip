// Strip-walk kernel shapes (kernels_walk_impl.h): complex values, 1-2 near distances and the long-pair shapes of three-dimensional grids.
#include "kernels_walk_impl.h"

namespace qp {

bool walk_launch_c128_lo(hipStream_t s, dim3 grid, const double2* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                             const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy) {
  return launch_shape<double2, 1>(s, grid, uvals, x, P, G, H, nrows, op, ntm, sy);
}

}  // namespace qp
