// Strip-walk kernel shapes (kernels_walk_impl.h): the real copy of an all-real operator, the long-pair shapes beyond (one pair, one far distance) --
// two far distances and / or two long pairs: higher-order stencils of three-dimensional grids, four-dimensional grids.
#include "kernels_walk_impl.h"

namespace qp {

bool walk_launch_f64_xl(hipStream_t s, dim3 grid, const double* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy) {
  return launch_shape<double, 2>(s, grid, uvals, x, P, G, H, nrows, op, ntm, sy);
}

}  // namespace qp
