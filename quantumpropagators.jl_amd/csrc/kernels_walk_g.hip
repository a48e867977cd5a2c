// Strip-walk kernel shapes (kernels_walk_impl.h): complex values, diagonal far neighbours (nine-point stencils of two-dimensional grids:
// +-1, +-(g - 1), +-g, +-(g + 1)).
#include "kernels_walk_impl.h"

namespace qp {

bool walk_launch_c128_fd(hipStream_t s, dim3 grid, const double2* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy) {
  return launch_shape<double2, 3>(s, grid, uvals, x, P, G, H, nrows, op, ntm, sy);
}

}  // namespace qp
