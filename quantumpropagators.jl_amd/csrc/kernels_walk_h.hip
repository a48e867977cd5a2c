// Strip-walk kernel shapes (kernels_walk_impl.h): the real copy of an all-real operator, diagonal far neighbours (nine-point stencils of two-dimensional grids:
// +-1, +-(g - 1), +-g, +-(g + 1)).
#include "kernels_walk_impl.h"

namespace qp {

bool walk_launch_f64_fd(hipStream_t s, dim3 grid, const double* uvals, const double2* x, const WalkPlan& P, const WalkGeom& G,
                         const HrbArrays& H, int64_t nrows, const ChebyOp& op, int ntm, const SyncArgs& sy) {
  return launch_shape<double, 3>(s, grid, uvals, x, P, G, H, nrows, op, ntm, sy);
}

}  // namespace qp
