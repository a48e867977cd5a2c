// Two-term strip-walk kernel shapes (kernels_walk2_impl.h): 1 near distance.
#include "kernels_walk2_impl.h"

namespace qp {

QP_WALK2_DEFINE(walk2_launch_nn1, 1)

}  // namespace qp
