// Two-term strip-walk kernel shapes (kernels_walk2_impl.h): 2 near distances.
#include "kernels_walk2_impl.h"

namespace qp {

QP_WALK2_DEFINE(walk2_launch_nn2, 2)

}  // namespace qp
