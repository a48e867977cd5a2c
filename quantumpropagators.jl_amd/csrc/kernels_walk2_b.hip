// Two-term strip-walk kernel shapes (kernels_walk2_impl.h): 3 near distances.
#include "kernels_walk2_impl.h"

namespace qp {

QP_WALK2_DEFINE(walk2_launch_nn3, 3)

}  // namespace qp
