// neo_disp_opt2d_x.hip -- optimize_kernel on the 2-D reference map (nearest-cell lookup, esdf.py:53-82) in the all-fp32
// mode (NEO_FLAG_F32_SOLVE): fp32 sampling, fp32 coefficient solve (parallel cyclic reduction) / adjoint / optimiser
// vectors / stored pairs -- the arithmetic bench.py times on 3-D fields, here on the map the reference itself has, so that
// it can be held to the reference-generated fixtures (tests/golden g1 / g3 / g6) directly.  The lookup's index arithmetic
// stays fp64 with a true division, bit for bit int((y - origin.y) / res) of the fp32 position.
#include "neo_launch_opt.hpp"

namespace neo {

int launch_opt_2d_x(neo_ctx *c, int D, const OptArgs &a) {
  if (D == 2) return launch_opt<2, float, Map2D, Lookup2D<float>, 2, float>(c, a);
  return launch_opt<3, float, Map2D, Lookup2D<float>, 2, float>(c, a);
}

}  // namespace neo
