// neo_disp_opt2d.hip -- optimize_kernel on the 2-D reference map (nearest-cell lookup, esdf.py:53-82)
#include "neo_launch_opt.hpp"

namespace neo {

int launch_opt_2d(neo_ctx *c, int D, bool f32, const OptArgs &a) {
  if (D == 2)
    return f32 ? launch_opt<2, float, Map2D, Lookup2D<float>>(c, a) : launch_opt<2, double, Map2D, Lookup2D<double>>(c, a);
  return f32 ? launch_opt<3, float, Map2D, Lookup2D<float>>(c, a) : launch_opt<3, double, Map2D, Lookup2D<double>>(c, a);
}

// the reference's own shape in batches (D = 2): register allocation for two wavefronts per SIMD -- the planar problem
// needs 257 registers at the one-wave budget and fits 256 without spills; same arithmetic, same results
int launch_opt_2d_w2(neo_ctx *c, bool f32, const OptArgs &a) {
  return f32 ? launch_opt<2, float, Map2D, Lookup2D<float>, 2>(c, a) : launch_opt<2, double, Map2D, Lookup2D<double>, 2>(c, a);
}

}  // namespace neo
