// neo_disp_opt3d_f64.hip -- optimize_kernel on 3-D fields, fp64 sampling (parity mode), one wavefront per SIMD
#include "neo_launch_opt.hpp"

namespace neo {

int launch_opt_3d_f64(neo_ctx *c, int elem, int layout, const OptArgs &a) {
#define NEO_3D(LAY)                                                                                \
  if (elem == NEO_F32) return launch_opt<3, double, Map3D, Lookup3D<double, float, LAY>>(c, a);      \
  return launch_opt<3, double, Map3D, Lookup3D<double, __half, LAY>>(c, a);
  if (layout == 0) { NEO_3D(0) }
  if (layout == 2) { NEO_3D(2) }
  if (layout == 3) { NEO_3D(3) }
  NEO_3D(1)
#undef NEO_3D
}

// the same in the two-wavefronts-per-SIMD register allocation (256 registers: the lane = (piece, dimension) kernels
// spill 25 - 31 of them), for batches that queue for the SIMDs anyway
int launch_opt_3d_f64_w2(neo_ctx *c, int elem, int layout, const OptArgs &a) {
#define NEO_3D(LAY)                                                                                   \
  if (elem == NEO_F32) return launch_opt<3, double, Map3D, Lookup3D<double, float, LAY>, 2>(c, a);      \
  return launch_opt<3, double, Map3D, Lookup3D<double, __half, LAY>, 2>(c, a);
  if (layout == 0) { NEO_3D(0) }
  if (layout == 2) { NEO_3D(2) }
  if (layout == 3) { NEO_3D(3) }
  NEO_3D(1)
#undef NEO_3D
}

}  // namespace neo
