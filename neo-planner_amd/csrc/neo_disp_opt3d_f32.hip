// neo_disp_opt3d_f32.hip -- optimize_kernel on 3-D fields, fp32 sampling, one wavefront per SIMD
#include "neo_launch_opt.hpp"

namespace neo {

int launch_opt_3d_f32(neo_ctx *c, int elem, int layout, const OptArgs &a) {
#ifdef NEO_SLIM_BUILD  // kernel experiments (tools/probe/kstats.sh): only the cfg2 instantiation
  return launch_opt<3, float, Map3D, Lookup3D<float, float, 0>>(c, a);
#else
#define NEO_3D(LAY)                                                                                \
  if (elem == NEO_F32) return launch_opt<3, float, Map3D, Lookup3D<float, float, LAY>>(c, a);      \
  return launch_opt<3, float, Map3D, Lookup3D<float, __half, LAY>>(c, a);
  if (layout == 0) { NEO_3D(0) }
  if (layout == 2) { NEO_3D(2) }
  if (layout == 3) { NEO_3D(3) }
  NEO_3D(1)
#undef NEO_3D
#endif
}

}  // namespace neo
