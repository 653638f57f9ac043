// neo_disp_opt3d_w2.hip -- optimize_kernel on 3-D fields, fp32 sampling, register allocation for two wavefronts per SIMD
#include "neo_launch_opt.hpp"

namespace neo {

int launch_opt_3d_w2(neo_ctx *c, int elem, int layout, const OptArgs &a) {
#ifdef NEO_SLIM_BUILD  // kernel experiments (tools/probe/kstats.sh): only the cfg2 instantiation
  return launch_opt<3, float, Map3D, Lookup3D<float, float, 0>, 2>(c, a);
#else
#define NEO_3D2(LAY)                                                                                \
  if (elem == NEO_F32) return launch_opt<3, float, Map3D, Lookup3D<float, float, LAY>, 2>(c, a);    \
  return launch_opt<3, float, Map3D, Lookup3D<float, __half, LAY>, 2>(c, a);
  if (layout == 0) { NEO_3D2(0) }
  if (layout == 2) { NEO_3D2(2) }
  if (layout == 3) { NEO_3D2(3) }
  NEO_3D2(1)
#undef NEO_3D2
#endif
}

}  // namespace neo
