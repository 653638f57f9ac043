// neo_disp_opt3d_b.hip -- optimize_kernel with an evaluation budget per launch and resumable runs
// (neo_optimize_batch_budget_dev): 3-D fp32 fields in the linear or the corner-brick layout, every arithmetic mode, the
// register allocation of the throughput variants (bit-identical to the others by construction)
#include "neo_launch_opt.hpp"

namespace neo {

int launch_opt_3d_budget(neo_ctx *c, int elem, int layout, const OptArgs &a) {
  if (elem != NEO_F32 || (layout != NEO_LAYOUT_LINEAR && layout != NEO_LAYOUT_BRICK))
    return fail(c, NEO_ERR_INVALID, "budgeted launches: an fp32 field in the linear or the brick layout");
  const bool f32 = c->params.sample_dtype == NEO_F32;
  const bool x = f32 && (c->params.flags & NEO_FLAG_F32_SOLVE);
#define NEO_3DB(LAY)                                                                                          \
  if (x) return launch_opt<3, float, Map3D, Lookup3D<float, float, LAY>, 2, float, true>(c, a);               \
  if (f32) return launch_opt<3, float, Map3D, Lookup3D<float, float, LAY>, 2, double, true>(c, a);            \
  return launch_opt<3, double, Map3D, Lookup3D<double, float, LAY>, 2, double, true>(c, a);
  if (layout == NEO_LAYOUT_BRICK) { NEO_3DB(3) }
  NEO_3DB(0)
#undef NEO_3DB
}

}  // namespace neo
