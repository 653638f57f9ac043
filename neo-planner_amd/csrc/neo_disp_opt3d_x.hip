// neo_disp_opt3d_x.hip -- optimize_kernel on 3-D fields in the all-fp32 mode (NEO_FLAG_F32_SOLVE): fp32 sampling, fp32
// coefficient solve / adjoint / optimiser vectors / stored pairs, register allocation for two wavefronts per SIMD
#include "neo_launch_opt.hpp"

namespace neo {

int launch_opt_3d_x(neo_ctx *c, int elem, int layout, const OptArgs &a) {
#ifdef NEO_SLIM_BUILD  // kernel experiments (tools/probe/kstats.sh): only the cfg2 instantiation
  return launch_opt<3, float, Map3D, Lookup3D<float, float, 3>, 2, float>(c, a);  // (brick: the bench default since round 4)
#else
#ifdef NEO_X_ONE_WAVE  // experiment builds: a one-wavefront-per-SIMD allocation of the all-fp32 kernel (four samples a lane in
                      // flight) for NEO_FLAG_ONE_WAVE_PER_SIMD on fp32 brick fields -- the latency of a lone wavefront
  if ((c->params.flags & NEO_FLAG_ONE_WAVE_PER_SIMD) && elem == NEO_F32 && layout == 3)
    return launch_opt<3, float, Map3D, Lookup3D<float, float, 3>, 1, float>(c, a);
#endif
#define NEO_3DX(LAY)                                                                                       \
  if (elem == NEO_F32) return launch_opt<3, float, Map3D, Lookup3D<float, float, LAY>, 2, float>(c, a);    \
  return launch_opt<3, float, Map3D, Lookup3D<float, __half, LAY>, 2, float>(c, a);
  if (layout == 0) { NEO_3DX(0) }
  if (layout == 2) { NEO_3DX(2) }
  if (layout == 3) { NEO_3DX(3) }
  NEO_3DX(1)
#undef NEO_3DX
#endif
}

}  // namespace neo
