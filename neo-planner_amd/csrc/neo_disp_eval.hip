// neo_disp_eval.hip -- eval_kernel family: get_cost + get_grad for a batch (expert_planner.py:539-585)
#include "neo_host.hpp"
#include "neo_kernels.hpp"

namespace neo {

template <int D, typename Real, class MapT, class LookupT, typename Num = double>
int launch_eval(neo_ctx *c, const MapT &map, const EvalArgs &a) {
  const dim3 grid(a.B), blk(kWave);
#define NEO_EVAL_LG(NS, LG)                                                                                        \
  hipLaunchKernelGGL((eval_kernel<D, NS, Real, MapT, LookupT, LG, Num>), grid, blk, 0, c->stream, a.B, a.M, c->dev, map, \
                     a.x, a.head, a.tail, a.cost, a.costs4, a.grad, a.coeffs, a.status)
  const bool pd = D * a.M <= kWave && !(c->params.flags & 512);  // lane = (piece, dimension), as in launch_opt
#define NEO_EVAL(NS)                      \
  do {                                    \
    if (pd)                               \
      NEO_EVAL_LG(NS, WaveLanesPD<D>);    \
    else                                  \
      NEO_EVAL_LG(NS, WaveLanes);         \
  } while (0)
  switch (slots_for(a.M, D)) {
    case 1: NEO_EVAL(1); break;
    case 2: NEO_EVAL(2); break;
    case 3: NEO_EVAL_LG(3, WaveLanes); break;
    default: NEO_EVAL_LG(4, WaveLanes); break;
  }
#undef NEO_EVAL_LG
#undef NEO_EVAL
  return NEO_OK;
}

int dispatch_eval(neo_ctx *c, const MapEntry &e, int D, const EvalArgs &a) {
#ifdef NEO_SLIM_BUILD  // kernel experiments (tools/probe): only the cfg2 instantiation compiles, in 20 s
  if (e.kind != 0 && D == 3 && e.elem == NEO_F32 && e.m3.layout == 0 && c->params.sample_dtype == NEO_F32)
    return launch_eval<3, float, Map3D, Lookup3D<float, float, 0>>(c, e.m3, a);
  return fail(c, NEO_ERR_INVALID, "slim build: cfg2 kernels only");
#else
  const bool f32 = c->params.sample_dtype == NEO_F32;
  if (e.kind == 0) {
    if (f32 && (c->params.flags & NEO_FLAG_F32_SOLVE)) {  // all-fp32 mode on the 2-D map (neo_disp_opt2d_x.hip)
      if (D == 2) return launch_eval<2, float, Map2D, Lookup2D<float>, float>(c, e.m2, a);
      return launch_eval<3, float, Map2D, Lookup2D<float>, float>(c, e.m2, a);
    }
    if (D == 2)
      return f32 ? launch_eval<2, float, Map2D, Lookup2D<float>>(c, e.m2, a)
                 : launch_eval<2, double, Map2D, Lookup2D<double>>(c, e.m2, a);
    return f32 ? launch_eval<3, float, Map2D, Lookup2D<float>>(c, e.m2, a)
               : launch_eval<3, double, Map2D, Lookup2D<double>>(c, e.m2, a);
  }
  if (D != 3) return fail(c, NEO_ERR_INVALID, "a 3-D map needs D = 3");
  if (f32 && (c->params.flags & NEO_FLAG_F32_SOLVE)) {  // all-fp32 mode (neo_disp_opt3d_x.hip runs the optimiser)
#define NEO_3DX(LAY)                                                                                        \
  if (e.elem == NEO_F32) return launch_eval<3, float, Map3D, Lookup3D<float, float, LAY>, float>(c, e.m3, a); \
  return launch_eval<3, float, Map3D, Lookup3D<float, __half, LAY>, float>(c, e.m3, a);
    if (e.m3.layout == 0) { NEO_3DX(0) }
    if (e.m3.layout == 2) { NEO_3DX(2) }
    if (e.m3.layout == 3) { NEO_3DX(3) }
    NEO_3DX(1)
#undef NEO_3DX
  }
#define NEO_3D(LAY)                                                                                  \
  if (e.elem == NEO_F32)                                                                             \
    return f32 ? launch_eval<3, float, Map3D, Lookup3D<float, float, LAY>>(c, e.m3, a)               \
               : launch_eval<3, double, Map3D, Lookup3D<double, float, LAY>>(c, e.m3, a);            \
  return f32 ? launch_eval<3, float, Map3D, Lookup3D<float, __half, LAY>>(c, e.m3, a)                \
             : launch_eval<3, double, Map3D, Lookup3D<double, __half, LAY>>(c, e.m3, a);
  if (e.m3.layout == 0) { NEO_3D(0) }
  if (e.m3.layout == 2) { NEO_3D(2) }
  if (e.m3.layout == 3) { NEO_3D(3) }
  NEO_3D(1)
#undef NEO_3D
#endif
}

}  // namespace neo
