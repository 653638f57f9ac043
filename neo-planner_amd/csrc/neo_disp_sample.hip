// neo_disp_sample.hip -- sample_kernel family: the ESDF-lookup kernel (expert_planner.py:392-466)
#include "neo_host.hpp"
#include "neo_kernels.hpp"
#ifdef NEO_SAMPLE_EXPERIMENTS  // measured and not adopted (round 3): tools/probe/neo_sample_wg.hpp, neo_sample_chunk.hpp
#include "../../tools/probe/neo_sample_chunk.hpp"
#include "../../tools/probe/neo_sample_wg.hpp"
#include <cstdlib>
#endif

namespace neo {

#ifdef NEO_SAMPLE_EXPERIMENTS
// workgroup-per-trajectory form with LDS-staged gathers (neo_sample_wg.hpp): fp32 sampling on yz-quad fields
template <class LookupT, int WPT, int PF, int OCC>
int launch_sample_wg(neo_ctx *c, const Map3D &map, const SampleArgs &a) {
  hipLaunchKernelGGL((sample_wg_kernel<LookupT, WPT, PF, OCC>), dim3(a.B), dim3(kWave * WPT), 0, c->stream, a.B, a.M, c->dev,
                     map, a.coeffs, a.ts, a.costs2, a.grad_C, a.grad_T);
  return NEO_OK;
}
// variant = 100 * (wavefronts per SIMD) + 10 * (wavefronts per trajectory) + (rounds of gathers in flight per lane)
template <class LookupT>
int dispatch_sample_wg(neo_ctx *c, const Map3D &map, const SampleArgs &a, int variant) {
  switch (variant) {
    case 412: return launch_sample_wg<LookupT, 1, 2, 4>(c, map, a);
    case 414: return launch_sample_wg<LookupT, 1, 4, 4>(c, map, a);
    case 423: return launch_sample_wg<LookupT, 2, 3, 4>(c, map, a);
    case 443: return launch_sample_wg<LookupT, 4, 3, 4>(c, map, a);
    case 314: return launch_sample_wg<LookupT, 1, 4, 3>(c, map, a);
    case 316: return launch_sample_wg<LookupT, 1, 6, 3>(c, map, a);
    case 323: return launch_sample_wg<LookupT, 2, 3, 3>(c, map, a);
    case 343: return launch_sample_wg<LookupT, 4, 3, 3>(c, map, a);
    default: return launch_sample_wg<LookupT, 2, 5, 3>(c, map, a);
  }
}

#endif

#ifdef NEO_SAMPLE_EXPERIMENTS
// fp32 sampling: samples dealt to the lanes in contiguous chunks (neo_sample_chunk.hpp)
template <int D, class MapT, class LookupT>
int launch_sample_chunk(neo_ctx *c, const MapT &map, const SampleArgs &a) {
  hipLaunchKernelGGL((sample_chunk_kernel<D, MapT, LookupT>), dim3(a.B), dim3(kWave), chunk_lds_bytes<D>(a.M), c->stream, a.B,
                     a.M, c->dev, map, a.coeffs, a.ts, a.costs2, a.grad_C, a.grad_T);
  return NEO_OK;
}
#endif

template <int D, typename Real, class MapT, class LookupT>
int launch_sample(neo_ctx *c, const MapT &map, const SampleArgs &a) {
  hipLaunchKernelGGL((sample_kernel<D, Real, MapT, LookupT>), dim3(a.B), dim3(kWave),
#ifdef NEO_SAMPLE_SHARED_TAILS  // (experiment builds: tools/probe/neo_sample_shared.hpp)
                     sizeof(Real) == 4 ? sample_lds_bytes(a.M, D) : 0,
#else
                     0,
#endif
                     c->stream, a.B, a.M, c->dev,
                     map, a.coeffs, a.ts, a.costs2, a.grad_C, a.grad_T, (c->sample_order_B == a.B ? c->sample_order : nullptr));
  return NEO_OK;
}

int dispatch_sample(neo_ctx *c, const MapEntry &e, int D, const SampleArgs &a) {
#ifdef NEO_SLIM_BUILD  // kernel experiments (tools/probe): only the cfg2 instantiation compiles, in 20 s
  if (e.kind != 0 && D == 3 && e.elem == NEO_F32 && e.m3.layout == 0 && c->params.sample_dtype == NEO_F32)
    return launch_sample<3, float, Map3D, Lookup3D<float, float, 0>>(c, e.m3, a);
  return fail(c, NEO_ERR_INVALID, "slim build: cfg2 kernels only");
#else
  const bool f32 = c->params.sample_dtype == NEO_F32;
  if (e.kind == 0) {
    if (D == 2)
      return f32 ? launch_sample<2, float, Map2D, Lookup2D<float>>(c, e.m2, a)
                 : launch_sample<2, double, Map2D, Lookup2D<double>>(c, e.m2, a);
    return f32 ? launch_sample<3, float, Map2D, Lookup2D<float>>(c, e.m2, a)
               : launch_sample<3, double, Map2D, Lookup2D<double>>(c, e.m2, a);
  }
  if (D != 3) return fail(c, NEO_ERR_INVALID, "a 3-D map needs D = 3");
#ifdef NEO_SAMPLE_EXPERIMENTS
  {
    const char *ev = std::getenv("NEO_SAMPLE_VARIANT");  // dispatch_sample_wg; unset or 0 = the product's kernels
    const int variant = ev ? std::atoi(ev) : 0;
    if (f32 && e.m3.layout == NEO_LAYOUT_YZ4 && variant > 1) {
      if (e.elem == NEO_F32) return dispatch_sample_wg<Lookup3D<float, float, 1>>(c, e.m3, a, variant);
      return dispatch_sample_wg<Lookup3D<float, __half, 1>>(c, e.m3, a, variant);
    }
    if (f32 && e.m3.layout == NEO_LAYOUT_YZ4 && e.elem == NEO_F32 && variant == 1)
      return launch_sample_chunk<3, Map3D, Lookup3D<float, float, 1>>(c, e.m3, a);
  }
#endif
#define NEO_3D(LAY)                                                                                  \
  if (e.elem == NEO_F32)                                                                             \
    return f32 ? launch_sample<3, float, Map3D, Lookup3D<float, float, LAY>>(c, e.m3, a)             \
               : launch_sample<3, double, Map3D, Lookup3D<double, float, LAY>>(c, e.m3, a);          \
  return f32 ? launch_sample<3, float, Map3D, Lookup3D<float, __half, LAY>>(c, e.m3, a)              \
             : launch_sample<3, double, Map3D, Lookup3D<double, __half, LAY>>(c, e.m3, a);
  if (e.m3.layout == 0) { NEO_3D(0) }
  if (e.m3.layout == 2) { NEO_3D(2) }
  if (e.m3.layout == 3) { NEO_3D(3) }
  NEO_3D(1)
#undef NEO_3D
#endif
}

}  // namespace neo
