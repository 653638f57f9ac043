// neo_disp_group.hip -- optimize_group_kernel family: several small trajectories per wavefront (NEO_FLAG_LANE_GROUPS)
#include "neo_host.hpp"
#include "neo_kernels.hpp"
#include "neo_group_kernel.hpp"

namespace neo {

constexpr int kTicketRing = 64;

template <typename Real, class LookupT, int W, int NS, typename Num, int D = 3, class MapT = Map3D>
int launch_group_w(neo_ctx *c, const OptArgs &a) {
  constexpr int G = kWave / W;
  if (!c->tickets) {
    HIPCHK(c, hipMalloc((void **)&c->tickets, kTicketRing * sizeof(int)));
  }
  int *ticket = c->tickets + (c->ticket_next++ % kTicketRing);
  HIPCHK(c, hipMemsetAsync(ticket, 0, sizeof(int), c->stream));
  const int n = D * (a.M - 1) + a.M;
  const size_t dyn = (size_t)G * 2 * NEO_LBFGS_M * n * sizeof(Num);
  const int waves = std::min((a.B + G - 1) / G, 4096);  // persistent groups: they draw trajectories off the ticket
  hipLaunchKernelGGL((optimize_group_kernel<D, Real, MapT, LookupT, W, NS, Num>), dim3(waves), dim3(kWave), dyn, c->stream,
                     a.B, a.M, c->dev, static_cast<const MapT *>(a.table), a.x0 ? a.x0 : a.x, a.x, a.head, a.tail, a.costs4, a.costs4_last,
                     a.nit, a.nfev, a.status, c->sample_counter, (c->order_B == a.B ? c->dispatch_order : nullptr),
                     ticket);
  return NEO_OK;
}

// n <= 16: eight trajectories per wavefront when the pieces fit 8 lanes (flags bit 256: sixteen-lane groups, for
// comparison), else four
template <typename Real, class LookupT, typename Num>
int launch_group_n(neo_ctx *c, const OptArgs &a) {
  const int n = 3 * (a.M - 1) + a.M;
  if (n > 16) return launch_group_w<Real, LookupT, 16, 2, Num>(c, a);  // n <= 32 (M <= 8): four per wavefront, two slots
  if (a.M <= 8 && !(c->params.flags & 256)) return launch_group_w<Real, LookupT, 8, 2, Num>(c, a);
  return launch_group_w<Real, LookupT, 16, 1, Num>(c, a);
}
// NEO_FLAG_F32_SOLVE: solve, adjoint, optimiser vectors and pairs in fp32
template <typename Real, class LookupT>
int launch_group(neo_ctx *c, const OptArgs &a) {
  if (c->params.flags & NEO_FLAG_F32_SOLVE) return launch_group_n<Real, LookupT, float>(c, a);
  return launch_group_n<Real, LookupT, double>(c, a);
}

// the reference's own shape in batches: D = 2 on the nearest-cell map (esdf.py:53-82), M = 3 -> n = 7: eight replans per
// wavefront.  fp64 sampling (the parity arithmetic) or fp32; solve and optimiser in fp64.
template <typename Real>
int launch_group_2d(neo_ctx *c, const OptArgs &a) {
  const int n = 2 * (a.M - 1) + a.M;
  using LK = Lookup2D<Real>;
  if (n > 16) return launch_group_w<Real, LK, 16, 2, double, 2, Map2D>(c, a);
  if (a.M <= 8 && !(c->params.flags & 256)) return launch_group_w<Real, LK, 8, 2, double, 2, Map2D>(c, a);
  return launch_group_w<Real, LK, 16, 1, double, 2, Map2D>(c, a);
}
int launch_opt_groups_2d(neo_ctx *c, bool f32, const OptArgs &a) {
  return f32 ? launch_group_2d<float>(c, a) : launch_group_2d<double>(c, a);
}

int launch_opt_groups(neo_ctx *c, int elem, int layout, const OptArgs &a) {
  if (layout == NEO_LAYOUT_YZ4) {
    if (elem == NEO_F32) return launch_group<float, Lookup3D<float, float, 1>>(c, a);
    return launch_group<float, Lookup3D<float, __half, 1>>(c, a);
  }
  if (layout == NEO_LAYOUT_BRICK) {
    if (elem == NEO_F32) return launch_group<float, Lookup3D<float, float, 3>>(c, a);
    return launch_group<float, Lookup3D<float, __half, 3>>(c, a);
  }
  if (elem == NEO_F32) return launch_group<float, Lookup3D<float, float, 0>>(c, a);
  return launch_group<float, Lookup3D<float, __half, 0>>(c, a);
}

}  // namespace neo
