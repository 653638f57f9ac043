// neo_launch_opt.hpp -- launch of one optimize_kernel family member (shared by the neo_disp_opt*.hip units)
#pragma once
#include "neo_host.hpp"
#include "neo_kernels.hpp"

namespace neo {

// Num = float: the all-fp32 mode (NEO_FLAG_F32_SOLVE) -- solve, adjoint and optimiser vectors in fp32, pairs in fp32
template <int D, typename Real, class MapT, class LookupT, int WAVES = 1, typename Num = double, bool BUDGET = false>
int launch_opt(neo_ctx *c, const OptArgs &a) {
  // (a budgeted launch may cover a subset of the batch: workgroup i then works on trajectory subset[i])
  const int n_launch = a.subset ? a.n_subset : a.B;
  const int *launch_order = a.subset ? a.subset : (c->order_B == a.B ? c->dispatch_order : nullptr);
  const dim3 grid(n_launch), blk(kWave);
  const size_t pair_elems = (size_t)2 * NEO_LBFGS_M * (D * (a.M - 1) + a.M);  // L-BFGS pairs in LDS
  // staging in front of the pairs: the full size (with the rows of the per-piece fold) unless that costs the two-waves
  // variant occupancy -- eight wavefronts per CU want 160 KB / 8 each, less ~0.5 KB of static LDS.  The one-wave
  // variant follows the same rule so that both sum the partials in the same order (bit-identical results).
  // (static LDS beside the dynamic part: line-search state, cost terms, boundary states; the all-fp32 kernels'
  //  lane-assignment cache kSlCacheInts * 4 = 336 bytes more)
  // The chip hands out LDS in granules of 1 280 bytes (measured in round 5: a workgroup of 13 152 bytes runs eleven to a
  // CU, one of 12 704 twelve): the share of a wavefront is 160 KB / waves rounded DOWN to that.
  const size_t cache = sizeof(Num) == 4 ? kSlCacheInts * sizeof(int) : 0;
  const size_t statics = 400;  // line-search state 160 + cost terms 96 + boundary states 72 / 144 (fp32 / fp64), padded
  const size_t lds_share8 = (size_t)160 * 1024 / 8 / 1280 * 1280 - statics - cache,
               lds_share12 = (size_t)160 * 1024 / (4 * NEO_X_OCC) / 1280 * 1280 - statics - cache;
#define NEO_OPT_LG(NS, LG)                                                                                    \
  do {                                                                                                        \
    const size_t pairs = pair_elems * ((pairs_in_f32<Real, NS, WAVES>() || sizeof(Num) == 4) ? sizeof(float) : sizeof(double)); \
    const int full = stage_doubles<D, NS, Real>(), small = NS * kWave;                                        \
    const size_t lds_share = (sizeof(Num) == 4 && NS <= 2) ? lds_share12 : lds_share8; /* all-fp32: twelve per CU */ \
    int stage = pairs + (size_t)full * 8 <= lds_share ? full : small;                                         \
    size_t dyn = pairs + (size_t)stage * 8;                                                                   \
    int pcr_off = 0;                                                                                          \
    if (sizeof(Num) == 4 && LG::S > 1 && !(c->params.flags & 4096)) {                                             \
      /* all-fp32, lane = (piece, dimension): room for the reduction's multipliers next to the pairs when the fold   \
         runs on per-piece accumulators (80 B a piece at D = 3) instead of rows (96 B a lane); flags bit 4096: off (comparison runs) */ \
      const int acc = std::max(std::max(small, (a.M * fold_acc_stride(D) * 4 + 7) / 8), (pcr_xch_elems(a.M, 1) * 4 + 7) / 8);                                                    \
      const size_t off = ((size_t)acc * 8 + pairs + 15) / 16 * 2;                                             \
      const size_t need = off * 8 + (size_t)pcr_mult_elems(a.M) * sizeof(float);                              \
      if (need <= lds_share) {                                                                                \
        stage = acc;                                                                                          \
        pcr_off = (int)off;                                                                                   \
        dyn = need;                                                                                           \
      }                                                                                                       \
    }                                                                                                         \
    hipLaunchKernelGGL((optimize_kernel<D, NS, Real, MapT, LookupT, WAVES, LG, Num, BUDGET>), grid, blk,           \
                       dyn, c->stream, n_launch, a.M, c->dev,                                                 \
                       static_cast<const MapT *>(a.table), a.slots, a.nmaps, a.x0 ? a.x0 : a.x, a.x, a.head, a.tail, a.costs4,   \
                       a.costs4_last, a.nit, a.nfev, a.status, c->sample_counter,                             \
                       launch_order, c->trace, c->trace_xg, c->trace_cap, stage, pcr_off,                     \
                       BUDGET ? a.state : reinterpret_cast<double *>(c->progress) /* (plain launches: the progress counter) */, \
                       a.state_doubles,                                                                       \
                       a.budget, a.resume, a.traj_total);                                                     \
  } while (0)
  // lane = (piece, dimension) whenever D * M fits the wavefront (cfg2: 63 lanes busy in the PIECE-layout phases
  // instead of 21, a third of the per-dimension state per lane); lane = piece otherwise.  flags bit 512 forces the
  // latter (comparison runs).
  const bool pd = D * a.M <= kWave && !(c->params.flags & 512);
#define NEO_OPT(NS)                      \
  do {                                   \
    if (pd)                              \
      NEO_OPT_LG(NS, WaveLanesPD<D>);    \
    else                                 \
      NEO_OPT_LG(NS, WaveLanes);         \
  } while (0)
  if constexpr (BUDGET) {
    // the resumable form of the run exists for n <= 128 variables (neo_kernels.hpp NEO_SM_MAX_SLOTS)
    switch (slots_for(a.M, D)) {
      case 1: NEO_OPT(1); return NEO_OK;
      case 2: NEO_OPT(2); return NEO_OK;
      default: return fail(c, NEO_ERR_INVALID, "budgeted launches: n <= 128 variables");
    }
  } else
  switch (slots_for(a.M, D)) {
    case 1: NEO_OPT(1); break;
    case 2: NEO_OPT(2); break;
    case 3:
      if constexpr (WAVES == 1 || NEO_W2_MAX_SLOTS >= 4)
        NEO_OPT_LG(3, WaveLanes);  // (n > 128 means D * M > 64: lane = piece)
      else
        return fail(c, NEO_ERR_INVALID, "n > 128 variables: this build has no two-waves kernel for three FLAT slots");
      break;
    default:
      if constexpr (WAVES == 1 || NEO_W2_MAX_SLOTS >= 4)
        NEO_OPT_LG(4, WaveLanes);  // (two waves only up to NEO_W2_MAX_SLOTS)
      else
        return fail(c, NEO_ERR_INVALID, "n > 128 variables: this build has no two-waves kernel for four FLAT slots "
                                        "(NEO_W2_MAX_SLOTS < 4)");
      break;
  }
#undef NEO_OPT_LG
#undef NEO_OPT
  return NEO_OK;
}

}  // namespace neo
