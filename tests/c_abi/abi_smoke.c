/* tests/c_abi/abi_smoke.c -- the C ABI of include/neo_planner.h driven from plain C99 (no Python, no ctypes):
 * builds the 2-D ESDF of a recorded reference scenario on the device (neo_esdf_build_2d, esdf.py:11-33), runs
 * plan_once for it (neo_optimize_batch, expert_planner.py:205-237) and checks iterations, evaluations and the
 * final control points against the values the reference recorded (tests/golden/g3_trace_plan_s0.npz, exported
 * to a flat binary by tests/test_gpu_c_abi.py).  Also: neo_esdf_query, neo_cost_grad_batch, error paths.
 *
 *   gcc -std=c99 -I include tests/c_abi/abi_smoke.c -L <libdir> -lneo_planner_hip -lm -o abi_smoke
 *   ./abi_smoke fixture.bin
 *
 * fixture.bin (little endian): int32 W, H, M, D, nit, nfev; double res, ox, oy; int8 occ[H*W];
 *   double head[3*D], tail[3*D], x0[n], x_final[n], final_cost; */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "neo_planner.h"

#define CHECK(call)                                                                         \
  do {                                                                                      \
    int rc_ = (call);                                                                       \
    if (rc_ != NEO_OK) {                                                                    \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, ctx ? neo_last_error(ctx) : "no ctx"); \
      return 2;                                                                             \
    }                                                                                       \
  } while (0)

static int rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char **argv) {
  neo_ctx *ctx = NULL;
  if (argc < 2) return 64;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 66;
  int32_t hdr[6];
  double geo[3];
  if (rd(f, hdr, sizeof hdr) || rd(f, geo, sizeof geo)) return 65;
  const int W = hdr[0], H = hdr[1], M = hdr[2], D = hdr[3], ref_nit = hdr[4], ref_nfev = hdr[5];
  const int n = D * (M - 1) + M;
  int8_t *occ = malloc((size_t)W * H);
  double *head = malloc(sizeof(double) * 3 * D), *tail = malloc(sizeof(double) * 3 * D);
  double *x = malloc(sizeof(double) * n), *xf = malloc(sizeof(double) * n), ref_cost;
  if (rd(f, occ, (size_t)W * H) || rd(f, head, sizeof(double) * 3 * D) || rd(f, tail, sizeof(double) * 3 * D) ||
      rd(f, x, sizeof(double) * n) || rd(f, xf, sizeof(double) * n) || rd(f, &ref_cost, sizeof ref_cost))
    return 65;
  fclose(f);

  if (neo_abi_version() != NEO_ABI_VERSION) return 3;
  CHECK(neo_ctx_create(0, NULL, &ctx));
  neo_params p;
  CHECK(neo_params_default(&p)); /* the ROS YAML values, fp64 sampling = the reference's arithmetic */
  CHECK(neo_params_set(ctx, &p));

  /* map: occupancy -> EDT -> gradient on the device, arrays back for a spot check */
  double *dist = malloc(sizeof(double) * W * H);
  CHECK(neo_esdf_build_2d(ctx, 7, occ, W, H, geo[0], geo[1], geo[2], dist, NULL, NULL));
  double pt[2] = {geo[1] + 10.05 * geo[0], geo[2] + 20.05 * geo[0]}, dq, gq[2];
  CHECK(neo_esdf_query(ctx, 7, 1, pt, &dq, gq));
  if (dq != dist[20 * W + 10]) {
    fprintf(stderr, "query %g != map %g\n", dq, dist[20 * W + 10]);
    return 4;
  }

  /* one evaluation, then the whole run */
  double cost, costs4[4], last4[4], *grad = malloc(sizeof(double) * n);
  int32_t st = -1, nit = -1, nfev = -1;
  CHECK(neo_cost_grad_batch(ctx, 7, 1, M, D, x, head, tail, &cost, costs4, grad, NULL, &st));
  if (st != 0 || !(cost > 0.0)) return 5;
  CHECK(neo_optimize_batch(ctx, 7, NULL, 1, M, D, x, head, tail, costs4, last4, &nit, &nfev, &st));
  double err = 0.0, ref = 0.0;
  for (int i = 0; i < D * (M - 1); ++i) {
    err = fmax(err, fabs(x[i] - xf[i]));
    ref = fmax(ref, fabs(xf[i]));
  }
  double fc = 0.0;
  for (int k = 0; k < 4; ++k) fc += last4[k] * p.weights[k];
  printf("nit %d (ref %d)  nfev %d (ref %d)  status %d  x rel err %.3e  final cost %.9g (ref %.9g)\n", nit, ref_nit, nfev,
         ref_nfev, st, err / ref, fc, ref_cost);
  if (nit != ref_nit || nfev != ref_nfev) return 6;
  if (err > 1e-4 * ref) return 7; /* north_star tolerance; in practice 1e-15 */
  if (fabs(fc - ref_cost) > 1e-4 * fabs(ref_cost)) return 8;

  /* error paths: unknown scene, bad shape */
  if (neo_optimize_batch(ctx, 99, NULL, 1, M, D, x, head, tail, costs4, last4, &nit, &nfev, &st) != NEO_ERR_NO_MAP) return 9;
  if (neo_cost_grad_batch(ctx, 7, 1, 0, D, x, head, tail, &cost, costs4, grad, NULL, &st) != NEO_ERR_INVALID) return 10;
  if (strlen(neo_last_error(ctx)) == 0) return 11;
  CHECK(neo_esdf_drop(ctx, 7));
  CHECK(neo_ctx_destroy(ctx));
  printf("C ABI ok\n");
  return 0;
}
