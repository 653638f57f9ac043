// neo_host.hpp -- host-side state shared by the translation units of libneo_planner_hip.so: the context behind
// the opaque neo_ctx of include/neo_planner.h, the argument packs of the kernel families and the per-family
// dispatch entry points (defined in neo_disp_*.hip, one family per translation unit so they build in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/neo_planner.h"
#include "neo_device.hpp"

namespace neo {
struct MapEntry {
  int kind = -1;  // 0 = 2-D reference map, 1 = 3-D field
  int elem = NEO_F64;
  void *data = nullptr;  // device
  Map2D m2{};
  Map3D m3{};
  int slot = -1;  // index into the device-side map table
};

struct ProfileSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  int64_t launches = 0;
  double ms = 0.0;
};

}  // namespace neo

struct neo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t home_stream = nullptr;  // the stream of neo_ctx_create (neo_ctx_set_stream(NULL) returns to it)
  neo_params params{};
  neo::DevParams dev{};
  std::map<int, neo::MapEntry> maps;
  std::string err;
  std::recursive_mutex mu;  // recursive: the host-pointer entry points hold it across their *_dev call
  int *tickets = nullptr;  // ring of work counters for optimize_group_kernel launches (one per launch in flight)
  unsigned ticket_next = 0;
  // device-side table of maps (rebuilt when a map changes)
  void *table2d = nullptr, *table3d = nullptr;
  int n2d = 0, n3d = 0;
  bool table_dirty = true;
  // scratch for the host-pointer entry points
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  // pinned mirror of the scratch layout of neo_optimize_batch: ONE copy in and ONE copy out per call instead of four
  // and six from pageable memory (each of those stages through the runtime and waits)
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  bool profile = false;
  neo::ProfileSlot prof[NEO_KERNEL_COUNT];
  long long *sample_counter = nullptr;  // optional device array [B] (neo_optimize_sample_counter)
  int *progress = nullptr;              // optional device-accessible counter of finished trajectories (neo_optimize_progress_counter)
  const int *dispatch_order = nullptr;  // optional device permutation [B] (neo_optimize_dispatch_order)
  int order_B = 0;                      // batch size the permutation was given for (ignored for any other B)
  double *trace = nullptr;              // optional device array [B][trace_cap][4] (neo_optimize_trace)
  double *trace_xg = nullptr;           // optional device array [B][trace_cap][2][n] (neo_optimize_trace_xg)
  int trace_cap = 0;
  int *order_buf = nullptr;             // device copy of a host permutation (neo_optimize_dispatch_order_host)
  size_t order_cap = 0;
  int *sample_order = nullptr;          // context-owned device copy of the ESDF-lookup kernel's permutation
  size_t sample_order_cap = 0;          // (neo_sampled_terms_dispatch_order; never the optimiser's, never caller-owned)
  int sample_order_B = 0;
  int edt_flags = 0;                    // NEO_EDT_* (neo_esdf_build_config)
};

namespace neo {

inline int fail(neo_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}

#define HIPCHK(c, call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(c, NEO_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));       \
  } while (0)

// device allocation released on every exit path unless release()d into a longer-lived owner
struct DevBuf {
  void *p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() {
    if (p) hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  void *release() {
    void *q = p;
    p = nullptr;
    return q;
  }
  template <typename T>
  T *as() const { return static_cast<T *>(p); }
};

struct ProfScope {
  neo_ctx *c;
  int k;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(neo_ctx *c_, int k_) : c(c_), k(k_) {
    if (c->profile) {
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, c->stream);
    }
  }
  ~ProfScope() {
    if (c->profile) {
      hipEventRecord(b, c->stream);
      c->prof[k].pending.emplace_back(a, b);
    }
  }
};

struct EvalArgs {
  int B, M;
  const double *x, *head, *tail;
  double *cost, *costs4, *grad, *coeffs;
  int *status;
};

struct OptArgs {
  int B, M;
  const void *table;
  const int *slots;  // device array [B] of map-table slots, or NULL (all trajectories use table[0])
  int nmaps;         // entries of `table` (slots are checked against it on the device)
  const double *x0;  // start points (NULL: read from x, the in-place form)
  double *x;
  const double *head, *tail;
  double *costs4, *costs4_last;
  int *nit, *nfev, *status;
  // budgeted launches (neo_optimize_batch_budget_dev); all zero otherwise
  double *state = nullptr;       // [B][state_doubles] optimiser state of suspended runs
  int state_doubles = 0;
  int budget = 0;                // evaluations per trajectory and launch
  int resume = 0;                // continue suspended runs instead of starting from x0
  const int *subset = nullptr;   // device array of n_subset trajectory indices: launch these only (workgroup i -> subset[i])
  int n_subset = 0;
  int traj_total = 0;            // B of the arrays (subset entries are checked against it on the device)
};

struct SampleArgs {
  int B, M;
  const void *coeffs;  // [B][6M][D] doubles, or floats when io32
  const double *ts;
  double *costs2;
  void *grad_C, *grad_T;  // doubles, or floats when io32
  bool io32 = false;      // neo_sampled_terms_batch_f32_dev (fp32 sampling only)
};

// FLAT slots of the optimiser vectors: n <= 64, 128, 192 or 256 variables
inline int slots_for(int M, int D) {
  const int n = D * (M - 1) + M;
  const int ns = (n + kWave - 1) / kWave;
  return ns <= 3 ? (ns < 1 ? 1 : ns) : 4;  // (3: cfg5's n = 161 -- a quarter fewer optimiser-vector instructions than 4 slots)
}

// ---- per-family dispatch (neo_disp_*.hip)
int dispatch_eval(neo_ctx *c, const MapEntry &e, int D, const EvalArgs &a);
int dispatch_sample(neo_ctx *c, const MapEntry &e, int D, const SampleArgs &a);
// the families behind dispatch_opt (neo_abi.hip)
int launch_opt_2d(neo_ctx *c, int D, bool f32, const OptArgs &a);           // neo_disp_opt2d.hip
int launch_opt_3d_f32(neo_ctx *c, int elem, int layout, const OptArgs &a);  // neo_disp_opt3d_f32.hip
int launch_opt_3d_f64(neo_ctx *c, int elem, int layout, const OptArgs &a);  // neo_disp_opt3d_f64.hip
int launch_opt_3d_w2(neo_ctx *c, int elem, int layout, const OptArgs &a);   // neo_disp_opt3d_w2.hip
int launch_opt_3d_x(neo_ctx *c, int elem, int layout, const OptArgs &a);    // neo_disp_opt3d_x.hip
int launch_opt_groups(neo_ctx *c, int elem, int layout, const OptArgs &a);  // neo_disp_group.hip
int launch_opt_3d_f64_w2(neo_ctx *c, int elem, int layout, const OptArgs &a);
int launch_opt_3d_budget(neo_ctx *c, int elem, int layout, const OptArgs &a);  // neo_disp_opt3d_b.hip
int launch_opt_2d_w2(neo_ctx *c, bool f32, const OptArgs &a);
int launch_opt_2d_x(neo_ctx *c, int D, const OptArgs &a);                   // neo_disp_opt2d_x.hip (all-fp32 mode)
int launch_opt_groups_2d(neo_ctx *c, bool f32, const OptArgs &a);           // neo_disp_group.hip (D = 2, nearest-cell map)

}  // namespace neo
