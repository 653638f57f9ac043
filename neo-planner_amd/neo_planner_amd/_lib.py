"""ctypes binding of include/neo_planner.h.  Loading fails loudly: there is no CPU fallback."""
import ctypes
import os
import sys

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# NEO_PLANNER_LIB: another build of the same library (kernel experiments in tools/)
LIB_PATH = os.environ.get("NEO_PLANNER_LIB") or os.path.join(PKG, "libneo_planner_hip.so")

NEO_OK = 0
NEO_F64, NEO_F32, NEO_F16 = 0, 1, 2
NEO_LAYOUT_LINEAR, NEO_LAYOUT_YZ4, NEO_LAYOUT_CELL8, NEO_LAYOUT_BRICK = 0, 1, 2, 3
LAYOUTS = {"linear": NEO_LAYOUT_LINEAR, "yz4": NEO_LAYOUT_YZ4, "cell8": NEO_LAYOUT_CELL8, "brick": NEO_LAYOUT_BRICK}
NEO_TRAJ_CONVERGED_GRAD, NEO_TRAJ_CONVERGED_F, NEO_TRAJ_ABNORMAL = 0, 1, 2
NEO_TRAJ_MAXITER, NEO_TRAJ_NUMERIC_RANGE, NEO_TRAJ_NONFINITE, NEO_TRAJ_BAD_SCENE, NEO_TRAJ_SUSPENDED = 3, 4, 5, 6, 7
NEO_TRAJ_FLAG_COLLISION = 0x100
NEO_EDT_GENERIC_LINES = 1
NEO_KERNEL_EVAL, NEO_KERNEL_OPTIMIZE, NEO_KERNEL_ESDF_BUILD, NEO_KERNEL_ESDF_SAMPLE = 0, 1, 2, 3
NEO_FLAG_ONE_WAVE_PER_SIMD, NEO_FLAG_TWO_WAVES_PER_SIMD, NEO_FLAG_LANE_GROUPS = 32, 64, 128
NEO_FLAG_F32_SOLVE = 2048

# every symbol include/neo_planner.h declares (tests check the library exports them all)
EXPORTS = [
    "neo_abi_version", "neo_ctx_create", "neo_ctx_destroy", "neo_last_error", "neo_params_default",
    "neo_params_set", "neo_ctx_synchronize", "neo_esdf_upload_2d", "neo_esdf_build_2d",
    "neo_esdf_upload_3d", "neo_esdf_drop", "neo_esdf_query", "neo_cost_grad_batch",
    "neo_cost_grad_batch_dev", "neo_optimize_batch", "neo_optimize_batch_dev", "neo_scene_slot",
    "neo_optimize_workspace_bytes", "neo_eval_traj_batch", "neo_profile_enable", "neo_profile_read",
    "neo_profile_reset", "neo_optimize_sample_counter", "neo_optimize_dispatch_order",
    "neo_sampled_terms_batch", "neo_sampled_terms_batch_dev", "neo_esdf_build_3d",
    "neo_optimize_dispatch_order_host", "neo_ctx_set_stream", "neo_optimize_trace",
    "neo_optimize_batch_from_dev", "neo_optimize_trace_xg", "neo_sampled_terms_dispatch_order",
    "neo_esdf_build_config", "neo_pack_results_dev", "neo_optimize_state_bytes", "neo_optimize_batch_budget_dev",
    "neo_sampled_terms_batch_f32", "neo_sampled_terms_batch_f32_dev", "neo_effort_order_dev",
    "neo_optimize_progress_counter", "neo_effort_order_scratch_bytes",
]


class NeoParams(ctypes.Structure):
    _fields_ = [("v_max", ctypes.c_double), ("T_min", ctypes.c_double), ("T_max", ctypes.c_double),
                ("safe_dis", ctypes.c_double), ("delta_t", ctypes.c_double), ("weights", ctypes.c_double * 4),
                ("collision_cost_tol", ctypes.c_double), ("ftol", ctypes.c_double), ("gtol", ctypes.c_double),
                ("maxls", ctypes.c_int32), ("maxiter", ctypes.c_int32), ("maxfun", ctypes.c_int32),
                ("bugcompat_stale_T", ctypes.c_int32), ("sample_dtype", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


_lib = None


class NeoError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (no compute).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NeoError(f"{LIB_PATH} is missing: build it with `python -m neo_planner_amd.build` "
                       "(hipcc, gfx950).  There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP runtime under the same soname as /opt/rocm's.  Whichever is loaded
    # first serves the whole process; torch does not find its GPUs on the system one ("No HIP GPUs are
    # available"), while this library runs on either.  So: torch first, if it is installed at all.
    if "torch" not in sys.modules and not os.environ.get("NEO_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(LIB_PATH)
    c_p, c_i, c_d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.neo_abi_version.restype = c_i
    L.neo_ctx_create.argtypes = [c_i, c_p, ctypes.POINTER(c_p)]
    L.neo_ctx_destroy.argtypes = [c_p]
    L.neo_last_error.argtypes = [c_p]
    L.neo_last_error.restype = ctypes.c_char_p
    L.neo_params_default.argtypes = [ctypes.POINTER(NeoParams)]
    L.neo_params_set.argtypes = [c_p, ctypes.POINTER(NeoParams)]
    L.neo_ctx_synchronize.argtypes = [c_p]
    L.neo_ctx_set_stream.argtypes = [c_p, c_p]
    L.neo_esdf_upload_2d.argtypes = [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_d]
    L.neo_esdf_build_2d.argtypes = [c_p, c_i, c_p, c_i, c_i, c_d, c_d, c_d, c_p, c_p, c_p]
    L.neo_esdf_upload_3d.argtypes = [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_d, c_p, c_i, c_i]
    L.neo_esdf_build_3d.argtypes = [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_d, c_p, c_i, c_i, c_p]
    L.neo_esdf_drop.argtypes = [c_p, c_i]
    L.neo_esdf_query.argtypes = [c_p, c_i, c_i, c_p, c_p, c_p]
    L.neo_cost_grad_batch.argtypes = [c_p, c_i, c_i, c_i, c_i] + [c_p] * 8
    L.neo_cost_grad_batch_dev.argtypes = [c_p, c_i, c_i, c_i, c_i] + [c_p] * 8
    L.neo_optimize_batch.argtypes = [c_p, c_i, c_p, c_i, c_i, c_i] + [c_p] * 8
    L.neo_optimize_batch_dev.argtypes = [c_p, c_i, c_p, c_i, c_i, c_i] + [c_p] * 8
    L.neo_optimize_batch_from_dev.argtypes = [c_p, c_i, c_p, c_i, c_i, c_i] + [c_p] * 9
    L.neo_scene_slot.argtypes = [c_p, c_i]
    L.neo_optimize_workspace_bytes.argtypes = [c_i, c_i, c_i]
    L.neo_optimize_workspace_bytes.restype = ctypes.c_size_t
    L.neo_eval_traj_batch.argtypes = [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_d, c_i, c_p, c_p]
    L.neo_profile_enable.argtypes = [c_p, c_i]
    L.neo_profile_read.argtypes = [c_p, c_i, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(c_d)]
    L.neo_profile_reset.argtypes = [c_p]
    L.neo_optimize_sample_counter.argtypes = [c_p, c_p]
    L.neo_optimize_trace.argtypes = [c_p, c_p, c_i]
    L.neo_optimize_trace_xg.argtypes = [c_p, c_p, c_i]
    L.neo_optimize_dispatch_order.argtypes = [c_p, c_p, c_i]
    L.neo_optimize_dispatch_order_host.argtypes = [c_p, c_p, c_i]
    L.neo_sampled_terms_dispatch_order.argtypes = [c_p, c_p, c_i, c_i]
    L.neo_esdf_build_config.argtypes = [c_p, c_i]
    L.neo_pack_results_dev.argtypes = [c_p, c_i, c_i, c_p, c_p, c_p, c_p]
    L.neo_optimize_state_bytes.argtypes = [c_i, c_i]
    L.neo_optimize_state_bytes.restype = ctypes.c_size_t
    L.neo_optimize_batch_budget_dev.argtypes = [c_p, c_i, c_i, c_i, c_i] + [c_p] * 10 + [c_i, c_p, c_i, c_i]
    L.neo_sampled_terms_batch.argtypes = [c_p, c_i, c_i, c_i, c_i] + [c_p] * 5
    L.neo_sampled_terms_batch_dev.argtypes = [c_p, c_i, c_i, c_i, c_i] + [c_p] * 5
    L.neo_sampled_terms_batch_f32.argtypes = [c_p, c_i, c_i, c_i, c_i] + [c_p] * 5
    L.neo_sampled_terms_batch_f32_dev.argtypes = [c_p, c_i, c_i, c_i, c_i] + [c_p] * 5
    L.neo_effort_order_dev.argtypes = [c_p, c_i, c_i, c_i] + [c_p] * 5
    L.neo_optimize_progress_counter.argtypes = [c_p, c_p]
    L.neo_effort_order_scratch_bytes.argtypes = [c_i]
    L.neo_effort_order_scratch_bytes.restype = ctypes.c_size_t
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int or name in ("neo_abi_version",):
            fn.restype = c_i
    _lib = L
    return L


def ptr(a):
    """host pointer of a C-contiguous NumPy array (None -> NULL)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """one neo_ctx: a device, a HIP stream, the uploaded maps"""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = ctypes.c_void_p()
        self.device = int(device)
        rc = self.lib.neo_ctx_create(int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
        if rc != NEO_OK:
            raise NeoError(f"neo_ctx_create(device={device}) failed with {rc}: no usable MI355X? "
                           "(the product path has no CPU fallback)")
        self.h = h
        self.params = NeoParams()
        self.lib.neo_params_default(ctypes.byref(self.params))
        self._next_scene = 0

    def check(self, rc):
        if rc != NEO_OK:
            raise NeoError(f"libneo_planner_hip error {rc}: {self.lib.neo_last_error(self.h).decode()}")

    def set_params(self, **kw):
        for k, v in kw.items():
            if k == "weights":
                for i in range(4):
                    self.params.weights[i] = float(v[i])
            else:
                setattr(self.params, k, v)
        self.check(self.lib.neo_params_set(self.h, ctypes.byref(self.params)))

    def new_scene_id(self):
        self._next_scene += 1
        return self._next_scene

    def synchronize(self):
        self.check(self.lib.neo_ctx_synchronize(self.h))

    def set_stream(self, stream=None):
        """HIP stream handle for the calls that follow (None: the stream the context was created with)"""
        self.check(self.lib.neo_ctx_set_stream(self.h, ctypes.c_void_p(stream) if stream else None))

    def close(self):
        if self.h:
            self.lib.neo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
