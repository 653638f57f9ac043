"""
oracle/cpu_native -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

ctypes front end of minco_cpu.cpp, the C++ fp64 restatement of the reference's replan inner loop in the
reference's own formulation (banded 6M x 6M solve, per-sample loops; see that file's header for the
reference file:line of every function).  Three uses:

  * `NativePlanner`: cost / gradient in C++, optimiser = SciPy's own L-BFGS-B with the reference's options
    (expert_planner.py:213-225) -- a ~50x faster stand-in for oracle/minco_np.py in full-size parity tests;
  * `optimize_batch`: plan_once for a batch on host threads, optimiser = the restated L-BFGS-B control flow of
    csrc/neo_lbfgs.hpp -- bench.py's `cpu_native` ("fair CPU") figure;
  * the parity CONTROL (`sample_f32`, `coeff_eps`): the same runs with the sampled terms in fp32 arithmetic or the
    coefficients perturbed by an ulp, to measure how far two faithful implementations of this discontinuous
    objective drift apart (bench.py `parity_control`, tests/test_cpu_native.py).

Parity status: PINNED (tests/test_cpu_native.py: G1 per-evaluation fixtures and G3 runs captured from the real
reference; agreement with oracle/minco_np.py).
Only tests/, __graft_entry__.py and bench.py's CPU legs may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "minco_cpu.cpp")
LIB = os.path.join(HERE, "libminco_cpu.so")
CSRC = os.path.join(REPO, "neo-planner_amd", "csrc")


class Params(ctypes.Structure):
    _fields_ = [("v_max", ctypes.c_double), ("T_min", ctypes.c_double), ("T_max", ctypes.c_double),
                ("safe_dis", ctypes.c_double), ("delta_t", ctypes.c_double), ("w", ctypes.c_double * 4),
                ("coll_tol", ctypes.c_double), ("ftol", ctypes.c_double), ("gtol", ctypes.c_double),
                ("maxls", ctypes.c_int32), ("maxiter", ctypes.c_int32), ("maxfun", ctypes.c_int32),
                ("stale_T", ctypes.c_int32), ("sample_f32", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("coeff_eps", ctypes.c_double), ("grad_eps", ctypes.c_double)]


class Map(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("W", ctypes.c_int32), ("H", ctypes.c_int32),
                ("dist", ctypes.c_void_p), ("gx", ctypes.c_void_p), ("gy", ctypes.c_void_p),
                ("res", ctypes.c_double), ("ox", ctypes.c_double), ("oy", ctypes.c_double), ("oz", ctypes.c_double),
                ("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("field", ctypes.c_void_p)]


def build(force=False):
    """g++ -> oracle/cpu_native/libminco_cpu.so (git-ignored; travels to the GPU box with the snapshot)"""
    deps = [SRC, os.path.join(CSRC, "neo_lbfgs.hpp"), os.path.join(CSRC, "neo_linesearch.hpp")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-ffp-contract=off", "-I", CSRC, SRC, "-o",
           LIB + ".tmp"]
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        c_p, c_i, c_d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.mc_create.restype = c_p
        L.mc_create.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(Map), c_i, c_i, c_p, c_p]
        L.mc_destroy.argtypes = [c_p]
        L.mc_cost.argtypes = [c_p, c_p, c_p, c_p]
        L.mc_grad.argtypes = [c_p, c_p, c_p, c_p, c_p, c_p, c_p]
        L.mc_optimize_batch.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(Map), c_i, c_i, c_i] + [c_p] * 8 + \
                                       [c_i, c_d, c_p, c_p, c_i]
        L.mc_eval_points.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(Map), c_i, c_i, c_p, c_p, c_i] + [c_p] * 5
        _lib = L
    return _lib


def make_params(cfg=None, stale_T=True, sample_f32=False, coeff_eps=0.0, grad_eps=0.0):
    """cfg: anything with the PlannerConfig attribute names (oracle.minco_np.PlannerParams by default)"""
    if cfg is None:
        from oracle import minco_np
        cfg = minco_np.PlannerParams()
    p = Params()
    p.v_max, p.T_min, p.T_max, p.safe_dis, p.delta_t = cfg.v_max, cfg.T_min, cfg.T_max, cfg.safe_dis, cfg.delta_t
    for k in range(4):
        p.w[k] = float(cfg.weights[k])
    p.coll_tol = float(cfg.collision_cost_tol)
    p.ftol = p.gtol = 1e-4                              # tol=1e-4 (expert_planner.py:218)
    p.maxls, p.maxiter, p.maxfun = 20, 15000, 15000     # (:221-224)
    p.stale_T = int(bool(stale_T))
    p.sample_f32 = int(bool(sample_f32))
    p.coeff_eps = float(coeff_eps)
    p.grad_eps = float(grad_eps)
    return p


class NativeMap:
    """keeps the arrays alive next to the C struct"""

    def __init__(self, m, keep):
        self.c = m
        self._keep = keep

    @classmethod
    def from_grid2d(cls, g):
        """g: oracle.minco_np.GridESDF (or anything with its attributes)"""
        d = np.ascontiguousarray(g.esdf_map, dtype=np.float64)
        gx = np.ascontiguousarray(g.esdf_grad_x, dtype=np.float64)
        gy = np.ascontiguousarray(g.esdf_grad_y, dtype=np.float64)
        m = Map()
        m.kind, m.W, m.H = 0, int(g.map_width), int(g.map_height)
        m.dist, m.gx, m.gy = d.ctypes.data, gx.ctypes.data, gy.ctypes.data
        m.res, m.ox, m.oy = float(g.map_resolution), float(g.origin_x), float(g.origin_y)
        return cls(m, (d, gx, gy))

    @classmethod
    def from_field3d(cls, dist, res, origin):
        """dist [nz][ny][nx] (stored as float32, the element type of the HIP path's fp32 fields; an fp16 field is
        passed as its values widened to float32)"""
        f = np.ascontiguousarray(dist, dtype=np.float32)
        m = Map()
        m.kind = 1
        m.nz, m.ny, m.nx = f.shape
        m.res, m.ox, m.oy, m.oz = float(res), float(origin[0]), float(origin[1]), float(origin[2])
        m.field = f.ctypes.data
        return cls(m, (f,))


class NativePlanner:
    """MinJerkPlanner-shaped object on the C++ evaluation (names as oracle.minco_np.OraclePlanner)"""

    def __init__(self, config=None, stale_T=True, sample_f32=False, coeff_eps=0.0):
        from oracle import minco_np
        self.cfg = config if config is not None else minco_np.PlannerParams()
        self.params = make_params(self.cfg, stale_T, sample_f32, coeff_eps)
        self.weights = np.array(self.cfg.weights, dtype=np.float64)
        self.collision_cost_tol = self.cfg.collision_cost_tol
        self.T_min, self.T_max = self.cfg.T_min, self.cfg.T_max
        self.h = None
        self.last_result = None
        self.iter_num = 0
        self.L = load()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mc_destroy(self.h)
            self.h = None

    def read_planning_conditions(self, map, head_state, tail_state, int_wpts, ts):
        """map: NativeMap"""
        self.D = head_state.shape[1]
        self.M = ts.shape[0]
        hs = np.zeros((3, self.D))
        tl = np.zeros((3, self.D))
        hs[:min(3, head_state.shape[0])] = head_state[:3]
        tl[:min(3, tail_state.shape[0])] = tail_state[:3]
        self.head_state, self.tail_state = hs, tl
        self.int_wpts, self.ts = np.asarray(int_wpts, dtype=np.float64), np.asarray(ts, dtype=np.float64)
        self.map = map
        if self.h:
            self.L.mc_destroy(self.h)
        self.h = self.L.mc_create(ctypes.byref(self.params), ctypes.byref(map.c), self.M, self.D, hs.ctypes.data,
                                  tl.ctypes.data)
        if not self.h:
            raise ValueError("bad problem shape")
        self.n = self.D * (self.M - 1) + self.M
        self.costs = np.zeros(4)
        self.coeffs = np.zeros((6 * self.M, self.D))
        self.grad_C = np.zeros((6 * self.M, self.D))
        self.grad_T = np.zeros(self.M)

    def get_cost(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = ctypes.c_double()
        st = self.L.mc_cost(self.h, x.ctypes.data, ctypes.addressof(f), self.costs.ctypes.data)
        if st:
            raise OverflowError("math range error")
        return f.value

    def get_grad(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros(self.n)
        tsb = np.zeros(self.M)
        st = self.L.mc_grad(self.h, x.ctypes.data, g.ctypes.data, self.coeffs.ctypes.data, self.grad_C.ctypes.data,
                            self.grad_T.ctypes.data, tsb.ctypes.data)
        if st:
            raise OverflowError("math range error")
        self.ts_eval = tsb
        return g

    def map_T2tau(self, ts):
        return np.array([-np.log((self.T_max - self.T_min) / (t - self.T_min) - 1) for t in ts])

    def map_tau2T(self, tau):
        return np.array([(self.T_max - self.T_min) / (1 + np.exp(-t)) + self.T_min for t in tau])

    def plan_once(self, trace=None):
        """expert_planner.py:205-237 with SciPy's own L-BFGS-B"""
        from scipy import optimize as sciopt
        nq = self.D * (self.M - 1)
        self.tau = self.map_T2tau(self.ts)
        x0 = np.concatenate((np.reshape(self.int_wpts, (nq,)), self.tau))
        cb = None
        if trace is not None:
            def cb(intermediate_result):
                trace.append((np.array(intermediate_result.x), float(intermediate_result.fun)))
        res = sciopt.minimize(self.get_cost, x0, method='L-BFGS-B', jac=self.get_grad, bounds=None, tol=1e-4,
                              callback=cb, options={'maxcor': 10, 'maxfun': 15000, 'maxiter': 15000, 'maxls': 20})
        self.last_result = res
        self.int_wpts = np.reshape(res.x[:nq], (self.D, self.M - 1))
        self.tau = res.x[nq:]
        self.ts = self.map_tau2T(self.tau)
        self.iter_num += res.nit
        self.weighted_cost = self.costs * self.weights      # costs of the LAST evaluated x (:233)
        self.final_cost = self.weighted_cost.sum()
        if self.weighted_cost[3] > self.collision_cost_tol:
            raise ValueError("collision cost too large")


def eval_points(map, x, head, tail, M, D, params=None):
    """cost / cost terms / gradient at the points x [E][n] of one trajectory (head, tail [3][D]); status 4 where the
    reference raises OverflowError.  Returns dict(f [E], costs [E][4], grad [E][n], status [E])."""
    L = load()
    p = params if params is not None else make_params()
    x = np.ascontiguousarray(x, dtype=np.float64)
    E, n = x.shape
    head = np.ascontiguousarray(head, dtype=np.float64)
    tail = np.ascontiguousarray(tail, dtype=np.float64)
    f = np.zeros(E); c4 = np.zeros((E, 4)); g = np.zeros((E, n)); st = np.zeros(E, dtype=np.int32)
    if L.mc_eval_points(ctypes.byref(p), ctypes.byref(map.c), M, D, head.ctypes.data, tail.ctypes.data, E, x.ctypes.data,
                        f.ctypes.data, c4.ctypes.data, g.ctypes.data, st.ctypes.data) != 0:
        raise ValueError("bad arguments")
    return dict(f=f, costs=c4, grad=g, status=st)


def optimize_batch(map, x0, head, tail, M, D, params=None, threads=1, limit_s=0.0, trace_cap=0):
    """plan_once for every row of x0 [B][n] on `threads` host threads (optimiser: csrc/neo_lbfgs.hpp).
    Returns dict(x, costs, costs_last, nit, nfev, status, done, finished[, trace]); trace [B][trace_cap][4] =
    (f, step, samples, iteration) per counted evaluation."""
    L = load()
    p = params if params is not None else make_params()
    x = np.array(x0, dtype=np.float64, order="C")
    B = x.shape[0]
    head = np.ascontiguousarray(head, dtype=np.float64)
    tail = np.ascontiguousarray(tail, dtype=np.float64)
    c4 = np.zeros((B, 4))
    c4l = np.zeros((B, 4))
    nit = np.zeros(B, dtype=np.int32)
    nfev = np.zeros(B, dtype=np.int32)
    st = np.zeros(B, dtype=np.int32)
    done = np.zeros(B, dtype=np.uint8)
    trace = np.zeros((B, trace_cap, 4)) if trace_cap > 0 else None
    fin = L.mc_optimize_batch(ctypes.byref(p), ctypes.byref(map.c), B, M, D, x.ctypes.data, head.ctypes.data,
                              tail.ctypes.data, c4.ctypes.data, c4l.ctypes.data, nit.ctypes.data, nfev.ctypes.data,
                              st.ctypes.data, int(threads), float(limit_s), done.ctypes.data,
                              trace.ctypes.data if trace is not None else None, int(trace_cap))
    if fin < 0:
        raise ValueError("bad arguments")
    out = dict(x=x, costs=c4, costs_last=c4l, nit=nit, nfev=nfev, status=st, done=done.astype(bool), finished=fin)
    if trace is not None:
        out["trace"] = trace
    return out
