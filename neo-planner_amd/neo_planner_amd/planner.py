"""
Host side of the MI355X planner: the reference's Python interface on top of the C ABI.

`MinJerkPlanner` keeps the names, argument meaning, side effects and exceptions of
traj_planner/expert_planner.py:28-585 (+ traj_utils.py evaluation helpers), so that
ros_node/traj_planner_node.py-style callers can switch without edits:

    planner = MinJerkPlanner(config)            # PlannerConfig-like object
    planner.plan(map, head_state, tail_state)   # or warm_start_plan / batch_plan / plan_once
    planner.int_wpts, planner.ts, planner.iter_num, planner.get_full_state_cmd(hz) ...

All arithmetic of the replan loop (coefficient solve, cost, gradient, L-BFGS-B) runs on the
GPU through libneo_planner_hip.so; this file only packs arguments, keeps the reference's
retry/exception control flow, and unpacks results.  `BatchPlanner` is the batched entry the
reference does not have: B independent replans in one launch.
"""
import ctypes
import math

import numpy as np

from . import _lib
from .esdf import ESDF, ESDF3D


class PlannerConfig:
    """parameter bag with the attribute names MinJerkPlanner reads (expert_planner.py:33-56,
    ros_node/traj_planner_node.py:32-46); defaults = launch/config/planner_config.yaml:2-13."""

    def __init__(self, **kw):
        self.v_max = 1.0
        self.T_min = 0.5
        self.T_max = 5.0
        self.safe_dis = 0.7
        self.delta_t = 0.1
        self.weights = [1.0, 1.0, 1.0, 10000.0]
        self.init_wpts_mode = 'fixed'
        self.init_seg_len = 2.0
        self.init_wpts_num = 2
        self.init_T = 2.5
        self.collision_cost_tol = 5
        self.opt_tol = 1e-2
        self.des_pos_z = 2.0
        for k, v in kw.items():
            setattr(self, k, v)


def _push_params(ctx, p, sample_dtype, stale_T=True, flags=0):
    ctx.set_params(v_max=float(p.v_max), T_min=float(p.T_min), T_max=float(p.T_max), safe_dis=float(p.safe_dis),
                   delta_t=float(p.delta_t), weights=[float(w) for w in p.weights],
                   collision_cost_tol=float(p.collision_cost_tol), ftol=1e-4, gtol=1e-4, maxls=20,
                   maxiter=15000, maxfun=15000, bugcompat_stale_T=int(bool(stale_T)),
                   sample_dtype={"f64": _lib.NEO_F64, "f32": _lib.NEO_F32}[sample_dtype], flags=int(flags))


def _map_scene(ctx, map, cache):
    """scene id of `map` on the device.  Our own ESDF classes are already resident; a foreign
    object with the reference's attribute protocol (esdf.py:17-33) is snapshotted."""
    if isinstance(map, (ESDF, ESDF3D)):
        if map.ctx is not ctx:
            raise ValueError("map and planner live on different contexts")
        return map.scene_id
    snap = ESDF.from_arrays(map.esdf_map, map.esdf_grad_x, map.esdf_grad_y, map.map_resolution,
                            (map.map_origin.x, map.map_origin.y), ctx=ctx)
    old = cache.get("foreign")
    if old is not None:
        ctx.lib.neo_esdf_drop(ctx.h, old.scene_id)
    cache["foreign"] = snap
    return snap.scene_id


class MinJerkPlanner:
    """MI355X-backed stand-in for expert_planner.py:MinJerkPlanner"""

    def __init__(self, config=None, ctx=None, sample_dtype="f64", stale_T=True):
        config = config if config is not None else PlannerConfig()
        self._ctx = ctx             # created on first device use: host-only helpers work without a GPU
        self.s = 3
        self.v_max = config.v_max
        self.T_min = config.T_min
        self.T_max = config.T_max
        self.safe_dis = config.safe_dis
        self.collision_cost_tol = config.collision_cost_tol
        self.weights = np.array(config.weights, dtype=np.float64)
        self.delta_t = config.delta_t
        self.opt_tol = config.opt_tol
        self.init_wpts_mode = config.init_wpts_mode
        self.init_seg_len = config.init_seg_len
        self.init_wpts_num = int(config.init_wpts_num)
        self.init_T = config.init_T
        self.batch_num = 3
        self.iter_num = 0
        self.opt_running_times = 0
        self.coeffs = []
        # "f32x": everything in fp32 (NEO_FLAG_F32_SOLVE) -- the arithmetic bench.py times, here behind the reference's own
        # interface so that the recorded reference runs (tests/golden g3 / g6) can be replayed in it
        self.flags = _lib.NEO_FLAG_F32_SOLVE if sample_dtype == "f32x" else 0
        self.sample_dtype = "f32" if sample_dtype == "f32x" else sample_dtype
        self.stale_T = stale_T
        self._cache = {}
        self._scene = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    # ------------------------------------------------------------ initial guesses (:82-140)
    def generate_init_variables(self, head_state, tail_state, seed=0):
        start, target = head_state[0], tail_state[0]
        if self.init_wpts_mode == 'adaptive':
            dist = np.linalg.norm(target - start)
            count = max(math.ceil(dist / self.init_seg_len - 1), 1)
        elif self.init_wpts_mode == 'fixed':
            count = self.init_wpts_num
        stride = (target - start) / (count + 1)
        wpts = np.linspace(start + stride, target, count, endpoint=False)
        if seed != 0:
            wpts += np.random.normal(0, 0.5, wpts.shape)   # the reference's unseeded global RNG (:94)
        ts = self.init_T * np.ones((count + 1,))
        ts[0] *= 1.5
        ts[-1] *= 1.5
        return wpts.T, ts

    def batch_generate_init_variables(self, head_state, tail_state):
        if self.init_wpts_mode != 'fixed':
            print("Error! init_wpts_mode must be 'fixed'")
        start, target = head_state[0], tail_state[0]
        forward = (target - start) / np.linalg.norm(target - start)
        sideways = np.array([[forward[1], -forward[0]], [-forward[1], forward[0]]])
        count = self.init_wpts_num
        out = np.zeros((self.batch_num, count, head_state.shape[1]))
        stride = (target - start) / (count + 1)
        out[0] = np.linspace(start + stride, target, count, endpoint=False)
        which = 0
        for i in range(1, self.batch_num):
            out[i] = out[0] + 0.6 * sideways[which]
            which = 1 - which
        ts = self.init_T * np.ones((count + 1,))
        ts[0] *= 1.5
        ts[-1] *= 1.5
        return np.transpose(out, (0, 2, 1)), ts

    # ------------------------------------------------------------ entry points (:62-80, :142-237)
    def read_planning_conditions(self, map, head_state, tail_state, int_wpts, ts, _resnapshot=True):
        self.map = map
        self.D = head_state.shape[1]
        self.M = ts.shape[0]
        self.head_state = np.zeros((self.s, self.D))
        self.tail_state = np.zeros((self.s, self.D))
        for i in range(min(self.s, head_state.shape[0])):
            self.head_state[i] = head_state[i]
        for i in range(min(self.s, tail_state.shape[0])):
            self.tail_state[i] = tail_state[i]
        self.int_wpts = int_wpts
        self.ts = ts
        # snapshot the map now: the reference reads it unlocked while a subscriber thread may
        # be rewriting it
        if _resnapshot or self._scene is None:
            self._scene = _map_scene(self.ctx, map, self._cache)

    def plan(self, map, head_state, tail_state):
        int_wpts, ts = self.generate_init_variables(head_state, tail_state)
        self.warm_start_plan(map, head_state, tail_state, int_wpts, ts)

    def warm_start_plan(self, map, head_state, tail_state, int_wpts, ts):
        self.read_planning_conditions(map, head_state, tail_state, int_wpts, ts)
        seed = 0
        while seed < 5:
            try:
                self.plan_once()
                return
            except Exception as ex:
                print(f"Re-planning for {ex}, current seed: {seed}")
                seed += 1
                self.int_wpts, self.ts = self.generate_init_variables(head_state, tail_state, seed)
        raise Exception("No solution for the given target")

    def batch_plan(self, map, head_state, tail_state):
        """expert_planner.py:142-168.  The three lateral candidates are optimised in ONE launch of three trajectories
        (a trajectory's result does not depend on its batch neighbours: bit-identical to three launches of one); the
        reference's loop -- including its success check INSIDE the loop (:160-168) and the order of its side effects on
        iter_num / opt_running_times / int_wpts -- then runs on the host over the three results."""
        cands, ts = self.batch_generate_init_variables(head_state, tail_state)
        best_wpts = np.zeros(cands.shape)
        best_ts = np.zeros((self.batch_num, len(ts)))
        cost = np.zeros(self.batch_num)
        results = None
        for i in range(self.batch_num):
            try:
                self.read_planning_conditions(map, head_state, tail_state, cands[i], ts, _resnapshot=(i == 0))
                if results is None:
                    results = self._launch_plan_once([cands[k] for k in range(self.batch_num)], ts)
                self._finish_plan_once(*[r[i:i + 1] for r in results])
                best_wpts[i] = self.int_wpts
                best_ts[i] = self.ts
                cost[i] = self.weighted_cost.sum()
                print(f"batch_cost[{i}] = {cost[i]}")
            except Exception as ex:
                print(f"The {i}th attempt is deprecated for {ex}")
                cost[i] = np.inf
            # as in the reference the check sits inside the loop (:160-168)
            if np.min(cost) < np.inf:
                k = np.argmin(cost)
                self.int_wpts = best_wpts[k]
                self.ts = best_ts[k]
                self.final_cost = cost[k]
            else:
                print("All attempts are infeasible! Start re-planning from scratch...")
                self.warm_start_plan(map, head_state, tail_state, cands[0], ts)

    def _pack_x(self):
        nq = self.D * (self.M - 1)
        return np.concatenate((np.reshape(self.int_wpts, (nq,)), self.tau), axis=0)

    def _unpack_x(self, x):
        nq = self.D * (self.M - 1)
        self.int_wpts = np.reshape(x[:nq], (self.D, self.M - 1))
        self.tau = x[nq:]
        self.ts = self.map_tau2T(self.tau)

    def _sync_params(self):
        cfg = PlannerConfig(v_max=self.v_max, T_min=self.T_min, T_max=self.T_max, safe_dis=self.safe_dis,
                            delta_t=self.delta_t, weights=self.weights,
                            collision_cost_tol=self.collision_cost_tol)
        _push_params(self.ctx, cfg, self.sample_dtype, self.stale_T, self.flags)

    def _launch_plan_once(self, wpts_list, ts):
        """one launch of len(wpts_list) trajectories that share head / tail / durations: (x, costs, last, nit, nfev, st)"""
        nb = len(wpts_list)
        nq = self.D * (self.M - 1)
        tau = self.map_T2tau(ts)
        x = np.stack([np.concatenate((np.reshape(_lib.as_f64(w), (nq,)), tau), axis=0) for w in wpts_list])
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._sync_params()
        c = self.ctx
        costs = np.zeros((nb, 4)); last = np.zeros((nb, 4))
        nit = np.zeros(nb, np.int32); nfev = np.zeros(nb, np.int32); st = np.zeros(nb, np.int32)
        head = np.ascontiguousarray(np.broadcast_to(_lib.as_f64(self.head_state), (nb, 3, self.D)))
        tail = np.ascontiguousarray(np.broadcast_to(_lib.as_f64(self.tail_state), (nb, 3, self.D)))
        c.check(c.lib.neo_optimize_batch(c.h, self._scene, None, nb, self.M, self.D, _lib.ptr(x), _lib.ptr(head),
                                         _lib.ptr(tail), _lib.ptr(costs), _lib.ptr(last), _lib.ptr(nit),
                                         _lib.ptr(nfev), _lib.ptr(st)))
        return x, costs, last, nit, nfev, st

    def _finish_plan_once(self, x, costs, last, nit, nfev, st):
        """what plan_once does with the optimiser's answer (:226-237): exceptions first, then the side effects"""
        self.tau = self.map_T2tau(self.ts)
        code = int(st[0]) & 0xff
        self.last_status, self.last_nit, self.last_nfev = code, int(nit[0]), int(nfev[0])
        if code == _lib.NEO_TRAJ_NUMERIC_RANGE:
            raise OverflowError("math range error")          # what math.exp raises at :481
        if code == _lib.NEO_TRAJ_NONFINITE:
            # the reference leaves minimize() through Python-float overflow in the same situation
            # (e.g. (jerk)**2 at :382 once a line-search trial point blows the coefficients up)
            raise OverflowError(34, "Numerical result out of range")
        self._unpack_x(x[0])
        self.iter_num += int(nit[0])
        self.opt_running_times += 1
        self.costs = last[0].copy()                           # costs of the last evaluated x (:233)
        self.costs_at_x = costs[0].copy()
        self.weighted_cost = self.costs * self.weights
        self.final_cost = self.weighted_cost.sum()
        if self.weighted_cost[3] > self.collision_cost_tol:
            raise ValueError("collision cost too large")

    def plan_once(self):
        """expert_planner.py:205-237 -- the L-BFGS-B run happens in one kernel launch"""
        self._finish_plan_once(*self._launch_plan_once([self.int_wpts], self.ts))

    optimize = plan_once

    # ------------------------------------------------------------ callbacks (:539-585)
    def _device_eval(self, x, want_coeffs=True):
        self._sync_params()
        c = self.ctx
        xx = _lib.as_f64(x).reshape(1, -1)
        n = xx.shape[1]
        cost = np.zeros(1); costs = np.zeros((1, 4)); grad = np.zeros((1, n))
        coeffs = np.zeros((1, 6 * self.M, self.D)); st = np.zeros(1, np.int32)
        head = _lib.as_f64(self.head_state).reshape(1, 3, self.D)
        tail = _lib.as_f64(self.tail_state).reshape(1, 3, self.D)
        c.check(c.lib.neo_cost_grad_batch(c.h, self._scene, 1, self.M, self.D, _lib.ptr(xx), _lib.ptr(head),
                                          _lib.ptr(tail), _lib.ptr(cost), _lib.ptr(costs), _lib.ptr(grad),
                                          _lib.ptr(coeffs), _lib.ptr(st)))
        if int(st[0]) == _lib.NEO_TRAJ_NUMERIC_RANGE:
            raise OverflowError("math range error")
        return cost[0], costs[0], grad[0], coeffs[0]

    def get_cost(self, x):
        x = np.asarray(x, dtype=np.float64)
        self._unpack_x(x)
        cost, self.costs, self._grad, self.coeffs = self._device_eval(x)
        return cost

    def get_grad(self, x):
        x = np.asarray(x, dtype=np.float64)
        self._unpack_x(x)
        _, _, grad, self.coeffs = self._device_eval(x)      # get_grad leaves self.costs alone (:572)
        return grad

    def get_coeffs(self, int_wpts, ts):
        """expert_planner.py:261-336 (coefficients only; the device never forms the 6M x 6M matrix)"""
        self.int_wpts, self.ts = int_wpts, ts
        self.tau = self.map_T2tau(np.asarray(ts, dtype=np.float64))
        _, self._eval_costs, _, self.coeffs = self._device_eval(self._pack_x())

    def reset_cost(self):
        self.costs = np.zeros(len(self.weights))

    # standalone use after get_coeffs(), as all_planner_demo.py:46-51 does
    def add_energy_cost(self):
        self.costs[0] += self._eval_costs[0]

    def add_time_cost(self):
        self.costs[1] += self._eval_costs[1]

    def add_sampled_cost(self):
        self.costs[2] += self._eval_costs[2]
        self.costs[3] += self._eval_costs[3]

    # ------------------------------------------------------------ time map (:468-483)
    def map_T2tau(self, ts):
        tau = np.zeros(self.M)
        for i in range(self.M):
            tau[i] = -math.log((self.T_max - self.T_min) / (ts[i] - self.T_min) - 1)
        return tau

    def map_tau2T(self, tau):
        ts = np.zeros(self.M)
        for i in range(self.M):
            ts[i] = (self.T_max - self.T_min) / (1 + math.exp(-tau[i])) + self.T_min
        return ts

    # ------------------------------------------------------------ evaluation (traj_utils.py:85-250)
    def _states(self, hz):
        ts = np.asarray(self.ts, dtype=np.float64)
        K = len(np.arange(0, sum(ts), 1 / hz))
        self.tau = self.map_T2tau(ts)
        x = _lib.as_f64(self._pack_x()).reshape(1, -1)
        state = np.zeros((1, max(K, 1), 3, self.D))
        cnt = np.zeros(1, np.int32)
        head = _lib.as_f64(self.head_state).reshape(1, 3, self.D)
        tail = _lib.as_f64(self.tail_state).reshape(1, 3, self.D)
        self._sync_params()
        c = self.ctx
        if K > 0:
            c.check(c.lib.neo_eval_traj_batch(c.h, 1, self.M, self.D, _lib.ptr(x), _lib.ptr(head), _lib.ptr(tail),
                                              float(hz), K, _lib.ptr(state), _lib.ptr(cnt)))
        return state[0, :K]

    def get_full_state_cmd(self, hz=300):
        return self._states(hz)

    def get_pos_array(self):
        return self._states(10.0)[:, 0, :]       # np.arange(0, sum(ts), 0.1): 1/10.0 == 0.1

    def get_vel_array(self):
        return self._states(10.0)[:, 1, :]

    def get_acc_array(self):
        return self._states(10.0)[:, 2, :]

    # single-time evaluators (traj_utils.py:85-179).  Not on the hot path (visualisation helpers): plain
    # polynomial evaluation of the device-computed coefficients, with the reference's piece search.
    def _at(self, t, order):
        if isinstance(self.coeffs, list) and self.coeffs == []:
            self.get_coeffs(self.int_wpts, self.ts)
        total = sum(self.ts)
        if t > total:
            t = total
        k = 0
        while sum(self.ts[:k + 1]) < t:
            k += 1
        T = t - sum(self.ts[:k])
        beta = [np.array([1, T, T**2, T**3, T**4, T**5]), np.array([0, 1, 2*T, 3*T**2, 4*T**3, 5*T**4]),
                np.array([0, 0, 2, 6*T, 12*T**2, 20*T**3]), np.array([0, 0, 0, 6, 24*T, 60*T**2])][order]
        return np.dot(np.asarray(self.coeffs)[6 * k:6 * k + 6, :].T, beta)[None, :]

    def get_pos(self, t):
        return self._at(t, 0)

    def get_vel(self, t):
        return self._at(t, 1)

    def get_acc(self, t):
        return self._at(t, 2)

    def get_jerk(self, t):
        return self._at(t, 3)

    def get_jer_array(self):
        self.get_coeffs(self.int_wpts, self.ts)
        return np.concatenate([self.get_jerk(t) for t in np.arange(0, sum(self.ts), 0.1)], axis=0)

    def print_results(self):
        print("-----------------------Final intermediate waypoints-----------------------")
        print(np.asarray(self.int_wpts).T)
        print("-----------------------Final T--------------------------------------------")
        print(self.ts)
        self.weighted_cost = self.costs * self.weights
        print("Energy cost: %f, Time cost: %f, Feasibility cost: %f, Collision cost: %f" % tuple(self.weighted_cost))


class BatchPlanner:
    """B independent replans per call (host arrays in, host arrays out).  For device-resident
    buffers use `optimize_dev` with torch tensors."""

    def __init__(self, config=None, ctx=None, sample_dtype="f64", stale_T=True, waves_per_simd=None, lane_groups=False):
        """waves_per_simd: None (the library decides by batch size), 1 (shortest evaluations) or 2 (highest
        throughput when several batches are in flight) -- include/neo_planner.h NEO_FLAG_*; same results.
        sample_dtype: "f64" (parity mode), "f32" (fp32 sampled terms, fp64 solve and optimiser) or "f32x" (everything
        in fp32, NEO_FLAG_F32_SOLVE: 3-D fields and the reference's 2-D map)"""
        self.cfg = config if config is not None else PlannerConfig()
        self._ctx = ctx
        self.all_f32 = sample_dtype == "f32x"
        if self.all_f32:
            sample_dtype = "f32"
        self.sample_dtype = sample_dtype
        self.stale_T = stale_T
        self.flags = {None: 0, 1: _lib.NEO_FLAG_ONE_WAVE_PER_SIMD, 2: _lib.NEO_FLAG_TWO_WAVES_PER_SIMD}[waves_per_simd]
        if lane_groups:     # small problems: eight trajectories per wavefront (NEO_FLAG_LANE_GROUPS; fp32-rounding-level
            self.flags |= _lib.NEO_FLAG_LANE_GROUPS   # differences to the default kernel)
        if self.all_f32:
            self.flags |= _lib.NEO_FLAG_F32_SOLVE

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    def _sync(self):
        _push_params(self.ctx, self.cfg, self.sample_dtype, self.stale_T, self.flags)

    def pack_x(self, int_wpts, ts):
        """int_wpts (B, D, M-1), ts (B, M) -> x (B, n) with tau = map_T2tau(ts) (:468-475)"""
        B = ts.shape[0]
        tau = -np.log((self.cfg.T_max - self.cfg.T_min) / (ts - self.cfg.T_min) - 1.0)
        return np.concatenate([np.asarray(int_wpts, dtype=np.float64).reshape(B, -1), tau], axis=1)

    def unpack_x(self, x, M, D):
        B = x.shape[0]
        nq = D * (M - 1)
        ts = (self.cfg.T_max - self.cfg.T_min) / (1.0 + np.exp(-x[:, nq:])) + self.cfg.T_min
        return x[:, :nq].reshape(B, D, M - 1), ts

    def cost_grad(self, map, x, head, tail, want_coeffs=False):
        self._sync()
        c = self.ctx
        x = _lib.as_f64(x); head = _lib.as_f64(head); tail = _lib.as_f64(tail)
        B, n = x.shape
        D = head.shape[2]
        M = (n + D) // (D + 1)
        cost = np.zeros(B); costs = np.zeros((B, 4)); grad = np.zeros((B, n)); st = np.zeros(B, np.int32)
        coeffs = np.zeros((B, 6 * M, D)) if want_coeffs else None
        c.check(c.lib.neo_cost_grad_batch(c.h, map.scene_id, B, M, D, _lib.ptr(x), _lib.ptr(head), _lib.ptr(tail),
                                          _lib.ptr(cost), _lib.ptr(costs), _lib.ptr(grad), _lib.ptr(coeffs),
                                          _lib.ptr(st)))
        return dict(cost=cost, costs=costs, grad=grad, coeffs=coeffs, status=st)

    def sampled_terms(self, map, coeffs, ts, order=None, io32=False):
        """add_sampled_cost + add_sampled_grad_CT (:392-466) for B trajectories: coeffs (B, 6M, D), ts (B, M).
        `order`: optional permutation (B,) the ESDF-lookup kernel dispatches its workgroups in (`spatial_order`); the
        results stay in the caller's order, bit-identical.  `io32`: hand the coefficients over and take the partials back
        as float32 (neo_sampled_terms_batch_f32; fp32 sampling only)"""
        self._sync()
        c = self.ctx
        ts = _lib.as_f64(ts)
        B, M = ts.shape
        D = np.shape(coeffs)[2]
        perm = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
        c.check(c.lib.neo_sampled_terms_dispatch_order(c.h, _lib.ptr(perm), 0, 0 if perm is None else B))
        costs2 = np.zeros((B, 2))
        if io32:
            coeffs = np.ascontiguousarray(coeffs, dtype=np.float32)
            gC = np.zeros((B, 6 * M, D), np.float32); gT = np.zeros((B, M), np.float32)
            fn = c.lib.neo_sampled_terms_batch_f32
        else:
            coeffs = _lib.as_f64(coeffs)
            gC = np.zeros((B, 6 * M, D)); gT = np.zeros((B, M))
            fn = c.lib.neo_sampled_terms_batch
        c.check(fn(c.h, map.scene_id, B, M, D, _lib.ptr(coeffs), _lib.ptr(ts), _lib.ptr(costs2), _lib.ptr(gC), _lib.ptr(gT)))
        return dict(costs2=costs2, grad_C=gC, grad_T=gT)

    def optimize(self, map, x0, head, tail, scene_ids=None, order=True):
        """map: one map for all trajectories (scene_ids None) or any map of the right kind plus
        scene_ids (B,) int32 of per-trajectory scene ids.  `order`: start the runs expected to be long
        first (`expected_effort_order`); results are in the caller's order either way."""
        self._sync()
        c = self.ctx
        x = _lib.as_f64(x0).copy(); head = _lib.as_f64(head); tail = _lib.as_f64(tail)
        B, n = x.shape
        D = head.shape[2]
        M = (n + D) // (D + 1)
        perm = None
        if order and B > 1024:
            _, ts0 = self.unpack_x(x, M, D)
            perm = self.expected_effort_order(head, tail, ts0)
        c.check(c.lib.neo_optimize_dispatch_order_host(c.h, _lib.ptr(perm), 0 if perm is None else B))
        costs = np.zeros((B, 4)); last = np.zeros((B, 4))
        nit = np.zeros(B, np.int32); nfev = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
        sid = None if scene_ids is None else np.ascontiguousarray(scene_ids, dtype=np.int32)
        c.check(c.lib.neo_optimize_batch(c.h, map.scene_id, _lib.ptr(sid), B, M, D, _lib.ptr(x), _lib.ptr(head),
                                         _lib.ptr(tail), _lib.ptr(costs), _lib.ptr(last), _lib.ptr(nit),
                                         _lib.ptr(nfev), _lib.ptr(st)))
        w = np.asarray(self.cfg.weights, dtype=np.float64)
        return dict(x=x, costs=costs, costs_last=last, nit=nit, nfev=nfev, status=st & 0xff,
                    collision=(st & _lib.NEO_TRAJ_FLAG_COLLISION) != 0, final_cost=(costs * w).sum(axis=1))

    def optimize_budgeted_dev(self, map, x0, x, head, tail, costs, costs_last, nit, nfev, status, state, eval_budget,
                              max_launches=4096):
        """optimize_dev with an evaluation budget per launch (neo_optimize_batch_budget_dev): the first launch gives every
        trajectory `eval_budget` evaluations, then the trajectories still running (status NEO_TRAJ_SUSPENDED) are
        re-launched COMPACTED -- one workgroup per straggler -- until none is left.  Finished results are valid as soon as
        their launch has ended; the finals are bit for bit those of `optimize_dev`.  torch CUDA tensors; `state` a uint8
        tensor of B * neo_optimize_state_bytes(M, D) bytes.  Returns the launch sizes."""
        import torch
        self._sync()
        c = self.ctx
        B, n = x.shape
        D = head.shape[2]
        M = (n + D) // (D + 1)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())
        sizes = []
        subset, resume = None, 0
        for _ in range(max_launches):
            c.check(c.lib.neo_optimize_batch_budget_dev(
                c.h, map.scene_id, B, M, D, pp(x0), pp(x), pp(head), pp(tail), pp(costs), pp(costs_last), pp(nit), pp(nfev),
                pp(status), pp(state), int(eval_budget), None if subset is None else pp(subset),
                0 if subset is None else int(subset.numel()), resume))
            sizes.append(B if subset is None else int(subset.numel()))
            c.synchronize()
            # (the statuses decide the next launch: this is the host round trip a budget costs)
            subset = torch.nonzero(status == _lib.NEO_TRAJ_SUSPENDED).flatten().to(torch.int32)
            resume = 1
            if subset.numel() == 0:
                return sizes
        raise RuntimeError(f"{int(subset.numel())} trajectories still suspended after {max_launches} launches")

    def optimize_budgeted(self, map, x0, head, tail, eval_budget):
        """host-array form of optimize_budgeted_dev: the dict of `optimize` plus `launch_sizes`"""
        import torch
        c = self.ctx
        dev = torch.device("cuda", c.device)
        x0 = _lib.as_f64(x0); head = _lib.as_f64(head); tail = _lib.as_f64(tail)
        B, n = x0.shape
        D = head.shape[2]
        M = (n + D) // (D + 1)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_x0, d_h, d_t = t(x0), t(head), t(tail)
        d_x = torch.empty_like(d_x0)
        costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
        nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); st = torch.zeros_like(nit)
        state = torch.empty(B * int(c.lib.neo_optimize_state_bytes(M, D)), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)        # (the context has its own stream: the buffers above are ready before it starts)
        sizes = self.optimize_budgeted_dev(map, d_x0, d_x, d_h, d_t, costs, last, nit, nfev, st, state, eval_budget)
        stn = st.cpu().numpy()
        w = np.asarray(self.cfg.weights, dtype=np.float64)
        cn_ = costs.cpu().numpy()
        return dict(x=d_x.cpu().numpy(), costs=cn_, costs_last=last.cpu().numpy(), nit=nit.cpu().numpy(), nfev=nfev.cpu().numpy(),
                    status=stn & 0xff, collision=(stn & _lib.NEO_TRAJ_FLAG_COLLISION) != 0, final_cost=(cn_ * w).sum(axis=1),
                    launch_sizes=sizes)

    def optimize_progressive(self, map, x0, head, tail, shares=(0.8,)):
        """`optimize` as a generator that hands results over as they finish (round 6, neo_optimize_progress_counter): ONE plain
        launch; the host polls the launch's counter of finished trajectories through a side stream and, each time it passes
        the next share of the batch, copies status first and then the results -- what status marks finished is final.
        Yields (indices, dict) -- indices int64 of the trajectories first seen finished, dict with their x, costs, costs_last,
        nit, nfev, status, collision, final_cost -- once per share and once more for the rest when the launch has ended.
        Every trajectory is yielded exactly once, with the bits `optimize` returns for it."""
        import torch
        c = self.ctx
        self._sync()
        dev = torch.device("cuda", c.device)
        x0 = _lib.as_f64(x0); head = _lib.as_f64(head); tail = _lib.as_f64(tail)
        B, n = x0.shape
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_x0, d_h, d_t = t(x0), t(head), t(tail)
        bufs = dict(x=torch.empty_like(d_x0), costs=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                    costs_last=torch.zeros(B, 4, dtype=torch.float64, device=dev), nit=torch.zeros(B, dtype=torch.int32, device=dev),
                    nfev=torch.zeros(B, dtype=torch.int32, device=dev), status=torch.full((B,), -1, dtype=torch.int32, device=dev))
        host = {k: torch.empty(tuple(v.shape), dtype=v.dtype).pin_memory() for k, v in bufs.items()}
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        h_cnt = torch.zeros(1, dtype=torch.int32).pin_memory()
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)        # (the context has its own stream: the buffers above are ready before it starts)
        w = np.asarray(self.cfg.weights, dtype=np.float64)
        seen = np.zeros(B, dtype=bool)

        def snapshot():
            with torch.cuda.stream(side):
                for k in ["status"] + [k_ for k_ in bufs if k_ != "status"]:     # status FIRST (include/neo_planner.h)
                    host[k].copy_(bufs[k], non_blocking=True)
                    side.synchronize()
            st = host["status"].numpy()
            new = np.flatnonzero((st != -1) & ~seen)
            seen[new] = True
            cn_ = host["costs"].numpy()[new]
            return new, dict(x=host["x"].numpy()[new].copy(), costs=cn_.copy(), costs_last=host["costs_last"].numpy()[new].copy(),
                             nit=host["nit"].numpy()[new].copy(), nfev=host["nfev"].numpy()[new].copy(), status=st[new] & 0xff,
                             collision=(st[new] & _lib.NEO_TRAJ_FLAG_COLLISION) != 0, final_cost=(cn_ * w).sum(axis=1))

        try:
            self.optimize_dev(map, bufs["x"], d_h, d_t, bufs["costs"], bufs["costs_last"], bufs["nit"], bufs["nfev"],
                              bufs["status"], x0=d_x0, progress=counter)
            for share in sorted(float(s_) for s_ in shares):
                need = min(B, int(np.ceil(share * B)))
                while True:
                    with torch.cuda.stream(side):
                        h_cnt.copy_(counter, non_blocking=True)
                        side.synchronize()
                    if int(h_cnt[0]) >= need:
                        break
                new, res = snapshot()
                if new.size:
                    yield new, res
            c.synchronize()
            new, res = snapshot()
            if new.size:
                yield new, res
        finally:
            c.synchronize()
            c.check(c.lib.neo_optimize_progress_counter(c.h, None))

    def init_guess(self, head, tail, count, rng=None, noise=0.0):
        """generate_init_variables (:82-101) for a batch: `count` waypoints on the straight line from start to target,
        durations init_T with the first and last piece 1.5 times as long; noise > 0 adds the N(0, noise) jitter of the
        reference's re-seeded attempts (:94; a numpy Generator here, the reference draws from the global RNG)"""
        head = np.asarray(head, dtype=np.float64); tail = np.asarray(tail, dtype=np.float64)
        start, target = head[:, 0], tail[:, 0]
        f = (np.arange(1, count + 1) / (count + 1))[None, None, :]
        wp = start[:, :, None] + (target - start)[:, :, None] * f                     # (B, D, count)
        if noise > 0.0:
            wp = wp + (rng if rng is not None else np.random.default_rng()).normal(0.0, noise, wp.shape)
        ts = np.full((head.shape[0], count + 1), float(self.cfg.init_T))
        ts[:, 0] *= 1.5
        ts[:, -1] *= 1.5
        return wp, ts

    def plan(self, map, head, tail, int_wpts=None, ts=None, waypoints=None, max_attempts=5, rng=None, scene_ids=None,
             seed=None, return_launch_sizes=False):
        """warm_start_plan (:186-203) for a batch: every request gets up to `max_attempts` plan_once runs.  An attempt that
        ends the way the reference raises on -- OverflowError statuses or `collision cost too large` (:235-237) -- is
        re-seeded like the reference's retries (straight line + N(0, 0.5), :94, :201) and optimised again; only the
        failed requests are launched again (compacted re-launches).  The jitter of request i at attempt a comes from its
        OWN stream, SeedSequence(seed, i, a): a retry does not depend on which other requests of the batch failed, as the
        reference's warm_start_plan of one request does not depend on other requests (`seed`: an int; None draws one from
        `rng` or from the OS).  Returns the optimiser's dict plus `attempts` (B,) and `solved` (B,): a request with solved
        False is one the reference answers with Exception("No solution for the given target")."""
        if (int_wpts is None) != (ts is None):
            raise ValueError("BatchPlanner.plan: give both int_wpts and ts, or neither")
        head = _lib.as_f64(head); tail = _lib.as_f64(tail)
        B, D = head.shape[0], head.shape[2]
        if int_wpts is None:
            int_wpts, ts = self.init_guess(head, tail, waypoints if waypoints is not None else int(self.cfg.init_wpts_num))
        count = np.asarray(int_wpts).shape[2]
        out = self.optimize(map, self.pack_x(int_wpts, ts), head, tail, scene_ids=scene_ids)
        if np.any(out["status"] == _lib.NEO_TRAJ_BAD_SCENE):
            # not a planning failure: the request named a map-table slot that does not exist -- retrying cannot help
            raise _lib.NeoError("BatchPlanner.plan: requests %s name a scene without a map (NEO_TRAJ_BAD_SCENE)"
                                % np.flatnonzero(out["status"] == _lib.NEO_TRAJ_BAD_SCENE)[:8].tolist())
        out["attempts"] = np.ones(B, dtype=np.int32)
        failed = lambda r: ((r["status"] > _lib.NEO_TRAJ_MAXITER) & (r["status"] != _lib.NEO_TRAJ_BAD_SCENE)) | r["collision"]
        todo = np.flatnonzero(failed(out))
        if seed is None:
            seed = int((rng if rng is not None else np.random.default_rng()).integers(0, 2 ** 62))
        launches = [B]
        for attempt in range(1, max_attempts):
            if todo.size == 0:
                break
            noise = np.stack([np.random.default_rng([int(seed), int(i), attempt]).normal(0.0, 0.5, (D, count)) for i in todo])
            wp_n, ts_n = self.init_guess(head[todo], tail[todo], count)
            r = self.optimize(map, self.pack_x(wp_n + noise, ts_n), head[todo], tail[todo],
                              scene_ids=None if scene_ids is None else np.asarray(scene_ids)[todo])
            launches.append(int(todo.size))
            for k in ("x", "costs", "costs_last", "nit", "nfev", "status", "collision", "final_cost"):
                out[k][todo] = r[k]
            out["attempts"][todo] += 1
            todo = todo[failed(r)]
        out["solved"] = ~failed(out)
        if return_launch_sizes:
            out["launch_sizes"] = launches
        return out

    def expected_effort_order(self, head, tail, ts):
        """permutation that starts the runs expected to be long first.  Proxy: time slack of the initial
        guess, sum(ts) * v_max / distance -- a guess that is far too slow needs many iterations to shed
        duration (measured on the cfg2 batch: rank correlation 0.5 with the evaluation count)."""
        head = np.asarray(head); tail = np.asarray(tail); ts = np.asarray(ts)
        dist = np.linalg.norm(tail[:, 0] - head[:, 0], axis=1)
        slack = ts.sum(axis=1) * self.cfg.v_max / np.maximum(dist, 1e-9)
        return np.argsort(-slack, kind="stable").astype(np.int32)

    def expected_effort_order_dev(self, x0, head, tail):
        """the same order computed on the device from RESIDENT torch tensors x0 (B, n), head / tail (B, 3, D) -- a keys kernel
        and a radix sort on the context's stream (neo_effort_order_dev); returns (order int32 [B], keys float64 [B]) device tensors"""
        import ctypes
        import torch
        self._sync()
        c = self.ctx
        B, n = x0.shape
        D = head.shape[2]
        M = (n + D) // (D + 1)
        scratch = torch.empty(int(c.lib.neo_effort_order_scratch_bytes(B)) // 8 + 1, dtype=torch.float64, device=x0.device)   # (keys first)
        order = torch.empty(B, dtype=torch.int32, device=x0.device)
        pp = lambda t: ctypes.c_void_p(t.data_ptr())
        c.check(c.lib.neo_effort_order_dev(c.h, B, M, D, pp(x0), pp(head), pp(tail), pp(scratch), pp(order)))
        return order, scratch[:B]

    @staticmethod
    def spatial_order(head, tail, xcds=8, cell=1.0, key=None, chunk=None):
        """XCD-aware spatial dispatch order for the ESDF-lookup kernel (neo_sampled_terms_dispatch_order): requests are keyed by
        a coarse Morton code of where they fly -- the midpoint of start and goal in cells of `cell` metres -- sorted, and
        the sorted list is dealt so that workgroups i, i + 8, i + 16, ... (one XCD: the hardware dispatches workgroups
        round-robin over the 8 XCDs, each with its own 4 MB L2) hold contiguous runs of it.  Requests whose
        corridors overlap then share an L2 and run at about the same time.  `chunk`: length of the runs dealt to the XCDs
        in turn (None: min(512, B / xcds); runs shorter than B / xcds make every XCD sweep the whole scene, which evens out
        their loads).  Results are in the caller's order and bit-identical whatever the order."""
        head = np.asarray(head); tail = np.asarray(tail)
        B = head.shape[0]
        if key is None:
            mid = 0.5 * (head[:, 0] + tail[:, 0])
            q = np.floor((mid - mid.min(axis=0)) / cell).astype(np.int64)
            bits = max(1, int(np.ceil(np.log2(max(int(q.max()) + 1, 2)))))
            key = np.zeros(B, dtype=np.int64)
            D = q.shape[1]
            for b in range(bits):
                for d in range(D):
                    key |= ((q[:, d] >> b) & 1) << (D * b + d)
        srt = np.argsort(key, kind="stable")
        if xcds <= 1:
            return srt.astype(np.int32)
        # default: runs of at most 512 requests -- what an XCD holds at four wavefronts per SIMD -- dealt to the XCDs in turn
        # (measured on the MI355X, cfg2, 163 840 requests in one launch: one run of B / 8 per XCD 0.385 of the roofline
        # figure, runs of 64 ... 512 0.401 ... 0.406, sorted without the deal 0.398; a 4096 launch has 512 per XCD either way)
        chunk = int(chunk) if chunk else min(512, -(-B // xcds))
        # run c of the sorted list goes to XCD c mod xcds; queue the runs per XCD, then workgroup i takes the next request
        # of XCD i mod xcds (XCDs that run out early are served from the others' leftovers: still a permutation)
        queues = [[] for _ in range(xcds)]
        for c, lo in enumerate(range(0, B, chunk)):
            queues[c % xcds].append(srt[lo:lo + chunk])
        queues = [np.concatenate(q_) if q_ else np.zeros(0, dtype=srt.dtype) for q_ in queues]
        out = np.full(B, -1, dtype=np.int64)
        left = []
        for k in range(xcds):
            slots = np.arange(k, B, xcds)
            m = min(len(slots), len(queues[k]))
            out[slots[:m]] = queues[k][:m]
            left.append(queues[k][m:])
        rest = np.concatenate(left)
        out[out < 0] = rest
        return out.astype(np.int32)

    def optimize_dev(self, map, x, head, tail, costs, costs_last, nit, nfev, status, slots=None, x0=None, progress=None):
        """torch CUDA tensors (float64 / int32), asynchronous on the context's stream.
        `slots`: optional int32 device tensor of map-table slots (Context.lib.neo_scene_slot).
        `x0`: optional start points (only read; results go to x) -- None: x is optimised in place.
        `progress`: optional int32 device tensor (one element, zeroed by the caller) that counts the trajectories as they
        finish (neo_optimize_progress_counter): poll it through a copy on another stream to use the finished results while
        the launch's long runs are still going"""
        B, n = x.shape
        D = head.shape[2]
        M = (n + D) // (D + 1)
        c = self.ctx
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        c.check(c.lib.neo_optimize_progress_counter(c.h, p(progress)))
        c.check(c.lib.neo_optimize_batch_from_dev(c.h, map.scene_id, p(slots), B, M, D, p(x0 if x0 is not None else x),
                                                  p(x), p(head), p(tail), p(costs), p(costs_last), p(nit), p(nfev),
                                                  p(status)))
