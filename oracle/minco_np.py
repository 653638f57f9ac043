"""
oracle/minco_np.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

A CPU restatement, in NumPy/SciPy with the reference's per-sample Python loop
granularity, of neo-planner's replan inner loop:

  * MINCO minimum-jerk coefficient solve          (expert_planner.py:261-336)
  * energy / time / feasibility / collision cost  (expert_planner.py:345-422)
  * their gradients + adjoint propagation         (expert_planner.py:361-390, 424-537)
  * sigmoid time re-parametrisation               (expert_planner.py:468-492)
  * SciPy L-BFGS-B driver and retry wrappers      (expert_planner.py:62-237)
  * 2-D nearest-cell ESDF                         (map_server/esdf.py:11-82)
  * trajectory evaluation                         (traj_utils.py:85-222)

Paths are relative to /root/reference/src/planner/scripts/.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against fixtures captured from the real reference (tools/gen_golden.py, run in
the build container against /root/reference with NumPy 2.2.6 / SciPy 1.15.3).
The 3-D trilinear lookup (`Grid3DESDF`) has no counterpart in the reference:
for that mode this file is the definition ("parity unpinned" against the
reference; tied back to the 2-D mode by tests/test_oracle_golden.py).

Bug-compatible behaviours kept on purpose:
stale T in the last-piece time gradient, ESDF gradient in metres-per-cell,
int() truncation of cell indices, fixed-step quadrature that drops the last
partial interval, `final_cost` taken from the last *evaluated* x.
"""
import math

import numpy as np
from scipy import ndimage
from scipy import optimize as _sciopt

S_ORDER = 3                      # minimum-jerk: s = 3 (expert_planner.py:36)
OOB_DISTANCE = 10000             # esdf.py:64-65
HAS_COLLISION_DIS = 0.5          # esdf.py:4, 50-51


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
class PlannerParams:
    """Field names follow PlannerConfig (ros_node/traj_planner_node.py:32-46) /
    DefaultConfig (expert_planner.py:12-25).  Defaults are the ROS YAML values
    (launch/config/planner_config.yaml:2-13), not the unusable in-file
    DefaultConfig."""

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
        self.opt_tol = 1e-2          # read but never used by the reference (:49, :218)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


# --------------------------------------------------------------------------
# maps
# --------------------------------------------------------------------------
class GridESDF:
    """2-D ESDF with nearest-cell lookup (map_server/esdf.py:7-82)."""

    def __init__(self, occupancy, resolution, width, height, origin_xy):
        self.build(occupancy, resolution, width, height, origin_xy)

    def build(self, occupancy, resolution, width, height, origin_xy):
        # esdf.py:16-33 -- only the value 100 is an obstacle, unknown (-1) is free
        self.map_resolution = resolution
        self.map_width = int(width)
        self.map_height = int(height)
        self.origin_x = float(origin_xy[0])
        self.origin_y = float(origin_xy[1])
        occ = (np.asarray(occupancy).reshape(-1) == 100).astype(np.int64)
        self.occupancy_2d = occ.reshape(self.map_height, self.map_width)
        self.esdf_map = ndimage.distance_transform_edt(1 - self.occupancy_2d) * self.map_resolution
        # unit spacing on a metric map: metres per *cell* 
        self.esdf_grad_y, self.esdf_grad_x = np.gradient(self.esdf_map)

    def _cell(self, pos):
        # esdf.py:61-62 -- int() truncates toward zero
        row = int((pos[1] - self.origin_y) / self.map_resolution)
        col = int((pos[0] - self.origin_x) / self.map_resolution)
        inside = 0 <= row < self.map_height and 0 <= col < self.map_width
        return row, col, inside

    def get_edt_dis(self, pos):                      # esdf.py:53-67
        row, col, inside = self._cell(pos)
        return self.esdf_map[row, col] if inside else OOB_DISTANCE

    def get_edt_grad(self, pos):                     # esdf.py:69-82
        row, col, inside = self._cell(pos)
        if not inside:
            return [0, 0]
        return [self.esdf_grad_x[row, col], self.esdf_grad_y[row, col]]

    def has_collision(self, pos):                    # esdf.py:50-51
        return self.get_edt_dis(pos) < HAS_COLLISION_DIS

    def is_occupied(self, pos):                      # esdf.py:35-48
        row, col, inside = self._cell(pos)
        return bool(self.occupancy_2d[row, col]) if inside else False


class Grid3DESDF:
    """3-D ESDF, trilinear distance + analytic gradient (north-star mode; no
    reference counterpart -- this class IS the definition the HIP kernel is
    checked against).

    dist[iz, iy, ix] is the value at the voxel centre
    origin + (i + 0.5) * res.  A query outside [origin, origin + n*res) on any
    axis returns OOB_DISTANCE / zero gradient, like esdf.py:64-67.  Inside the
    domain the interpolation cell index is clamped to [0, n-2] and the fraction
    to [0, 1] (constant extrapolation over the outer half voxel).  The gradient
    is the derivative of the interpolant in metres per metre.
    """

    def __init__(self, dist, resolution, origin_xyz):
        self.dist = np.ascontiguousarray(dist)
        self.nz, self.ny, self.nx = self.dist.shape
        self.res = float(resolution)
        self.origin = np.asarray(origin_xyz, dtype=np.float64)

    def _cell(self, pos):
        n = (self.nx, self.ny, self.nz)
        idx, frac = [], []
        for a in range(3):
            u = (pos[a] - self.origin[a]) / self.res
            if not (u >= 0.0 and u < n[a]):
                return None
            u -= 0.5
            i0 = int(math.floor(u))
            i0 = min(max(i0, 0), n[a] - 2)
            f = min(max(u - i0, 0.0), 1.0)
            idx.append(i0)
            frac.append(f)
        return idx, frac

    def lookup(self, pos):
        """returns (distance, [gx, gy, gz])"""
        cell = self._cell(pos)
        if cell is None:
            return float(OOB_DISTANCE), [0.0, 0.0, 0.0]
        (ix, iy, iz), (fx, fy, fz) = cell
        d = self.dist
        c000 = float(d[iz, iy, ix]);         c100 = float(d[iz, iy, ix + 1])
        c010 = float(d[iz, iy + 1, ix]);     c110 = float(d[iz, iy + 1, ix + 1])
        c001 = float(d[iz + 1, iy, ix]);     c101 = float(d[iz + 1, iy, ix + 1])
        c011 = float(d[iz + 1, iy + 1, ix]); c111 = float(d[iz + 1, iy + 1, ix + 1])
        # lerp along x, then y, then z
        c00 = c000 + fx * (c100 - c000)
        c10 = c010 + fx * (c110 - c010)
        c01 = c001 + fx * (c101 - c001)
        c11 = c011 + fx * (c111 - c011)
        c0 = c00 + fy * (c10 - c00)
        c1 = c01 + fy * (c11 - c01)
        val = c0 + fz * (c1 - c0)
        # d/dx of the interpolant: lerp of the x-differences
        dx00 = c100 - c000; dx10 = c110 - c010; dx01 = c101 - c001; dx11 = c111 - c011
        dx0 = dx00 + fy * (dx10 - dx00)
        dx1 = dx01 + fy * (dx11 - dx01)
        gx = (dx0 + fz * (dx1 - dx0)) / self.res
        dy0 = c10 - c00
        dy1 = c11 - c01
        gy = (dy0 + fz * (dy1 - dy0)) / self.res
        gz = (c1 - c0) / self.res
        return val, [gx, gy, gz]

    def get_edt_dis(self, pos):
        return self.lookup(pos)[0]

    def get_edt_grad(self, pos):
        return self.lookup(pos)[1]


# --------------------------------------------------------------------------
# MINCO pieces
# --------------------------------------------------------------------------
def monomial_table(T_max, delta_t):
    """expert_planner.py:250-259: rows k -> t = k*delta_t, [4 derivative orders, 6 powers]."""
    ts = np.arange(0, T_max, delta_t)
    tab = np.zeros((len(ts), 4, 6))
    for k, t in enumerate(ts):
        tab[k, 0] = [1, t, t**2, t**3, t**4, t**5]
        tab[k, 1] = [0, 1, 2*t, 3*t**2, 4*t**3, 5*t**4]
        tab[k, 2] = [0, 0, 2, 6*t, 12*t**2, 20*t**3]
        tab[k, 3] = [0, 0, 0, 6, 24*t, 60*t**2]
    return tab


def assemble_system(int_wpts, ts, head_state, tail_state):
    """expert_planner.py:261-334.  int_wpts (D, M-1), ts (M,), head/tail (3, D).
    Row layout: 3 head rows; per joint i: waypoint row 6i+3, then position ..
    snap continuity rows 6i+4 .. 6i+8; 3 tail rows."""
    M = ts.shape[0]
    D = head_state.shape[1]
    N = 6 * M
    A = np.zeros((N, N))
    b = np.zeros((N, D))
    b[0:3] = head_state
    b[N - 3:] = tail_state
    A[0, 0] = 1.0
    A[1, 1] = 1.0
    A[2, 2] = 2.0
    wp = int_wpts.T
    # powers taken on the whole duration vector, as the reference does (:268-272)
    P1, P2, P3, P4, P5 = ts, ts**2, ts**3, ts**4, ts**5
    for i in range(M - 1):
        r, c = 6 * i + 3, 6 * i
        p = [1.0, P1[i], P2[i], P3[i], P4[i], P5[i]]
        A[r, c:c + 6] = p
        A[r + 1, c:c + 6] = p
        A[r + 1, c + 6] = -1.0
        A[r + 2, c + 1:c + 6] = [1.0, 2 * P1[i], 3 * P2[i], 4 * P3[i], 5 * P4[i]]
        A[r + 2, c + 7] = -1.0
        A[r + 3, c + 2:c + 6] = [2.0, 6 * P1[i], 12 * P2[i], 20 * P3[i]]
        A[r + 3, c + 8] = -2.0
        A[r + 4, c + 3:c + 6] = [6.0, 24.0 * P1[i], 60.0 * P2[i]]
        A[r + 4, c + 9] = -6.0
        A[r + 5, c + 4:c + 6] = [24.0, 120.0 * P1[i]]
        A[r + 5, c + 10] = -24.0
        b[r] = wp[i]
    A[N - 3, N - 6:] = [1.0, P1[-1], P2[-1], P3[-1], P4[-1], P5[-1]]
    A[N - 2, N - 5:] = [1.0, 2 * P1[-1], 3 * P2[-1], 4 * P3[-1], 5 * P4[-1]]
    A[N - 1, N - 4:] = [2.0, 6 * P1[-1], 12 * P2[-1], 20 * P3[-1]]
    return A, b


def jerk_gram(T):
    """closed-form integral of beta3 beta3^T over [0, T] (expert_planner.py:353-358)."""
    Q = np.zeros((6, 6))
    Q[3, 3:] = [36 * T, 72 * T**2, 120 * T**3]
    Q[4, 3:] = [72 * T**2, 192 * T**3, 360 * T**4]
    Q[5, 3:] = [120 * T**3, 360 * T**4, 720 * T**5]
    return Q


class OraclePlanner:
    """Restatement of MinJerkPlanner (expert_planner.py:28-585) + TrajUtils
    evaluation (traj_utils.py:85-222).  Same public names and side effects."""

    def __init__(self, config=None, stale_T=True):
        cfg = config if config is not None else PlannerParams()
        self.s = S_ORDER
        self.v_max = cfg.v_max
        self.T_min = cfg.T_min
        self.T_max = cfg.T_max
        self.safe_dis = cfg.safe_dis
        self.collision_cost_tol = cfg.collision_cost_tol
        self.weights = np.array(cfg.weights, dtype=np.float64)
        self.delta_t = cfg.delta_t
        self.beta_full = monomial_table(self.T_max, self.delta_t)
        self.opt_tol = cfg.opt_tol
        self.init_wpts_mode = cfg.init_wpts_mode
        self.init_seg_len = cfg.init_seg_len
        self.init_wpts_num = int(cfg.init_wpts_num)
        self.init_T = cfg.init_T
        self.batch_num = 3
        self.iter_num = 0
        self.opt_running_times = 0
        self.stale_T = stale_T        # False = mathematically consistent last-piece gradient
        self.coeffs = []
        self.last_result = None

    # ---- initial guesses (expert_planner.py:82-140) ----
    def generate_init_variables(self, head_state, tail_state, seed=0):
        start, target = head_state[0], tail_state[0]
        length = np.linalg.norm(target - start)
        if self.init_wpts_mode == 'adaptive':
            num = max(math.ceil(length / self.init_seg_len - 1), 1)
        else:
            num = self.init_wpts_num
        step = (target - start) / (num + 1)
        wpts = np.linspace(start + step, target, num, endpoint=False)
        if seed != 0:
            wpts += np.random.normal(0, 0.5, wpts.shape)     # global, unseeded RNG (:94)
        ts = self.init_T * np.ones((num + 1,))
        ts[0] *= 1.5
        ts[-1] *= 1.5
        return wpts.T, ts

    def batch_generate_init_variables(self, head_state, tail_state):
        start, target = head_state[0], tail_state[0]
        along = (target - start) / np.linalg.norm(target - start)
        lateral = np.array([[along[1], -along[0]], [-along[1], along[0]]])
        num = self.init_wpts_num
        cands = np.zeros((self.batch_num, num, head_state.shape[1]))
        step = (target - start) / (num + 1)
        cands[0] = np.linspace(start + step, target, num, endpoint=False)
        side = 0
        for i in range(1, self.batch_num):
            cands[i] = cands[0] + 0.6 * lateral[side]
            side = 1 - side
        ts = self.init_T * np.ones((num + 1,))
        ts[0] *= 1.5
        ts[-1] *= 1.5
        return np.transpose(cands, (0, 2, 1)), ts

    # ---- entry points (expert_planner.py:62-80, 142-237) ----
    def read_planning_conditions(self, map, head_state, tail_state, int_wpts, ts):
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
            except Exception:
                seed += 1
                self.int_wpts, self.ts = self.generate_init_variables(head_state, tail_state, seed)
        raise Exception("No solution for the given target")

    def batch_plan(self, map, head_state, tail_state):
        cands, ts = self.batch_generate_init_variables(head_state, tail_state)
        opt_wpts = np.zeros(cands.shape)
        opt_ts = np.zeros((self.batch_num, len(ts)))
        cost = np.zeros(self.batch_num)
        for i in range(self.batch_num):
            try:
                self.read_planning_conditions(map, head_state, tail_state, cands[i], ts)
                self.plan_once()
                opt_wpts[i] = self.int_wpts
                opt_ts[i] = self.ts
                cost[i] = self.weighted_cost.sum()
            except Exception:
                cost[i] = np.inf
            # the success check sits inside the loop (expert_planner.py:160-168);
            # not-yet-run candidates still hold cost 0 there.
            if np.min(cost) < np.inf:
                best = np.argmin(cost)
                self.int_wpts = opt_wpts[best]
                self.ts = opt_ts[best]
                self.final_cost = cost[best]
            else:
                self.warm_start_plan(map, head_state, tail_state, cands[0], ts)

    def plan_once(self, trace=None):
        self.tau = self.map_T2tau(self.ts)
        nq = self.D * (self.M - 1)
        x0 = np.concatenate((np.reshape(self.int_wpts, (nq,)), self.tau), axis=0)
        cb = None
        if trace is not None:
            def cb(intermediate_result):
                trace.append((np.array(intermediate_result.x), float(intermediate_result.fun)))
        res = _sciopt.minimize(self.get_cost, x0, method='L-BFGS-B', jac=self.get_grad,
                               bounds=None, tol=1e-4, callback=cb,
                               options={'maxcor': 10, 'maxfun': 15000, 'maxiter': 15000, 'maxls': 20})
        self.last_result = res
        self.int_wpts = np.reshape(res.x[:nq], (self.D, self.M - 1))
        self.tau = res.x[nq:]
        self.ts = self.map_tau2T(self.tau)
        self.iter_num += res.nit
        self.opt_running_times += 1
        self.weighted_cost = self.costs * self.weights      # costs of the LAST evaluated x (:233)
        self.final_cost = self.weighted_cost.sum()
        if self.weighted_cost[3] > self.collision_cost_tol:
            raise ValueError("collision cost too large")

    optimize = plan_once

    # ---- linear system (expert_planner.py:261-336) ----
    def get_coeffs(self, int_wpts, ts):
        self.A, b = assemble_system(int_wpts, ts, self.head_state, self.tail_state)
        self.coeffs = np.linalg.solve(self.A, b)

    def reset_cost(self):
        self.costs = np.zeros(len(self.weights))

    def reset_grad_CT(self):
        self.grad_C = np.zeros((6 * self.M, self.D))
        self.grad_T = np.zeros(self.M)

    # ---- energy / time (expert_planner.py:345-390) ----
    def add_energy_cost(self):
        for i in range(self.M):
            c = self.coeffs[6 * i:6 * i + 6, :]
            self.costs[0] += np.trace(c.T @ jerk_gram(self.ts[i]) @ c)

    def add_energy_grad_CT(self):
        for i in range(self.M):
            c = self.coeffs[6 * i:6 * i + 6, :]
            T = self.ts[i]
            jerk_row = np.array([0, 0, 0, 6, 24 * T, 60 * T**2])
            self.grad_C[6 * i:6 * i + 6, :] += self.weights[0] * 2 * jerk_gram(T) @ c
            for d in range(self.D):
                self.grad_T[i] += self.weights[0] * float(np.dot(c[:, d], jerk_row))**2

    def add_time_cost(self):
        self.costs[1] += np.sum(self.ts)

    def add_time_grad_CT(self):
        self.grad_T += self.weights[1] * np.ones(self.M)

    # ---- sampled terms (expert_planner.py:392-466) ----
    def _map_query_pos(self, pos):
        # the reference projects to the first two axes (:416); a 3-D map takes all of pos
        return pos if isinstance(self.map, Grid3DESDF) else pos[:2]

    def add_sampled_cost(self):
        for i in range(self.M):
            c = self.coeffs[6 * i:6 * i + 6, :]
            n_i = int(self.ts[i] / self.delta_t)
            for j in range(n_i):
                beta = self.beta_full[j]
                pos = np.dot(c.T, beta[0])
                vel = np.dot(c.T, beta[1])
                omg = 0.5 if j in (0, n_i - 1) else 1
                violate_vel = sum(vel**2) - self.v_max**2
                if violate_vel > 0.0:
                    self.costs[2] += omg * self.delta_t * violate_vel**3
                dis = self.map.get_edt_dis(self._map_query_pos(pos))
                violate_dis = self.safe_dis - dis
                if violate_dis > 0.0:
                    self.costs[3] += omg * self.delta_t * violate_dis**3

    def add_sampled_grad_CT(self):
        w = self.weights
        for i in range(self.M):
            c = self.coeffs[6 * i:6 * i + 6, :]
            n_i = int(self.ts[i] / self.delta_t)
            for j in range(n_i):
                beta = self.beta_full[j]
                pos = np.dot(c.T, beta[0])
                vel = np.dot(c.T, beta[1])
                omg = 0.5 if j in (0, n_i - 1) else 1
                violate_vel = sum(vel**2) - self.v_max**2
                if violate_vel > 0.0:
                    acc = np.dot(c.T, beta[2])
                    dK = 3 * self.delta_t * omg * violate_vel**2
                    self.grad_C[6 * i:6 * i + 6, :] += w[2] * dK * 2 * np.outer(beta[1], vel)
                    self.grad_T[i] += w[2] * (omg * violate_vel**3 / n_i
                                              + dK * 2 * float(np.dot(acc, vel)) * j / n_i)
                qpos = self._map_query_pos(pos)
                violate_dis = self.safe_dis - self.map.get_edt_dis(qpos)
                if violate_dis > 0.0:
                    g = np.zeros(self.D)
                    gm = self.map.get_edt_grad(qpos)
                    g[:len(gm)] = gm
                    dK = 3 * self.delta_t * omg * violate_dis**2
                    self.grad_C[6 * i:6 * i + 6, :] += w[3] * dK * (-np.outer(beta[0], g))
                    self.grad_T[i] += w[3] * (omg * violate_dis**3 / n_i
                                              + dK * (-float(np.dot(g, vel))) * j / n_i)

    # ---- time map (expert_planner.py:468-492) ----
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

    def get_grad_T2tau(self, grad_T):
        out = np.zeros(self.M)
        for i in range(self.M):
            e = math.exp(-self.tau[i])
            out[i] = grad_T[i] * (self.T_max - self.T_min) * e / (1 + e)**2
        return out

    # ---- adjoint (expert_planner.py:494-537) ----
    @staticmethod
    def _dE_joint(T):
        # d/dT of the six joint rows (waypoint, pos, vel, acc, jerk, snap continuity)
        return np.array([[0, 1, 2*T, 3*T**2, 4*T**3, 5*T**4],
                         [0, 1, 2*T, 3*T**2, 4*T**3, 5*T**4],
                         [0, 0, 2, 6*T, 12*T**2, 20*T**3],
                         [0, 0, 0, 6, 24*T, 60*T**2],
                         [0, 0, 0, 0, 24, 120*T],
                         [0, 0, 0, 0, 0, 120]])

    def propagate_grad_q_tau(self):
        M = self.M
        G = np.linalg.solve(self.A.T, self.grad_C)
        self.G = G
        grad_q = np.zeros((self.D, M - 1))
        grad_T = np.zeros(M)
        for i in range(M - 1):
            grad_q[:, i] = G[6 * i + 3, :]
        T = None
        for i in range(M - 1):
            T = self.ts[i]
            Gi = G[6 * i + 3:6 * i + 9, :]
            ci = self.coeffs[6 * i:6 * i + 6, :]
            grad_T[i] = self.grad_T[i] - np.trace(Gi.T @ self._dE_joint(T) @ ci)
        # tail rows: the reference reuses the loop's last T, i.e. ts[M-2] (:528-533)
        if not self.stale_T or T is None:
            T = self.ts[M - 1]
        dE_tail = self._dE_joint(T)[1:4]
        grad_T[M - 1] = self.grad_T[M - 1] - np.trace(G[6 * M - 3:, :].T @ dE_tail @ self.coeffs[6 * M - 6:, :])
        self.grad_T_total = grad_T
        return grad_q, self.get_grad_T2tau(grad_T)

    # ---- callbacks (expert_planner.py:539-585) ----
    def _unpack(self, x):
        nq = self.D * (self.M - 1)
        self.int_wpts = np.reshape(x[:nq], (self.D, self.M - 1))
        self.tau = x[nq:]
        self.ts = self.map_tau2T(self.tau)

    def get_cost(self, x):
        self._unpack(x)
        self.get_coeffs(self.int_wpts, self.ts)
        self.reset_cost()
        self.add_energy_cost()
        self.add_time_cost()
        self.add_sampled_cost()
        return np.dot(self.costs, self.weights)

    def get_grad(self, x):
        self._unpack(x)
        self.get_coeffs(self.int_wpts, self.ts)
        self.reset_grad_CT()
        self.add_energy_grad_CT()
        self.add_time_grad_CT()
        self.add_sampled_grad_CT()
        grad_q, grad_tau = self.propagate_grad_q_tau()
        return np.concatenate((np.reshape(grad_q, (self.D * (self.M - 1),)), grad_tau), axis=0)

    # ---- evaluation (traj_utils.py:85-222) ----
    def _piece_and_local_t(self, t):
        total = sum(self.ts)
        if t > total:
            t = total
        k = 0
        while sum(self.ts[:k + 1]) < t:
            k += 1
        return k, t - sum(self.ts[:k])

    def _eval(self, t, order):
        k, T = self._piece_and_local_t(t)
        c = self.coeffs[6 * k:6 * k + 6, :]
        if order == 0:
            beta = np.array([1, T, T**2, T**3, T**4, T**5])
        elif order == 1:
            beta = np.array([0, 1, 2*T, 3*T**2, 4*T**3, 5*T**4])
        elif order == 2:
            beta = np.array([0, 0, 2, 6*T, 12*T**2, 20*T**3])
        else:
            beta = np.array([0, 0, 0, 6, 24*T, 60*T**2])
        return np.dot(c.T, beta)

    def get_pos(self, t):
        return self._eval(t, 0)[None, :]

    def get_vel(self, t):
        return self._eval(t, 1)[None, :]

    def get_acc(self, t):
        return self._eval(t, 2)[None, :]

    def get_jerk(self, t):
        return self._eval(t, 3)[None, :]

    def get_full_state_cmd(self, hz=300):
        self.get_coeffs(self.int_wpts, self.ts)
        t_samples = np.arange(0, sum(self.ts), 1 / hz)
        out = np.zeros((t_samples.shape[0], 3, self.D))
        for i, t in enumerate(t_samples):
            for order in range(3):
                out[i, order] = self._eval(t, order)
        return out

    def _sample_array(self, order):
        self.get_coeffs(self.int_wpts, self.ts)
        t_samples = np.arange(0, sum(self.ts), 0.1)
        out = np.zeros((t_samples.shape[0], self.D))
        for i, t in enumerate(t_samples):
            out[i] = self._eval(t, order)
        return out

    def get_pos_array(self):
        return self._sample_array(0)

    def get_vel_array(self):
        return self._sample_array(1)

    def get_acc_array(self):
        return self._sample_array(2)
