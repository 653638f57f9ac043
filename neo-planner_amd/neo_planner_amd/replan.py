"""
ROS-free replan loop: the orchestration that sits directly above the optimiser in
ros_node/traj_planner_node.py, with perfect tracking in place of PX4/Gazebo.

  set_local_target          traj_planner_node.py:450-488   5 m ahead toward the goal, lateral steps out
                                                           of obstacles, target speed 0.8 v_max
  get_drone_state_ahead     :527-537                       state `planning_time_ahead` s ahead on the
                                                           current command array
  try_local_planning/replan :421-448, :539-578             <= 11 re-targeted attempts, splice the new
                                                           command array at the look-ahead index
  first_plan                :490-525

Works with any planner object that has the reference's interface (`plan`, `batch_plan`,
`get_full_state_cmd`, `int_wpts`, `ts`, `iter_num`): neo_planner_amd.MinJerkPlanner on the GPU, or the
CPU oracle in tests.  This is BASELINE.json configs[0] (one trajectory at a time, goal (30, 0)).
"""
import contextlib
import io

import numpy as np


class _State:
    def __init__(self, pos, vel):
        self.global_pos = np.asarray(pos, dtype=np.float64)
        self.global_vel = np.asarray(vel, dtype=np.float64)


class ReplanLoop:
    def __init__(self, planner, map, goal=(30.0, 0.0), v_max=1.0, des_pos_z=2.0, cmd_hz=60,
                 replan_period=1.0, planning_time_ahead=1.0, longitu_step_dis=5.0, lateral_step_length=1.0,
                 target_reach_threshold=0.2, mode="basic", quiet=True):
        self.planner, self.map = planner, map
        self.global_target = np.asarray(goal, dtype=np.float64)
        self.move_vel = 0.8 * v_max                               # :87
        self.des_pos_z, self.cmd_hz = des_pos_z, cmd_hz
        self.replan_period, self.planning_time_ahead = replan_period, planning_time_ahead
        self.longitu_step_dis, self.lateral_step_length = longitu_step_dis, lateral_step_length
        self.target_reach_threshold = target_reach_threshold
        self.mode, self.quiet = mode, quiet
        self.near_global_target = False
        self.des_state_index = 0
        self.n_plans = self.n_failed_attempts = 0

    # ---- :450-488
    def set_local_target(self, current_pos, seed=0):
        self.target_state = np.zeros((2, 2))
        goal = self.global_target
        if np.linalg.norm(goal - current_pos) < self.longitu_step_dis:
            self.target_state[0] = goal
            self.near_global_target = True
            return
        ahead = (goal - current_pos) / np.linalg.norm(goal - current_pos)
        side = np.array([[ahead[1], -ahead[0]], [-ahead[1], ahead[0]]])
        which, shift = 0, self.lateral_step_length
        if seed > 1e-3:
            p = current_pos + self.longitu_step_dis * ahead + np.random.normal(0, 1, 2)
        else:
            p = current_pos + self.longitu_step_dis * ahead
        while self.map.has_collision(p):
            p = p + shift * side[which]
            which = 1 - which
            shift += self.lateral_step_length
        to_goal = (goal - p) / np.linalg.norm(goal - p)
        self.target_state = np.array([p, self.move_vel * to_goal])

    def _call_planner(self, start):
        start_2d = np.array([start.global_pos[:2], start.global_vel[:2]])
        sink = io.StringIO() if self.quiet else None
        with (contextlib.redirect_stdout(sink) if sink else contextlib.nullcontext()):
            if self.mode == "batch":
                self.planner.batch_plan(self.map, start_2d, self.target_state)
            else:
                self.planner.plan(self.map, start_2d, self.target_state)
        self.n_plans += 1

    def _plan_with_retargeting(self, start_fn, current_pos):
        """:399-448: up to 11 targets (the first deterministic, the rest randomly shifted)"""
        seed = 0
        self.set_local_target(current_pos, seed)
        while True:
            try:
                self._call_planner(start_fn())
                return True
            except Exception:
                self.n_failed_attempts += 1
                seed += 1
                self.set_local_target(current_pos, seed)
                if seed > 10:
                    return False

    # ---- :527-537
    def get_drone_state_ahead(self):
        self.future_index = min(int(self.planning_time_ahead * self.cmd_hz) + self.des_state_index,
                                self.des_state_array.shape[0] - 1)
        return _State(np.append(self.des_state_array[self.future_index, 0, :], self.des_pos_z),
                      np.append(self.des_state_array[self.future_index, 1, :], 0.0))

    def run(self, start_pos=(0.0, 0.0), start_vel=(0.0, 0.0), max_replans=60):
        """fly to the goal; returns a dict of what the reference's metrics file records (:288-308)"""
        drone = _State(np.append(np.asarray(start_pos, float), self.des_pos_z), np.append(np.asarray(start_vel, float), 0.0))
        if not self._plan_with_retargeting(lambda: drone, drone.global_pos[:2]):      # first_plan :490-525
            return dict(success=False, replans=self.n_plans, path=np.zeros((0, 2)))
        self.des_state_array = self.planner.get_full_state_cmd(self.cmd_hz)
        self.des_state_index = 0
        step = int(round(self.replan_period * self.cmd_hz))
        ok = True
        for _ in range(max_replans):
            if self.near_global_target:
                break
            # perfect tracking: one replan period later the drone is `step` commands further
            self.des_state_index = min(self.des_state_index + step, self.des_state_array.shape[0] - 1)
            cur = self.des_state_array[self.des_state_index, 0, :]
            ok = self._plan_with_retargeting(self.get_drone_state_ahead, cur)
            if not ok:
                break
            new = self.planner.get_full_state_cmd(self.cmd_hz)
            self.des_state_array = np.concatenate((self.des_state_array[:self.future_index], new), axis=0)   # :577
        path = self.des_state_array[:, 0, :]
        reached = ok and np.linalg.norm(path[-1] - self.global_target) < self.target_reach_threshold
        dists = np.array([self.map.get_edt_dis(p) for p in path[::6]])
        return dict(success=bool(reached), replans=self.n_plans, failed_attempts=self.n_failed_attempts,
                    iter_num=self.planner.iter_num, opt_runs=self.planner.opt_running_times, path=path,
                    min_clearance=float(dists.min()), duration=path.shape[0] / self.cmd_hz,
                    max_speed=float(np.linalg.norm(self.des_state_array[:, 1, :], axis=1).max()))
