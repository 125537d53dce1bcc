"""Host-side step scheduler of the GRU-ODE rollout.

The reference decides, inside ``NNFOwithBayesianJumps.forward`` (streamingflow/layers/
temporal_ode_bayes.py:508-620), when to propagate the ODE, when to apply an observation jump and
which visited state answers each target timestamp — interleaved with device work and with one
device->host sync per comparison.  The schedule is a pure float64 function of
``(observation times, target times, delta_t, USE_VARIABLE_ODE_STEP)``; this module computes it up
front so that the whole rollout can be enqueued (and hipGraph-captured) without host syncs.
Python floats are IEEE doubles, i.e. the same arithmetic as the reference's 0-d float64 tensors.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from ._lib import OP_JUMP, OP_STEP, SF_COEF_STRIDE

DRAWS_PER_STEP = {"euler": 1, "midpoint": 2, "rk4": 4}


@dataclass
class Schedule:
    ops: List[Tuple[int, int]] = field(default_factory=list)   # (OP_JUMP, obs index) | (OP_STEP, step index)
    dts: List[float] = field(default_factory=list)             # float64 dt of every step
    sel_nops: List[int] = field(default_factory=list)          # per target: number of ops applied to the chosen state
    n_draws: int = 0                                           # eps draws consumed (reference order)
    path_t: List[float] = field(default_factory=list)

    @property
    def n_steps(self):
        return len(self.dts)

    @property
    def n_jumps(self):
        return sum(1 for k, _ in self.ops if k == OP_JUMP)

    def key(self):
        """Structure of the rollout (what a captured hipGraph depends on; dt values do not)."""
        return (tuple(self.ops), tuple(self.sel_nops))

    def ops_array(self):
        return np.asarray(self.ops, dtype=np.int32).reshape(-1)

    def coef_array(self):
        """fp32 coefficient records (SF_COEF_STRIDE per step), each rounded once from float64 —
        the reference multiplies fp32 tensors by float64 scalars, i.e. by fp32(dt), fp32(dt/2)."""
        out = np.zeros((max(1, len(self.dts)), SF_COEF_STRIDE), dtype=np.float32)
        for i, dt in enumerate(self.dts):
            out[i] = [dt, dt / 2, dt / 6, dt / 3, dt / 2, dt / 6, dt / 2, dt / 3, dt, dt / 3, dt / 6, 0.0]
        return out


def merge_observations(camera_ts, lidar_ts):
    """models/future_prediction_ode.py:37-49 for one sample: the dict is keyed by 0-d tensors
    (hashed by identity, so equal times are NOT merged) and sorted by time with a stable sort =>
    camera before lidar on ties.  Returns (times, [(source, index)]) with source 0=camera, 1=lidar."""
    items = [(float(t), 0, i) for i, t in enumerate(camera_ts)] + [(float(t), 1, i) for i, t in enumerate(lidar_ts)]
    items.sort(key=lambda v: v[0])
    return [v[0] for v in items], [(v[1], v[2]) for v in items]


def build_schedule(times, targets, delta_t, variable, solver="euler") -> Schedule:
    """Restates temporal_ode_bayes.py:508-620.  ``times``: sorted observation times (float64),
    ``targets``: target times in the order given (not necessarily sorted)."""
    if len(times) == 0:
        raise ValueError("at least one observation is required (reference: times.min() of an empty tensor)")
    delta_t = float(delta_t)
    sch = Schedule()
    per_step = DRAWS_PER_STEP[solver]
    current_time = float(min(times))                                       # :508
    path_t, path_n = [], []

    def step(dt):
        nonlocal current_time
        sch.ops.append((OP_STEP, len(sch.dts)))
        sch.dts.append(dt)
        sch.n_draws += per_step
        current_time = current_time + dt                                   # :458

    for i, obs_time in enumerate(times):                                   # :539
        obs_time = float(obs_time)
        while current_time <= (obs_time - delta_t):                        # :541
            step((obs_time - current_time) if variable else delta_t)       # :546-549
        sch.ops.append((OP_JUMP, i))                                       # :565
        sch.n_draws += 1                                                   # :574
        path_t.append(obs_time)
        path_n.append(len(sch.ops))                                        # :578-581

    for predict_time in targets:                                           # :585
        predict_time = float(predict_time)
        while current_time < predict_time:                                 # :586
            step((predict_time - current_time) if variable else delta_t)   # :590-593
            if predict_time - 0.5 * delta_t < current_time < predict_time + 0.5 * delta_t:
                path_t.append(current_time)                                # :601-604
                path_n.append(len(sch.ops))

    pt = np.array(path_t)
    for time_stamp in targets:                                             # :610-620
        time_stamp = float(time_stamp)
        A = np.where(pt > time_stamp - 0.5 * delta_t)[0]
        B = np.where(pt < time_stamp + 0.5 * delta_t)[0]
        both = A[np.isin(A, B)]
        idx = int(np.max(both)) if both.size else int(np.argmin(np.abs(pt - time_stamp)))
        sch.sel_nops.append(path_n[idx])
    sch.path_t = path_t
    return sch
