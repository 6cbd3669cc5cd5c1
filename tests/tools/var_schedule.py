"""Why the adaptive-step kernel is slower per attempt at 1e6 packets than at 1e7 (DESIGN.md
section 3): rk5 attempts per packet from the C oracle (TEST INFRASTRUCTURE, hence under
tests/tools/) for the bench's variable-step workload, and the makespan of the kernel's lane-refill
schedule (every free lane takes the next packet of the queue) in units of attempts, for the queue
orders one can build without knowing the answer.  Runs on the CPU (about 15 s for 1e5 packets on 8
threads).   python tests/tools/var_schedule.py > profiles/r04_var_schedule.json"""
import contextlib
import heapq
import io
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import np_oracle as O          # noqa: E402
from oracle.c_oracle import COracle        # noqa: E402
from oracle_context import OracleContext   # noqa: E402
from nexoclom_amd import Input, Output     # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
inputs.options.step_size, inputs.options.resolution = 0., 1e-4
with contextlib.redirect_stdout(io.StringIO()):
    out = Output(inputs, n, seed=1234, integrate=False, save=False, context=OracleContext())
X0 = np.ascontiguousarray(out.x0_soa().T)
forces = O.Forces(**out.forces_kwargs())
attempts = np.zeros(n, dtype=np.int64)


def count(rows):
    oracle = COracle()
    for i in rows:
        attempts[i] = oracle.integrate_var(forces, X0[i:i + 1], 1e-4, inputs.options.outeredge)[2]


with ThreadPoolExecutor(8) as pool:
    list(pool.map(count, [range(k, n, 8) for k in range(8)]))


def makespan(sequence, lanes):
    free_at = [0]*lanes
    for a in sequence:
        heapq.heappush(free_at, heapq.heappop(free_at) + int(a))
    return max(free_at)


lanes = max(64, int(256*768*n/1e6))       # the chip's resident lanes, scaled to n of 1e6 packets
speed = np.linalg.norm(X0[:, 4:7], axis=1)
_, step8, _, _ = COracle().integrate_var(forces, X0, 1e-4, inputs.options.outeredge, max_steps=8)
orders = {'as sampled': np.arange(n), 'fastest first (the product until round 4)': np.argsort(-speed),
          't_remaining / speed, descending (the product)': np.argsort(-X0[:, 0]/speed),
          't_remaining, descending': np.argsort(-X0[:, 0]), 'slowest first': np.argsort(speed),
          't_remaining / step size after 8 attempts (a pilot pass)': np.argsort(-X0[:, 0]/step8),
          'longest first (needs the answer)': np.argsort(-attempts)}
print(json.dumps({
    'workload': f'{n} packets of the bench inputfile, step_size 0, resolution 1e-4, random ages',
    'attempts': {'mean': float(attempts.mean()), 'median': float(np.median(attempts)),
                 'p99': float(np.percentile(attempts, 99)), 'p99.9': float(np.percentile(attempts, 99.9)),
                 'max': int(attempts.max())},
    'correlation_with_attempts': {'speed': float(np.corrcoef(speed, attempts)[0, 1]),
                                  't_remaining': float(np.corrcoef(X0[:, 0], attempts)[0, 1])},
    'lanes': lanes, 'packets_per_lane': n/lanes, 'mean_attempts_per_lane': float(attempts.sum()/lanes),
    'makespan_in_attempts': {k: makespan(attempts[o], lanes) for k, o in orders.items()},
    'makespan_in_attempts_at_50_packets_per_lane': {k: makespan(attempts[o], max(1, lanes//10))
                                                    for k, o in orders.items()},
    'mean_attempts_per_lane_at_50_packets_per_lane': float(attempts.sum()/max(1, lanes//10)),
    'note': 'the kernel cannot end before its longest packet has made its attempts one after the '
            'other; with 5 packets per lane that chain is twice the mean load of a lane.  The long '
            'chains belong to slow (bound) packets with much time left, so remaining time over '
            'launch speed orders the queue almost as well as the answer would; the model treats '
            'lanes as independent, the kernel hands out the queue 32 packets at a time to waves of '
            '64 lanes and gains less (35.0 -> 33.7 ms at 1e6 packets, 159 -> 151 ms at 1e7)'}, indent=1))
