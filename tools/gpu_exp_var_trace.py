"""Where the time of a k_var launch goes (library built with -DNXC_VAR_TRACE): per wave, when its
queue drained, when it was down to 8 live lanes, when it ended, and the trips in between."""
import contextlib, ctypes, io, os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from nexoclom_amd import Input, Output, hip_api

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ctx = hip_api.Context(0)
inputs = Input(bench.INFILE); inputs.options.step_size = 0.; inputs.options.resolution = 1e-4
leg = bench.variable_leg(ctx, inputs, n, passes=1)
print('%d packets: k_var %.2f ms' % (n, leg['kernel_ms']))
tr = np.zeros((4096, 8), dtype=np.uint64)
lib = ctx.lib
assert lib.nxc_debug_var_trace(ctx._h, tr.ctypes.data_as(ctypes.c_void_p)) == 0
tr = tr[tr[:, 3] > 0].astype(np.int64)
t0 = tr[:, 0].min()
ms = lambda c: (c - t0)/1e5
start, drain, low, end = ms(tr[:, 0]), ms(tr[:, 1]), ms(tr[:, 2]), ms(tr[:, 3])
trips, trips_d, trips_l = tr[:, 4], tr[:, 5], tr[:, 6]
simd, slot = (tr[:, 7] >> 4) & 3, tr[:, 7] & 15          # HW_ID: the SIMD and the wave slot on it
q = lambda a: ' '.join('%7.2f' % v for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
print('waves %d   percentiles 0 10 50 90 99 100' % len(tr))
print('waves per SIMD id:', np.bincount(simd), ' per wave slot:', np.bincount(slot))
print('start     ms:', q(start))
print('drained   ms:', q(drain))
print('<=8 live  ms:', q(low[tr[:, 2] > 0]))
print('end       ms:', q(end))
# a SIMD issues for its oldest wave first: rank the waves of each (CU, SIMD) by their start
before = (drain - start)*1e3/np.maximum(trips_d, 1)
order = np.argsort(before)
third = len(tr)//3
for name, sel in (('fastest third before the drain', order[:third]), ('middle third', order[third:2*third]),
                  ('slowest third', order[2*third:])):
    d = sel[tr[sel, 1] > 0]
    print('%-32s us per trip before the drain: %s' % (name, q(before[d])))
    print('%-32s us per trip after it:         %s' % ('', q((end[d] - drain[d])*1e3/np.maximum(trips[d] - trips_d[d], 1))))
    print('%-32s trips in all: %s   end ms: %s' % ('', q(trips[d]), q(end[d])))
l = tr[:, 2] > 0
print('us per trip once <= 8 lanes live:', q((end[l] - low[l])*1e3/np.maximum(trips[l] - trips_l[l], 1)))
print('trips after the drain:', q(trips - trips_d), '  once <= 8 live:', q(trips[l] - trips_l[l]))
