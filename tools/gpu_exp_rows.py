#!/usr/bin/env python3
"""Where the time of the trajectory-rows protocol goes for the 13 reference-sized chunks of
Input.run(1e6) taken in one launch: pass 1 (count), pass 2 (write records), transposition, copies.
Prints one JSON line."""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, Output, hip_api          # noqa: E402
from nexoclom_amd.Output import n_output_steps            # noqa: E402

nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n = 80467
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
opt = inputs.options
nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
ctx = hip_api.Context(0)
with contextlib.redirect_stdout(io.StringIO()):
    outs = [Output(inputs, n, seed=1 + k, integrate=False, save=False, context=ctx)
            for k in range(nchunks)]
ctx.set_forces(**outs[0].forces_kwargs())
ctx.set_bounce(None)
ctx.set_bodies(None)
soa = np.concatenate([o.x0_soa() for o in outs], axis=1)
res = {}
t0 = time.perf_counter()
ctx.upload_soa(soa)
res['upload_ms'] = (time.perf_counter() - t0)*1e3
for rep in range(3):
    t0 = time.perf_counter()
    ctx.integrate_const(opt.step_size, n_iter, opt.outeredge, want_steps=True, want_final=True)
    res['pass1_call_ms'] = (time.perf_counter() - t0)*1e3
    res['pass1_kernel_ms'] = ctx.last_kernel_ms()
    t0 = time.perf_counter()
    r = ctx.integrate_const_rows(opt.step_size, n_iter, opt.outeredge, narrow=True, resident=True)
    res['rows_call_ms'] = (time.perf_counter() - t0)*1e3
    res['pass2_kernel_ms'] = ctx.last_kernel_ms()
    res['records'] = r['store'].total
    t0 = time.perf_counter()
    rows, idx = r['store'].download()
    res['download_ms'] = (time.perf_counter() - t0)*1e3
    res['download_GB'] = (rows.nbytes + idx.nbytes)/1e9
    r['store'].free()
res['particle_steps'] = ctx.counters()['particle_steps']
res['packets'] = soa.shape[1]
print(json.dumps(res))
ctx.close()
