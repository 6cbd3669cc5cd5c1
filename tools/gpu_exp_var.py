"""Timing of the variable-step driver kernel (a-4) and the single-step kernels."""
import os, sys, io, contextlib, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, Output, hip_api
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
inputs.options.step_size = 0.
inputs.options.resolution = 1e-4
ctx = hip_api.Context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
with contextlib.redirect_stdout(io.StringIO()):
    out = Output(inputs, n, seed=3, integrate=False, save=False, context=ctx)
ctx.set_forces(**out.forces_kwargs())
ctx.upload_soa(out.x0_soa())
for _ in range(3):
    t0 = time.time(); final, hs = ctx.integrate_var(1e-4, 25.); t1 = time.time()
    ms = ctx.last_kernel_ms(); c = ctx.counters()
    print(f'k_var: {n} packets, {c["particle_steps"]} rk5 attempts in {ms:.2f} ms -> {c["particle_steps"]/ms/1e6:.2f} G attempts/s; '
          f'unfinished {c["unfinished"]} bad {c["bad_step"]} alive {(final[:,7]>0).sum()} (call {1e3*(t1-t0):.0f} ms)')
X = np.ascontiguousarray(out.x0_soa().T)
h = np.full(n, 30.0)
for want in (False, True):
    t0 = time.time(); ctx.rk5_step(X, h, want_delta=want); t1 = time.time()
    print(f'nxc_rk5_step(delta={want}) {n} packets: call {1e3*(t1-t0):.1f} ms incl. PCIe')
