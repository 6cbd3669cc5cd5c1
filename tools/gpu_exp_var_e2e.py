"""The adaptive-step driver through the public API: Input.run(N) with options.step_size = 0 and the
reference's default chunking (packs_per_it = 1e6, Input.py:216-217).  Input.run integrates all the
chunks of a pass in ONE k_var launch (Output.integrate_batch); prints the kernel's rate and the
wall time.  usage: python tools/gpu_exp_var_e2e.py [N]"""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, hip_api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = hip_api.Context(0)
for batch in (True, False):
    inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
    inputs.options.step_size = 0.
    inputs.options.resolution = 1e-4
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        inputs.run(n, seed=7, context=ctx, batch=batch)
    t1 = time.time()
    ctr, k_ms = ctx.counters(), ctx.last_kernel_ms()
    print(json.dumps({'Input.run': n, 'mode': 'variable step', 'batch': batch,
                      'outputs': len(inputs._catalogue), 'wall_s': t1 - t0,
                      'last_k_var_launch': {'rk5_attempts': ctr['particle_steps'], 'kernel_ms': k_ms,
                                            'attempts_per_s': ctr['particle_steps']/(k_ms*1e-3)}}),
          flush=True)
    del inputs
ctx.close()
