"""Time the adaptive-step kernel for several builds of the library (NEXOCLOM_HIP_LIB), each in its
own process, and print a checksum of the final states (every build must give the same bits):
python tools/gpu_exp_var.py [npackets] lib1.so lib2.so ..."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n):
    import contextlib, io, numpy as np
    from nexoclom_amd import Input, Output, hip_api
    inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
    inputs.options.step_size = 0.
    inputs.options.resolution = 1e-4
    ctx = hip_api.Context(0)
    with contextlib.redirect_stdout(io.StringIO()):
        out = Output(inputs, n, seed=1234, integrate=False, save=False, context=ctx)
    ctx.set_forces(**out.forces_kwargs())
    ctx.upload_soa(out.x0_soa())
    ms = []
    for it in range(4):
        final, hs = ctx.integrate_var(1e-4, inputs.options.outeredge)[:2]
        if it:
            ms.append(ctx.last_kernel_ms())
    ctr = ctx.counters()
    digest = hashlib.sha1(np.ascontiguousarray(final).tobytes() + np.ascontiguousarray(hs).tobytes()).hexdigest()[:12]
    print(f'n={n} k_var {np.mean(ms):.2f} ms (min {min(ms):.2f}) attempts {ctr["particle_steps"]} '
          f'= {ctr["particle_steps"]/np.mean(ms)/1e-3:.3e}/s sha {digest}')


if __name__ == '__main__':
    if sys.argv[1:2] == ['--child']:
        child(int(float(sys.argv[2])))
        sys.exit(0)
    args = sys.argv[1:]
    n = '1e6'
    if args and not args[0].endswith('.so'):
        n = args.pop(0)
    for lib in args or ['']:
        env = dict(os.environ)
        if lib:
            env['NEXOCLOM_HIP_LIB'] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', n], env=env,
                           capture_output=True, text=True, timeout=600)
        out = [l for l in r.stdout.splitlines() if l.startswith('n=')]
        print(os.path.basename(lib) or 'default', '|', out[0] if out else (r.stdout + r.stderr)[-600:])
