"""Time the fused kernel (with / without image) for several builds of the library
(NEXOCLOM_HIP_LIB), each in its own process: python tools/gpu_exp_variants.py lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:] or ['']:
    env = dict(os.environ)
    if lib:
        env['NEXOCLOM_HIP_LIB'] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gpu_experiments.py'), '1e7'],
                       env=env, capture_output=True, text=True, timeout=300)
    out = [l for l in r.stdout.splitlines() if l.startswith('n=')]
    print(lib or 'default', '|', out[0] if out else r.stderr[-400:])
