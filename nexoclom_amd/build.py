"""Build libnexoclom_hip.so (gfx950) in-tree with hipcc.

``python -m nexoclom_amd.build`` or ``nexoclom_amd.build.build()``.  hipcc cross-compiles without a
GPU; the resulting .so is git-ignored but travels to the GPU box with the snapshot.

Flags that matter for parity: ``-ffp-contract=off -fno-fast-math`` (one rounding per fp64
operation, as NumPy) and ``-munsafe-fp-atomics`` (hardware global_atomic_add_f64 for the image).
Flag that matters for speed: ``-mllvm -disable-machine-licm``.  Machine LICM hoists the 64-bit
constants of the exp/log/division code out of the persistent loop into scalar registers, which
then spill (30-60 SGPRs reloaded with v_readlane inside the loop); without it the fused kernel has
no spills and 13 fewer VGPRs.  ``-DNXC_EXPERIMENT_KNOBS`` (tools/ only) compiles the timing
switches of the image path in.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'nxc_api.hip')
DEPS = [os.path.join(HERE, 'csrc', f) for f in
        ('nxc_api.hip', 'nxc_kernels.hpp', 'nxc_device.hpp', 'nxc_math.hpp', 'nxc_log_table.hpp')]
DEPS.append(os.path.join(os.path.dirname(HERE), 'include', 'nexoclom_hip.h'))
OUT = os.path.join(HERE, 'lib', 'libnexoclom_hip.so')
# the same library with NumPy's two roundings per tableau term (rk5.py:33-35,41-43) instead of the
# fused multiply-adds: not what the package loads -- the yardstick the parity suite keeps alive
# (tests/test_gpu_two_roundings.py, against the C checker built with -DORACLE_TABLEAU_TWO_ROUNDINGS)
OUT_TWO_ROUNDINGS = os.path.join(HERE, 'lib', 'libnexoclom_hip_2r.so')

FLAGS = ['-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-fast-math',
         '-munsafe-fp-atomics', '-mllvm', '-disable-machine-licm', '-fPIC', '-shared',
         '-std=c++17', '-Wall', '-Wno-unused-function']
if os.environ.get('NXC_EXPERIMENT_KNOBS'):
    FLAGS.append('-DNXC_EXPERIMENT_KNOBS')
FLAGS += os.environ.get('NXC_EXTRA_FLAGS', '').split()      # experiments only


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; the HIP library cannot be built')
    return exe


def up_to_date(out=OUT):
    if not os.path.exists(out):
        return False
    t = os.path.getmtime(out)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force=False, verbose=False, two_roundings=False):
    out = OUT_TWO_ROUNDINGS if two_roundings else OUT
    if not force and up_to_date(out):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    extra = ['-DNXC_TABLEAU_TWO_ROUNDINGS'] if two_roundings else []
    cmd = [hipcc()] + FLAGS + extra + [SRC, '-o', out, '-ldl']
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f'hipcc failed building {os.path.basename(out)}')
    return out


def build_all(force=False, verbose=False):
    """Both libraries, side by side (two hipcc processes)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as pool:
        jobs = [pool.submit(build, force, verbose, two) for two in (False, True)]
        return [j.result() for j in jobs]


if __name__ == '__main__':
    if '--all' in sys.argv:
        print('\n'.join(build_all(force='--force' in sys.argv, verbose=True)))
    else:
        print(build(force='--force' in sys.argv, verbose=True, two_roundings='--two-roundings' in sys.argv))
