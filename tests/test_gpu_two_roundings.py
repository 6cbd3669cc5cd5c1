"""The kernels built with -DNXC_TABLEAU_TWO_ROUNDINGS (libnexoclom_hip_2r.so: NumPy's two roundings per
tableau term, rk5.py:33-35,41-43 -- the arithmetic the product had until its tableau sums became fused
multiply-adds) against the C checker built with -DORACLE_TABLEAU_TWO_ROUNDINGS: the pair must stay
bit-identical like the product pair, so that "what did the fused terms cost in parity" stays a
question one can measure (tests/tools/arith_contract.py) instead of a build that has rotted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_roundings_kernels_equal_the_two_roundings_checker():
    from nexoclom_amd import build
    lib = build.build(two_roundings=True)            # built by __graft_entry__.build(); a no-op here
    env = dict(os.environ, NEXOCLOM_HIP_LIB=lib)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'tools', 'two_roundings_worker.py')],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out['step_bit_exact'] and out['step_differs_from_fused']
    assert out['step_vs_numpy'] < 1e-14                     # the reference's roundings but for pow / exp / log
    assert out['const_bit_exact'] and out['const_work'] and out['image_rel'] < 1e-11
    assert out['var_bit_exact'] and out['var_work']
