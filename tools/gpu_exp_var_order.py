"""What a perfect queue order would buy k_var (library built with -DNXC_VAR_TRACE, which returns each
packet's attempts in place of its stored step and lets NXC_TEST_VAR_NO_ORDER take the packets as
uploaded for the queue): the bench's variable-step packets in the product's order, sorted by their
true number of attempts (longest first), and shuffled."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from nexoclom_amd import Input, Output, hip_api

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ctx = hip_api.Context(0)
inputs = Input(bench.INFILE); inputs.options.step_size = 0.; inputs.options.resolution = 1e-4
with contextlib.redirect_stdout(io.StringIO()):
    out = Output(inputs, n, seed=bench.SEED, integrate=False, save=False, context=ctx)
ctx.set_forces(**out.forces_kwargs()); ctx.set_bodies(None); ctx.set_bounce(None)
soa = out.x0_soa()
res, edge = 1e-4, inputs.options.outeredge
def run(cols, label):
    ctx.upload_soa(np.ascontiguousarray(cols)); ms = []
    for it in range(3):
        fin, att = ctx.integrate_var(res, edge); ms.append(ctx.last_kernel_ms())
    print('%-45s %7.2f ms' % (label, min(ms[1:])), flush=True)
    return att
for variant in ('plain', 'fair'):
    os.environ['NXC_TEST_VAR_VARIANT'] = variant
    os.environ.pop('NXC_TEST_VAR_NO_ORDER', None)
    att = run(soa, variant + ': the product (flight key)')
    os.environ['NXC_TEST_VAR_NO_ORDER'] = '1'
    run(soa, variant + ': as sampled')
    o = np.argsort(-att)
    run(soa[:, o], variant + ': longest first (the answer)')
    # the answer with a tenth of the long ones misplaced at random
    rng = np.random.default_rng(1); o2 = o.copy(); k = n//50
    miss = rng.random(k) < 0.1; pos = rng.integers(0, n, miss.sum())
    o2[np.nonzero(miss)[0]], o2[pos] = o[pos], o[np.nonzero(miss)[0]]
    run(soa[:, o2], variant + ': same, 10% of the top 2% misplaced')
