"""What one rk5 attempt of k_var costs a wave that has its SIMD to itself: K copies of the
longest chain among 20 000 packets of the bench's variable-step workload, alone on the chip
(K = 1: one lane of one wave; 64: one full wave; ...), for both compiled forms of the kernel.
ms / max attempts per packet = the serial latency the tail of a 1e6-packet launch runs at."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from nexoclom_amd import Input, Output, hip_api

ctx = hip_api.Context(0)
inputs = Input(bench.INFILE); inputs.options.step_size = 0.; inputs.options.resolution = 1e-4
with contextlib.redirect_stdout(io.StringIO()):
    out = Output(inputs, 200_000, seed=1234, integrate=False, save=False, context=ctx)
ctx.set_forces(**out.forces_kwargs()); ctx.set_bodies(None); ctx.set_bounce(None)
soa = out.x0_soa()                                   # (8, n)
key = soa[0]/np.linalg.norm(soa[4:7], axis=0)
top = np.argsort(-key)
res, edge = 1e-4, inputs.options.outeredge
# attempts of each of the top 256 alone -> the longest chain among them
att = []
for i in range(int(os.environ.get('SINGLES', 3000))):                               # one launch per packet: its attempts from the counter
    ctx.upload_soa(np.ascontiguousarray(soa[:, i:i + 1])); ctx.integrate_var(res, edge)
    att.append(ctx.counters()['particle_steps'])
att = np.array(att)
print('attempts of the first packets, each alone: max %d median %d mean %.0f' % (att.max(), np.median(att), att.mean()), flush=True)
longest = int(np.argmax(att))
top = np.argsort(-att)
for variant in ('plain', 'fair'):
    os.environ['NXC_TEST_VAR_VARIANT'] = variant
    for K, label in ((1, 'the longest alone'), (64, 'copies of it: one full wave'), (64*4, '4 waves'),
                     (64*16, '16 waves'), (64*64, '64 waves'), (64*256, '256 waves'), (64*512, '512 waves'),
                     (64*1024, 'copies: one wave per SIMD'), (64*1024*3, 'copies: three waves per SIMD')):
        cols = np.ascontiguousarray(np.repeat(soa[:, longest:longest + 1], K, axis=1))
        ctx.upload_soa(cols)
        ms = []
        for it in range(3):
            ctx.integrate_var(res, edge); ms.append(ctx.last_kernel_ms())
        a = ctx.counters()['particle_steps']//K
        print('%s  %8d %-30s %8.3f ms  %6d attempts  %.3f us per attempt' % (variant, K, label, min(ms), a, min(ms)*1e3/a), flush=True)
    # sparse: 1024 DIFFERENT packets (one live lane pattern like the tail's)
    cols = np.ascontiguousarray(soa[:, top[:2048]])
    ctx.upload_soa(cols)
    ms = []
    for it in range(3):
        ctx.integrate_var(res, edge); ms.append(ctx.last_kernel_ms())
    print('%s  the 2048 longest of the 20 000: %.3f ms, %d attempts in all' % (variant, min(ms), ctx.counters()['particle_steps']), flush=True)

    for K in (64*256, 64*1024, 64*1024*3):
        ctx.upload_soa(np.ascontiguousarray(soa[:, :K])); ms = []
        for it in range(3):
            ctx.integrate_var(res, edge); ms.append(ctx.last_kernel_ms())
        print('%s  the first %d packets as sampled: %.3f ms, %d attempts in all' % (variant, K, min(ms), ctx.counters()['particle_steps']), flush=True)
