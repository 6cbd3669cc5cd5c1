"""The pipelined upload pass with a large LAST piece (NXC_STREAM_TAIL = its share of the packets):
the pieces are put into queue order one by one, so the last one's long-lived packets start when
the queue is almost drained and the pass ends with a tail of few busy lanes; one big last piece is
ordered as a whole after the upload and its long-lived packets start early enough."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
from tools.gpu_experiments import setup
from nexoclom_amd.Output import n_output_steps
inputs, ctx, out, img = setup(10_000_000)
opt = inputs.options
nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
soa = out.x0_soa()
ctx.upload_soa(soa)
for rep in range(2):
    ctx.image_clear(); ctx.integrate_const_async(30., n_iter, 25., image=True); ctx.synchronize()
res = ctx.last_kernel_ms()
walls = []
for rep in range(4):
    ctx.image_clear(); ctx.synchronize()
    t0 = time.perf_counter()
    ctx.integrate_const_streamed(soa, 30., n_iter, 25., image=True, pieces=int(sys.argv[1]))
    ctx.synchronize()
    walls.append((time.perf_counter() - t0)*1e3)
c = ctx.counters()
print('tail', os.environ.get('NXC_STREAM_TAIL', '0'), 'pieces', sys.argv[1], 'resident %%.2f' %% res,
      'streamed wall', ' '.join('%%.2f' %% w for w in walls[1:]), 'unfinished', c['unfinished'], flush=True)
''' % ROOT
for tail in ('0', '0.25', '0.35', '0.45', '0.55'):
    for pieces in ('16',):
        env = dict(os.environ, NXC_STREAM_TAIL=tail)
        subprocess.run([sys.executable, '-c', code, pieces], env=env, check=True)
