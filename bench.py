#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Workload (BASELINE.json configs[2]): Na at Mercury, taa = 1.3, gravity +
radiation pressure + photoionisation, endtime 50000 s, constant step 30 s (1668 stored records),
outeredge 25 R, uniform/flat(2.5 +- 2 km/s)/isotropic source, 1e7 packets PER GPU, every stored
record binned into a 512 x 512 radiance image (width 8 x 8 R).  One "step" = one full pass: clear
the image, integrate all resident packets to the end with the fused persistent kernel, and (N > 1)
sum the per-GPU image pairs over RCCL.  Inputs (X0, tables) are resident in HBM before the timed
region.

metric = particle*steps/s, whole job: sum over ranks of rk5 steps actually taken (active packets
only, Output.py:385) / max-over-ranks wall time.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Packets shard by index with no data-path collective except the image sum (weak scaling: the
per-GPU packet count is fixed).  torch.distributed (gloo, CPU) is only the control plane here:
rendezvous of the RCCL unique id, barrier, max/sum of scalars.  The image reduce itself is
ncclAllReduce called from libnexoclom_hip.so on the handle's stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PARTICLE_STEP = 128      # 8 fp64 read + 8 fp64 written
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--packets', type=int, default=10_000_000, help='packets per GPU')
    p.add_argument('--dims', type=int, default=512)
    p.add_argument('--quantity', default='radiance')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-packets', type=int, default=150_000)
    return p.parse_args()


def cpu_baseline(args, inputs):
    """The reference's CPU path, timed beside the GPU number on the same box: the NumPy oracle's
    constant-step driver (bit-identical to the reference's rk5/state arithmetic) on a bounded
    sample of the same workload, one process = one core; plus, for context, the C oracle on all
    host threads."""
    from nexoclom_amd import Output
    from nexoclom_amd.Output import n_output_steps
    from oracle import np_oracle as O
    from oracle.c_oracle import COracle
    from tests import helpers as H
    n = args.cpu_packets
    out = Output(inputs, n, seed=4242, integrate=False, save=False)
    X0 = np.ascontiguousarray(out.x0_soa().T)
    f = H.mercury_forces('Na', 1.3)
    opt = inputs.options
    nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
    t0 = time.time()
    _, _, work = O.constant_step_driver(f, X0, opt.endtime.value, opt.step_size, opt.outeredge)
    t_np = time.time() - t0
    co = COracle()
    threads = co.max_threads()
    nc = min(20*n, 2_000_000)
    outc = Output(inputs, nc, seed=4243, integrate=False, save=False)
    X0c = np.ascontiguousarray(outc.x0_soa().T)
    t0 = time.time()
    resc = co.integrate_const(f, X0c, opt.step_size, n_iter, opt.outeredge, threads=threads,
                              want_final=False)
    t_c = time.time() - t0
    return {'value': work/t_np, 'unit': 'particle*steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'NumPy oracle constant-step driver (reference arithmetic), {n} packets of '
                      f'the same workload, {work} particle*steps in {t_np:.1f} s',
            'c_port_value': resc['work']/t_c, 'c_port_cores': threads,
            'c_port_sample': f'C oracle (OpenMP), {nc} packets, {resc["work"]} particle*steps '
                             f'in {t_c:.1f} s'}


def main():
    args = parse()
    from nexoclom_amd import Input, Output, ModelImage, hip_api
    from nexoclom_amd.distributed import ControlPlane
    world = int(os.environ.get('WORLD_SIZE', '1')) if 'RANK' in os.environ else 1
    cp = ControlPlane(world)
    rank = cp.rank
    from nexoclom_amd.Output import n_output_steps

    infile = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
    inputs = Input(infile)
    opt = inputs.options
    nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)

    ndev = hip_api.device_count()
    if ndev < 1:
        raise SystemExit('bench.py needs a HIP device; there is no CPU fallback')
    ctx = hip_api.Context(cp.local_rank % ndev)

    # ---- set-up (untimed): sample this rank's shard, make everything resident --------------
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        out = Output(inputs, args.packets, seed=1234 + rank, integrate=False, save=False,
                     context=ctx)
        params = {'quantity': args.quantity, 'dims': f'{args.dims},{args.dims}', 'width': '8,8',
                  'center': '0,0'}
        img = ModelImage(inputs, params, context=ctx)      # parses params; no packets yet
    ctx.set_forces(**out.forces_kwargs())
    aplanet, vrplanet = out.aplanet, out.vrplanet
    img._set_image(ctx, aplanet, vrplanet, True)
    ctx.upload_soa(out.x0_soa())
    del out

    reduce_mode = 'none'
    if world > 1:
        reduce_mode = 'rccl-allreduce' if cp.init_rccl(ctx) else 'gloo-host-fallback'

    def one_step():
        ctx.image_clear()
        ctx.integrate_const_async(opt.step_size, n_iter, opt.outeredge, image=True)
        if reduce_mode == 'rccl-allreduce':
            ctx.image_allreduce()
        elif world > 1:          # only if RCCL could not be brought up: sum on the host
            cp.allreduce_images_host(*ctx.image_download())

    for _ in range(args.warmup):
        one_step()
        ctx.synchronize()

    kernel_ms = []
    cp.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        ctx.synchronize()
        kernel_ms.append(ctx.last_kernel_ms())     # HIP events on the handle's stream
    ctx.synchronize()
    cp.barrier()
    elapsed = time.perf_counter() - t0

    ctr = ctx.counters()
    elapsed = cp.reduce(elapsed, 'MAX')
    work_all = cp.reduce(ctr['particle_steps'], 'SUM')
    samples_all = cp.reduce(ctr['samples'], 'SUM')
    sec_per_step = elapsed/args.steps
    value = work_all/sec_per_step

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        achieved = ALGO_BYTES_PER_PARTICLE_STEP*ctr['particle_steps']/(k_ms*1e-3)/1e9
        traffic = None
        secondary = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                prof = json.load(open(tfile))
                traffic = prof.get('k_const_fused_bytes_per_launch')
                # the ceilings that actually bind this kernel, from the committed PMC profile:
                # fp64 VALU issue (one wave64 fp64 instruction per 4 cycles per SIMD) and the
                # memory-side scattered-atomic request rate (tools/ubench_atomics.hip)
                cus = 256
                insts = prof.get('k_const_fused_valu_wave_insts_per_launch')
                atoms = prof.get('k_const_fused_atomic_requests_per_launch')
                if insts and atoms:
                    valu_floor_ms = insts/(cus*4*2.4e9/4)*1e3
                    atomic_floor_ms = atoms/2.4e10*1e3
                    secondary = {'valu_wave_insts_per_launch': insts,
                                 'valu_issue_floor_ms': valu_floor_ms,
                                 'valu_issue_frac': valu_floor_ms/k_ms,
                                 'atomic_requests_per_launch': atoms,
                                 'atomic_floor_ms': atomic_floor_ms,
                                 'atomic_frac': atomic_floor_ms/k_ms,
                                 'source': 'profiles/' + str(prof.get('tag', '')) + '_pmc.json'}
            except Exception:
                traffic = None
        line = {
            'metric': 'particle*steps/s', 'value': value, 'unit': 'particle*steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': sec_per_step*1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'Na at Mercury (taa 1.3), gravity+radpres+photoionisation, '
                                   f'{args.packets} packets/GPU x {n_iter} steps of 30 s, fused '
                                   f'{args.dims}x{args.dims} {args.quantity} image '
                                   f'(BASELINE configs[2])',
                       'packets_per_gpu': args.packets, 'n_iter': n_iter, 'nsteps': nsteps,
                       'image': f'{args.dims}x{args.dims}', 'parallelism': f'packet-shard x{world}',
                       'image_reduce': reduce_mode},
            'particle_steps_per_pass': work_all, 'samples_per_pass': samples_all,
            'los_pixels_per_s': args.dims*args.dims*world/sec_per_step,
            'samples_per_s': samples_all/sec_per_step,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'k_const_fused<IMAGE>', 'kernel_ms': k_ms,
                         'algorithmic_bytes_per_particle_step': ALGO_BYTES_PER_PARTICLE_STEP,
                         'binding_ceilings': secondary,
                         'note': 'fused persistent kernel keeps packet state in registers: real '
                                 'HBM traffic is far below the algorithmic figure; the binding '
                                 'resource is fp64 VALU issue (see DESIGN.md)'},
            'device': ctx.device_name(),
        }
        if world == 1:
            # the same pass with the other image quantity (configs[2] words it as a "column"
            # image; the headline above uses the costlier radiance weighting), for reference
            other = 'column' if args.quantity != 'column' else 'radiance'
            with contextlib.redirect_stdout(io.StringIO()):
                img2 = ModelImage(inputs, dict(params, quantity=other), context=ctx)
            img2._set_image(ctx, aplanet, vrplanet, True)
            ms2 = []
            for it in range(3):
                ctx.image_clear()
                ctx.integrate_const_async(opt.step_size, n_iter, opt.outeredge, image=True)
                ctx.synchronize()
                if it:
                    ms2.append(ctx.last_kernel_ms())
            line['other_quantity'] = {'quantity': other, 'kernel_ms': float(np.mean(ms2)),
                                      'value': ctr['particle_steps']/(float(np.mean(ms2))*1e-3),
                                      'unit': 'particle*steps/s'}
        if world == 1 and not args.no_cpu_baseline:
            with contextlib.redirect_stdout(io.StringIO()):
                line['cpu_baseline'] = cpu_baseline(args, inputs)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if reduce_mode == 'rccl-allreduce':
        ctx.comm_destroy()
    ctx.close()
    cp.close()


if __name__ == '__main__':
    main()
