#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Workload (BASELINE.json configs[2]): Na at Mercury, taa = 1.3, gravity +
radiation pressure + photoionisation, endtime 50000 s, constant step 30 s (1668 stored records),
outeredge 25 R, uniform/flat(2.5 +- 2 km/s)/isotropic source, 1e7 packets PER GPU, every stored
record binned into a 512 x 512 radiance image (width 8 x 8 R).  One "step" = one full pass: clear
the image, integrate all resident packets to the end with the fused persistent kernel, and (N > 1)
sum the per-GPU image pairs over RCCL.  Inputs (X0, tables) are resident in HBM before the timed
region; ``ms_per_step_incl_h2d`` is the same pass preceded by the upload of X0 from host memory
and the on-device queue ordering, reported next to it.

metric = particle*steps/s, whole job: sum over ranks of rk5 steps actually taken (active packets
only, Output.py:385) / max-over-ranks wall time.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --mode variable            # the adaptive-step driver (Output.py:221-366)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Packets shard by index with no data-path collective except the image sum (weak scaling: the
per-GPU packet count is fixed; rank r owns chunk r of the global chunk grid, i.e. the packets the
host sampler draws from seed 1234 + r -- nexoclom_amd.distributed.chunk_plan).  torch is not
imported: ``torch.distributed.run`` is only the launcher whose RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*
environment is read; rendezvous of the RCCL unique id goes over the package's TCP control plane,
and barrier / max / sum of scalars as well as the image reduce are RCCL calls made from
libnexoclom_hip.so on the handle's stream.  If the communicator cannot be created the bench
prints ``"value": null`` with the reason and exits non-zero: nothing else is timed in its place.
"""
import argparse
import contextlib
import io
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
SEED = 1234


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--packets', type=int, default=None,
                   help='packets per GPU (default 1e7 for both drivers)')
    p.add_argument('--dims', type=int, default=512)
    p.add_argument('--quantity', default='radiance')
    p.add_argument('--mode', choices=('constant', 'variable'), default='constant')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-extras', action='store_true',
                   help='skip the untimed side measurements (other quantity, variable step)')
    p.add_argument('--cpu-packets', type=int, default=150_000)
    return p.parse_args()


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def cpu_baseline(args, inputs, variable=False):
    """The reference's CPU path, timed beside the GPU number on the same box: the NumPy oracle's
    constant-step driver (bit-identical to the reference's rk5/state arithmetic) on a bounded
    sample of the same workload, one process = one core; plus, for context, the C oracle on all
    host threads.  variable=True: the oracle's adaptive driver on 8000 packets instead."""
    from nexoclom_amd import Output
    from nexoclom_amd.Output import n_output_steps
    from oracle import np_oracle as O
    from oracle.c_oracle import COracle
    from tests import helpers as H
    if variable:
        n = 8000
        out = Output(inputs, n, seed=4242, integrate=False, save=False)
        X0 = np.ascontiguousarray(out.x0_soa().T)
        f = H.mercury_forces('Na', 1.3)
        t0 = time.time()
        _, _, work = O.variable_step_driver(f, X0, float(inputs.options.resolution),
                                            inputs.options.outeredge)
        t_np = time.time() - t0
        return {'value': work/t_np, 'unit': 'rk5 attempts/s', 'cores': 1, 'kind': 'port',
                'sample': f'NumPy oracle variable-step driver (reference arithmetic), {n} '
                          f'packets, {work} rk5 attempts in {t_np:.1f} s'}
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


def profile_ceilings(k_ms):
    """HBM traffic and the two ceilings that bind the fused kernel, from the committed PMC
    profile (profiles/traffic.json): fp64 VALU issue (one wave64 fp64 instruction per 4 cycles
    per SIMD) and the memory-side scattered-atomic request rate (tools/ubench_atomics.hip)."""
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(tfile):
        return None, None
    try:
        prof = json.load(open(tfile))
    except (OSError, ValueError):
        return None, None
    traffic = prof.get('k_const_fused_bytes_per_launch')
    insts = prof.get('k_const_fused_valu_wave_insts_per_launch')
    atoms = prof.get('k_const_fused_atomic_requests_per_launch')
    secondary = None
    if insts and atoms:
        valu_floor_ms = insts/(256*4*2.4e9/4)*1e3
        atomic_floor_ms = atoms/2.4e10*1e3
        secondary = {'valu_wave_insts_per_launch': insts, 'valu_issue_floor_ms': valu_floor_ms,
                     'valu_issue_frac': valu_floor_ms/k_ms,
                     'atomic_requests_per_launch': atoms, 'atomic_floor_ms': atomic_floor_ms,
                     'atomic_frac': atomic_floor_ms/k_ms,
                     'source': 'profiles/' + str(prof.get('tag', '')) + '_pmc.json'}
    return traffic, secondary


def variable_leg(ctx, inputs_var, n, passes=3):
    """The adaptive-step driver over n resident packets: rk5 attempts/s from HIP events."""
    from nexoclom_amd import Output
    with quiet():
        out = Output(inputs_var, n, seed=SEED, integrate=False, save=False, context=ctx)
    ctx.set_forces(**out.forces_kwargs())
    ctx.set_bodies(None)
    ctx.set_bounce(None)
    ctx.upload_soa(out.x0_soa())
    ms, call = [], []
    for it in range(passes + 1):
        t0 = time.perf_counter()
        ctx.integrate_var(float(inputs_var.options.resolution), inputs_var.options.outeredge)
        dt = time.perf_counter() - t0
        if it:
            ms.append(ctx.last_kernel_ms())
            call.append(dt*1e3)
    ctr = ctx.counters()
    k_ms = float(np.mean(ms))
    return {'packets': n, 'rk5_attempts': ctr['particle_steps'], 'kernel': 'k_var',
            'kernel_ms': k_ms, 'call_ms_incl_d2h': float(np.mean(call)),
            'value': ctr['particle_steps']/(k_ms*1e-3), 'unit': 'rk5 attempts/s',
            'unfinished': ctr['unfinished'],
            'roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
                         'achieved': ALGO_BYTES_PER_PARTICLE_STEP*ctr['particle_steps']/(k_ms*1e-3)/1e9,
                         'frac': ALGO_BYTES_PER_PARTICLE_STEP*ctr['particle_steps']/(k_ms*1e-3)/1e9/HBM_PEAK_GBS}}


def fail_line(args, world, reason):
    return {'metric': 'particle*steps/s', 'value': None, 'unit': 'particle*steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic', 'error': reason}


def main():
    args = parse()
    from nexoclom_amd import Input, Output, ModelImage, hip_api
    from nexoclom_amd.distributed import ControlPlane, chunk_plan, pick_device
    from nexoclom_amd.Output import n_output_steps
    world = int(os.environ.get('WORLD_SIZE', '1')) if 'RANK' in os.environ else 1
    cp = ControlPlane(world)
    rank = cp.rank
    variable = args.mode == 'variable'
    packets = args.packets or 10_000_000

    infile = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
    inputs = Input(infile)
    inputs_var = Input(infile)
    inputs_var.options.step_size = 0.
    inputs_var.options.resolution = 1e-4
    opt = inputs.options
    nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)

    if hip_api.device_count() < 1:
        raise SystemExit('bench.py needs a HIP device; there is no CPU fallback')
    # every rank needs a device of its own; the ranks agree on that before anything is set up
    ctx, err = None, ''
    try:
        ctx = hip_api.Context(pick_device(cp))
    except hip_api.HipError as exc:
        err = str(exc)
    problems = [p.decode() for p in cp.allgather_bytes(err.encode())]
    if any(problems):
        if rank == 0:
            print(json.dumps(fail_line(args, world, '; '.join(p for p in problems if p))))
        sys.stderr.write(f'[bench rank {rank}] {err or "another rank has no device"}\n')
        cp.close()
        sys.exit(1)

    # ---- set-up (untimed): this rank's shard = chunk `rank` of the global chunk grid ----------
    (k, c0, clen, a, b), = chunk_plan(packets*world, packets, rank*packets, (rank + 1)*packets)
    assert (k, clen, a, b) == (rank, packets, c0, c0 + packets)
    run_inputs = inputs_var if variable else inputs
    with quiet():
        out = Output(run_inputs, packets, seed=SEED + k, integrate=False, save=False, context=ctx)
        params = {'quantity': args.quantity, 'dims': f'{args.dims},{args.dims}', 'width': '8,8',
                  'center': '0,0'}
        img = ModelImage(inputs, params, context=ctx)      # parses params; no packets yet
    ctx.set_forces(**out.forces_kwargs())
    aplanet, vrplanet = out.aplanet, out.vrplanet
    img._set_image(ctx, aplanet, vrplanet, True)
    x0 = out.x0_soa()
    ctx.upload_soa(x0)
    ctx.set_first_index(a)
    del out

    reduce_mode = 'none'
    if world > 1:
        try:
            cp.init_rccl(ctx)
        except hip_api.HipError as err:
            if rank == 0:
                print(json.dumps(fail_line(args, world, str(err))))
            sys.stderr.write(f'[bench rank {rank}] {err}\n')
            cp.close()
            ctx.close()
            sys.exit(1)
        reduce_mode = 'rccl-allreduce'

    def job_barrier():
        if world > 1:
            ctx.barrier()           # RCCL all-reduce of one double on the handle's stream

    def one_step():
        if variable:
            ctx.integrate_var(float(run_inputs.options.resolution), opt.outeredge)
            return
        ctx.image_clear()
        ctx.integrate_const_async(opt.step_size, n_iter, opt.outeredge, image=True)
        if world > 1:
            ctx.image_allreduce()

    for _ in range(args.warmup):
        one_step()
        ctx.synchronize()

    kernel_ms = []
    job_barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        ctx.synchronize()
        kernel_ms.append(ctx.last_kernel_ms())     # HIP events on the handle's stream
    ctx.synchronize()
    job_barrier()
    elapsed = time.perf_counter() - t0

    ctr = ctx.counters()
    if world > 1:
        elapsed = ctx.allreduce_max(elapsed)
        work_all = ctx.allreduce_sum(float(ctr['particle_steps']))
        samples_all = ctx.allreduce_sum(float(ctr['samples']))
    else:
        work_all, samples_all = float(ctr['particle_steps']), float(ctr['samples'])
    sec_per_step = elapsed/args.steps
    value = work_all/sec_per_step

    # ---- the same pass with X0 coming from host memory (untimed for the headline) ------------
    incl = []
    for _ in range(2):
        job_barrier()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.upload_soa(x0)                  # H2D of 64 B/packet + on-device queue ordering
        ctx.set_first_index(a)
        one_step()
        ctx.synchronize()
        job_barrier()
        incl.append(time.perf_counter() - t0)
    incl_s = min(incl)
    if world > 1:
        incl_s = ctx.allreduce_max(incl_s)
    del x0

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        achieved = ALGO_BYTES_PER_PARTICLE_STEP*ctr['particle_steps']/(k_ms*1e-3)/1e9
        traffic, secondary = (None, None) if variable else profile_ceilings(k_ms)
        unit = 'rk5 attempts/s' if variable else 'particle*steps/s'
        if variable:
            workload = (f'Na at Mercury (taa 1.3), gravity+radpres+photoionisation, {packets} '
                        f'packets/GPU at random ages, variable-step driver, resolution 1e-4 '
                        f'(BASELINE configs[1] forces, Output.py:221-366)')
            kernel = 'k_var'
        else:
            workload = (f'Na at Mercury (taa 1.3), gravity+radpres+photoionisation, {packets} '
                        f'packets/GPU x {n_iter} steps of 30 s, fused {args.dims}x{args.dims} '
                        f'{args.quantity} image (BASELINE configs[2])')
            kernel = 'k_const_fused<IMAGE>'
        line = {
            'metric': 'particle*steps/s', 'value': value, 'unit': unit,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': sec_per_step*1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload, 'mode': args.mode, 'packets_per_gpu': packets,
                       'n_iter': n_iter, 'nsteps': nsteps, 'image': f'{args.dims}x{args.dims}',
                       'parallelism': f'packet-shard x{world}', 'image_reduce': reduce_mode,
                       'control_plane': 'tcp+rccl' if world > 1 else 'none'},
            'particle_steps_per_pass': work_all, 'samples_per_pass': samples_all,
            'ms_per_step_incl_h2d': incl_s*1e3,
            'value_incl_h2d': work_all/incl_s,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': kernel, 'kernel_ms': k_ms,
                         'algorithmic_bytes_per_particle_step': ALGO_BYTES_PER_PARTICLE_STEP,
                         'binding': 'fp64 VALU issue', 'binding_ceilings': secondary,
                         'note': 'persistent kernel keeps packet state in registers: real HBM '
                                 'traffic is far below the algorithmic figure; the binding '
                                 'resource is fp64 VALU issue (see DESIGN.md)'},
            'device': ctx.device_name(),
        }
        if not variable:
            line['los_pixels_per_s'] = args.dims*args.dims*world/sec_per_step
            line['samples_per_s'] = samples_all/sec_per_step
        if world == 1 and not variable and not args.no_extras:
            # the same pass with the other image quantity (configs[2] words it as a "column"
            # image; the headline above uses the costlier radiance weighting), for reference
            other = 'column' if args.quantity != 'column' else 'radiance'
            with quiet():
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
            # the adaptive-step driver (a-4), for the record: at the reference's chunk of 1e6
            # packets (Input.py:218; five packets per lane -- the kernel then lasts as long as its
            # longest packet's chain of attempts, profiles/r02_var_schedule.json) and at 1e7
            line['variable_step'] = variable_leg(ctx, inputs_var, 1_000_000)
            line['variable_step_1e7'] = variable_leg(ctx, inputs_var, 10_000_000, passes=2)
        if world == 1 and not args.no_cpu_baseline:
            with quiet():
                line['cpu_baseline'] = cpu_baseline(args, inputs_var if variable else inputs,
                                                    variable)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        ctx.comm_destroy()
    ctx.close()
    cp.close()


if __name__ == '__main__':
    main()
