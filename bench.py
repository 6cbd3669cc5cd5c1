#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Workload (BASELINE.json configs[2]): Na at Mercury, taa = 1.3, gravity +
radiation pressure + photoionisation, endtime 50000 s, constant step 30 s (1668 stored records),
outeredge 25 R, uniform/flat(2.5 +- 2 km/s)/isotropic source, 1e7 packets PER GPU, every stored
record binned into a 512 x 512 radiance image (width 8 x 8 R).  One "step" = one full pass: clear
the image, integrate all resident packets to the end with the fused persistent kernel, and (N > 1)
sum the per-GPU image pairs over RCCL.  Inputs (X0, tables) are resident in HBM before the timed
region; ``value_incl_h2d`` / ``ms_per_step_incl_h2d`` is the same pass with X0 coming from host
memory through the streamed upload ("incl. H2D of X0"), reported next to it.

metric = particle*steps/s, whole job: sum over ranks of rk5 steps actually taken (active packets
only, Output.py:385) / max-over-ranks wall time.

Untimed extras of the N = 1 line (each with its own roofline object): the other image quantity,
stored samples through the LDS tiles at 512 x 512 and at the reference's default 800 x 800, 512
spacecraft lines of sight over the same stored samples, the adaptive driver at 1e6 and 1e7 packets.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --with-comm                # N = 1, but through every N > 1 branch (RCCL world of one)
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

``run_rank`` is everything one rank does; tests/test_bench_ranks.py runs it on two CPU processes
with the C oracle standing in for the device, so that no line of the N > 1 path meets hardware
unexecuted.
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
GUIDE_COPY_GBS = 6290.0                 # same guide: float4 copy, measured
SEED = 1234
INFILE = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--packets', type=int, default=None,
                   help='packets per GPU (default 1e7 for both drivers)')
    p.add_argument('--dims', type=int, default=512)
    p.add_argument('--quantity', default='radiance')
    p.add_argument('--mode', choices=('constant', 'variable'), default='constant')
    p.add_argument('--with-comm', action='store_true',
                   help='create the RCCL communicator even for one rank and take every N > 1 '
                        'branch (image all-reduce inside each timed pass, RCCL barrier / max / '
                        'sum); the plain pass is timed beside it')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-extras', action='store_true',
                   help='skip the untimed side measurements (other quantity, variable step, '
                        'stream-copy ceiling, clock)')
    p.add_argument('--no-h2d-pass', action='store_true',
                   help='skip the pass that includes the upload of X0 (counter-mode profilers '
                        'serialise kernels, and the pipelined pass needs two running side by side)')
    p.add_argument('--cpu-packets', type=int, default=150_000)
    return p.parse_args(argv)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


@contextlib.contextmanager
def stdout_to_stderr():
    """File descriptor 1 points at stderr inside the block: librccl prints a version banner on
    stdout when a communicator is created, and stdout carries exactly one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


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


def profile_ceilings(k_ms, particle_steps, clock_mhz, wave_trips=None, binned=None):
    """HBM traffic and the ceilings that bind the fused kernel.

    Counted in THIS run where the kernel can count it itself: ``wave_trips`` (trips of a wave
    through the persistent step loop, nxc_counters.wave_trips) times the static VALU mix of one
    trip (profiles/cost_model.json, from tools/isa_census.py) gives the VALU wave-instructions;
    the binned samples are the memory-side atomic requests (one pair-atomic instruction per binned
    sample, image_add_pairs); ``lanes_active`` = particle_steps / (64 x wave_trips).  A regression
    in lane occupancy, refill behaviour or sample counts therefore shows in the driver's line.
    Only ``traffic`` (HBM bytes: FETCH_SIZE / WRITE_SIZE need the PMC counters) and
    ``valu_busy_frac_pmc`` come from the committed profile (profiles/traffic.json), scaled to
    this run's particle*steps and marked ``from_profile``.
    ``valu_issue_floor_ms``: per wave trip the cost model's fp64 instructions at one 4-cycle issue
    slot, v_rcp/v_rsq_f64 at 3.3 slots, 32-bit instructions at half a slot, at the clock THIS run
    held (``clock_mhz``: in-kernel stamps of a diagnostic launch, or the profile's)."""
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    cfile = os.path.join(ROOT, 'profiles', 'cost_model.json')
    try:
        prof = json.load(open(tfile))
        model = json.load(open(cfile))
    except (OSError, ValueError):
        return None, None
    scale = particle_steps/float(prof.get('particle_steps_per_launch', particle_steps))
    traffic = prof.get('k_const_fused_bytes_per_launch')
    mix = model['per_wave_trip']
    per_trip = mix['fp64'] + mix['transcendental'] + mix['valu32']
    if wave_trips:
        insts, insts_src = wave_trips*per_trip, 'wave trips counted in this run x the static ' \
            f'per-trip VALU mix of profiles/cost_model.json ({per_trip} instructions)'
    else:
        insts = prof.get('k_const_fused_valu_wave_insts_per_launch')
        insts = insts*scale if insts else None
        insts_src = 'profiles/' + str(prof.get('tag', '')) + '_pmc.json, scaled'
    if binned is not None:
        atoms, atoms_src = float(binned), 'samples binned in this run (one pair-atomic each)'
    else:
        atoms = prof.get('k_const_fused_atomic_requests_per_launch')
        atoms = atoms*scale if atoms else None
        atoms_src = 'profiles/' + str(prof.get('tag', '')) + '_pmc.json, scaled'
    if not (insts and atoms):
        return traffic, None
    slots = (mix['fp64'] + mix['transcendental']*model['slot_cost']['transcendental']
             + mix['valu32']*model['slot_cost']['valu32'])
    per_inst = slots/per_trip
    clock = (clock_mhz or model['profile_clock_mhz'])*1e6
    n_simd = 256*4
    valu_floor_ms = insts*per_inst*4/(n_simd*clock)*1e3
    atomic_floor_ms = atoms/model['atomic_requests_per_s']*1e3
    secondary = {'valu_wave_insts_per_launch': insts, 'valu_wave_insts_source': insts_src,
                 'wave_trips': wave_trips,
                 'lanes_active': particle_steps/(64.0*wave_trips) if wave_trips else None,
                 'issue_slots_per_instruction': per_inst, 'clock_mhz': clock/1e6,
                 'clock_source': 'in-kernel stamps, this run' if clock_mhz else 'profile',
                 'valu_issue_floor_ms': valu_floor_ms, 'valu_issue_frac': valu_floor_ms/k_ms,
                 'valu_busy_frac_pmc': prof.get('k_const_fused_valu_busy_frac'),
                 'valu_busy_frac_pmc_from_profile': True,
                 'atomic_requests_per_launch': atoms, 'atomic_requests_source': atoms_src,
                 'atomic_floor_ms': atomic_floor_ms, 'atomic_frac': atomic_floor_ms/k_ms,
                 'source': 'this run (wave trips, binned samples, clock); profiles/' +
                           str(prof.get('tag', '')) + '_pmc.json for valu_busy_frac_pmc; '
                           'profiles/cost_model.json for the per-trip mix and slot costs'}
    return (traffic*scale if traffic else None), secondary


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
    ach = ALGO_BYTES_PER_PARTICLE_STEP*ctr['particle_steps']/(k_ms*1e-3)/1e9
    return {'packets': n, 'rk5_attempts': ctr['particle_steps'], 'kernel': 'k_var',
            'kernel_ms': k_ms, 'call_ms_incl_d2h': float(np.mean(call)),
            'value': ctr['particle_steps']/(k_ms*1e-3), 'unit': 'rk5 attempts/s',
            'unfinished': ctr['unfinished'],
            'roofline': {'bound': 'valu', 'contract_bound': 'hbm', 'unit': 'GB/s',
                         'peak': HBM_PEAK_GBS, 'achieved': ach, 'frac': ach/HBM_PEAK_GBS}}


def stored_samples_leg(ctx, inputs, imgs, aplanet, vrplanet, quantity, n=1_000_000):
    """a-6..a-8 over STORED samples (the reference's two-stage flow, ModelImage.py:229-274): the
    float32 rows of n packets stay in HBM (what Input.run leaves there) and are binned by k_image
    (one global atomic pair per binned sample) and by the tiled image (k_image_bin +
    k_image_tiles); kernel time by HIP events, 40 / 32 algorithmic bytes per sample.
    ``imgs``: ModelImages by label -- BASELINE's 512 x 512 and the reference's default 800 x 800
    (ModelImage.py:53) over the same rows."""
    from nexoclom_amd import Output
    from nexoclom_amd.Output import n_output_steps
    opt = inputs.options
    _, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
    with quiet():
        out = Output(inputs, n, seed=SEED, integrate=False, save=False, context=ctx)
    ctx.set_forces(**out.forces_kwargs())
    ctx.upload_soa(out.x0_soa())
    store = ctx.integrate_const_rows(opt.step_size, n_iter, opt.outeredge, narrow=True,
                                     resident=True)['store']
    per_sample = 40 if quantity == 'radiance' else 32
    legs = {}
    for label, img in imgs.items():
        img._set_image(ctx, aplanet, vrplanet, False)
        leg = {'samples': int(store.total), 'quantity': quantity, 'image': label,
               'algorithmic_bytes_per_sample': per_sample}
        for mode in ('atomics', 'tiles'):
            ctx.image_mode(mode)
            ms = []
            for it in range(4):
                ctx.image_clear()
                ctx.image_accumulate_rows(store)
                if it:
                    ms.append(ctx.last_kernel_ms())
            k_ms = float(np.mean(ms))
            ach = per_sample*store.total/(k_ms*1e-3)/1e9
            leg[mode] = {'kernel': 'k_image' if mode == 'atomics' else 'k_image_bin + k_image_tiles',
                         'kernel_ms': k_ms, 'samples_per_s': store.total/(k_ms*1e-3),
                         'pixels_per_s': img.dims[0]*img.dims[1]/(k_ms*1e-3),
                         'binned': ctx.counters()['samples_binned'],
                         'roofline': {'bound': 'atomic requests' if mode == 'atomics' else 'hbm',
                                      'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'achieved': ach,
                                      'frac': ach/HBM_PEAK_GBS}}
        ctx.image_mode('auto')
        try:        # HBM bytes really moved by the two tile passes (PMC, profiles/traffic.json)
            per = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
            key = 'image_tiles' if label == '512x512' else 'image_tiles_' + label
            leg['tiles']['roofline']['traffic'] = per[key + '_bytes_per_sample']*leg['samples']
            leg['tiles']['roofline']['traffic_from_profile'] = True
            leg['tiles']['roofline']['traffic_source'] = per[key + '_source']
        except (OSError, ValueError, KeyError):
            leg['tiles']['roofline']['traffic'] = None
        legs[label] = leg
    legs['_store'] = (store, out)
    return legs


def synthetic_orbit(nspec, seed=0):
    """Spacecraft positions on an eccentric ring (1.6 .. 2.9 R) with boresights in assorted
    directions (a third roughly planetward): the MESSENGERuvvs stand-in of the kernel benches."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0, 2*np.pi, nspec, endpoint=False)
    r = 1.6 + 1.3*np.cos(th)**2
    pos = np.stack([0.3*r*np.cos(th), r*np.sin(th)*0.6 - 0.4, r*np.sin(th)*0.8], 1)
    look = rng.normal(size=(nspec, 3))
    look[::3] = -pos[::3] + 0.9*rng.normal(size=(len(pos[::3]), 3))
    look /= np.linalg.norm(look, axis=1)[:, None]
    return pos, look


def line_of_sight_leg(ctx, inputs, store, out, n_spectra=512):
    """f-1 over STORED samples (data_simulation/compute_iteration.py:90-240): the resident float32
    rows of the stored-samples leg against `n_spectra` synthetic lines of sight of 1 degree
    half-angle; kernel time (k_los_blocks + k_los + k_los_pairs) by HIP events."""
    from nexoclom_amd import LOSResult, SpacecraftData
    from nexoclom_amd.LOSResult import arccos_threshold, los_geometry
    pos, look = synthetic_orbit(n_spectra)
    sc = SpacecraftData(pos[:, 0], pos[:, 1], pos[:, 2], look[:, 0], look[:, 1], look[:, 2])
    with quiet():
        los = LOSResult(sc, inputs, dphi=np.radians(1.0), context=ctx)
    dist, lengths, ladder = los_geometry(sc.data, inputs.options.outeredge, los.dphi)
    scarr = np.stack([sc.data.x, sc.data.y, sc.data.z, sc.data.xbore, sc.data.ybore,
                      sc.data.zbore, dist, lengths.astype(float)])
    args = (los.dphi, np.sin(los.dphi), np.sin(2*los.dphi), arccos_threshold(los.dphi),
            float(out.vrplanet_Rs()), out.unit_km*1e5, los.g_tables(float(out.aplanet)), ladder,
            scarr)
    ms = []
    for it in range(4):
        res = ctx.los_accumulate(*args, rows=(store, 0, store.total, 0), n_index=out.npackets)
        if it:
            ms.append(ctx.last_kernel_ms())
    k_ms = float(np.mean(ms))
    P, S = int(store.total), n_spectra
    ach = 40.0*P/(k_ms*1e-3)/1e9
    return {'samples': P, 'spectra': S, 'kernel': 'k_los_blocks + k_los + k_los_pairs',
            'kernel_ms': k_ms, 'pairs_decided_per_s': P*S/(k_ms*1e-3),
            'samples_per_s': P/(k_ms*1e-3), 'pairs_inside_cones': int(res['npackets'].sum()),
            'bounding_sphere_tests': int(ctx.counters()['samples']),
            'roofline': {'bound': 'valu (bounding-sphere tests, 58 % busy) beside wave-synchronous LDS stages', 'contract_bound': 'hbm',
                         'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'achieved': ach,
                         'frac': ach/HBM_PEAK_GBS,
                         'note': '40 algorithmic bytes per stored sample, read once for all '
                                 f'{S} spectra of the launch'}}


def fail_line(args, world, reason):
    return {'metric': 'particle*steps/s', 'value': None, 'unit': 'particle*steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic', 'error': reason}


def run_rank(args, cp, make_context, emit=print, hard_exit_after=None):
    """Everything one rank of the bench does.  ``cp``: the control plane (rank, world,
    allgather_bytes, init_rccl, watch, announce_failure, close); ``make_context()``: this rank's
    device context (raises hip_api.HipError when there is none).  Rank 0 hands the JSON line to
    ``emit``.  Returns the process exit code.

    No rank can hang the job: every wait on a collective carries a deadline (nxc_comm_set_timeout,
    NXC_COLLECTIVE_TIMEOUT_S, default 120 s) after which the communicator is aborted and the
    call raises; a rank that raises -- anything, anywhere after the control plane is up -- tells
    the others through the control plane's failure channel, whose watcher ends their wait at
    once (``comm_request_abort``); a rank that dies without a word is noticed by its closed
    socket.  Rank 0 then prints the ``"value": null`` line with the reason and every rank returns
    1.  ``hard_exit_after`` (seconds; main() sets it): a rank whose main thread still has not
    come back that long after a peer's failure was announced (it sits in a call that cannot be
    interrupted, e.g. ncclCommInitRank) prints the line and leaves with os._exit(1); so does a
    job of several ranks that has not finished after NXC_BENCH_DEADLINE_S (default 420 s)."""
    import traceback
    state = {'ctx': None, 'emitted': False, 'done': False}

    def emit_once(text):
        if not state['emitted']:
            state['emitted'] = True
            emit(text)

    def leave_now(reason):
        if state['done']:
            return
        if cp.rank == 0:
            emit_once(json.dumps(fail_line(args, cp.world, reason)))
        sys.stderr.write(f'[bench rank {cp.rank}] leaving: {reason}\n')
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)

    def on_peer_failure(reason):
        ctx = state['ctx']
        if ctx is not None and hasattr(ctx, 'comm_request_abort'):
            ctx.comm_request_abort()
        if hard_exit_after is not None:
            import threading
            timer = threading.Timer(hard_exit_after, leave_now, args=(reason,))
            timer.daemon = True
            timer.start()

    limit = float(os.environ.get('NXC_BENCH_DEADLINE_S', '420' if cp.world > 1 else '0'))
    if hard_exit_after is not None and limit > 0:
        # last line of defence for a job of several ranks: whatever else fails to end it
        import threading
        watchdog = threading.Timer(limit, leave_now,
                                   args=(f'the bench did not finish within {limit:.0f} s',))
        watchdog.daemon = True
        watchdog.start()
    try:
        rc = _run_rank(args, cp, make_context, emit_once, state, on_peer_failure)
        state['done'] = True
        return rc
    except Exception as exc:                        # noqa: BLE001 -- whatever it is, nobody hangs
        why = f'{type(exc).__name__}: {exc}'
        peer = getattr(cp, 'failure', None)
        if peer and peer not in why:
            why = f'{peer} (seen on rank {cp.rank} as {why})'
        sys.stderr.write(f'[bench rank {cp.rank}] {why}\n{traceback.format_exc()}')
        if hasattr(cp, 'announce_failure'):
            cp.announce_failure(why)
        ctx = state['ctx']
        if ctx is not None and hasattr(ctx, 'comm_abort'):
            try:
                ctx.comm_abort()
            except Exception:                       # noqa: BLE001
                pass
        if cp.rank == 0:
            emit_once(json.dumps(fail_line(args, cp.world, why)))
        state['done'] = True
        cp.close()
        return 1


def _run_rank(args, cp, make_context, emit, state, on_peer_failure):
    from nexoclom_amd import Input, Output, ModelImage, hip_api
    from nexoclom_amd.distributed import chunk_plan
    from nexoclom_amd.Output import n_output_steps
    world, rank = cp.world, cp.rank
    variable = args.mode == 'variable'
    packets = args.packets or 10_000_000
    comm = world > 1 or args.with_comm

    inputs = Input(INFILE)
    inputs_var = Input(INFILE)
    inputs_var.options.step_size = 0.
    inputs_var.options.resolution = 1e-4
    opt = inputs.options
    nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)

    # every rank needs a device of its own; the ranks agree on that before anything is set up
    ctx, err = None, ''
    try:
        ctx = make_context()
    except hip_api.HipError as exc:
        err = str(exc)
    state['ctx'] = ctx
    if hasattr(cp, 'watch'):
        cp.watch(on_peer_failure)
    problems = [p.decode() for p in cp.allgather_bytes(err.encode())]
    if any(problems):
        if rank == 0:
            emit(json.dumps(fail_line(args, world, '; '.join(p for p in problems if p))))
        sys.stderr.write(f'[bench rank {rank}] {err or "another rank has no device"}\n')
        if ctx is not None:
            ctx.close()
        cp.close()
        return 1

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
    if comm:
        try:
            with stdout_to_stderr():
                cp.init_rccl(ctx)
        except hip_api.HipError as exc:
            if rank == 0:
                emit(json.dumps(fail_line(args, world, str(exc))))
            sys.stderr.write(f'[bench rank {rank}] {exc}\n')
            cp.close()
            ctx.close()
            return 1
        reduce_mode = 'rccl-allreduce'

    def job_barrier(on):
        if on:
            ctx.barrier()           # RCCL all-reduce of one double on the handle's stream

    def one_step(on):
        if variable:
            ctx.integrate_var(float(run_inputs.options.resolution), opt.outeredge)
            return
        ctx.image_clear()
        ctx.integrate_const_async(opt.step_size, n_iter, opt.outeredge, image=True)
        if on:
            ctx.image_allreduce()

    def timed(on, steps):
        """K passes bracketed by barrier + synchronize on both sides."""
        kernel_ms = []
        job_barrier(on)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step(on)
            ctx.synchronize()
            kernel_ms.append(ctx.last_kernel_ms())     # HIP events on the handle's stream
        ctx.synchronize()
        job_barrier(on)
        return time.perf_counter() - t0, kernel_ms

    for _ in range(args.warmup):
        one_step(comm)
        ctx.synchronize()
    elapsed, kernel_ms = timed(comm, args.steps)

    ctr = ctx.counters()
    trips = ctx.wave_trips() if hasattr(ctx, 'wave_trips') else None
    if comm:
        elapsed = ctx.allreduce_max(elapsed)
        work_all = ctx.allreduce_sum(float(ctr['particle_steps']))
        samples_all = ctx.allreduce_sum(float(ctr['samples']))
    else:
        work_all, samples_all = float(ctr['particle_steps']), float(ctr['samples'])
    sec_per_step = elapsed/args.steps
    value = work_all/sec_per_step

    # --with-comm on one rank: the plain pass beside it, so the line shows what the collectives cost
    plain = None
    if comm and world == 1:
        plain_s, _ = timed(False, args.steps)
        plain = plain_s/args.steps

    # ---- the same pass with X0 coming from host memory (never the headline) ---
    incl, pipeline = [], 'sequential: upload, then the pass' if variable else \
        'pipelined: one persistent launch consumes the queue while it crosses PCIe'
    for _ in range(0 if args.no_h2d_pass else 2):
        job_barrier(comm)
        ctx.synchronize()
        t0 = time.perf_counter()
        if variable:
            ctx.upload_soa(x0)              # H2D of 64 B/packet + on-device queue ordering
            one_step(comm)
        else:
            # the pipelined pass: piece p + 1 crosses PCIe and is ordered while piece p is integrated
            ctx.image_clear()
            ctx.integrate_const_streamed(x0, opt.step_size, n_iter, opt.outeredge, image=True)
            if comm:
                ctx.image_allreduce()
        stalled = False
        try:
            ctx.synchronize()
        except hip_api.HipError as exc:
            if getattr(exc, 'code', None) != hip_api.NXC_ERR_INCOMPLETE:
                raise
            stalled = True
        job_barrier(comm)
        incl.append(time.perf_counter() - t0)
        if stalled:
            # the kernel gave up waiting for its queue: the ordering kernels did not get to run
            # beside it (a profiler in counter mode serialises kernels).  Say so and time the two
            # steps one after the other instead.
            pipeline = 'sequential: upload, then the pass (the pipelined pass stalled -- are ' \
                       'kernels being serialised by a profiler?)'
            job_barrier(comm)
            t0 = time.perf_counter()
            ctx.upload_soa(x0)
            one_step(comm)
            ctx.synchronize()
            job_barrier(comm)
            incl[-1] = time.perf_counter() - t0
    if incl and not variable:
        ctr_incl = ctx.counters()
        assert ctr_incl['particle_steps'] == ctr['particle_steps'], 'the H2D pass is another run'
        ctx.set_first_index(a)
    incl_s = min(incl) if incl else float('nan')
    if comm and incl:
        incl_s = ctx.allreduce_max(incl_s)
    del x0

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        achieved = ALGO_BYTES_PER_PARTICLE_STEP*ctr['particle_steps']/(k_ms*1e-3)/1e9
        extras = world == 1 and not args.no_extras
        clock_mhz = ctx.shader_clock_mhz() if extras and hasattr(ctx, 'shader_clock_mhz') else None
        copy_gbs = ctx.stream_copy_gbs() if extras and hasattr(ctx, 'stream_copy_gbs') else None
        traffic, secondary = (None, None) if variable else \
            profile_ceilings(k_ms, ctr['particle_steps'], clock_mhz, trips,
                             ctr.get('samples_binned'))
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
                       'control_plane': 'tcp+rccl' if comm else 'none'},
            'particle_steps_per_pass': work_all, 'samples_per_pass': samples_all,
            # the pass including the host-to-device copy of X0
            'value_incl_h2d': work_all/incl_s if incl else None,
            'ms_per_step_incl_h2d': incl_s*1e3 if incl else None,
            'h2d_pass': pipeline if incl else 'skipped',
            'roofline': {'bound': 'valu', 'contract_bound': 'hbm',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'peak_measured': copy_gbs,
                         'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBS,
                         'frac_of_measured_peak': achieved/copy_gbs if copy_gbs else None,
                         # MI355X_MICROARCH.md's own float4-copy measurement, for a box-independent
                         # reading of the same fraction
                         'peak_guide_copy': GUIDE_COPY_GBS, 'frac_of_guide_copy': achieved/GUIDE_COPY_GBS,
                         'traffic': traffic, 'traffic_from_profile': traffic is not None,
                         'kernel': kernel, 'kernel_ms': k_ms,
                         'algorithmic_bytes_per_particle_step': ALGO_BYTES_PER_PARTICLE_STEP,
                         'binding': 'fp64 VALU issue', 'binding_ceilings': secondary,
                         'note': 'achieved/peak/frac are the HBM contract of SURVEY.md 8(d) '
                                 '(128 algorithmic bytes per particle*step against 8 TB/s; '
                                 'peak_measured = this box\'s streaming-copy rate).  The '
                                 'persistent kernel keeps the packet state in registers, so its '
                                 'real HBM traffic is far below the algorithmic figure and what '
                                 'binds it is fp64 VALU issue (bound; see DESIGN.md section 3)'},
            'device': ctx.device_name(),
        }
        if plain is not None:
            line['with_comm'] = {'ms_per_step_with_collectives': sec_per_step*1e3,
                                 'ms_per_step_plain': plain*1e3,
                                 'delta_ms_per_step': (sec_per_step - plain)*1e3,
                                 'note': 'RCCL communicator of one rank: image all-reduce in every '
                                         'timed pass, barrier / max / sum over RCCL'}
        if not variable:
            line['los_pixels_per_s'] = args.dims*args.dims*world/sec_per_step
            line['samples_per_s'] = samples_all/sec_per_step
        if extras and not variable:
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
            # a-6..a-8 over stored samples: the reference's two-stage flow (Input.produce_image)
            with quiet():
                img800 = ModelImage(inputs, dict(params, dims='800,800'), context=ctx)
            legs = stored_samples_leg(ctx, inputs, {f'{args.dims}x{args.dims}': img,
                                                    '800x800': img800}, aplanet, vrplanet,
                                      args.quantity)
            line['stored_samples_image'] = legs[f'{args.dims}x{args.dims}']
            # the reference's default image size (ModelImage.py:53)
            line['stored_samples_image_800'] = legs['800x800']
            # the other half of data_simulation: spacecraft lines of sight over the same rows
            store, out_rows = legs['_store']
            line['line_of_sight'] = line_of_sight_leg(ctx, inputs, store, out_rows)
            store.free()
            # the adaptive-step driver (a-4), for the record: at the reference's chunk of 1e6
            # packets (Input.py:218) and at 1e7, which is what Input.run launches at once
            line['variable_step'] = variable_leg(ctx, inputs_var, 1_000_000)
            line['variable_step_1e7'] = variable_leg(ctx, inputs_var, 10_000_000, passes=2)
        if world == 1 and not args.no_cpu_baseline:
            with quiet():
                line['cpu_baseline'] = cpu_baseline(args, inputs_var if variable else inputs,
                                                    variable)
        else:
            line['cpu_baseline'] = None
        emit(json.dumps(line))
    if comm:
        ctx.comm_destroy()
    ctx.close()
    cp.close()
    return 0


def main():
    args = parse()
    from nexoclom_amd import hip_api
    from nexoclom_amd.distributed import ControlPlane, pick_device
    world = int(os.environ.get('WORLD_SIZE', '1')) if 'RANK' in os.environ else 1
    rank = int(os.environ.get('RANK', '0')) if world > 1 else 0
    try:
        cp = ControlPlane(world, timeout=float(os.environ.get('NXC_CONTROL_TIMEOUT_S', '180')))
    except (TimeoutError, OSError, ValueError) as exc:
        # some rank never started: there is no job to measure
        if rank == 0:
            print(json.dumps(fail_line(args, world, f'control plane: {exc}')), flush=True)
        sys.stderr.write(f'[bench rank {rank}] control plane: {exc}\n')
        sys.exit(1)
    if hip_api.device_count() < 1:
        cp.announce_failure('no HIP device')
        if rank == 0:
            print(json.dumps(fail_line(args, world, 'no HIP device (there is no CPU fallback)')), flush=True)
        raise SystemExit('bench.py needs a HIP device; there is no CPU fallback')
    rc = run_rank(args, cp, lambda: hip_api.Context(pick_device(cp)), hard_exit_after=45.0)
    sys.stdout.flush()
    sys.stderr.flush()
    if rc:
        # a failed job leaves without tearing the device state down: a stream may still hold the
        # kernel of an aborted collective (a fresh exit -- nothing is re-executed)
        os._exit(rc)
    sys.exit(0)


if __name__ == '__main__':
    main()
