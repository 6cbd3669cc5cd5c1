#!/usr/bin/env python3
"""Timing + roofline line for every kernel of the hot path OTHER than the bench's fused one:

  k_image        a-6..a-8  stored samples -> image (40 B/sample radiance, 32 B column)
  k_var          a-4       adaptive-step driver (128 B per rk5 attempt)
  k_const_fused<ROWS>  a-3/f-3   compact trajectory rows (72 B per live record) -- one chunk and
                       the 13 chunks of Input.run(1e6) in one launch; the dense trajectory
  k_los          f-1       line-of-sight cones ((spectrum, sample) pair tests)
  k_sample       f-4       initial states on the device (64 B written per packet), Philox and PCG64
  k_stream_copy            the box's streaming-copy ceiling; k_clock: its clock under fp64 load
  k_speed_max / k_order_hist / k_order_scatter   queue order of the resident packets
  k_const_fused  a-1..a-3  the STRESS vector of SURVEY 8(d): every packet alive for all 1667 steps

One JSON line per kernel: HIP-event time of the launch on the handle's stream, units, the
algorithmic bytes of SURVEY.md section 8(d) (or of DESIGN.md section 3 for the rows this survey
has no figure for) and the fraction of the 8 TB/s HBM peak they amount to.  tools/profile_kernels.sh
runs this under rocprofv3 (kernel trace + PMC passes); profiles/ keeps the summaries.
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input, ModelImage, Output, hip_api          # noqa: E402
from nexoclom_amd.LOSResult import (LOSResult, SpacecraftData, arccos_threshold,   # noqa: E402
                                    los_geometry)
from nexoclom_amd.Output import n_output_steps                        # noqa: E402

HBM = 8000.0


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def line(kernel, ms, units, unit_name, bytes_per_unit, note=''):
    ach = bytes_per_unit*units/(ms*1e-3)/1e9
    print(json.dumps({'kernel': kernel, 'kernel_ms': ms, 'units': units, 'unit': unit_name,
                      'value': units/(ms*1e-3), 'algorithmic_bytes_per_unit': bytes_per_unit,
                      'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM, 'unit': 'GB/s',
                                   'frac': ach/HBM}, 'note': note}))


def synthetic_orbit(nspec, seed=0):
    rng = np.random.default_rng(seed)
    th = np.linspace(0, 2*np.pi, nspec, endpoint=False)
    r = 1.6 + 1.3*np.cos(th)**2
    pos = np.stack([0.3*r*np.cos(th), r*np.sin(th)*0.6 - 0.4, r*np.sin(th)*0.8], 1)
    look = rng.normal(size=(nspec, 3))
    look[::3] = -pos[::3] + 0.9*rng.normal(size=(len(pos[::3]), 3))
    look /= np.linalg.norm(look, axis=1)[:, None]
    return pos, look


def main():
    reps = 3
    infile = os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input')
    inputs = Input(infile)
    opt = inputs.options
    nsteps, n_iter = n_output_steps(opt.endtime.value, opt.step_size)
    ctx = hip_api.Context(0)

    # ---- stored samples of a reference-sized chunk (Input.py:219-222): rows, traj, image, LOS --
    n = 80467
    with quiet():
        out = Output(inputs, n, seed=1, integrate=False, save=False, context=ctx)
    ctx.set_forces(**out.forces_kwargs())
    ctx.set_bounce(None); ctx.set_bodies(None)
    ctx.upload_soa(out.x0_soa())
    for _ in range(reps):
        t0 = time.perf_counter()
        res = ctx.integrate_const_rows(opt.step_size, n_iter, opt.outeredge, narrow=True,
                                       resident=True)
        call_ms = (time.perf_counter() - t0)*1e3
        ms_rows = ctx.last_kernel_ms()          # pass 2: k_const_fused<ROWS>
        store = res['store']
        P = store.total
        rows = store.download(index=False)[0]
        if _ < reps - 1:
            store.free()
    los_store = store                           # stays in HBM for the line-of-sight line below
    work = ctx.counters()["particle_steps"]
    line('k_const_fused<ROWS> one chunk', ms_rows, P, 'live records written', 72,
         f'{n} packets (ONE reference chunk: fewer packets than the chip has lanes, so the launch '
         f'lasts as long as its longest packet), {P} live records of {n*nsteps} slots, {work} '
         f'particle-steps re-integrated; whole resident call (pass 1, offsets, pass 2, '
         f'transpose to float32 columns) {call_ms:.2f} ms')
    # all 13 chunks of Input.run(1e6) (Input.py:216-246) in ONE launch, as Input.run does here
    nchunks = 13
    with quiet():
        many = [Output(inputs, n, seed=1 + k, integrate=False, save=False, context=ctx).x0_soa()
                for k in range(nchunks)]
    ctx.upload_soa(np.concatenate(many, axis=1))
    del many
    for _ in range(reps):
        t0 = time.perf_counter()
        res = ctx.integrate_const_rows(opt.step_size, n_iter, opt.outeredge, narrow=True,
                                       resident=True)
        call_ms = (time.perf_counter() - t0)*1e3
        ms_batch = ctx.last_kernel_ms()
        Pb = res['store'].total
        if _ < reps - 1:
            res['store'].free()
    batch_store = res['store']                  # stays in HBM for the image lines below
    workb = ctx.counters()["particle_steps"]
    line(f'k_const_fused<ROWS> {nchunks} chunks in one launch', ms_batch, Pb,
         'live records written', 72,
         f'{nchunks} x {n} packets, {Pb} live records, {workb} particle-steps; per reference chunk '
         f'{ms_batch/nchunks:.3f} ms; whole resident call {call_ms:.2f} ms '
         f'({call_ms/nchunks:.2f} per chunk)')
    nt = 20000
    ctx.upload_soa(np.ascontiguousarray(out.x0_soa()[:, :nt]))
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.integrate_const(opt.step_size, n_iter, opt.outeredge, nrec=nsteps)
        call_ms = (time.perf_counter() - t0)*1e3
        ms_traj = ctx.last_kernel_ms()
    line('dense trajectory: k_const_fused<ROWS> + k_rows_densify', ms_traj, nt*nsteps,
         'record slots', 64,
         f'{nt} packets x {nsteps} records, dense (compress=False); kernel_ms = the rows pass; '
         f'whole call incl. the {8*nt*nsteps*8/1e9:.2f} GB device-to-host copy {call_ms:.1f} ms')

    x, y, z, vy, frac = (np.ascontiguousarray(rows[c]) for c in (1, 2, 3, 5, 7))
    # replicate the samples to a bench-sized stream (the image kernel is linear in the samples)
    k = max(1, int(2.6e7)//P)
    xs, ys, zs, vys, fs = (np.tile(a, k) for a in (x, y, z, vy, frac))
    for q in ('radiance', 'column'):
        with quiet():
            img = ModelImage(inputs, {'quantity': q, 'dims': '512,512'}, context=ctx)
        img._set_image(ctx, float(out.aplanet), float(out.vrplanet), False)
        per = 40 if q == 'radiance' else 32
        ctx.image_mode('atomics')
        for _ in range(reps):
            ctx.image_clear()
            ctx.image_accumulate(xs, ys, zs, vys, fs)
            ms = ctx.last_kernel_ms()
        c = ctx.counters()
        line(f'k_image[{q}]', ms, len(xs), 'samples', per,
             f'{c["samples_binned"]} of {c["samples"]} samples inside the 512x512 image; one '
             f'global atomic pair per binned sample')
        # the resident rows of the 13-chunk launch above, both ways
        for mode, name in (('atomics', f'k_image[{q}] resident rows'),
                           ('tiles', f'k_image_bin + k_image_tiles[{q}] resident rows')):
            ctx.image_mode(mode)
            for _ in range(reps):
                ctx.image_clear()
                ctx.image_accumulate_rows(batch_store)
                ms = ctx.last_kernel_ms()
            c = ctx.counters()
            line(name, ms, batch_store.total, 'samples', per,
                 f'{c["samples_binned"]} of {c["samples"]} samples inside the 512x512 image'
                 + ('; filed by image tile in LDS, summed per tile in LDS, one global atomic pair '
                    'per touched pixel and tile group' if mode == 'tiles' else ''))
        ctx.image_mode('auto')
    batch_store.free()

    S = 512
    pos, look = synthetic_orbit(S)
    sc = SpacecraftData(pos[:, 0], pos[:, 1], pos[:, 2], look[:, 0], look[:, 1], look[:, 2])
    with quiet():
        los = LOSResult(sc, inputs, dphi=np.radians(1.0), context=ctx)
    dist, lengths, ladder = los_geometry(sc.data, 25., los.dphi)
    scarr = np.stack([sc.data.x, sc.data.y, sc.data.z, sc.data.xbore, sc.data.ybore,
                      sc.data.zbore, dist, lengths.astype(float)])
    args = (los.dphi, np.sin(los.dphi), np.sin(2*los.dphi), arccos_threshold(los.dphi),
            float(out.vrplanet), out.unit_km*1e5, los.g_tables(float(out.aplanet)), ladder, scarr,
            x, y, z, vy, frac)
    # the resident rows with their packet-index column (what LOSResult hands over): blocks follow
    # the packets; and the same samples as bare host columns (no index: blocks of 8 rows as they come)
    for name, kw in (('k_los', dict(rows=(los_store, 0, los_store.total, 0), n_index=n)),
                     ('k_los[host columns, no index]', dict(zip(('x', 'y', 'z', 'vy', 'frac'),
                                                              (x, y, z, vy, frac))))):
        for _ in range(reps):
            r = ctx.los_accumulate(*args[:9], **kw)
            ms = ctx.last_kernel_ms()
        tests = ctx.counters()['samples']
        line(name, ms, P*S, '(sample, spectrum) pairs decided', 40.0/S,
             f'{P} samples x {S} spectra, {int(r["npackets"].sum())} pairs inside cones, '
             f'{tests} bounding-sphere tests (groups of 8 blocks, their halves, then blocks of 8 rows: one test '
             f'per {P*S/max(tests, 1):.0f} pairs; round 3 tested every (block, spectrum): '
             f'{(P + 7)//8*S}); the samples are read once by k_los_blocks (16 B) and again only where a '
             f'block passes a cone')
    los_store.free()

    # ---- bench-sized resident set: sampler, ordering, variable-step driver ---------------------
    N = 10_000_000
    with quiet():
        big = Output(inputs, 1000, seed=1, integrate=False, save=False, context=ctx)
    src = big.source_desc()
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.sample_packets(N, 1234, 0, **src)
        call_ms = (time.perf_counter() - t0)*1e3
    # last_kernel_ms of sample_packets times k_sample itself (the ordering follows it)
    line('k_sample', ctx.last_kernel_ms(), N, 'packets', 64,
         f'uniform/flat/isotropic source; whole call incl. queue ordering {call_ms:.2f} ms')
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.sample_packets(N, 1234, 0, pcg64=(N, 0), **src)
        call_ms = (time.perf_counter() - t0)*1e3
    line('k_sample[pcg64]', ctx.last_kernel_ms(), N, 'packets', 64,
         f"the same source following NumPy's PCG64 stream of the seed (128-bit jump-ahead per "
         f'packet); whole call incl. queue ordering {call_ms:.2f} ms')
    print(json.dumps({'kernel': 'k_stream_copy', 'GBps_read_plus_written': ctx.stream_copy_gbs(),
                      'clock_mhz_under_fp64_load': ctx.shader_clock_mhz(),
                      'note': "the box's own streaming ceiling and shader clock (bench.py prints "
                              'them as roofline.peak_measured / binding_ceilings.clock_mhz)'}))
    soa = ctx.sample_packets(2_000_000, 1234, 0, download=True, **src)
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.upload_soa(soa)
        up_ms = (time.perf_counter() - t0)*1e3
    print(json.dumps({'kernel': 'nxc_packets_upload', 'call_ms': up_ms, 'packets': soa.shape[1],
                      'note': 'H2D of 64 B/packet from pageable memory + k_speed_max + '
                              'k_order_hist + k_order_scan + k_order_scatter'}))

    # ---- SURVEY 8(d)'s stress vector: no early deaths, so no refill, no empty lanes, no queue
    # order -- the step loop's own rate.  Packets start at rest 30 R from the planet, outside its
    # shadow, and the outer edge is moved out of reach; radiation pressure carries them off.
    ns = 5*256*768          # five packets per resident lane: every lane does the same work
    rng = np.random.default_rng(5)
    phi = rng.uniform(0, 2*np.pi, ns)
    stress = np.zeros((8, ns))
    stress[0] = opt.endtime.value
    stress[1], stress[2], stress[3] = 30*np.cos(phi), rng.uniform(-5, 5, ns), 30*np.sin(phi)
    stress[5] = rng.uniform(-2, 2, ns)/out.unit_km          # a spread of Doppler shifts
    stress[7] = 1.0
    ctx.set_forces(**out.forces_kwargs())
    ctx.upload_soa(stress)
    for _ in range(reps):
        ctx.integrate_const(opt.step_size, n_iter, 1e9)
        ms = ctx.last_kernel_ms()
    c = ctx.counters()
    line('k_const_fused<no image> stress', ms, c['particle_steps'], 'particle*steps', 128,
         f'{ns} packets x {n_iter} steps, none dies (SURVEY 8d stress vector)')
    with quiet():
        img = ModelImage(inputs, {'quantity': 'radiance', 'dims': '512,512'}, context=ctx)
    img._set_image(ctx, float(out.aplanet), float(out.vrplanet), True)
    for _ in range(reps):
        ctx.image_clear()
        ctx.integrate_const(opt.step_size, n_iter, 1e9, image=True)
        ms = ctx.last_kernel_ms()
    c = ctx.counters()
    line('k_const_fused<IMAGE> stress', ms, c['particle_steps'], 'particle*steps', 128,
         f'same packets, every sample located, {c["samples_binned"]} of {c["samples"]} inside '
         f'the image (the packets are far from it): the locate stage without weights and atomics')

    inputs.options.step_size = 0.
    inputs.options.resolution = 1e-4
    nv = 1_000_000
    with quiet():
        outv = Output(inputs, nv, seed=3, integrate=False, save=False, context=ctx)
    ctx.set_forces(**outv.forces_kwargs())
    ctx.upload_soa(outv.x0_soa())
    for _ in range(reps):
        ctx.integrate_var(1e-4, 25.)
        ms = ctx.last_kernel_ms()
    c = ctx.counters()
    line('k_var', ms, c['particle_steps'], 'rk5 attempts', 128,
         f'{nv} packets at random ages, resolution 1e-4')
    ctx.close()


if __name__ == '__main__':
    main()
