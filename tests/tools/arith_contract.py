"""Where does "bit-exact counts against the reference's arithmetic" end?  (CPU only.)

The kernels (and, call for call, the C checker oracle/c/oracle.c) differ from the reference's
NumPy arithmetic in two places: pow / exp / log are deterministic 1-ulp functions of their own
(NumPy's are 1-ulp libm kernels too), and since round 3 the 14 x 7 tableau terms of a step are
fused multiply-adds (one rounding per term where rk5.py:33-35,41-43 has two).  States therefore
agree to ~1e-10, not bit for bit, and a sample whose float32 value sits within that distance of
a rounding boundary AND within one float32 ulp of a pixel edge may be counted in the neighbouring
pixel.  This tool measures how often: N seeded packets of the bench workload (chunks of 20 000,
seed 7000 + chunk) through

    * the NumPy oracle (= the reference: pinned bit for bit to rk5.py / state.py / Histogram2d
      by oracle/make_golden.py),
    * the C checker as built for the kernels (tableau terms fused),
    * the C checker with -DORACLE_TABLEAU_TWO_ROUNDINGS (NumPy's two roundings per term),

all 1667 steps, 512 x 512 radiance image of the float32 samples, and reports per build: packets
whose step count differs, samples binned into a different pixel, pixels of the cumulative image
whose packet count differs, worst relative state and brightness difference.  With --variable the
adaptive driver as well (attempt counts and final step sizes, 2000 packets per chunk).

    python tests/tools/arith_contract.py --packets 1000000 --procs 4 \
        > profiles/r04_arith_contract_1e6.txt        # ~15 min on 8 vCPU, ~6 GB per process
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHUNK, ENDTIME, STEP, EDGE, DIMS = 20000, 50000., 30., 25., (512, 512)


def one_chunk(job):
    k, n, threads = job
    from oracle import np_oracle as O
    from oracle.c_oracle import COracle
    from tests import helpers as H
    f = H.mercury_forces('Na', 1.3)
    X0 = H.sample_x0(n, 7000 + k, ENDTIME)
    nsteps, n_iter = O.n_output_steps(ENDTIME, STEP)
    t0 = time.time()
    results, _, work = O.constant_step_driver(f, X0, ENDTIME, STEP, EDGE)
    t_np = time.time() - t0
    last = (results[:, 7, :n_iter] > 0).sum(axis=1)
    fin = results[np.arange(n), :, last]
    s = O.samples_from_results(results, compress=True, downcast=True)
    del results
    im = H.image_setup(f, 'radiance', dims=DIMS)
    ref_img, ref_cnt, _, _ = O.create_image(s['x'], s['y'], s['z'], s['vy'], s['frac'], f.vrplanet,
                                            im['M'], 'radiance', im['g_tables'], im['dims'],
                                            im['xrange'], im['zrange'], im['apix'], matmul=False)
    del s
    out = {'chunk': k, 'packets': n, 'work': int(work), 'binned': int(ref_cnt.sum()),
           'numpy_s': round(t_np, 1)}
    ref_cnt = ref_cnt.astype(np.int64)
    for name, kw in (('fused', {}), ('two_roundings', {'two_roundings': True})):
        co = COracle(**kw)
        desc = co.image_desc(im['M'], f.vrplanet, im['apix'], 'radiance', im['g_tables'],
                             im['xedges'], im['zedges'], downcast=True)
        c = co.integrate_const(f, X0, STEP, n_iter, EDGE, img=desc, threads=threads)
        diff = c['counts'].astype(np.int64) - ref_cnt
        nz = np.flatnonzero(diff.ravel())
        lit = ref_img > 0
        out[name] = {
            'work_equal': bool(c['work'] == work),
            'step_counts_differing': int((c['steps'] != last).sum()),
            'samples_in_another_pixel': int(np.abs(diff).sum()//2),
            'diff_pix': nz.tolist(), 'diff_val': diff.ravel()[nz].tolist(),
            'max_rel_state': float(np.nanmax(np.abs(c['final'] - fin) /
                                             np.maximum(np.abs(fin), 1e-3))),
            'max_rel_brightness': float(np.max(np.abs(c['image'][lit] - ref_img[lit]) /
                                               ref_img[lit])) if not len(nz) else None}
    return out


def one_var_chunk(job):
    k, n = job
    from oracle import np_oracle as O
    from oracle.c_oracle import COracle
    from tests import helpers as H
    f = H.mercury_forces('Na', 1.3)
    X0 = H.sample_x0(n, 9000 + k, 20000.)
    X0[:, 0] = np.random.default_rng(90 + k).random(n)*20000.
    fin, hs, work = O.variable_step_driver(f, X0, 1e-4, EDGE)
    out = {'chunk': k, 'packets': n, 'attempts': int(work)}
    for name, kw in (('fused', {}), ('two_roundings', {'two_roundings': True})):
        cfin, chs, cwork, bad = COracle(**kw).integrate_var(f, X0, 1e-4, EDGE)
        out[name] = {'attempts': int(cwork), 'bad': int(bad),
                     'final_step_differs_beyond_1e-9': int((~np.isclose(chs, hs, rtol=1e-9, atol=0)).sum()),
                     'max_rel_state': float(np.nanmax(np.abs(cfin - fin)/np.maximum(np.abs(fin), 1e-3)))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--packets', type=int, default=1_000_000)
    ap.add_argument('--procs', type=int, default=4)
    ap.add_argument('--variable', type=int, default=0, help='packets through the adaptive driver')
    args = ap.parse_args()
    jobs = [(k, min(CHUNK, args.packets - k*CHUNK), max(1, 8//args.procs))
            for k in range((args.packets + CHUNK - 1)//CHUNK)]
    t0 = time.time()
    total = {n: {'step_counts_differing': 0, 'samples_in_another_pixel': 0, 'max_rel_state': 0.0,
                 'max_rel_brightness': 0.0, 'work_equal': True, 'first_flip_at_packets': None}
             for n in ('fused', 'two_roundings')}
    cum = {n: np.zeros(DIMS[0]*DIMS[1], dtype=np.int64) for n in total}
    done = work = binned = 0
    with mp.get_context('spawn').Pool(args.procs) as pool:
        for res in pool.imap(one_chunk, jobs):
            done += res['packets']
            work += res['work']
            binned += res['binned']
            for n, t in total.items():
                r = res[n]
                t['step_counts_differing'] += r['step_counts_differing']
                t['samples_in_another_pixel'] += r['samples_in_another_pixel']
                t['work_equal'] &= r['work_equal']
                t['max_rel_state'] = max(t['max_rel_state'], r['max_rel_state'])
                if r['max_rel_brightness'] is not None:
                    t['max_rel_brightness'] = max(t['max_rel_brightness'], r['max_rel_brightness'])
                if r['diff_pix']:
                    cum[n][r['diff_pix']] += r['diff_val']
                    if t['first_flip_at_packets'] is None:
                        t['first_flip_at_packets'] = done
            sys.stderr.write(f"chunk {res['chunk']:3d}  {done:9d} packets  {time.time() - t0:6.0f} s  "
                             + '  '.join(f"{n}: {t['step_counts_differing']} steps, "
                                         f"{t['samples_in_another_pixel']} samples"
                                         for n, t in total.items()) + '\n')
    for n, t in total.items():
        t['pixels_differing_in_the_cumulative_image'] = int(np.count_nonzero(cum[n]))
    report = {'packets': done, 'particle_steps': work, 'samples_binned': binned,
              'image': '512x512 radiance, float32 samples', 'chunk': CHUNK,
              'seconds': round(time.time() - t0), 'constant_step': total}
    if args.variable:
        vjobs = [(k, min(2000, args.variable - 2000*k)) for k in range((args.variable + 1999)//2000)]
        vt = {n: {'attempts_differing': 0, 'final_step_differs_beyond_1e-9': 0, 'max_rel_state': 0.0}
              for n in ('fused', 'two_roundings')}
        attempts = 0
        with mp.get_context('spawn').Pool(args.procs) as pool:
            for res in pool.imap(one_var_chunk, vjobs):
                attempts += res['attempts']
                for n, t in vt.items():
                    t['attempts_differing'] += abs(res[n]['attempts'] - res['attempts'])
                    t['final_step_differs_beyond_1e-9'] += res[n]['final_step_differs_beyond_1e-9']
                    t['max_rel_state'] = max(t['max_rel_state'], res[n]['max_rel_state'])
        report['variable_step'] = {'packets': args.variable, 'rk5_attempts': attempts, **vt}
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
