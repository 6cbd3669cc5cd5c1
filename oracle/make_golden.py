"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE'S OWN CODE and pin the
oracle against them.  Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

What is loaded from the reference (by path, oracle/ref_loader.py): particle_tracking/rk5.py,
particle_tracking/state.py, math/histogram.py (Histogram2d), math/rotation_matrix.py -- the files
of the hot path that import nothing but numpy.  Everything else on the path needs astropy /
PostgreSQL and is re-stated in oracle/np_oracle.py; the drivers are exercised here AROUND the
reference's imported rk5 (np_oracle's loops with ``rk5_fn`` = the reference function).

The script asserts, before writing anything, that oracle/np_oracle.py reproduces the reference
bit for bit on this machine, so the committed .npz files are at once the reference's outputs and
the oracle's.  Only inputs and expected outputs are stored (no reference source).

Vectors
  g1_state.npz      state(): 6 force/loss configurations x 96 packets incl. shadow / rho==1 /
                    y==0 / off-table velocities
  g2_rk5.npz        rk5(): Na, Ca, Mg tables; h = 30 and per-packet h; result and delta
  g3_const.npz      constant driver (Gravity.input-like and Na-reference forces), 256 packets:
                    final records, active steps, alive count and sum(frac) per step, a few full
                    trajectories, 64x64 radiance/column images + packet counts (fp64 and with the
                    float32 save/restore round trip)
  g4_var.npz        variable driver around the reference rk5, 128 packets
  g5_hist.npz       Histogram2d edge cases at 512x512 (on-edge, right edge, outside)
  g6_rotation.npz   rotation_matrix / image_rotation for several sub-observer points
  g9_var2000.npz    the adaptive driver around the reference's rk5 on 2000 packets of the bench
                    workload at random ages: attempts in total, final states and stored steps
  g8_const20k.npz   the bench workload at BASELINE's image geometry, around the reference's rk5:
                    20 000 packets (X0 = tests.helpers.sample_x0(20000, 8008, 50000.), not
                    stored) x all 1667 steps: per-packet step counts, alive count and sum(frac)
                    per step, the 512 x 512 packet-count image of the float32 samples (sparse),
                    and of the radiance / column images their row sums, column sums and every
                    16th pixel -- what the -m gpu suite holds the fused-tableau kernels to
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import np_oracle as O          # noqa: E402
from oracle import ref_loader              # noqa: E402
from tests import helpers as H             # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    assert ref_loader.available(), 'reference tree not found'
    rk5m, statem, histm, rotm = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)

    # ---- G1 state -----------------------------------------------------------------------------
    X = H.random_cloud(96, 11)
    cfgs = [(True, True, 0.0), (True, False, 0.0), (False, True, 0.0), (True, True, 3600.0),
            (True, True, -7200.0), (False, False, 0.0)]
    g1 = dict(X=X, cfgs=np.array(cfgs, dtype=float))
    for k, (grav, rad, life) in enumerate(cfgs):
        f = H.mercury_forces('Na', 1.3, grav, rad, life)
        a_ref, i_ref = statem.state(X, ref_loader.duck_output(f, 30.0))
        a_o, i_o = O.state(X, f)
        assert np.array_equal(a_ref, a_o) and np.array_equal(i_ref, i_o), 'oracle state != reference'
        g1[f'accel{k}'], g1[f'ioniz{k}'] = a_ref, i_ref
    np.savez_compressed(os.path.join(OUT, 'g1_state.npz'), **g1)

    # ---- G2 rk5 -------------------------------------------------------------------------------
    g2 = {}
    for sp, taa in (('Na', 1.3), ('Ca', 0.0), ('Mg', 3.14)):
        f = H.mercury_forces(sp, taa)
        X = H.random_cloud(64, 5)
        hvar = np.random.default_rng(2).uniform(1, 120, 64)
        r_ref, d_ref = rk5m.rk5(ref_loader.duck_output(f, 0), X.copy(), hvar)
        r_o, d_o = O.rk5(f, X, hvar, want_delta=True)
        assert np.array_equal(r_ref, r_o) and np.array_equal(d_ref, d_o), 'oracle rk5 != reference'
        h30 = np.zeros(64) + 30.0
        r30, none = rk5m.rk5(ref_loader.duck_output(f, 30.0), X.copy(), h30)
        assert none is None and np.array_equal(r30, O.rk5(f, X, h30)[0])
        g2.update({f'{sp}_X': X, f'{sp}_h': hvar, f'{sp}_result': r_ref, f'{sp}_delta': d_ref,
                   f'{sp}_result30': r30})
    np.savez_compressed(os.path.join(OUT, 'g2_rk5.npz'), **g2)

    # ---- G3 constant driver -------------------------------------------------------------------
    g3 = {}
    cases = {'grav': dict(f=H.mercury_forces('Na', 3.14, True, False, 0.0), endtime=20000.,
                          step=30., edge=1e30, vprob=4., delv=4.),
             'na': dict(f=H.mercury_forces('Na', 1.3), endtime=50000., step=30., edge=25.,
                        vprob=2.5, delv=2.)}
    for name, c in cases.items():
        f = c['f']
        X0 = H.sample_x0(256, 1234, c['endtime'], c['vprob'], c['delv'])
        out = ref_loader.duck_output(f, c['step'])
        res_ref, _, work_ref = O.constant_step_driver(
            f, X0, c['endtime'], c['step'], c['edge'], rk5_fn=lambda X, h: rk5m.rk5(out, X, h))
        res_o, _, work_o = O.constant_step_driver(f, X0, c['endtime'], c['step'], c['edge'])
        assert work_ref == work_o and np.array_equal(res_ref, res_o), 'oracle driver != reference'
        alive = res_ref[:, 7, :] > 0
        steps = alive.sum(axis=1) - 1 + (~alive[:, -1])        # iterations each packet was active
        steps = np.minimum(steps, res_ref.shape[2]-1)
        last = np.minimum(steps, res_ref.shape[2]-1)
        final = res_ref[np.arange(256), :, last]
        g3.update({f'{name}_X0': X0, f'{name}_final': final, f'{name}_steps': steps,
                   f'{name}_alive_per_step': alive.sum(axis=0),
                   f'{name}_fracsum_per_step': res_ref[:, 7, :].sum(axis=0),
                   f'{name}_work': np.int64(work_ref),
                   f'{name}_traj_ids': np.array([0, 17, 101, 255]),
                   f'{name}_traj': res_ref[[0, 17, 101, 255]],
                   f'{name}_params': np.array([c['endtime'], c['step'], c['edge']])})
        if name == 'na':
            for q in ('radiance', 'column'):
                im = H.image_setup(f, q, dims=(64, 64))
                for dc in (False, True):
                    s = O.samples_from_results(res_ref, compress=True, downcast=dc)
                    # image through the reference's own Histogram2d wrapper
                    img_o, cnt_o, _, _ = O.create_image(
                        s['x'], s['y'], s['z'], s['vy'], s['frac'], f.vrplanet, im['M'], q,
                        im['g_tables'], im['dims'], im['xrange'], im['zrange'], im['apix'])
                    # redo the histogram step with the reference class on the oracle's
                    # intermediate quantities to pin the binning rule
                    pts = np.stack([s['x'], s['y'], s['z']], 1)
                    pobs = np.array(np.matmul(im['M'], pts.T).T)
                    hh = histm.Histogram2d(pobs[:, 0], pobs[:, 2], bins=im['dims'],
                                           range=[list(im['xrange']), list(im['zrange'])])
                    assert np.array_equal(hh.histogram, cnt_o)
                    tag = f'{name}_{q}_{"f32" if dc else "f64"}'
                    g3[tag + '_image'], g3[tag + '_counts'] = img_o, cnt_o
    np.savez_compressed(os.path.join(OUT, 'g3_const.npz'), **g3)

    # ---- G8: the bench workload, 20 000 packets, 512 x 512, around the imported rk5 -------------
    f = H.mercury_forces('Na', 1.3)
    n8, seed8, endtime8, step8, edge8 = 20000, 8008, 50000., 30., 25.
    X0 = H.sample_x0(n8, seed8, endtime8)
    out = ref_loader.duck_output(f, step8)
    res_ref, _, work_ref = O.constant_step_driver(
        f, X0, endtime8, step8, edge8, rk5_fn=lambda X, h: rk5m.rk5(out, X, h))
    res_o, _, work_o = O.constant_step_driver(f, X0, endtime8, step8, edge8)
    assert work_ref == work_o and np.array_equal(res_ref, res_o), 'oracle driver != reference (g8)'
    del res_o
    n_iter8 = res_ref.shape[2] - 1
    g8 = dict(params=np.array([n8, seed8, endtime8, step8, edge8]), work=np.int64(work_ref),
              steps=(res_ref[:, 7, :n_iter8] > 0).sum(axis=1).astype(np.uint16),
              alive_per_step=(res_ref[:, 7, :] > 0).sum(axis=0).astype(np.int32),
              fracsum_per_step=res_ref[:, 7, :].sum(axis=0))
    s8 = O.samples_from_results(res_ref, compress=True, downcast=True)
    del res_ref
    for q in ('radiance', 'column'):
        im = H.image_setup(f, q, dims=(512, 512))
        img, cnt, _, _ = O.create_image(s8['x'], s8['y'], s8['z'], s8['vy'], s8['frac'],
                                        f.vrplanet, im['M'], q, im['g_tables'], im['dims'],
                                        im['xrange'], im['zrange'], im['apix'])
        pts = np.stack([s8['x'], s8['y'], s8['z']], 1)
        pobs = np.array(np.matmul(im['M'], pts.T).T)
        hh = histm.Histogram2d(pobs[:, 0], pobs[:, 2], bins=im['dims'],
                               range=[list(im['xrange']), list(im['zrange'])])
        assert np.array_equal(hh.histogram, cnt), 'oracle binning != reference Histogram2d (g8)'
        if 'count_pix' in g8:
            assert np.array_equal(cnt.ravel()[g8['count_pix']], g8['count_val']) and \
                cnt.sum() == g8['count_val'].sum(), 'radiance and column count images differ'
        else:
            nz = np.flatnonzero(cnt.ravel())
            g8['count_pix'], g8['count_val'] = nz.astype(np.uint32), cnt.ravel()[nz].astype(np.uint16)
            assert cnt.max() < 65536
        g8[q + '_rowsum'], g8[q + '_colsum'] = img.sum(axis=1), img.sum(axis=0)
        g8[q + '_every16'] = img.ravel()[::16].copy()
    np.savez_compressed(os.path.join(OUT, 'g8_const20k.npz'), **g8)

    # ---- G4 variable driver -------------------------------------------------------------------
    f = H.mercury_forces('Na', 1.3)
    X0 = H.sample_x0(128, 4321, 20000.)
    X0[:, 0] = np.random.default_rng(8).random(128)*20000.
    out = ref_loader.duck_output(f, 0)
    fin_ref, hs_ref, w_ref = O.variable_step_driver(
        f, X0, 1e-4, 25.0, rk5_fn=lambda X, h: rk5m.rk5(out, X, h))
    fin_o, hs_o, w_o = O.variable_step_driver(f, X0, 1e-4, 25.0)
    assert w_ref == w_o and np.array_equal(fin_ref, fin_o) and np.array_equal(hs_ref, hs_o)
    np.savez_compressed(os.path.join(OUT, 'g4_var.npz'), X0=X0, final=fin_ref, step_size=hs_ref,
                        work=np.int64(w_ref), params=np.array([1e-4, 25.0]))

    # ---- G9: the adaptive driver on 2000 packets of the bench workload ---------------------------
    f = H.mercury_forces('Na', 1.3)
    X0 = H.sample_x0(2000, 909, 50000.)
    X0[:, 0] = np.random.default_rng(99).random(2000)*50000.
    out = ref_loader.duck_output(f, 0)
    fin_ref, hs_ref, w_ref = O.variable_step_driver(
        f, X0, 1e-4, 25.0, rk5_fn=lambda X, h: rk5m.rk5(out, X, h))
    fin_o, hs_o, w_o = O.variable_step_driver(f, X0, 1e-4, 25.0)
    assert w_ref == w_o and np.array_equal(fin_ref, fin_o) and np.array_equal(hs_ref, hs_o)
    np.savez_compressed(os.path.join(OUT, 'g9_var2000.npz'), X0=X0, final=fin_ref,
                        step_size=hs_ref, work=np.int64(w_ref), params=np.array([1e-4, 25.0]))

    # ---- G5 histogram edge cases --------------------------------------------------------------
    rng = np.random.default_rng(5)
    edges = np.linspace(-4, 4, 513)
    px = np.concatenate([edges, rng.uniform(-4.5, 4.5, 4000), [4.0]*8, [-4.0]*8,
                         np.nextafter(edges[100:110], np.inf), np.nextafter(edges[100:110], -np.inf)])
    pz = np.concatenate([edges[::-1], rng.uniform(-4.5, 4.5, 4000), rng.uniform(-4, 4, 8),
                         [4.0]*8, rng.uniform(-4, 4, 20)])
    w = rng.uniform(0, 1, px.size)
    hw = histm.Histogram2d(px, pz, weights=w, bins=[512, 512], range=[[-4, 4], [-4, 4]])
    hc = histm.Histogram2d(px, pz, bins=[512, 512], range=[[-4, 4], [-4, 4]])
    nz = np.nonzero(hc.histogram)
    np.savez_compressed(os.path.join(OUT, 'g5_hist.npz'), px=px, pz=pz, w=w, nz_i=nz[0],
                        nz_j=nz[1], counts=hc.histogram[nz], weights=hw.histogram[nz],
                        xcenters=hw.x, dx=hw.dx)

    # ---- G6 rotation --------------------------------------------------------------------------
    pts = [(0.0, np.pi/2), (0.4, 1.1), (np.pi, 0.0), (1.0, -0.3), (0.0, 0.0)]
    mats = []
    for slon, slat in pts:
        M = O.image_rotation(slon, slat)
        pSun = np.array([0., -1., 0.])
        pObs = np.array([np.sin(slon)*np.cos(slat), -np.cos(slon)*np.cos(slat), np.sin(slat)])
        if not np.array_equal(pSun, pObs):
            costh = np.dot(pSun, pObs)/np.linalg.norm(pSun)/np.linalg.norm(pObs)
            Mref = np.asarray(rotm.rotation_matrix(np.arccos(np.clip(costh, -1, 1)),
                                                   np.cross(pSun, pObs)))
            assert np.array_equal(M, Mref), 'oracle rotation != reference'
        mats.append(M)
    np.savez_compressed(os.path.join(OUT, 'g6_rotation.npz'), subobs=np.array(pts),
                        M=np.array(mats))

    for fn in sorted(os.listdir(OUT)):
        print(f'{fn:20s} {os.path.getsize(os.path.join(OUT, fn))/1024:8.1f} KiB')
    print('oracle == reference on every vector; golden files written')


if __name__ == '__main__':
    main()
