"""Pins of the C oracle's own building blocks (oracle/c/oracle_math.h) and of the exactness
arguments the HIP kernels rely on."""
from fractions import Fraction

import numpy as np
import pytest

from oracle.c_oracle import COracle


def ulps(a, b):
    return np.abs(a - b)/np.spacing(np.maximum(np.abs(a), np.abs(b)))


def test_cube_is_correctly_rounded(coracle):
    rng = np.random.default_rng(0)
    r = np.concatenate([rng.uniform(0.5, 40, 20000), 10**rng.uniform(-3, 3, 2000)])
    exact = np.array([float(Fraction(v)**3) for v in r])
    assert np.array_equal(coracle.math('cube', r), exact)


def test_exp_log_within_one_ulp_of_libm(coracle):
    libm = COracle(libm=True)
    rng = np.random.default_rng(1)
    x = np.concatenate([-rng.uniform(0, 40, 200000), rng.uniform(-1e-9, 1e-9, 100),
                        [0.0, -0.34, -0.35, -1.03, -1.04, -700.0]])
    assert ulps(coracle.math('exp', x), libm.math('exp', x)).max() <= 1.0
    f = np.concatenate([rng.uniform(1e-10, 1, 200000), 10**rng.uniform(-300, 300, 2000),
                        [1.0, 0.5, 2.0, 1e-10, np.nextafter(1.0, 0), np.nextafter(1.0, 2)]])
    assert ulps(coracle.math('log', f), libm.math('log', f)).max() <= 1.0
    assert coracle.math('log', np.array([1.0]))[0] == 0.0
    assert coracle.math('exp', np.array([0.0]))[0] == 1.0


def test_sqrt_threshold_equivalences():
    """sqrt(s) > 1 <=> s > 1+2^-52 and sqrt(s) < 1 <=> s < 1 for correctly rounded sqrt: the
    kernels test s directly instead of taking the root (nxc_device.hpp sunlit / apply_fate)."""
    one = 1.0
    s = np.array([np.nextafter(one, 0), one, np.nextafter(one, 2),
                  np.nextafter(np.nextafter(one, 2), 2), 1 - 1e-15, 1 + 1e-15, 0.25, 4.0])
    root = np.sqrt(s)
    assert np.array_equal(root > 1, s > float.fromhex('0x1.0000000000001p+0'))
    assert np.array_equal((root - 1.0) < 0, s < 1.0)
    rng = np.random.default_rng(2)
    t = 1 + rng.integers(-64, 64, 4096)*2.0**-53
    assert np.array_equal(np.sqrt(t) > 1, t > float.fromhex('0x1.0000000000001p+0'))
    assert np.array_equal(np.sqrt(t) < 1, t < 1.0)


def test_libm_variant_agrees_to_tolerance(coracle):
    from tests import helpers as H
    libm = COracle(libm=True)
    f = H.mercury_forces('Na', 1.3)
    X = H.random_cloud(2048, 3)
    h = np.zeros(len(X)) + 30.
    a, _ = coracle.rk5(f, X, h)
    b, _ = libm.rk5(f, X, h)
    np.testing.assert_allclose(a, b, rtol=1e-14, atol=1e-20)


def test_threads_do_not_change_states(coracle):
    from oracle import np_oracle as O
    from tests import helpers as H
    f = H.mercury_forces('Na', 1.3)
    X0 = H.sample_x0(1500, 77, 50000.)
    nsteps, n_iter = O.n_output_steps(50000., 30.)
    a = coracle.integrate_const(f, X0, 30., n_iter, 25., threads=1)
    b = coracle.integrate_const(f, X0, 30., n_iter, 25., threads=4)
    assert a['work'] == b['work'] and np.array_equal(a['final'], b['final'])
    assert np.array_equal(a['steps'], b['steps'])


@pytest.mark.parametrize('quantity', ['radiance', 'column'])
def test_c_oracle_against_the_reference_arithmetic_at_3000_packets(coracle, quantity):
    """The C checker rounds like the kernels (deterministic cube / exp / log, tableau terms fused),
    the NumPy oracle like the reference (pinned bit for bit to the reference's own rk5.py /
    state.py).  Twelve times the golden fixture -- 3000 seeded packets, all 1667 steps, a 256 x 256
    image of the float32 samples -- the two still agree in what north_star asks to be exact:
    every packet's step count and the packet count of every pixel; states and brightness agree
    far inside its 1e-6."""
    from oracle import np_oracle as O
    from tests import helpers as H
    f = H.mercury_forces('Na', 1.3)
    n, endtime, step, edge = 3000, 50000., 30., 25.
    X0 = H.sample_x0(n, 4321, endtime)
    nsteps, n_iter = O.n_output_steps(endtime, step)
    results, _, work = O.constant_step_driver(f, X0, endtime, step, edge)
    im = H.image_setup(f, quantity, dims=(256, 256))
    desc = coracle.image_desc(im['M'], f.vrplanet, im['apix'], quantity, im['g_tables'],
                              im['xedges'], im['zedges'], downcast=True)
    c = coracle.integrate_const(f, X0, step, n_iter, edge, img=desc, threads=coracle.max_threads())
    assert c['work'] == work
    # a packet is stepped at iteration ct while its record ct - 1 is alive (Output.py:385)
    last = (results[:, 7, :n_iter] > 0).sum(axis=1)
    assert np.array_equal(c['steps'], last)
    fin = results[np.arange(n), :, last]
    np.testing.assert_allclose(c['final'], fin, rtol=1e-9, atol=1e-13)
    s = O.samples_from_results(results, compress=True, downcast=True)
    ref_img, ref_cnt, _, _ = O.create_image(s['x'], s['y'], s['z'], s['vy'], s['frac'], f.vrplanet,
                                            im['M'], quantity, im['g_tables'], im['dims'],
                                            im['xrange'], im['zrange'], im['apix'], matmul=False)
    assert ref_cnt.sum() > 1e5
    assert np.array_equal(c['counts'], ref_cnt.astype(np.uint64))        # bit-exact packet counts
    np.testing.assert_allclose(c['image'], ref_img, rtol=1e-6, atol=0)   # north_star
    np.testing.assert_allclose(c['image'], ref_img, rtol=1e-9, atol=0)   # what we get


def _g8():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g8_const20k.npz'),
                   allow_pickle=False)


@pytest.mark.parametrize('quantity', ['radiance', 'column'])
def test_c_oracle_against_g8_the_reference_rk5_at_20000_packets(coracle, quantity):
    """g8_const20k.npz was produced by the reference's OWN rk5.py / state.py / Histogram2d (loaded
    by path, oracle/make_golden.py) on 20 000 packets of the bench workload with BASELINE's 512 x
    512 image.  The C checker -- the kernels' arithmetic, tableau terms fused -- must reproduce
    every step count and every pixel's packet count; brightness to 1e-9 (north_star: 1e-6)."""
    from oracle import np_oracle as O
    from tests import helpers as H
    g = _g8()
    n, seed, endtime, step, edge = g['params']
    f = H.mercury_forces('Na', 1.3)
    X0 = H.sample_x0(int(n), int(seed), endtime)
    nsteps, n_iter = O.n_output_steps(endtime, step)
    im = H.image_setup(f, quantity, dims=(512, 512))
    desc = coracle.image_desc(im['M'], f.vrplanet, im['apix'], quantity, im['g_tables'],
                              im['xedges'], im['zedges'], downcast=True)
    c = coracle.integrate_const(f, X0, step, n_iter, edge, img=desc, threads=coracle.max_threads())
    assert c['work'] == int(g['work'])
    assert np.array_equal(c['steps'], g['steps'].astype(np.int64))
    ref_cnt = np.zeros(512*512, dtype=np.uint64)
    ref_cnt[g['count_pix']] = g['count_val']
    assert ref_cnt.sum() > 1.2e6
    assert np.array_equal(c['counts'].ravel(), ref_cnt)
    np.testing.assert_allclose(c['image'].sum(axis=1), g[quantity + '_rowsum'], rtol=1e-9)
    np.testing.assert_allclose(c['image'].sum(axis=0), g[quantity + '_colsum'], rtol=1e-9)
    np.testing.assert_allclose(c['image'].ravel()[::16], g[quantity + '_every16'], rtol=1e-9)


def test_two_roundings_build_is_the_numpy_arithmetic_to_1e14():
    """-DORACLE_TABLEAU_TWO_ROUNDINGS (with -DNXC_TABLEAU_TWO_ROUNDINGS in the kernels) restores
    NumPy's two roundings per tableau term (rk5.py:33-35,41-43).  That build must stay alive and
    stay closer to the reference than the fused one: one step agrees with the NumPy oracle to 1e-14
    (what is left are the 1-ulp pow / exp / log), the fused build to 1e-13; a whole run keeps
    every step count and lands closer to the NumPy trajectory than the fused build does."""
    from oracle import np_oracle as O
    from tests import helpers as H
    two, fused = COracle(two_roundings=True), COracle()
    f = H.mercury_forces('Na', 1.3)
    X = H.random_cloud(4096, 9)
    h = np.random.default_rng(3).uniform(1, 120, len(X))
    ref, dref = O.rk5(f, X, h, want_delta=True)
    a, da = two.rk5(f, X, h, want_delta=True)
    b, _ = fused.rk5(f, X, h, want_delta=True)
    scale = np.maximum(np.abs(ref), 1e-3)
    assert (np.abs(a - ref)/scale).max() < 1e-14
    assert (np.abs(b - ref)/scale).max() < 1e-13
    assert not np.array_equal(a, b)                      # the switch does switch something
    np.testing.assert_allclose(da, dref, rtol=1e-9, atol=1e-22)
    n, endtime, step, edge = 1500, 50000., 30., 25.
    X0 = H.sample_x0(n, 5150, endtime)
    nsteps, n_iter = O.n_output_steps(endtime, step)
    results, _, work = O.constant_step_driver(f, X0, endtime, step, edge)
    last = (results[:, 7, :n_iter] > 0).sum(axis=1)
    fin = results[np.arange(n), :, last]
    err = {}
    for name, co in (('two', two), ('fused', fused)):
        c = co.integrate_const(f, X0, step, n_iter, edge, threads=co.max_threads())
        assert c['work'] == work and np.array_equal(c['steps'], last), name
        err[name] = np.nanmax(np.abs(c['final'] - fin)/np.maximum(np.abs(fin), 1e-3))
    assert err['two'] < 1e-9 and err['fused'] < 1e-9


def test_variable_driver_step_counts_against_the_numpy_arithmetic(coracle):
    """The adaptive driver decides accept / reject on delta, which the fused tableau terms also
    touch (Output.py:281-342): attempts per packet and final step sizes of the C checker against
    the NumPy oracle (the reference's arithmetic) on 1500 packets.  A decision that lands within
    rounding of its threshold may flip, so equality is not asserted, only counted: at most 1 packet
    in 500 may differ in its attempt count, and every state agrees to 1e-6 (north_star) -- in
    practice none differs."""
    from oracle import np_oracle as O
    from tests import helpers as H
    f = H.mercury_forces('Na', 1.3)
    n = 1500
    X0 = H.sample_x0(n, 616, 20000.)
    X0[:, 0] = np.random.default_rng(61).random(n)*20000.
    fin, hs, work = O.variable_step_driver(f, X0, 1e-4, 25.0)
    for co in (coracle, COracle(two_roundings=True)):
        cfin, chs, cwork, bad = co.integrate_var(f, X0, 1e-4, 25.0)
        assert bad == 0
        assert abs(cwork - work) <= max(2, work//100000)
        differ = int((chs != hs).sum())
        close = np.isclose(chs, hs, rtol=1e-6, atol=0)
        assert (~close).sum() <= n//500, (differ, int((~close).sum()))
        np.testing.assert_allclose(cfin[close], fin[close], rtol=1e-6, atol=1e-9)
