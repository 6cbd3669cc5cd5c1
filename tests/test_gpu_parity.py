"""GPU parity tests: the HIP path (through the C ABI) against the oracles.

Contract (DESIGN.md "Parity"):
  * HIP vs C oracle: BIT-EXACT for every state column and every integer (steps, counts); the
    weighted image differs only by fp64 atomic summation order (rtol 1e-12).
  * HIP vs NumPy oracle (= the reference's arithmetic, pinned by tests/golden): rtol 1e-12 per
    step; the only differences are NumPy's own 1-ulp pow/exp/log kernels.
"""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_device_math_is_ieee_exact(ctx, coracle):
    rng = np.random.default_rng(0)
    # nxc_sqrt / nxc_div (the streamlined chains the kernels use) must stay correctly rounded:
    # operational range, wide range, and beyond +-2^200 where they fall back to the full sequence
    for lo, hi, n in ((-3, 3, 2000000), (-60, 60, 1000000), (-300, 300, 400000)):
        x = 10**rng.uniform(lo, hi, n)*rng.choice([1.0, 1.0, 0.999999], n)
        y = 10**rng.uniform(lo/2, hi/2, n)*rng.choice([-1.0, 1.0], n)
        assert np.array_equal(ctx.math('sqrt', x), np.sqrt(x))
        with np.errstate(over='ignore', under='ignore'):
            assert np.array_equal(ctx.math('div', x, y), x/y)
    m = 1 + rng.integers(0, 2**52, 1000000)*2.0**-52        # dense mantissas
    assert np.array_equal(ctx.math('sqrt', m), np.sqrt(m))
    assert np.array_equal(ctx.math('sqrt', 2*m), np.sqrt(2*m))
    d = 1 + rng.integers(0, 2**52, 1000000)*2.0**-52
    assert np.array_equal(ctx.math('div', m, d), m/d)
    assert np.array_equal(ctx.math('div', np.zeros(8), d[:8]), np.zeros(8))
    r = rng.uniform(0.5, 40, 400000)
    assert np.array_equal(ctx.math('cube', r), coracle.math('cube', r))
    e = -rng.uniform(0, 40, 400000)
    assert np.array_equal(ctx.math('exp', e), coracle.math('exp', e))
    fr = np.concatenate([rng.uniform(1e-10, 1, 400000), 10**rng.uniform(-12, 3, 1000)])
    assert np.array_equal(ctx.math('log', fr), coracle.math('log', fr))
    # exp: every argument range, both signs, the neighbourhood (+-64 ulp) of the thresholds of its
    # rare path and of the reduction (multiples of ln 2 / 2); log: dense around 1, every magnitude,
    # subnormals, and +-64 ulp around every one of the 92 bin edges of its table at four scales
    def around(v, n=64):
        out = [v]
        lo = hi = v
        for _ in range(n):
            lo, hi = np.nextafter(lo, -np.inf), np.nextafter(hi, np.inf)
            out += [lo, hi]
        return np.array(out)
    e = np.concatenate([rng.uniform(-0.4, 0.4, 200000), rng.uniform(-1.2, 1.2, 200000),
                        rng.uniform(-745.5, 710, 200000), -rng.uniform(700, 746, 20000),
                        rng.choice([-1., 1.], 50000)*10**rng.uniform(-14, 0, 50000),
                        np.array([0.0, -0.0, 709.782712893384, -745.1332191019411])] +
                       [sgn*around(t) for t in (0.34657359027997264, 1.0397207708399179,
                                                3.725290298461914e-09, 0.6931471805599453,
                                                2.0794415416798357, 22.180709777918249)
                        for sgn in (-1., 1.)])
    with np.errstate(over='ignore', under='ignore'):
        assert np.array_equal(ctx.math('exp', e), coracle.math('exp', e))
    mant = lambda h: np.frombuffer(np.array([(0x3ff00000 | h) << 32], dtype=np.uint64).tobytes(),
                                   dtype=np.float64)[0]
    lg = np.concatenate([1 + rng.choice([-1., 1.], 100000)*10**rng.uniform(-16, -0.5, 100000),
                         10**rng.uniform(-320, 300, 100000), rng.uniform(0.25, 4, 400000),
                         np.array([1.0, 0.5, 2.0, 5e-324, 2.2250738585072014e-308])] +
                        [sc*around(mant(h)) for h in (0x6147a, 0x6b851, 0x6a09f, 0x00002, 0xffffd)
                         for sc in (0.25, 0.5, 1.0, 2.0)])
    lg = lg[lg > 0]
    assert np.array_equal(ctx.math('log', lg), coracle.math('log', lg))
    edges = np.concatenate([sc*around((181 + 2*i)/256.0) for i in range(92)
                            for sc in (2.0**-30, 0.5, 1.0, 2.0**40)])
    assert np.array_equal(ctx.math('log', edges), coracle.math('log', edges))
    # and both stay within 1 ulp of the correctly rounded values (NumPy's own are only that too)
    with np.errstate(over='ignore', under='ignore'):
        xs = np.concatenate([lg[np.isfinite(lg)], edges]).astype(np.longdouble)
        ref = np.log(xs)
        got = ctx.math('log', xs.astype(np.float64)).astype(np.longdouble)
        ulp = np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)
        ok = ref != 0
        assert np.max(np.abs(got[ok] - ref[ok])/ulp[ok]) < 1.0
        ee = e[np.abs(e) < 700].astype(np.longdouble)
        refe = np.exp(ee)
        gote = ctx.math('exp', ee.astype(np.float64)).astype(np.longdouble)
        ulpe = np.spacing(refe.astype(np.float64)).astype(np.longdouble)
        assert np.max(np.abs(gote - refe)/ulpe) < 1.0


@pytest.mark.parametrize('gravity,radpres,lifetime', [
    (True, True, 0.0), (True, False, 0.0), (False, True, 0.0), (True, True, 3600.0),
    (True, True, -7200.0), (False, False, 0.0)])
def test_state_matches_oracles(ctx, coracle, gravity, radpres, lifetime):
    f = H.mercury_forces('Na', 1.3, gravity, radpres, lifetime)
    H.set_ctx_forces(ctx, f)
    X = H.random_cloud(4096, 11)
    a_g, i_g = ctx.state(X[:, 1], X[:, 2], X[:, 3], X[:, 5])
    a_c, i_c = coracle.state(f, X[:, 1], X[:, 2], X[:, 3], X[:, 5])
    assert np.array_equal(a_g, a_c) and np.array_equal(i_g, i_c)
    a_n, i_n = O.state(X, f)
    assert np.array_equal(i_g, i_n)
    np.testing.assert_allclose(a_g, a_n, rtol=2e-15, atol=1e-22)


@pytest.mark.parametrize('species,taa', [('Na', 1.3), ('Ca', 0.0), ('Mg', 3.14)])
def test_rk5_step_matches_oracles(ctx, coracle, species, taa):
    f = H.mercury_forces(species, taa)
    H.set_ctx_forces(ctx, f)
    X = H.random_cloud(8192, 5)
    h = np.random.default_rng(2).uniform(1, 120, len(X))
    r_g, d_g = ctx.rk5_step(X, h, want_delta=True)
    r_c, d_c = coracle.rk5(f, X, h, want_delta=True)
    assert np.array_equal(r_g, r_c)
    assert np.array_equal(d_g, d_c)
    r_n, d_n = O.rk5(f, X, h, want_delta=True)
    # (the tableau terms are fused: one rounding where NumPy has two; a component whose terms
    # cancel -- one velocity of 65536 values here -- shows it as 3e-17 absolute)
    np.testing.assert_allclose(r_g, r_n, rtol=1e-13, atol=1e-16)
    r_g2, none = ctx.rk5_step(X, 30.0)
    assert none is None
    r_c2, _ = coracle.rk5(f, X, 30.0)
    assert np.array_equal(r_g2, r_c2)


def _const_case(ctx, coracle, f, n, seed, endtime, step, outeredge, image=None):
    H.set_ctx_forces(ctx, f)
    X0 = H.sample_x0(n, seed, endtime)
    nsteps, n_iter = O.n_output_steps(endtime, step)
    ctx.upload_packets(X0)
    return X0, nsteps, n_iter


def test_const_driver_trajectory_bit_exact(ctx, coracle):
    f = H.mercury_forces('Na', 1.3)
    endtime, step, edge = 6000.0, 30.0, 25.0
    X0, nsteps, n_iter = _const_case(ctx, coracle, f, 3000, 1234, endtime, step, edge)
    g = ctx.integrate_const(step, n_iter, edge, nrec=nsteps, want_final=True, want_steps=True)
    c = coracle.integrate_const(f, X0, step, n_iter, edge, nrec=nsteps, threads=4)
    assert np.array_equal(g['traj'], c['traj'])
    assert np.array_equal(g['steps'], c['steps'])
    assert np.array_equal(g['final'], c['final'])
    assert ctx.counters()['particle_steps'] == c['work']
    # and against the NumPy restatement of the reference driver
    res, _, work = O.constant_step_driver(f, X0[:400], endtime, step, edge)
    np.testing.assert_allclose(g['traj'][:, :, :400].transpose(2, 0, 1), res, rtol=1e-9,
                               atol=1e-12)


def test_const_driver_lane_refill_bit_exact(ctx, coracle):
    """Persistent lane-refill kernel == lock-step kernel == C oracle, packet for packet."""
    f = H.mercury_forces('Na', 1.3)
    endtime, step, edge = 50000.0, 30.0, 25.0
    X0, nsteps, n_iter = _const_case(ctx, coracle, f, 20000, 99, endtime, step, edge)
    g = ctx.integrate_const(step, n_iter, edge, want_final=True, want_steps=True)
    c = coracle.integrate_const(f, X0, step, n_iter, edge, threads=8)
    assert np.array_equal(g['steps'], c['steps'])
    assert np.array_equal(g['final'], c['final'])
    assert ctx.counters()['particle_steps'] == c['work']


@pytest.mark.parametrize('quantity,downcast', [('radiance', False), ('column', False),
                                               ('radiance', True)])
def test_fused_image_matches_oracle(ctx, coracle, quantity, downcast):
    f = H.mercury_forces('Na', 1.3)
    endtime, step, edge = 50000.0, 30.0, 25.0
    X0, nsteps, n_iter = _const_case(ctx, coracle, f, 20000, 7, endtime, step, edge)
    im = H.image_setup(f, quantity, dims=(128, 96), width=(8., 6.), sublon=0.4, sublat=1.1)
    ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                  im['g_tables'], downcast_f32=downcast)
    ctx.integrate_const(step, n_iter, edge, image=True)
    image, counts = ctx.image_download()
    desc = coracle.image_desc(im['M'], f.vrplanet, im['apix'], quantity, im['g_tables'],
                              im['xedges'], im['zedges'], downcast=downcast)
    c = coracle.integrate_const(f, X0, step, n_iter, edge, img=desc, threads=8)
    assert np.array_equal(counts, c['counts'])
    np.testing.assert_allclose(image, c['image'], rtol=1e-11, atol=0)
    ctr = ctx.counters()
    assert ctr['samples_binned'] == int(counts.sum())
    assert ctr['nonfinite'] == 0


def test_image_kernel_matches_numpy_histogram(ctx, coracle):
    f = H.mercury_forces('Na', 1.3)
    rng = np.random.default_rng(5)
    p = 300000
    X = H.random_cloud(p, 21)
    x, y, z, vy, frac = X[:, 1], X[:, 2], X[:, 3], X[:, 5], X[:, 7]
    im = H.image_setup(f, 'radiance', dims=(512, 512))
    # samples exactly on edges, on the right-most edge and outside
    x[:513] = im['xedges']; z[:513] = im['zedges'][::-1]
    x[600:610] = 4.0; z[600:610] = rng.uniform(-4, 4, 10)
    x[700:710] = 4.0000001
    ctx.set_image(im['M'], f.vrplanet, im['apix'], 'radiance', im['xedges'], im['zedges'],
                  im['g_tables'])
    ctx.image_accumulate(x, y, z, vy, frac)
    image, counts = ctx.image_download()
    ref_img, ref_cnt, _, _ = O.create_image(x, y, z, vy, frac, f.vrplanet, im['M'], 'radiance',
                                            im['g_tables'], im['dims'], im['xrange'],
                                            im['zrange'], im['apix'], matmul=False)
    assert np.array_equal(counts, ref_cnt.astype(np.uint64))
    np.testing.assert_allclose(image, ref_img, rtol=1e-12, atol=0)


@pytest.mark.parametrize('quantity', ['radiance', 'column'])
def test_image_kernel_on_float32_samples_equals_the_restored_path(ctx, quantity):
    """nxc_image_accumulate_f32 (samples as Output.save() stores them) against restore()'s 64-bit
    up-cast followed by nxc_image_accumulate: same pixels, same counts, the same weights (one
    sample per pixel-and-lane order aside: atomics), NaN and out-of-range samples included."""
    f = H.mercury_forces('Na', 1.3)
    p = 200001
    X = H.random_cloud(p, 33).astype(np.float32)
    cols = [np.ascontiguousarray(X[:, c]) for c in (1, 2, 3, 5, 7)]
    cols[0][:7] = [np.nan, np.inf, -np.inf, 4.0, -4.0, 3.9999998, 1e30]
    im = H.image_setup(f, quantity, dims=(200, 120), width=(8., 6.))
    ctx.set_image(im['M'], f.vrplanet, im['apix'], quantity, im['xedges'], im['zedges'],
                  im['g_tables'])
    ctx.image_accumulate(*cols)                                    # float32 entry point
    img32, cnt32 = ctx.image_download()
    ctr32 = ctx.counters()
    ctx.image_clear()
    ctx.image_accumulate(*(c.astype(np.float64) for c in cols))
    img64, cnt64 = ctx.image_download()
    assert ctr32 == ctx.counters() and ctr32['samples'] == p
    assert np.array_equal(cnt32, cnt64) and cnt64.sum() > 1000
    np.testing.assert_allclose(img32, img64, rtol=1e-12, atol=0)


@pytest.mark.parametrize('variant', ['by size', 'fair', 'plain'])
def test_variable_driver_bit_exact(ctx, coracle, variant, monkeypatch):
    """k_var has two launch forms of one arithmetic: with clock-rotated wave priorities and a merged
    tail (the live packets of sparse waves handed to one keeper wave per SIMD through LDS) when the
    launch is mostly tail, plain when every lane has dozens of packets.  The library picks by
    size; NXC_TEST_VAR_VARIANT forces one: the same bits whichever runs."""
    if variant != 'by size':
        monkeypatch.setenv('NXC_TEST_VAR_VARIANT', variant)
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    n, endtime = 5000, 20000.0
    X0 = H.sample_x0(n, 4321, endtime)
    X0[:, 0] = np.random.default_rng(8).random(n)*endtime
    ctx.upload_packets(X0)
    g_final, g_hs = ctx.integrate_var(1e-4, 25.0)
    c_final, c_hs, work, bad = coracle.integrate_var(f, X0, 1e-4, 25.0)
    assert bad == 0
    assert np.array_equal(g_final, c_final)
    assert np.array_equal(g_hs, c_hs)
    ctr = ctx.counters()
    assert ctr['particle_steps'] == work
    assert ctr['bad_step'] == ctr['nonfinite'] == ctr['neg_frac'] == ctr['unfinished'] == 0


@pytest.mark.parametrize('variant', ['fair', 'plain'])
def test_variable_driver_attempt_cap_travels_with_the_packet(ctx, coracle, variant, monkeypatch):
    """max_steps ends a packet after that many attempts, wherever they were made: in the fair form
    a packet's attempt count moves with it when a sparse wave hands it to its SIMD's keeper.  4e4
    packets, a cap that a third of them reach: states, stored steps and total work as the C
    oracle's, `unfinished` = the packets still alive at the cap."""
    monkeypatch.setenv('NXC_TEST_VAR_VARIANT', variant)
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    n, endtime, cap = 40_000, 30000.0, 400
    X0 = H.sample_x0(n, 99, endtime)
    X0[:, 0] = np.random.default_rng(3).random(n)*endtime
    ctx.upload_packets(X0)
    g_final, g_hs = ctx.integrate_var(1e-4, 25.0, max_steps=cap)
    ctr = ctx.counters()
    c_final, c_hs, work, bad = coracle.integrate_var(f, X0, 1e-4, 25.0, max_steps=cap)
    assert ctr['particle_steps'] == work and work > 5e6
    assert np.array_equal(g_final, c_final) and np.array_equal(g_hs, c_hs)
    capped = int(((c_final[:, 0] > 1e-4) & (c_final[:, 7] > 0)).sum())     # still flying at the cap
    assert ctr['unfinished'] == capped and n//10 < capped < n


def test_rccl_single_rank_allreduce(ctx, coracle):
    """The RCCL path (dlopen librccl, communicator, fp64 + u64 all-reduce on the handle's stream)
    with a world of one: the image pair must come back unchanged."""
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    X0 = H.sample_x0(5000, 3, 50000.)
    nsteps, n_iter = O.n_output_steps(50000., 30.)
    im = H.image_setup(f, 'radiance', dims=(64, 64))
    ctx.set_image(im['M'], f.vrplanet, im['apix'], 'radiance', im['xedges'], im['zedges'],
                  im['g_tables'])
    ctx.upload_packets(X0)
    ctx.integrate_const(30., n_iter, 25., image=True)
    image0, counts0 = ctx.image_download()
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 0, 1)
    ctx.image_allreduce()
    ctx.barrier()
    assert ctx.allreduce_max(3.25) == 3.25
    # the work / sample totals of bench.py go through nxc_allreduce_sum_f64: exact for every
    # integer-valued double a counter can hold, and for a sum that is not representable in fp32
    for v in (0.0, 1.0, 1280065134.0, 2.0**53 - 1, -7.25, 1e-300, 0.1 + 0.2):
        assert ctx.allreduce_sum(v) == v and ctx.allreduce_max(v) == v
    image1, counts1 = ctx.image_download()
    ctx.comm_destroy()
    assert np.array_equal(image0, image1) and np.array_equal(counts0, counts1)
    assert counts0.sum() > 0
    # after nxc_comm_destroy every collective refuses instead of hanging or falling back
    from nexoclom_amd import hip_api
    for call in (ctx.image_allreduce, ctx.barrier, lambda: ctx.allreduce_sum(1.0)):
        with pytest.raises(hip_api.HipError, match='nxc_comm_init'):
            call()


def test_a_collective_that_never_completes_ends_at_its_deadline(ctx):
    """nxc_comm_set_timeout: a wait on a collective is bounded.  The hang is injected
    (nxc_comm_test_stall holds the stream as a lost peer would); the waiting call must come back
    with NXC_ERR_RCCL once the deadline has passed, the communicator must be gone (ncclCommAbort),
    and the handle must stay usable once the stream has drained.  (ncclCommAbort waits for the
    device: an RCCL kernel leaves as soon as it sees the abort flag, the injected stall only when
    its time is up -- so here the call returns when the stall ends, not at 0.4 s.)"""
    import time
    from nexoclom_amd import hip_api
    f = H.mercury_forces('Na', 1.3)
    H.set_ctx_forces(ctx, f)
    im = H.image_setup(f, 'radiance', dims=(32, 32))
    ctx.set_image(im['M'], f.vrplanet, im['apix'], 'radiance', im['xedges'], im['zedges'],
                  im['g_tables'])
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    # the healthy path goes through the same polling wait
    ctx.image_allreduce()
    ctx.synchronize()
    assert ctx.allreduce(np.array([1.5, -2.0, 7.0])).tolist() == [1.5, -2.0, 7.0]
    ctx.comm_set_timeout(0.4)
    ctx.comm_test_stall(2.0)
    t0 = time.perf_counter()
    with pytest.raises(hip_api.HipError, match='did not complete within 0.4 s') as err:
        ctx.synchronize()
    waited = time.perf_counter() - t0
    assert err.value.code == hip_api.NXC_ERR_RCCL and 0.35 < waited < 8.0     # (2 s + a loaded box)
    with pytest.raises(hip_api.HipError, match='nxc_comm_init'):      # no communicator any more
        ctx.image_allreduce()
    ctx.synchronize()                                                  # the stall ends by itself
    assert time.perf_counter() - t0 > 1.9
    # ... and a peer's failure reported from ANOTHER thread ends the wait at once
    import threading
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    ctx.comm_set_timeout(30.0)
    ctx.comm_test_stall(2.0)
    threading.Timer(0.2, ctx.comm_request_abort).start()
    t0 = time.perf_counter()
    with pytest.raises(hip_api.HipError, match='peer rank reported a failure') as err:
        ctx.synchronize()
    assert err.value.code == hip_api.NXC_ERR_RCCL and time.perf_counter() - t0 < 8.0
    ctx.synchronize()
    # the handle works on: a fresh communicator, a collective, a pass
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    assert ctx.allreduce_sum(4.0) == 4.0
    ctx.comm_destroy()
    X0 = H.sample_x0(512, 3, 50000.)
    ctx.upload_packets(X0)
    nsteps, n_iter = O.n_output_steps(50000., 30.)
    ctx.image_clear()
    ctx.integrate_const(30., n_iter, 25., image=True)
    assert ctx.counters()['particle_steps'] > 0


@pytest.mark.parametrize('bounce', [False, True])
def test_compact_rows_equal_the_filtered_dense_trajectory(ctx, coracle, bounce):
    """nxc_integrate_const_rows/nxc_rows_fetch == the frac > 0 records of the dense trajectory in
    packet-major order (what Output.save() keeps, Output.py:523-524), bit for bit, with lossfrac
    accumulated in the reference's association (Output.py:420-421)."""
    f = H.mercury_forces('Na', 1.3)
    endtime, step = 6000., 30.
    n = 3000
    X0 = H.sample_x0(n, 21, endtime)
    X0[::17, 7] = 0.0                                    # packets that start dead: no rows
    nsteps, n_iter = O.n_output_steps(endtime, step)
    H.set_ctx_forces(ctx, f)
    if bounce:
        from nexoclom_amd.surface import SurfaceInteraction
        import types
        sint = types.SimpleNamespace(sticktype='constant', stickcoef=0.4, accomfactor=0.0)
        cfg = dict(GM=f.GM, unit_km=f.R_km, accomfactor=0.0, stickcoef=0.4, A=(0., 0., 0.), t0=100.,
                   t1=600., temp_dependent=False, tx=np.zeros(8), ty=np.zeros(8), coef=np.zeros(16),
                   seed=77)
        ctx.set_bounce(cfg)
    try:
        ctx.upload_packets(X0)
        dense = ctx.integrate_const(step, n_iter, 6.0, nrec=nsteps)['traj']      # (8, nsteps, N)
        ctx.upload_packets(X0)
        res = ctx.integrate_const_rows(step, n_iter, 6.0)
        assert ctx.counters()['unfinished'] == 0
        narrow = ctx.integrate_const_rows(step, n_iter, 6.0, narrow=True)      # nxc_rows_fetch_f32
    finally:
        ctx.set_bounce(None)
    frac = dense[7].T                                                            # (N, nsteps)
    live = frac > 0
    assert np.array_equal(res['lengths'], live.sum(1))
    # live records are a prefix of each packet's step axis
    assert np.array_equal(live, np.arange(nsteps)[None, :] < res['lengths'][:, None])
    for c in range(8):
        assert np.array_equal(res['rows'][c], dense[c].T[live])
    lossfrac = np.zeros_like(frac)
    for ct in range(1, nsteps):
        act = frac[:, ct-1] > 0
        lossfrac[act, ct] = (lossfrac[act, ct-1] + frac[act, ct-1]) - frac[act, ct]
    assert np.array_equal(res['rows'][8], lossfrac[live])
    # the rows narrowed on the device = save()'s float32 down-cast of the same rows (Output.py:528-543)
    assert narrow['rows'].dtype == np.float32 and np.array_equal(narrow['lengths'], res['lengths'])
    assert np.array_equal(narrow['rows'], res['rows'].astype(np.float32))
    if not bounce:
        c = coracle.integrate_const(f, X0, step, n_iter, 6.0, nrec=nsteps)
        assert np.array_equal(dense, c['traj'])


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_lookup_table_equals_np_interp_on_adversarial_tables(ctx, seed):
    """The LDS lookup (clamp-free cell index, host-bisected cell table, two-row probe, rare walk)
    against np.interp itself, through nxc_state with gravity off (ay = interp(vy + vrplanet) for a
    sunlit packet): tables with nodes clustered far below the cell width, two-point tables and a
    Na-sized irregular one; abscissae at the nodes, one ulp either side of them, between them, at
    and beyond both ends, huge, and NaN.  Bit-exact."""
    rng = np.random.default_rng(seed)
    n = [2, 40, 827, 1200][seed]
    if seed == 1:                       # clusters: many nodes inside single cells
        base = np.sort(rng.uniform(-1, 1, 8))
        xp = np.unique(np.concatenate([base + k*1e-9 for k in range(5)]))
    elif seed == 0:
        xp = np.array([-0.37, 0.91])
    else:
        xp = np.unique(np.cumsum(rng.choice([1e-6, 3e-3, 1e-2, 0.2], size=n, p=[.05, .4, .4, .15])))
        xp = xp - xp.mean()
    fp = rng.normal(size=len(xp))
    vr = 0.125
    f = O.Forces(GM=-1.5e-6, vrplanet=vr, gravity=False, radpres=True, lifetime=0., photo=None,
                 v_tab=xp, a_tab=fp)
    H.set_ctx_forces(ctx, f)
    span = xp[-1] - xp[0]
    xs = np.concatenate([xp, np.nextafter(xp, np.inf), np.nextafter(xp, -np.inf),
                         rng.uniform(xp[0] - 0.3*span, xp[-1] + 0.3*span, 200000),
                         (xp[:-1] + xp[1:])/2, [xp[0] - 1e-300, xp[-1] + 1e300, -1e308, 1e308,
                                                0.0, -0.0, np.nan]])
    vy = xs - vr
    keep = (vy + vr == xs) | np.isnan(xs)          # only abscissae the kernel reconstructs exactly
    xs, vy = xs[keep], vy[keep]
    m = len(xs)
    # a sunlit position: y < 0
    a, ion = ctx.state(np.zeros(m), -2*np.ones(m), np.zeros(m), vy)
    want = np.interp(xs, xp, fp)
    assert keep.sum() > 100000
    assert np.array_equal(a[:, 1], want, equal_nan=True)
    assert np.array_equal(a[:, 0], np.zeros(m)) and np.array_equal(ion, np.zeros(m))


@pytest.mark.parametrize('narrow', [False, True])
def test_resident_rows_equal_the_fetched_rows_and_feed_the_image(ctx, narrow):
    """nxc_rows_build keeps pass 2's rows in HBM: what comes back through nxc_rows_download (whole
    store and a row range) is what nxc_rows_fetch[_f32] delivers, the index column is the packet
    number of every row (the reference's X.Index, Output.py:438), and nxc_image_accumulate_rows
    over the store gives the image of the same columns sent from the host."""
    f = H.mercury_forces('Na', 1.3)
    endtime, step = 9000., 30.
    n = 5000
    X0 = H.sample_x0(n, 33, endtime)
    X0[::23, 7] = 0.0
    nsteps, n_iter = O.n_output_steps(endtime, step)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None)
    ctx.upload_packets(X0)
    fetched = ctx.integrate_const_rows(step, n_iter, 8.0, narrow=narrow)
    res = ctx.integrate_const_rows(step, n_iter, 8.0, narrow=narrow, resident=True)
    assert ctx.counters()['unfinished'] == 0
    store = res['store']
    assert store.total == fetched['rows'].shape[1] == res['lengths'].sum() and store.narrow == narrow
    rows, index = store.download()
    assert rows.dtype == fetched['rows'].dtype and np.array_equal(rows, fetched['rows'])
    assert index.dtype == (np.int32 if narrow else np.int64)
    assert np.array_equal(index, np.repeat(np.arange(n), res['lengths']))
    part, pidx = store.download(1234, 40000)
    assert np.array_equal(part, rows[:, 1234:41234]) and np.array_equal(pidx, index[1234:41234])
    # the image of rows [a, b) from HBM == the image of the same columns sent from the host
    im = H.image_setup(f, 'radiance', dims=(96, 96))
    a, b = 777, store.total - 999
    for src in ('host', 'store'):
        ctx.set_image(im['M'], f.vrplanet, im['apix'], 'radiance', im['xedges'], im['zedges'],
                      im['g_tables'])
        if src == 'host':
            ctx.image_accumulate(*(rows[c, a:b] for c in (1, 2, 3, 5, 7)))
            want, want_counts = ctx.image_download()
            want_ctr = ctx.counters()
        else:
            ctx.image_accumulate_rows(store, a, b - a)
            got, got_counts = ctx.image_download()
            assert ctx.counters() == want_ctr
    assert want_counts.sum() > 1e5 and np.array_equal(got_counts, want_counts)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
    store.free()
    with pytest.raises(Exception):
        store.download()


def test_streamed_pass_equals_upload_then_integrate(ctx):
    """nxc_integrate_const_streamed (upload, queue order and integration pipelined over pieces on
    two compute streams + a copy stream) is the same run as nxc_packets_upload +
    nxc_integrate_const: the same work and sample counters, the same packet-count image, weights
    to fp64 summation order -- for one piece, several, more pieces than fit evenly, and pieces
    of a single packet; and the handle stays usable (a plain pass over the now-resident packets)."""
    f = H.mercury_forces('Na', 1.3)
    nsteps, n_iter = O.n_output_steps(20000., 30.)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None)
    ctx.set_bodies(None)
    im = H.image_setup(f, 'radiance', dims=(128, 128))
    for n, pieces in ((200003, 8), (200003, 1), (70001, 3), (9, 32), (5000, 32)):
        X0 = H.sample_x0(n, 77 + pieces, 20000.)
        soa = np.ascontiguousarray(X0.T)
        ctx.set_image(im['M'], f.vrplanet, im['apix'], 'radiance', im['xedges'], im['zedges'],
                      im['g_tables'], downcast_f32=True)
        ctx.upload_soa(soa)
        ctx.integrate_const(30., n_iter, 25., image=True)
        want_ctr = ctx.counters()
        want, want_counts = ctx.image_download()
        ctx.image_clear()
        ctx.integrate_const_streamed(soa, 30., n_iter, 25., image=True, pieces=pieces)
        ctx.synchronize()
        assert ctx.counters() == want_ctr, (n, pieces)
        got, got_counts = ctx.image_download()
        assert np.array_equal(got_counts, want_counts) and want_counts.sum() > 100
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=0)
        ctx.image_clear()
        ctx.integrate_const(30., n_iter, 25., image=True)       # the packets are resident now
        assert ctx.counters() == want_ctr
        assert np.array_equal(ctx.image_download()[1], want_counts)


def test_a_streamed_pass_that_gives_up_is_reported_and_the_handle_lives_on(ctx, monkeypatch):
    """The pipelined pass is the one launch whose kernel may end by itself before its work is
    done: a wave that has waited three seconds for the next piece of the queue leaves.  Forced
    here by never publishing the last piece (NXC_TEST_WITHHOLD_LAST_PIECE): the synchronize that
    follows must raise NXC_ERR_INCOMPLETE -- not hand over the partial image -- and the handle
    must work on: upload + plain pass give the full result."""
    from nexoclom_amd import hip_api
    f = H.mercury_forces('Na', 1.3)
    nsteps, n_iter = O.n_output_steps(20000., 30.)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None); ctx.set_bodies(None)
    im = H.image_setup(f, 'radiance', dims=(64, 64))
    ctx.set_image(im['M'], f.vrplanet, im['apix'], 'radiance', im['xedges'], im['zedges'],
                  im['g_tables'], downcast_f32=True)
    soa = np.ascontiguousarray(H.sample_x0(60000, 12, 20000.).T)
    ctx.upload_soa(soa)
    ctx.integrate_const(30., n_iter, 25., image=True)
    want_ctr, (want, want_counts) = ctx.counters(), ctx.image_download()
    ctx.image_clear()
    monkeypatch.setenv('NXC_TEST_WITHHOLD_LAST_PIECE', '1')
    ctx.integrate_const_streamed(soa, 30., n_iter, 25., image=True, pieces=4)
    monkeypatch.delenv('NXC_TEST_WITHHOLD_LAST_PIECE')
    with pytest.raises(hip_api.HipError, match='gave up waiting for its queue') as err:
        ctx.synchronize()
    assert err.value.code == hip_api.NXC_ERR_INCOMPLETE
    part = ctx.counters()
    assert part['unfinished'] > 0 and 0 < part['particle_steps'] < want_ctr['particle_steps']
    assert ctx.n_packets == 0
    with pytest.raises(hip_api.HipError, match='no resident packets'):
        ctx.integrate_const(30., n_iter, 25., image=True)
    ctx.synchronize()                                            # the error is reported once
    # the sequential form, and then the pipelined pass again, unharmed
    ctx.image_clear()
    ctx.upload_soa(soa)
    ctx.integrate_const(30., n_iter, 25., image=True)
    assert ctx.counters() == want_ctr and np.array_equal(ctx.image_download()[1], want_counts)
    ctx.image_clear()
    ctx.integrate_const_streamed(soa, 30., n_iter, 25., image=True, pieces=4)
    ctx.synchronize()
    assert ctx.counters() == want_ctr and np.array_equal(ctx.image_download()[1], want_counts)


def test_rows_of_runs_without_live_records_and_of_a_single_packet(ctx):
    """Edge cases of the rows protocol: packets that all start dead (no row at all: an empty store
    that can still be downloaded, binned and freed), and a single packet (one lane of one wave;
    the launch is sized to the packets, not to the chip)."""
    f = H.mercury_forces('Na', 1.3)
    nsteps, n_iter = O.n_output_steps(3000., 30.)
    H.set_ctx_forces(ctx, f)
    ctx.set_bounce(None)
    ctx.set_bodies(None)
    im = H.image_setup(f, 'column', dims=(32, 32))
    ctx.set_image(im['M'], f.vrplanet, im['apix'], 'column', im['xedges'], im['zedges'], [])
    dead = H.sample_x0(300, 3, 3000.)
    dead[:, 7] = 0.0
    ctx.upload_packets(dead)
    for narrow in (False, True):
        res = ctx.integrate_const_rows(30., n_iter, 25., narrow=narrow, resident=True)
        assert res['lengths'].sum() == 0 and res['store'].total == 0
        rows, index = res['store'].download()
        assert rows.shape == (9, 0) and index.shape == (0,)
        ctx.image_accumulate_rows(res['store'])
        assert ctx.image_download()[1].sum() == 0
        res['store'].free()
        assert ctx.integrate_const_rows(30., n_iter, 25., narrow=narrow)['rows'].shape == (9, 0)
    dense = ctx.integrate_const(30., n_iter, 25., nrec=nsteps)['traj']
    assert np.array_equal(dense[:, 0, :], dead.T) and not dense[:, 1:, :].any()
    one = H.sample_x0(1, 5, 3000.)
    ctx.upload_packets(one)
    res = ctx.integrate_const_rows(30., n_iter, 25., resident=True)
    rows, index = res['store'].download()
    dense = ctx.integrate_const(30., n_iter, 25., nrec=nsteps)['traj']
    live = dense[7, :, 0] > 0
    assert res['lengths'][0] == live.sum() == rows.shape[1] >= 1 and not index.any()
    for c in range(8):
        assert np.array_equal(rows[c], dense[c, live, 0])
