"""CPU side of the moons / torus extension (BASELINE config 5; no reference implementation --
Output.py:153-155 asserts -- so PARITY UNPINNED): the two oracles agree, and the host set-up
(inputfile keys, moon placement, launch from a moon) is self-consistent."""
import os

import numpy as np
import pytest

import nexoclom_amd
from nexoclom_amd import Input
from nexoclom_amd.Output import Output
from oracle import np_oracle as O
from tests import helpers as H

INFILE = os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles', 'Na.io.torus.input')


def test_numpy_and_c_oracle_agree_with_moons(coracle):
    f = H.mercury_forces('Na', 1.3)
    b = O.Bodies(gm=(f.GM*0.01, f.GM*0.003), radius=(0.05, 0.03), a=(3.0, 5.0),
                 omega=(2e-4, 1e-4), phi=(0.3, 2.0), t0=3000.0, chx_on=True, chx_k0=1e-4,
                 chx_rho0=3.0, chx_width=1.0, chx_height=0.5, chx_omega=3e-4)
    X0 = H.sample_x0(300, 5, 3000.0)
    nsteps, n_iter = O.n_output_steps(3000.0, 30.0)
    res, work = O.constant_step_driver_bodies(f, b, X0, 3000.0, 30.0, 8.0)
    c = coracle.integrate_const(f, X0, 30.0, n_iter, 8.0, nrec=nsteps, bodies=b)
    assert work == c['work']
    np.testing.assert_allclose(c['traj'].transpose(2, 0, 1), res, rtol=1e-11, atol=1e-13)
    plain = coracle.integrate_const(f, X0, 30.0, n_iter, 8.0, nrec=nsteps)
    assert not np.array_equal(plain['traj'], c['traj'])


def test_moon_positions_follow_the_documented_phase_convention():
    # docs/nexoclom/inputfiles.rst:72-77: 0 = superior conjunction (behind the planet seen from
    # the Sun, which sits at -y), pi/2 = over the dawn terminator (-x)
    b = O.Bodies(gm=(0.,), radius=(0.,), a=(2.0,), omega=(1e-3,), phi=(0.0,), t0=0.)
    assert np.allclose(O.moon_xy(b, 0, 0, 0, 1.0), (0.0, 2.0))
    b.phi = (np.pi/2,)
    assert np.allclose(O.moon_xy(b, 0, 0, 0, 1.0), (-2.0, 0.0))
    # earlier in time (t_remaining > 0) the moon is at a smaller phase
    b.t0 = 100.
    x, y = O.moon_xy(b, 0, 0, 0, 1.0)
    assert np.allclose((x, y), (-2*np.sin(np.pi/2 - 0.1), 2*np.cos(np.pi/2 - 0.1)))


def test_config5_inputfile_sets_up_io_europa_and_the_torus():
    inputs = Input(INFILE)
    assert inputs.geometry.startpoint == 'Io'
    assert {o.object for o in inputs.geometry.objects} == {'Jupiter', 'Io', 'Europa'}
    assert inputs.options.chx['k0'] == 5e-6 and inputs.options.chx['corotation']
    out = Output(inputs, 2000, seed=3, integrate=False, save=False)
    bd = out._bodies
    assert [m['name'] for m in bd['moons']] == ['Io', 'Europa']
    io = bd['moons'][0]
    # Kepler: omega^2 a^3 = |GM_planet| to the accuracy of the tabulated period
    assert abs(io['omega']**2*io['a']**3/(-out.GM) - 1) < 2e-3
    assert abs(bd['chx']['omega'] - 2*np.pi/(9.925*3600)) < 1e-9
    # packets sit on the exobase sphere around Io at launch and move with it
    X = out.X0
    ang = io['phi'] - io['omega']*X['time'].values
    mx, my = -io['a']*np.sin(ang), io['a']*np.cos(ang)
    d = np.sqrt((X.x - mx)**2 + (X.y - my)**2 + X.z**2)/io['radius']
    np.testing.assert_allclose(d, 1.3, rtol=1e-9)
    vmx, vmy = -io['a']*io['omega']*np.cos(ang), -io['a']*io['omega']*np.sin(ang)
    spin_x, spin_y = -io['omega']*(X.y - my), io['omega']*(X.x - mx)
    vrel = np.sqrt((X.vx - vmx - spin_x)**2 + (X.vy - vmy - spin_y)**2 + X.vz**2)*out.unit_km
    assert vrel.min() >= 0.5 - 1e-9 and vrel.max() <= 4.5 + 1e-9       # flat 2.5 +- 2 km/s
    np.testing.assert_allclose(vrel, X.v*out.unit_km, rtol=1e-9)


def test_single_body_inputs_have_no_bodies_and_variable_step_refuses_them():
    inputs = Input(os.path.join(os.path.dirname(nexoclom_amd.__file__), 'inputfiles',
                                'Na.mercury.bench.input'))
    out = Output(inputs, 10, seed=1, integrate=False, save=False)
    assert out._bodies is None
    jup = Input(INFILE)
    jup.options.step_size = 0.
    with pytest.raises(NotImplementedError):
        Output(jup, 10, seed=1, integrate=False, save=False)


def test_collinear_point_dynamics_in_the_c_oracle(coracle):
    """The physics pin of tests/test_gpu_bodies.py::test_collinear_point_of_the_planet_moon_system
    on the CPU side: the oracle's moon model too reproduces the closed-form dynamics around the
    inner collinear point (equilibrium, growth rate and direction of the unstable mode)."""
    case = H.collinear_case()
    f = O.Forces(GM=case['GM'], vrplanet=0.0, gravity=True, radpres=False, lifetime=0.0, photo=None)
    b = O.Bodies(gm=(case['gm'],), radius=(1e-3,), a=(case['a'],), omega=(case['omega'],),
                 phi=(case['phi'],), t0=case['T'], chx_on=False)
    c = coracle.integrate_const(f, case['X0'], case['step'], case['n_iter'], 1e6,
                                nrec=case['n_iter'] + 1, bodies=b)
    H.check_collinear_run(case, c['traj'])
