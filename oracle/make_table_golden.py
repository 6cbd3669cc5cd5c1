"""Golden vectors for the set-up tables, computed in the build
container STRAIGHT FROM THE REFERENCE'S TEXT DATA FILES by a restatement that shares no code with
nexoclom_amd/ (TEST INFRASTRUCTURE; run where /root/reference exists):

    python oracle/make_table_golden.py        ->  tests/golden/g7_tables.npz

Sources: nexoclom/data/g-values/g-values_old.csv (the reference's dump of the table its
gValue / RadPresConst read, atomicdata/g_values.py:72-94,141-160), nexoclom/data/Loss/Photo/*.dat
(atomicdata/initialize_atomicdata.py:66-89 -> photolossrates.py:84-86) and
nexoclom/data/PlanetaryConstants.dat (solarsystem/SSObject.py:102-114, planet_dist.py:36-67).
The reference's own classes need astropy and cannot be imported here, so
these arrays are "the reference's data through an independent restatement of its formulas", which
is what tests/test_host.py::test_tables_match_the_text_file_restatement compares the product's
tables with; together with the survey's recorded Mercury numbers and the reference's PhotoRate
known answers that is the pin of row a-5 (beyond it: parity unpinned).
"""
import csv
import glob
import os

import numpy as np

REF = '/root/reference/nexoclom/data'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                   'g7_tables.npz')
H, AMU, AU_KM, G = 6.62607015e-34, 1.66053906660e-27, 1.495978707e8, 6.6743e-11
MASS = {'Na': 22.98976928, 'Ca': 40.078, 'Mg': 24.305}
LINES = {'Na': (3303., 5891., 5897.), 'Ca': (2722., 4227., 4567.), 'Mg': (2852.,)}
DISTANCES = (0.3, 0.35140097909804036, 1.5)


def g_rows():
    with open(os.path.join(REF, 'g-values', 'g-values_old.csv'), newline='') as fh:
        return [(r['species'], float(r['wavelength']), float(r['velocity']), float(r['gvalue']),
                 float(r['refpoint'])) for r in csv.DictReader(fh)]


def g_of_v(rows, sp, wave, a):
    sel = [(v, g, ref) for s, w, v, g, ref in rows if s == sp and w == wave]
    v, g, ref = (np.array(c) for c in zip(*sel))
    k = np.argsort(v)
    return v[k], (g*ref**2/a**2)[k]


def a_rad(rows, sp, a):
    waves = sorted({w for s, w, *_ in rows if s == sp})
    grid = np.array(sorted({v for s, _, v, *_ in rows if s == sp}))
    acc = np.zeros_like(grid)
    for w in waves:
        v, g = g_of_v(rows, sp, w, a)
        acc += H/(MASS[sp]*AMU)/(w*1e-10)*np.interp(grid, v, g)*1e-3
    return np.array(waves), grid, acc


def photo_rates():
    kappa = {}
    for f in sorted(glob.glob(os.path.join(REF, 'Loss', 'Photo', '*.dat'))):
        for line in open(f):
            p = line.split(':')
            if len(p) == 4:
                kappa.setdefault(p[0].strip(), []).append(float(p[2].strip()))
    return kappa


def mercury_orbit():
    for line in open(os.path.join(REF, 'PlanetaryConstants.dat')):
        p = [x.strip() for x in line.split(':')]
        if p[0] == 'Mercury':
            return dict(radius=float(p[2]), mass=float(p[3]), a=float(p[4]), e=float(p[5]),
                        period=float(p[8])*86400.)
    raise RuntimeError('Mercury not found')


def dist_and_vr(o, taa):
    a, e = o['a'], o['e']
    t = np.linspace(0, 1, 1000)*o['period']
    t = np.concatenate([[t[0]-t[1]], t])
    m = np.linspace(0, 2*np.pi, 1000)
    m = np.concatenate([[m[0]-m[1]], m])
    nu = m + (2*e - e**3/4)*np.sin(m) + 5/4*e**2*np.sin(2*m) + 13/12*e**3*np.sin(3*m)
    r = a*(1-e**2)/(1+e*np.cos(nu))
    drdt = (r[1:]-r[:-1])/(t[1:]-t[:-1])
    return a*(1-e**2)/(1+e*np.cos(taa)), np.interp(taa, nu[1:], drdt*AU_KM)


def main():
    rows = g_rows()
    out = {'distances': np.array(DISTANCES)}
    for sp in ('Na', 'Ca', 'Mg'):
        for k, a in enumerate(DISTANCES):
            waves, grid, acc = a_rad(rows, sp, a)
            assert tuple(waves) == LINES[sp]
            out[f'{sp}_radpres_v'] = grid
            out[f'{sp}_radpres_a{k}'] = acc
            for w in waves:
                v, g = g_of_v(rows, sp, w, a)
                out[f'{sp}_{int(w)}_v'] = v
                out[f'{sp}_{int(w)}_g{k}'] = g
    kappa = photo_rates()
    for sp in ('Na', 'Ca', 'Mg', 'K', 'O'):
        out[f'{sp}_photo'] = np.array([np.array([k/a**2 for k in kappa[sp]]).sum()
                                       for a in DISTANCES])
    o = mercury_orbit()
    taas = np.array([0., 0.7, 1.3, 3.14, 4.5])
    out['mercury_taa'] = taas
    out['mercury_r_vr'] = np.array([dist_and_vr(o, t) for t in taas])
    out['mercury_GM_R3'] = np.array(-o['mass']*G/(o['radius']*1e3)**3)
    # the survey's numbers, recorded from the reference itself
    r, vr = dist_and_vr(o, 1.3)
    assert r == 0.35140097909804036 and abs(vr/9.730746760831499 - 1) < 1e-12
    assert abs(out['Na_photo'][1]/5.8793685680196064e-05 - 1) < 1e-15
    assert len(out['Na_radpres_v']) == 827 and abs(out['Na_radpres_a1'].max()*1e5 - 361.09) < 0.01
    np.savez_compressed(OUT, **out)
    print(f'{OUT}: {len(out)} arrays, {os.path.getsize(OUT)/1024:.1f} KiB')


if __name__ == '__main__':
    main()
