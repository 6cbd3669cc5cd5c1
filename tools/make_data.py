"""Build the physical-data tables shipped in nexoclom_amd/data/ from the TEXT data files of
the reference tree (run in the build container only; /root/reference is absent on the GPU box).

Sources (all plain text; no pickle is read):
  nexoclom/data/g-values/g-values_old.csv   CSV mirror of the g-value table consumed by
        atomicdata/g_values.py:72-94,141-160 (2333 rows: species, wavelength [A], velocity [km/s],
        gvalue [1/s at refpoint], refpoint [au]); cross-checked below against the per-line .dat
        files that atomicdata/initialize_atomicdata.py:11-64 parses.
  nexoclom/data/Loss/Photo/*.dat            photo-reaction rates parsed with the rule of
        atomicdata/initialize_atomicdata.py:66-89 ("species : reaction : kappa : x" lines).
  nexoclom/data/PlanetaryConstants.dat      ':'-separated table read as solarsystem/SSObject.py:102-114.

Outputs are compact CSV files (data, not source): gvalues.csv, photorates.csv,
planetary_constants.csv, every double written with 17 significant digits.

Parsing note: the .dat files are read with pandas' DEFAULT float parser on purpose -- that is what
the reference's own table builders use (initialize_atomicdata.py, SSObject.py:102-114), and it is
not correctly rounded (about one value in five of the long g-value decimals lands 1 ulp off), so
the doubles in the reference's tables are these, not the nearest doubles of the printed decimals;
the cross-check against g-values_old.csv (the reference's dump of its own table) confirms it.  The
CSVs written here are then read back by nexoclom_amd with a correctly rounded parser
(atomicdata._lines, solarsystem._bodies), which returns exactly these doubles.
"""
import glob
import os
import sys

import numpy as np
import pandas as pd

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'nexoclom_amd', 'data')


def gvalues():
    src = os.path.join(REF, 'nexoclom', 'data', 'g-values')
    tab = pd.read_csv(os.path.join(src, 'g-values_old.csv'))
    tab['source'] = tab.filename.apply(lambda s: os.path.basename(s))
    tab = tab[['species', 'wavelength', 'velocity', 'gvalue', 'refpoint', 'source']]
    # cross-check every (species, wavelength) that still has a .dat file in the tree
    checked = 0
    for f in sorted(glob.glob(os.path.join(src, '*.dat'))):
        sp = os.path.basename(f).split('.')[0]
        with open(f) as fh:
            ref = float(fh.readline().split('=')[1])
        t = pd.read_csv(f, sep=':', skiprows=1)
        waves = [float(w) for w in t.columns[1:]]
        for k, w in enumerate(waves):
            sub = tab[(tab.species == sp) & (tab.wavelength == w)
                      & (tab.source == os.path.basename(f))]
            if len(sub) == 0:
                continue
            assert np.array_equal(sub.velocity.values, t.iloc[:, 0].values.astype(float)), (sp, w)
            assert np.array_equal(sub.gvalue.values, t.iloc[:, k+1].values.astype(float)), (sp, w)
            assert np.all(sub.refpoint.values == ref)
            checked += 1
    print(f'g-values: {len(tab)} rows, {checked} (species, line) tables cross-checked vs .dat')
    tab.to_csv(os.path.join(OUT, 'gvalues.csv'), index=False, float_format='%.17g')


def photorates():
    rows = []
    for f in sorted(glob.glob(os.path.join(REF, 'nexoclom', 'data', 'Loss', 'Photo', '*.dat'))):
        ref = ''
        for line in open(f):
            if 'reference' in line.lower():
                ref = line.split('//')[0].strip()
            elif len(line.split(':')) == 4:
                p = line.split(':')
                rows.append((p[0].strip(), p[1].strip(), float(p[2].strip()), ref))
    tab = pd.DataFrame(rows, columns=['species', 'reaction', 'kappa', 'reference'])
    print(f'photorates: {len(tab)} rows')
    tab.to_csv(os.path.join(OUT, 'photorates.csv'), index=False, float_format='%.17g')


def planets():
    f = os.path.join(REF, 'nexoclom', 'data', 'PlanetaryConstants.dat')
    tab = pd.read_csv(f, skipinitialspace=True, skip_blank_lines=True, comment='#', sep=':')
    tab.columns = [c.strip() for c in tab.columns]
    tab.Object = tab.Object.apply(lambda x: x.strip())
    tab.orbits = tab.orbits.apply(lambda x: x.strip())
    print(f'planetary constants: {len(tab)} rows')
    tab.to_csv(os.path.join(OUT, 'planetary_constants.csv'), index=False, float_format='%.17g')


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    gvalues()
    photorates()
    planets()
