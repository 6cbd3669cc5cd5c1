"""The reference's own parser fixtures: its 19 inputfiles (tests/test_data/inputfiles/, copied as
DATA to tests/golden/inputfiles/) and the expected section ``__dict__``s its regression test holds
(tests/unit_tests/Initial_state/test_input_classes.py:17-143, transcribed to
tests/golden/input_classes_expected.json: quantities as bare numbers, SSObjects as names)."""
import json
import os

import pytest

from nexoclom_amd import Input, SSObject
from nexoclom_amd.units import Quantity

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FILES = os.path.join(GOLD, 'inputfiles')
EXPECTED = json.load(open(os.path.join(GOLD, 'input_classes_expected.json')))


def plain(value):
    """A section attribute as the JSON fixture spells it."""
    if isinstance(value, SSObject):
        return value.object
    if isinstance(value, (set, frozenset)):
        return sorted(plain(v) for v in value)
    if isinstance(value, (tuple, list)):
        return [plain(v) for v in value]
    if isinstance(value, Quantity):
        return float(value)
    return value


def section_dict(filename, section):
    return {k: plain(v) for k, v in getattr(Input(os.path.join(FILES, filename)),
                                            section).__dict__.items()}


CASES = [(sec.replace('_per_code', ''), fn, want)
         for sec in ('geometry', 'surfaceinteraction', 'surfaceinteraction_per_code', 'forces',
                     'spatialdist')
         for fn, want in EXPECTED[sec].items()]


@pytest.mark.parametrize('section,filename,want', CASES, ids=[c[1] for c in CASES])
def test_section_matches_the_reference_expectation(section, filename, want):
    got = section_dict(filename, section)
    assert set(got) == set(want)
    for key, value in want.items():
        if isinstance(value, float) or (isinstance(value, list) and value
                                        and isinstance(value[0], float)):
            assert got[key] == pytest.approx(value, rel=1e-15), key
        else:
            assert got[key] == value, key


def test_units_of_the_parsed_quantities():
    g = Input(os.path.join(FILES, 'Geometry.01.input')).geometry
    assert all(p.unit == 'rad' for p in g.phi) and g.taa.unit == 'rad'
    assert all(p.unit == 'rad' for p in g.subsolarpoint)
    s = Input(os.path.join(FILES, 'Spatial.02.input')).spatialdist
    assert all(p.unit == 'rad' for p in s.longitude + s.latitude)


def test_section_equality_rules():
    """test_input_classes.py:55-57,75-76: a section equals itself and differs from the others."""
    g = [Input(os.path.join(FILES, f'Geometry.0{k}.input')).geometry for k in (1, 2, 3)]
    assert g[0] == g[0] and g[0] != g[1] and g[0] != g[2]
    s = [Input(os.path.join(FILES, f'SurfaceInteraction.0{k}.input')).surfaceinteraction
         for k in (1, 2)]
    assert s[0] == s[0] and s[0] != s[1]
    assert Input(os.path.join(FILES, 'Forces.01.input')) == Input(os.path.join(FILES, 'Forces.01.input'))
    assert Input(os.path.join(FILES, 'Forces.01.input')) != Input(os.path.join(FILES, 'Forces.02.input'))


def test_every_reference_inputfile_parses():
    """All 19 files build the seven sections; spot checks on the two reference runs
    (tests/system_tests/test_run_through.py uses Ca.reference / Na.reference)."""
    names = sorted(os.listdir(FILES))
    assert len(names) == 19
    for fn in names:
        inp = Input(os.path.join(FILES, fn))
        assert inp.options.endtime.value > 0 and inp.options.species in ('Na', 'Ca')
        assert str(inp).count('geometry.planet') == 1
    na = Input(os.path.join(FILES, 'Na.reference.input'))
    assert na.spatialdist.type == 'surface spot' and float(na.spatialdist.sigma) == 0.8726646259971648
    assert na.speeddist.type == 'maxwellian' and na.speeddist.temperature.value == 1200.
    assert na.speeddist.temperature.unit == 'K'
    assert na.options.step_size == 30. and na.options.outeredge == 25. and na.options.resolution is None
    assert na.forces.gravity and na.forces.radpres and na.options.lifetime.value == 0.
    ca = Input(os.path.join(FILES, 'Ca.reference.input'))
    assert ca.options.species == 'Ca'
    sm = Input(os.path.join(FILES, 'Ca.surfacemap.maxwellian.input'))
    assert sm.spatialdist.type == 'surface map' and sm.spatialdist.coordinate_system == 'solar-fixed'
    assert sm.spatialdist.mapfile == 'default' and sm.spatialdist.subsolarlon is None
    g = Input(os.path.join(FILES, 'Gravity.input'))
    assert g.forces.gravity is True and g.forces.radpres is False
