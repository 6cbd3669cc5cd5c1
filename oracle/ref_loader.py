"""Load the numpy-only files of the reference's hot path BY PATH (build container only).

TEST INFRASTRUCTURE: used by oracle/make_golden.py to pin the oracle and to generate the golden
vectors under tests/golden/.  ``import nexoclom`` itself cannot work here (its __init__ connects
to PostgreSQL and needs astropy/psycopg), but particle_tracking/rk5.py and state.py import only
numpy and each other, and math/histogram.py, math/rotation_matrix.py only numpy, so they load
under empty stub parent packages.  /root/reference does not exist on the GPU box: nothing in the
test-suite calls this module at run time.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get('NEXOCLOM_REFERENCE', '/root/reference')


def available():
    return os.path.isfile(os.path.join(REF_ROOT, 'nexoclom', 'particle_tracking', 'rk5.py'))


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load():
    """Return (rk5_module, state_module, histogram_module, rotation_matrix_module)."""
    for name in ('nexoclom', 'nexoclom.particle_tracking', 'nexoclom.math'):
        if name not in sys.modules:
            stub = types.ModuleType(name)
            stub.__path__ = []
            sys.modules[name] = stub
    state = _load('nexoclom.particle_tracking.state', 'nexoclom/particle_tracking/state.py')
    rk5 = _load('nexoclom.particle_tracking.rk5', 'nexoclom/particle_tracking/rk5.py')
    hist = _load('nexoclom.math.histogram', 'nexoclom/math/histogram.py')
    rot = _load('nexoclom.math.rotation_matrix', 'nexoclom/math/rotation_matrix.py')
    return rk5, state, hist, rot


class _Lifetime(float):
    """options.lifetime as the reference reads it: ``> 0`` and ``.value`` (state.py:44-46)."""
    @property
    def value(self):
        return float(self)


def duck_output(forces, step_size):
    """The attributes of ``output`` that rk5.py / state.py touch, filled from an oracle Forces."""
    ns = types.SimpleNamespace
    return ns(inputs=ns(forces=ns(gravity=forces.gravity, radpres=forces.radpres),
                        options=ns(lifetime=_Lifetime(forces.lifetime), step_size=step_size)),
              GM=forces.GM, vrplanet=forces.vrplanet,
              radpres=ns(velocity=forces.v_tab, accel=forces.a_tab),
              loss_info=ns(photo=forces.photo))
