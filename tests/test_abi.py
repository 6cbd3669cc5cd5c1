"""The C-ABI library loads without a GPU and exports every symbol include/nexoclom_hip.h
declares; the ctypes structures match the header's layout; GPU-less calls fail loudly."""
import ctypes as C
import os
import re

import pytest

from nexoclom_amd import hip_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nexoclom_hip.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nxc_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = hip_api.load_library()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f'{name} declared in nexoclom_hip.h but not exported'
    assert set(names) == set(hip_api.EXPORTS)
    assert lib.nxc_abi_version() == 1


def test_struct_layouts_match_header():
    # nxc_forces: 4 doubles, 4 int32, int64, 2 pointers
    assert C.sizeof(hip_api.nxc_forces) == 4*8 + 4*4 + 8 + 2*8
    # nxc_image_desc: 9+2 doubles, 4 int32, 2 int64, 2 ptr, 4 int64, 4+4 ptr
    assert C.sizeof(hip_api.nxc_image_desc) == 11*8 + 4*4 + 2*8 + 2*8 + 4*8 + 8*8
    assert C.sizeof(hip_api.nxc_counters) == 8*8
    assert hip_api.nxc_forces.n_tab.offset == 48
    assert hip_api.nxc_image_desc.nx.offset == 104


def test_no_gpu_means_loud_failure_not_fallback():
    if hip_api.device_count() > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(hip_api.HipError):
        hip_api.Context(0)
    lib = hip_api.load_library()
    assert lib.nxc_set_forces(None, None) != 0
    assert b'null' in lib.nxc_last_error_string()


def test_package_does_not_import_the_oracle():
    """Product code must never route through oracle/ (or any CPU integrator)."""
    pkg = os.path.join(ROOT, 'nexoclom_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), fn
                assert 'oracle_math.h' not in src and 'liboracle' not in src, fn
