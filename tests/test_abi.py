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
    assert lib.nxc_abi_version() == hip_api.ABI_VERSION == 3


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
    # tools/ are measurement helpers of the product: no oracle there either, directly or through
    # the test helpers (scripts that need them live in tests/tools/)
    for fn in os.listdir(os.path.join(ROOT, 'tools')):
        if fn.endswith('.py'):
            src = open(os.path.join(ROOT, 'tools', fn)).read()
            assert not re.search(r'^\s*(from|import)\s+(oracle|tests)\b', src, flags=re.M), fn
    # bench.py touches the oracle only inside its cpu_baseline leg; __graft_entry__ only in smoke()
    import ast
    for fn, allowed in (('bench.py', {'cpu_baseline'}), ('__graft_entry__.py', {'smoke'})):
        tree = ast.parse(open(os.path.join(ROOT, fn)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef):
                uses = any(isinstance(n, (ast.Import, ast.ImportFrom)) and
                           ('oracle' in (getattr(n, 'module', '') or '') or
                            any('oracle' in a.name for a in n.names))
                           for n in ast.walk(node))
                assert not uses or node.name in allowed, (fn, node.name)
        for node in tree.body:          # and never at module level
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                mod = getattr(node, 'module', '') or ''
                assert 'oracle' not in mod and all('oracle' not in a.name for a in node.names), fn


def test_ctypes_structs_match_the_compiled_header(tmp_path):
    """Every struct of include/nexoclom_hip.h as gcc lays it out (sizeof and the offset of each
    field) against the ctypes mirror in hip_api.py: a field added on one side only shows up here."""
    import subprocess
    import sys
    structs = ['nxc_forces', 'nxc_image_desc', 'nxc_counters', 'nxc_bounce_desc',
               'nxc_bodies_desc', 'nxc_source_desc', 'nxc_los_desc']
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){']
    for name in structs:
        ct = getattr(hip_api, name)
        lines.append(f'printf("{name} %zu", sizeof({name}));')
        for field, *_ in ct._fields_:
            lines.append(f'printf(" %zu", offsetof({name}, {field}));')
        lines.append('printf("\\n");')
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    assert len(out) == len(structs)
    for row in out:
        name, size, *offsets = row.split()
        ct = getattr(hip_api, name)
        assert int(size) == C.sizeof(ct), name
        assert [int(o) for o in offsets] == [getattr(ct, f).offset for f, *_ in ct._fields_], name
