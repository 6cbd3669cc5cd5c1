"""Static instruction census of one kernel of libnexoclom_hip.so, by source section.

    python tools/isa_census.py [kernel-substring] [--blocks]

Compiles nexoclom_amd/csrc/nxc_api.hip to gfx950 assembly with line tables (the product flags +
-gline-tables-only -S), walks the chosen kernel (default: the bench's k_const_fused<IMAGE, no
BOUNCE, FULL, no NBODY>) and attributes every instruction to the source function its .loc line
falls in (innermost inlined callee).  Blocks that only the rare paths reach -- the compiler's
full-range division / sqrt sequences (v_div_scale, v_div_fmas, v_div_fixup), the table and edge
walks -- are reported separately as "cold".  Output: per-section counts of VALU fp64 arithmetic,
transcendental (v_rcp/v_rsq_f64), conversions, compares/selects/integer VALU, LDS, VMEM, SALU.
The census is static (one pass over the loop body = one wave trip when every guarded region
executes); profiles/<tag>_pmc.json holds the dynamic SQ_INSTS_VALU it is compared with.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, 'nexoclom_amd', 'csrc')

# (file, first line, last line) -> section; filled from the sources so that edits do not shift it
FUNCS = {
    'nxc_device.hpp': ['lut_interp', 'lut_cell', 'lut_probe_cell', 'lut_probe_rows', 'lut_probe', 'lut_finish', 'lds_f64', 'lds_u16', 'lds_f64x2', 'sunlit',
                       'state_eval', 'rk5_step', 'apply_fate', 'bin_index', 'image_locate',
                       'image_weight', 'image_sample', 'image_add_pairs', 'push', 'pop', 'waiting',
                       'image_regs', 'wave_uniform', 'nxc_div_const', 'half_swap',
                       'f32_round_trip', 'lut_view', 'uniform_view', 'bodies_eval',
                       'moon_position', 'bounce_packet'],
    'nxc_math.hpp': ['nxc_sqrt_mid', 'nxc_sqrt', 'nxc_recip_seed', 'nxc_div_seeded', 'nxc_div_mid',
                     'nxc_div', 'nxc_cube', 'nxc_exp', 'nxc_log', 'nxc_mid_range'],
    'nxc_kernels.hpp': ['stage_tables', 'stage_tables_and_args', 'refill', 'k_const_fused',
                        'flush_counter', 'wave_sum', 'wave_bcast0', 'k_var', 'k_image'],
}


def function_spans():
    spans = {}
    for fn, names in FUNCS.items():
        lines = open(os.path.join(SRC, fn)).read().split('\n')
        starts = []
        for i, l in enumerate(lines, 1):
            m = re.match(r'^\s*(?:template\s*<[^>]*>\s*)?(?:NXC_DEV|__global__|static|inline)?\s*'
                         r'[\w:<>\*&\s]*?\b(\w+)\s*\([^;]*$', l)
            if m and m.group(1) in names and not l.strip().startswith(('return', 'if', '//')):
                starts.append((i, m.group(1)))
        # a function runs until the next top-level closing brace
        for i, name in starts:
            j = i
            while j < len(lines) and lines[j].rstrip() != '}' and lines[j].rstrip() != '};':
                j += 1
            spans.setdefault(fn, []).append((i, j + 1, name))
    return spans


def classify(op):
    if op in ('v_rcp_f64_e32', 'v_rsq_f64_e32', 'v_rcp_f64_e64', 'v_rsq_f64_e64'):
        return 'transc'
    if op.startswith(('v_cvt_',)):
        return 'cvt'
    if re.match(r'v_(mul|add|fma|fmac|max|min|ldexp|div_scale|div_fmas|div_fixup|frexp|trunc|floor|rndne|fract)_f64', op):
        return 'fp64'
    if op.startswith('v_cmp'):
        return 'cmp'
    if op.startswith('v_cndmask'):
        return 'select'
    if op.startswith('v_'):
        return 'valu32'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    want = next((a for a in sys.argv[1:] if not a.startswith('--')),
                'k_const_fusedILi2ELb0ELb1ELb0E')
    from nexoclom_amd import build as B
    flags = [f for f in B.FLAGS if f not in ('-shared', '-fPIC')]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'api.s')
        subprocess.run([B.hipcc()] + flags + ['-gline-tables-only', '-S', '--cuda-device-only',
                                              B.SRC, '-o', out], check=True,
                       capture_output=True)
        text = open(out).read().split('\n')
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"\s+md5', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
    start = next(i for i, l in enumerate(text) if l.startswith('_Z') and want in l and l.rstrip().endswith(tuple(':')) or (l.startswith('_Z') and want in l and ':' in l))
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith('.Lfunc_end'))
    spans = function_spans()

    def section(fid, line):
        fn = files.get(fid, '?')
        for a, b, name in spans.get(fn, []):
            if a <= line <= b:
                return name
        return fn + ':?'

    # pass 1: split into basic blocks, find the cold ones
    blocks, cur = [], {'label': 'entry', 'ins': [], 'depth': 0}
    loc = (0, 0)
    for l in text[start+1:end]:
        s = l.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        if re.match(r'^\.LBB\d+_\d+:', s):
            blocks.append(cur)
            m = re.search(r'Depth=(\d+)', s)
            cur = {'label': s.split(':')[0], 'ins': [], 'depth': int(m.group(1)) if m else 0}
            continue
        if not s or s.startswith((';', '.')):
            continue
        op = s.split()[0]
        if op.endswith(':'):
            continue
        cur['ins'].append((op, section(*loc), s))
        if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')):
            blocks.append(cur)                 # a branch ends the block even without a label
            cur = {'label': cur['label'] + "'", 'ins': [], 'depth': cur['depth']}
    blocks.append(cur)
    cold_ops = ('v_div_scale_f64', 'v_div_fmas_f64', 'v_div_fixup_f64')
    # depth 1 = the persistent loop's own blocks; depth >= 2 = loops inside it (queue refill, table
    # and edge walks: rare); depth 0 = prologue / epilogue (once per wave)
    for b in blocks:
        b['cold'] = any(op in cold_ops for op, _, _ in b['ins'])
    table = collections.defaultdict(collections.Counter)
    for b in blocks:
        for op, sec, _ in b['ins']:
            tag = ('cold: ' if b['cold'] else 'once: ' if b['depth'] == 0
                   else 'inner: ' if b['depth'] >= 2 else '')
            table[tag + sec][classify(op)] += 1
    cols = ['fp64', 'transc', 'cvt', 'cmp', 'select', 'valu32', 'lds', 'vmem', 'salu']
    print(f'kernel {text[start].split(":")[0]}')
    print(f'{"section":34s}' + ''.join(f'{c:>8s}' for c in cols) + f'{"VALU":>8s}')
    tot = collections.Counter()
    def rank(k):
        return (k.split(': ')[0] if ': ' in k else '', -sum(table[k].values()))
    for key in sorted(table, key=rank):
        row = table[key]
        valu = sum(row[c] for c in ('fp64', 'transc', 'cvt', 'cmp', 'select', 'valu32'))
        print(f'{key:34s}' + ''.join(f'{row[c]:8d}' for c in cols) + f'{valu:8d}')
        if ': ' not in key:
            tot.update(row)
    valu = sum(tot[c] for c in ('fp64', 'transc', 'cvt', 'cmp', 'select', 'valu32'))
    print(f'{"loop body total":34s}' + ''.join(f'{tot[c]:8d}' for c in cols) + f'{valu:8d}')
    if '--blocks' in sys.argv:
        for b in blocks:
            secs = collections.Counter(sec for _, sec, _ in b['ins'])
            print(b['label'], 'cold' if b['cold'] else '', len(b['ins']), dict(secs))


if __name__ == '__main__':
    main()
