"""Condense gpurun_out/prof_<tag>/ (rocprofv3 CSVs from tools/profile.sh) into profiles/<tag>_*.

Writes profiles/<tag>_kernel_stats.csv (the --stats summary), profiles/<tag>_pmc.json (per-launch
counter averages per kernel) and updates profiles/traffic.json (HBM bytes per launch of the
dominant kernel, FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM": gfx950 reports half the bytes
of a wide coalesced read; WRITE_SIZE taken as is; both are in KiB units -> x1024)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'   # --no-traffic: leave profiles/traffic.json alone
src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)



def newest(pattern):
    """One file per pass directory: the most recent run's (gpurun merges every call's files into
    the same local directory, so a tag profiled twice holds both runs)."""
    by_pass = {}
    for f in glob.glob(pattern, recursive=True):
        key = os.path.relpath(f, src).split(os.sep)[0]
        if key not in by_pass or os.path.getmtime(f) > os.path.getmtime(by_pass[key]):
            by_pass[key] = f
    return list(by_pass.values())


for f in newest(os.path.join(src, 'trace', '**', '*_kernel_stats.csv')):
    shutil.copy(f, os.path.join(dst, f'{tag}_kernel_stats.csv'))
bl = os.path.join(src, 'bench_line.json')
if os.path.exists(bl):
    shutil.copy(bl, os.path.join(dst, f'{tag}_bench_line.json'))

pmc = defaultdict(lambda: defaultdict(list))
for f in newest(os.path.join(src, '*', '**', '*_counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].split('(')[0]
        pmc[name][row['Counter_Name']].append(float(row['Counter_Value']))
summary = {k: {c: {'mean_per_launch': sum(v)/len(v), 'launches': len(v)} for c, v in d.items()}
           for k, d in pmc.items()}
json.dump(summary, open(os.path.join(dst, f'{tag}_pmc.json'), 'w'), indent=1, sort_keys=True)

dom = [] if '--no-traffic' in sys.argv else [k for k in summary if 'k_const_fused' in k]
traffic = {}
if dom:
    s = summary[dom[0]]
    if 'FETCH_SIZE' in s and 'WRITE_SIZE' in s:
        fetch = s['FETCH_SIZE']['mean_per_launch']*1024*2
        write = s['WRITE_SIZE']['mean_per_launch']*1024
        traffic = {'k_const_fused_bytes_per_launch': fetch + write, 'fetch_bytes_corrected': fetch,
                   'write_bytes': write, 'tag': tag,
                   'note': 'FETCH_SIZE x2 (gfx950 half-count of wide reads), KiB units x1024'}
        if 'SQ_INSTS_VALU' in s:
            traffic['k_const_fused_valu_wave_insts_per_launch'] = s['SQ_INSTS_VALU']['mean_per_launch']
        if 'TCC_EA0_ATOMIC_sum' in s:
            traffic['k_const_fused_atomic_requests_per_launch'] = \
                s['TCC_EA0_ATOMIC_sum']['mean_per_launch']
        if 'SQ_ACTIVE_INST_VALU' in s and 'GRBM_GUI_ACTIVE' in s:
            # quad-cycles in which a VALU instruction was issuing, summed over the 1024 SIMDs,
            # against the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
            traffic['k_const_fused_valu_busy_frac'] = \
                s['SQ_ACTIVE_INST_VALU']['mean_per_launch']*4/1024 / \
                (s['GRBM_GUI_ACTIVE']['mean_per_launch']/8)
        if os.path.exists(bl):
            try:
                traffic['particle_steps_per_launch'] = \
                    json.loads(open(bl).read().strip().splitlines()[-1])['particle_steps_per_pass']
            except (ValueError, KeyError, IndexError):
                pass
        tfile = os.path.join(dst, 'traffic.json')
        try:                                    # entries of other kernels stay
            keep = {k: v for k, v in json.load(open(tfile)).items() if k.startswith('image_tiles')}
        except (OSError, ValueError):
            keep = {}
        json.dump(dict(keep, **traffic), open(tfile, 'w'), indent=1)

# the tiled image of stored samples (tools/profile_kernels.sh): bytes really moved per sample
tiles = [k for k in summary if 'k_image_bin<float, true>' in k or 'k_image_tiles<true>' in k]
units = None
if os.path.exists(bl):
    for ln in open(bl):
        if 'k_image_bin + k_image_tiles[radiance]' in ln:
            units = json.loads(ln)['units']
if len(tiles) == 2 and units and all('FETCH_SIZE' in summary[k] and 'WRITE_SIZE' in summary[k]
                                     for k in tiles):
    total = sum(summary[k]['FETCH_SIZE']['mean_per_launch']*2048 +
                summary[k]['WRITE_SIZE']['mean_per_launch']*1024 for k in tiles)
    tfile = os.path.join(dst, 'traffic.json')
    try:
        cur = json.load(open(tfile))
    except (OSError, ValueError):
        cur = {}
    cur['image_tiles_bytes_per_sample'] = total/units
    cur['image_tiles_source'] = (f'profiles/{tag}_pmc.json: (FETCH_SIZE x 2 + WRITE_SIZE) x 1024 of '
                                 f'k_image_bin<float, true> + k_image_tiles<true>, {units} float32 '
                                 f'rows per launch pair (radiance and column launches averaged)')
    json.dump(cur, open(tfile, 'w'), indent=1)
print(json.dumps({'kernels': list(summary), 'traffic': traffic}, indent=1))
for k, d in summary.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:28s} {v["mean_per_launch"]:.6g}')
