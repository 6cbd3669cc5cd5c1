"""Build a variant of libnexoclom_hip.so with extra compiler flags (experiments only):
    python tools/build_variant.py build/exp/libT2.so -DNXC_REFILL_MIN=2"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nexoclom_amd import build as B
out, extra = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
cmd = [B.hipcc()] + B.FLAGS + extra + [B.SRC, '-o', out, '-ldl']
res = subprocess.run(cmd, capture_output=True, text=True)
sys.stderr.write(res.stderr[-3000:])
sys.exit(res.returncode)
