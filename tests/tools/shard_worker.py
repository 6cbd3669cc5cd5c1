"""One rank of a multi-process sharded_image run (started by tests/test_gpu_multiproc.py with RANK /
WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment, like torch.distributed.run would): the
product's own control plane, shard arithmetic, device sampler or host sampler, and merge.  Prints
one JSON line with a digest of the global packet-count image and the image's checksum."""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nexoclom_amd import Input                                  # noqa: E402
from nexoclom_amd.distributed import ControlPlane, sharded_image  # noqa: E402

npackets, seed, sampler, reduce = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
inputs.options.endtime = type(inputs.options.endtime)(9000., 's')
cp = ControlPlane()
with contextlib.redirect_stdout(io.StringIO()):
    img = sharded_image(inputs, {'quantity': 'radiance', 'dims': '96,96'}, npackets, seed, cp=cp,
                        device=0, sampler=sampler, packs_per_it=3000, reduce=reduce)
digest = hashlib.sha1(np.ascontiguousarray(img.packet_image).tobytes()).hexdigest()
print(json.dumps({'rank': cp.rank, 'world': cp.world, 'counts_sha1': digest,
                  'binned': float(img.packet_image.sum()), 'image_sum': float(img.image.sum()),
                  'totalsource': float(img.totalsource), 'npackets': int(img.npackets)}), flush=True)
cp.close()
