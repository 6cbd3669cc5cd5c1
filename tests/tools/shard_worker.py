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
flow = sys.argv[5] if len(sys.argv) > 5 else 'streaming'
inputs = Input(os.path.join(ROOT, 'nexoclom_amd', 'inputfiles', 'Na.mercury.bench.input'))
inputs.options.endtime = type(inputs.options.endtime)(9000., 's')
cp = ControlPlane()
extra = {}
with contextlib.redirect_stdout(io.StringIO()):
    if flow == 'streaming':
        img = sharded_image(inputs, {'quantity': 'radiance', 'dims': '96,96'}, npackets, seed,
                            cp=cp, device=0, sampler=sampler, packs_per_it=3000, reduce=reduce)
    else:
        # the reference's two-stage flow shared by the ranks: catalogued Outputs whose rows stay in
        # this rank's HBM, then the image and the lines of sight summed over the ranks
        from nexoclom_amd import LOSResult, SpacecraftData, hip_api
        ctx = hip_api.Context(0)
        kw = dict(sampler='device', generator='pcg64') if sampler == 'pcg64' else dict(sampler=sampler)
        inputs.run(npackets, packs_per_it=3000, seed=seed, context=ctx, cp=cp, **kw)
        img = inputs.produce_image({'quantity': 'radiance', 'dims': '96,96'}, context=ctx, cp=cp,
                                   reduce=reduce)
        th = np.linspace(0, 2*np.pi, 64, endpoint=False)
        pos = np.stack([np.cos(th), 1.2*np.sin(th) - 0.4, 1.6*np.sin(th)], 1)*1.5
        look = -pos + 0.5*np.random.default_rng(1).normal(size=pos.shape)
        look /= np.linalg.norm(look, axis=1)[:, None]
        sc = SpacecraftData(*pos.T, *look.T)
        los = LOSResult(sc, inputs, dphi=np.radians(3.0), context=ctx)
        los.simulate_data_from_inputs(sc, cp=cp, reduce=reduce)
        img.npackets = los.npackets
        extra = {'outputs_here': len(inputs._catalogue),
                 'los_counts_sha1': hashlib.sha1(los.npackets_los.values.tobytes()).hexdigest(),
                 'los_pairs': int(los.npackets_los.sum()),
                 'los_radiance_sum': float(los.radiance.sum())}
digest = hashlib.sha1(np.ascontiguousarray(img.packet_image).tobytes()).hexdigest()
print(json.dumps({'rank': cp.rank, 'world': cp.world, 'counts_sha1': digest,
                  'binned': float(img.packet_image.sum()), 'image_sum': float(img.image.sum()),
                  'totalsource': float(img.totalsource), 'npackets': int(img.npackets), **extra}),
      flush=True)
cp.close()
