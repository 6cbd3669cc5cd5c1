"""Input: the inputfile-driven front door of a model run.

Drop-in for the reference's initial_state/Input.py:27-272 on the hot path: ``Input(infile)`` turns
the ``section.key = value`` lines of an inputfile into seven section objects (``geometry``,
``surfaceinteraction``, ``forces``, ``spatialdist``, ``speeddist``, ``angulardist``,
``options``); ``run()`` integrates packets chunk by chunk; ``search()`` tells what has been run;
``produce_image()`` builds a ModelImage.  The file grammar is the reference's (Input.py:60-80): a
``;`` -- or, on lines without one, a ``#`` -- starts a comment; a line counts only if it holds
exactly one ``=`` and the left side exactly one ``.``; section and key are case-folded.

The reference catalogues runs in PostgreSQL; here the catalogue is the list of Output objects kept
on the Input (plus optional .npz files under ``savepath``).
"""
import math
import os
import time

from . import input_classes as spec

HOST_SAMPLER_THREADS = 8        # Outputs drawn side by side, ahead of the device (16: no faster)

# section name in the file -> (attribute on the Input, class that interprets it)
SECTIONS = (('geometry', spec.Geometry), ('surfaceinteraction', spec.SurfaceInteraction),
            ('forces', spec.Forces), ('spatialdist', spec.SpatialDist),
            ('speeddist', spec.SpeedDist), ('angulardist', spec.AngularDist),
            ('options', spec.Options))


def read_inputfile(path):
    """{section: {key: value text}} of an inputfile; later lines override earlier ones."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    table = {}
    with open(path, 'r') as handle:
        for raw in handle:
            mark = ';' if ';' in raw else '#'
            text = raw.split(mark, 1)[0]
            left, eq, right = text.partition('=')
            if not eq or '=' in right:
                continue
            section, dot, key = left.partition('.')
            if not dot or '.' in key:
                continue
            table.setdefault(section.casefold().strip(), {})[key.casefold().strip()] = right.strip()
    return table


class Input:
    def __init__(self, infile, savepath=None):
        self._inputfile, self.savepath = infile, savepath
        self._catalogue = []          # Output objects run with these inputs
        self._fused = []              # fused integrate+image results (ModelImage streaming mode)
        table = read_inputfile(infile)
        for name, cls in SECTIONS:
            setattr(self, name, cls(table.get(name, {})))

    def _sections(self):
        return [getattr(self, name) for name, _ in SECTIONS]

    def __copy__(self):
        # a copy shares the catalogue (ModelResult copies its inputs) but not the writer thread
        new = type(self).__new__(type(self))
        new.__dict__.update({k: v for k, v in self.__dict__.items()
                             if k not in ('_writer', '_pending')})
        return new

    def __eq__(self, other):
        return isinstance(other, type(self)) and self._sections() == other._sections()

    __hash__ = None

    def __str__(self):
        return '\n'.join(str(section) for section in self._sections())

    __repr__ = __str__

    # ---- catalogue ------------------------------------------------------------------------
    def search(self):
        """(ids, outputs-or-filenames, npackets, totalsource) of the runs made with these inputs
        (Input.py:121-172, without the database)."""
        runs = self._catalogue
        if not runs:
            return ([], [], 0, 0)
        return ([run.idnum for run in runs], [run.filename or run for run in runs],
                int(sum(run.npackets for run in runs)),
                float(sum(run.totalsource for run in runs)))

    def delete_files(self, filename=None):
        """Forget (and remove from disk) every catalogued run, or only the one saved as
        ``filename`` (Input.py:274-...)."""
        self.wait()
        doomed = [run for run in self._catalogue if filename is None or run.filename == filename]
        for run in doomed:
            if run.filename and os.path.exists(run.filename):
                os.remove(run.filename)
        self._catalogue = [run for run in self._catalogue if run not in doomed]
        self._fused = []

    def chunk_size(self, packs_per_it=None):
        """Packets per Output when the caller does not say (Input.py:216-227): a million for the
        adaptive driver; for the constant-step driver as many as keep the (N, 8, nsteps) history
        near 1024^3 doubles / 8."""
        if packs_per_it is not None:
            return int(packs_per_it)
        step = self.options.step_size
        if step == 0:
            return 1_000_000
        records = int(math.ceil(self.options.endtime.value/step) + 1)
        return int(math.ceil(1024**3/records/8))

    def run(self, npackets, packs_per_it=None, overwrite=False, compress=True,
            distribute=False, seed=None, *, device=0, keep_trajectory=True, context=None,
            sampler='numpy', batch=True, generator='philox', cp=None):
        """Integrate until the catalogue holds ``npackets`` packets (Input.py:175-268).

        Every pass plans ``ceil(todo / size)`` Outputs of ``size = min(todo, chunk_size)``
        packets -- the reference's arithmetic, so the last Output of a pass may overshoot.  With a
        ``seed`` the k-th Output of this call is drawn from ``seed + k``: Output 0 is the
        reference's stream; the reference itself passes the same seed to every Output
        (Input.py:246), which repeats identical packets.  ``sampler='device'`` draws the initial
        states on the GPU instead (counter-based: the k-th Output continues the index space of
        the ones before it under the one ``seed``; with ``generator='pcg64'`` the device follows
        the host sampler's seeded streams instead -- Output k from ``seed + k``, the same packets
        as sampler='numpy' to libm rounding, without the host drawing or uploading them).

        The Outputs of a pass are independent, so they are INTEGRATED TOGETHER (``batch``): one
        upload, one launch of each kernel over all their packets (Output.integrate_batch), each
        Output then owning its slice of the rows, which stay in HBM for produce_image /
        LOSResult.  With a ``savepath`` the files are written by a worker thread beside the next
        launch (``wait()`` joins it).

        ``cp`` (a distributed.ControlPlane of several ranks, one process per GPU): the run is
        SHARED.  The plan -- how many Outputs, of which size, from which seed or index range -- is
        the single-process plan; rank r makes the Outputs ``shard_range(passes, r, world)`` of
        every pass and catalogues only those, so Output k is the same Output whatever the number
        of ranks, and the union of the ranks' catalogues is the single-process catalogue.  The
        packet count that ends the run is summed over the ranks.  ``produce_image(..., cp=cp)`` and
        ``LOSResult.simulate_data_from_inputs(..., cp=cp)`` then sum over the ranks what the
        reference sums over the files (ModelImage.py:96-98, LOSResult.py:264-266)."""
        from .Output import Output
        started = time.time()
        rank, world = (cp.rank, cp.world) if cp is not None else (0, 1)
        if world > 1 and seed is None and sampler != 'device':
            # the host sampler seeds Output k with seed + k: without a seed the ranks would not
            # make the Outputs of one run
            seed = int.from_bytes(cp.bcast_bytes(os.urandom(4), 4), 'little')

        def report():
            local = self._report()
            return local if world == 1 else int(round(cp.reduce(float(local), 'SUM')))
        if distribute in (True, 'delay', 'delayed'):
            assert False, 'Dont do this'         # the reference's dask path is disabled too
        if overwrite:
            self.delete_files()
        have = report()
        want = int(npackets)
        made = 0
        drawn = have                             # device sampler: next free global packet index
        # re-emission draws are keyed by (the Output's seed, the packet's number): Outputs that
        # each have their own seed (host-sampled, or following the host streams with 'pcg64')
        # cannot share a launch when packets are re-emitted
        spec = self.surfaceinteraction
        sticks = spec.sticktype == 'constant' and spec.stickcoef == 1.
        one_key = sampler == 'device' and generator != 'pcg64'
        together = bool(batch) and keep_trajectory and (compress or self.options.step_size == 0) \
            and (sticks or one_key)
        if together and context is None:
            from . import hip_api
            context = hip_api.Context(device)
        if seed is None and sampler == 'device':
            from .Output import fresh_key
            seed = fresh_key()                   # one key for the whole (unseeded) run
            if world > 1:
                seed = int.from_bytes(cp.bcast_bytes(seed.to_bytes(8, 'little'), 8), 'little')
        while have < want:
            todo = want - have
            size = min(todo, self.chunk_size(packs_per_it))
            passes = -(-todo // size)
            print('Running Model')
            print(f'Will complete {passes} iterations of {size} packets.')
            number = 0
            stop = passes
            if world > 1:
                # this rank's Outputs of the pass; the counters move as if the others were made here
                from .distributed import shard_range
                number, stop = shard_range(passes, rank, world)
                made += number
                drawn += size*number
                print(f'Rank {rank} of {world}: iterations {number + 1} to {stop}.')
            pipeline = None                      # host-drawn Outputs on their way, in order
            pool = None
            launched = False
            submitted = number                   # next Output of the pass to hand to the drawers

            def draw_one(k):
                # Output k of the pass: an independent draw from seed + (its number in the run)
                s_ = None if seed is None else seed + made + (k - number)
                def job():
                    out = Output(self, size, compress=compress, device=device,
                                 keep_trajectory=keep_trajectory, context=context,
                                 integrate=False, save=False, seed=s_)
                    return out.prepare_for_launch()
                return pool.submit(job)

            try:
                while number < stop:
                    tick = time.time()
                    # as many Outputs per launch as HBM takes (rows: nsteps records per packet at
                    # most, far fewer in practice; the row store spills its oldest runs to the host)
                    limit = self._group_limit(size, context) if together else 1
                    group = min(stop - number, limit)
                    outs = []
                    if together and sampler == 'numpy' and (group > 1 or pipeline):
                        # The Outputs are drawn on a few threads (NumPy releases the GIL in its
                        # loops) a window ahead of the device, and a launch takes what is drawn by
                        # the time the device is free: the first launch starts after the first
                        # draws instead of after a whole group's, later ones find Outputs waiting.
                        if pool is None:
                            from collections import deque
                            from concurrent.futures import ThreadPoolExecutor
                            pool = ThreadPoolExecutor(max_workers=HOST_SAMPLER_THREADS)
                            pipeline = deque()
                        window = max(2*HOST_SAMPLER_THREADS, 2*limit)
                        while submitted < stop and len(pipeline) < window:
                            pipeline.append(draw_one(submitted))
                            submitted += 1
                        # A launch has a fixed cost of several milliseconds (two passes, row
                        # offsets, a store): the first takes a handful of Outputs so that the
                        # device starts early, the later ones wait for a third of what HBM takes
                        # (or the rest of the pass) and then add whatever else is drawn by then.
                        floor = max(1, min(limit, HOST_SAMPLER_THREADS)) if not launched else \
                            max(1, min(limit//3, HOST_SAMPLER_THREADS*2))
                        launched = True
                        outs = []
                        while pipeline and len(outs) < floor:
                            outs.append(pipeline.popleft().result())
                            while submitted < stop and len(pipeline) < window:
                                pipeline.append(draw_one(submitted))
                                submitted += 1
                        while pipeline and pipeline[0].done() and len(outs) < limit:
                            outs.append(pipeline.popleft().result())
                        number += len(outs)
                        drawn += size*len(outs)
                        made += len(outs)
                        group = 0
                    for g in range(group):
                        number += 1
                        print(f'Starting iteration #{number} of {passes}')
                        if sampler == 'device' and generator == 'pcg64':
                            draw = dict(seed=seed + made, sampler='device', generator='pcg64',
                                        first_index=drawn, presampled=together,
                                        materialize_x0=not together)
                        elif sampler == 'device':
                            draw = dict(seed=seed, sampler='device', first_index=drawn,
                                        presampled=together, materialize_x0=not together)
                        else:
                            draw = dict(seed=None if seed is None else seed + made)
                        out = Output(self, size, compress=compress, device=device,
                                     keep_trajectory=keep_trajectory, context=context,
                                     integrate=not together, save=not together, **draw)
                        context = out.context()      # every Output of the run shares one device
                        outs.append(out)
                        drawn += size
                        made += 1
                    if together and outs:
                        first = None if seed is None else seed + made - len(outs)
                        self._launch_group(outs, context, sampler, generator, seed, size,
                                           first_seed=first)
                    print(f'Completed iteration #{number} in {time.time() - tick} seconds.')
            finally:
                if pool is not None:
                    pool.shutdown(cancel_futures=True)
            made += passes - stop                # the Outputs of the ranks after this one
            drawn += size*(passes - stop)
            have = report()
        self.wait()                              # files of this run are on disk when it returns
        print(f'Model run completed in {time.time() - started:.2f} sec.')

    def _launch_group(self, outs, context, sampler, generator, seed, size, first_seed):
        """Draw (device samplers) and integrate the Outputs ``outs`` in one launch.  If their rows
        turn out not to fit in HBM beside what is resident -- the estimate of _group_limit assumes
        typical lifetimes -- the group is split in halves and tried again."""
        from . import hip_api
        from .Output import Output
        try:
            if sampler == 'device' and generator == 'pcg64':
                # every Output follows its own seeded stream: one piece of the resident set each
                src, whole = outs[0].source_desc(), size*len(outs)
                for g, out in enumerate(outs):
                    out._adopt_x0(context.sample_packets(
                        size, first_seed + g, outs[0]._first_index, download=True,
                        pcg64=(size, 0), piece=(g*size, whole), **src))
            elif sampler == 'device':
                lead = outs[0]
                soa = context.sample_packets(size*len(outs), seed, lead._first_index,
                                             download=True, **lead.source_desc())
                for g, out in enumerate(outs):
                    out._adopt_x0(soa[:, g*size:(g + 1)*size])
            Output.integrate_batch(outs, context)
        except hip_api.HipError as exc:
            if exc.code != hip_api.NXC_ERR_NOMEM or len(outs) < 2:
                raise
            half = len(outs)//2
            print(f'{len(outs)} Outputs in one launch need more HBM than is free: two launches')
            self._launch_group(outs[:half], context, sampler, generator, seed, size, first_seed)
            self._launch_group(outs[half:], context, sampler, generator, seed, size,
                               None if first_seed is None else first_seed + half)

    def _group_limit(self, size, context):
        """How many Outputs of ``size`` packets one launch may take: the packets' states and a
        generous estimate of their rows (a quarter of the dense history) within a third of the
        device's free memory; at least one."""
        step = self.options.step_size
        per_packet = 8*8*3
        if step != 0:
            records = int(math.ceil(self.options.endtime.value/step) + 1)
            per_packet += records*80//4
        free, _ = context.mem_info()
        return max(1, int(free//3//(per_packet*size)))

    # ---- files in the background ------------------------------------------------------------
    def _write_later(self, job, *args):
        """Run ``job(*args)`` on the writer thread (one at a time, in order)."""
        if getattr(self, '_writer', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._writer, self._pending = ThreadPoolExecutor(max_workers=1), []
        self._pending.append(self._writer.submit(job, *args))

    def wait(self):
        """Block until every file of the catalogue is on disk; re-raises a writer's error."""
        pending, self._pending = getattr(self, '_pending', []), []
        for job in pending:
            job.result()

    def _report(self):
        _, files, packets, _ = self.search()
        print(f'Found {len(files)} files with {packets} packets.')
        return packets

    def produce_image(self, format_, overwrite=False, distribute=None, *, cp=None, reduce='rccl',
                      **kwargs):
        """Input.py:270-272.  ``cp``: the control plane of a shared run (``run(..., cp=cp)``):
        every rank bins the Outputs of its own catalogue and the image pairs and source totals
        are summed over the ranks (reduce='rccl': one ncclAllReduce; 'host': over the control
        plane, for tests) -- every rank gets the image of the whole run."""
        from .ModelImage import ModelImage
        if cp is None or cp.world == 1:
            return ModelImage(self, format_, overwrite=overwrite, distribute=distribute, **kwargs)
        from .distributed import guarded, merge_catalogue
        image = ModelImage(self, format_, overwrite=overwrite, distribute=distribute,
                           finalize=False, **kwargs)
        with guarded(cp, image.context()):
            merge_catalogue(image, cp, image.context(), reduce)
        image.finalize()
        return image
