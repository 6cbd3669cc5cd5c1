"""Input: the inputfile-driven front door of a model run.

Mirrors the reference's initial_state/Input.py:27-272: ``Input(infile)`` parses
``category.parameter = value`` lines (';' / '#' comments, case-folded keys, exactly one '=' and
one '.', Input.py:60-80) into the seven spec objects; ``run()`` integrates packets in chunks;
``search()`` reports what has been run; ``produce_image()`` builds a ModelImage.  The PostgreSQL
catalogue of the reference is replaced by an in-memory list of Output objects on the Input (plus
optional .npz files under ``savepath``).
"""
import os
import time

import numpy as np

from .input_classes import (AngularDist, Forces, Geometry, Options, SpatialDist, SpeedDist,
                            SurfaceInteraction)


class Input:
    def __init__(self, infile, savepath=None):
        self._inputfile = infile
        self.savepath = savepath
        self._catalogue = []          # Output objects run with these inputs
        self._fused = []              # fused integrate+image results (ModelImage streaming mode)
        params = []
        if os.path.isfile(infile):
            for line in open(infile, 'r'):
                if ';' in line:
                    line = line[:line.find(';')]
                elif '#' in line:
                    line = line[:line.find('#')]
                if line.count('=') == 1:
                    param_, val_ = line.split('=')
                    if param_.count('.') == 1:
                        sec_, par_ = param_.split('.')
                        params.append((sec_.casefold().strip(), par_.casefold().strip(),
                                       val_.strip()))
        else:
            raise FileNotFoundError(infile)

        def extract_param(tag):
            return {b: c for (a, b, c) in params if a == tag}

        self.geometry = Geometry(extract_param('geometry'))
        self.surfaceinteraction = SurfaceInteraction(extract_param('surfaceinteraction'))
        self.forces = Forces(extract_param('forces'))
        self.spatialdist = SpatialDist(extract_param('spatialdist'))
        self.speeddist = SpeedDist(extract_param('speeddist'))
        self.angulardist = AngularDist(extract_param('angulardist'))
        self.options = Options(extract_param('options'))

    def __eq__(self, other):
        if not isinstance(other, type(self)):
            return False
        return all([self.geometry == other.geometry,
                    self.surfaceinteraction == other.surfaceinteraction,
                    self.forces == other.forces,
                    self.spatialdist == other.spatialdist,
                    self.speeddist == other.speeddist,
                    self.angulardist == other.angulardist,
                    self.options == other.options])

    def __str__(self):
        return '\n'.join(str(s) for s in (self.geometry, self.surfaceinteraction, self.forces,
                                          self.spatialdist, self.speeddist, self.angulardist,
                                          self.options))

    __repr__ = __str__

    # ---- catalogue ------------------------------------------------------------------------
    def search(self):
        """(ids, outputs-or-filenames, npackets, totalsource) of the runs made with these inputs
        (Input.py:121-172, without the database)."""
        if not self._catalogue:
            return [], [], 0, 0
        ids = [o.idnum for o in self._catalogue]
        files = [o.filename if o.filename else o for o in self._catalogue]
        return (ids, files, int(sum(o.npackets for o in self._catalogue)),
                float(sum(o.totalsource for o in self._catalogue)))

    def delete_files(self, filename=None):
        keep = []
        for o in self._catalogue:
            if filename is None or o.filename == filename:
                if o.filename and os.path.exists(o.filename):
                    os.remove(o.filename)
            else:
                keep.append(o)
        self._catalogue = keep
        self._fused = []

    def chunk_size(self, packs_per_it=None):
        """Packets per Output (Input.py:216-227): 1e6 in variable-step mode, else
        ceil(1024^3 / nsteps / 8)."""
        if (packs_per_it is None) and (self.options.step_size == 0):
            packs_per_it = 1000000
        elif packs_per_it is None:
            nsteps = int(np.ceil(self.options.endtime.value / self.options.step_size) + 1)
            packs_per_it = np.ceil(1024**3 / nsteps / 8)
        return int(packs_per_it)

    def run(self, npackets, packs_per_it=None, overwrite=False, compress=True,
            distribute=False, seed=None, *, device=0, keep_trajectory=True, context=None):
        """Run the model (Input.py:175-268).  Each chunk is one Output.  A given ``seed`` seeds
        chunk k with ``seed + k`` -- chunk 0 is the reference's stream; the reference itself
        re-uses the same seed for every chunk (Input.py:246), which repeats identical packets."""
        from .Output import Output
        t0 = time.time()
        distribute = distribute in (True, 'delay', 'delayed')
        if overwrite:
            self.delete_files()
            totalpackets = 0
        else:
            _, outputfiles, totalpackets, _ = self.search()
            print(f'Found {len(outputfiles)} files with {totalpackets} packets.')
        npackets = int(npackets)
        ntodo = npackets - totalpackets
        chunk = 0
        while ntodo > 0:
            per_it = int(np.min([ntodo, self.chunk_size(packs_per_it)]))
            nits = int(np.ceil(ntodo/per_it))
            print('Running Model')
            print(f'Will complete {nits} iterations of {per_it} packets.')
            if distribute:
                assert False, 'Dont do this'
            for it in range(nits):
                print(f'Starting iteration #{it+1} of {nits}')
                out = Output(self, per_it, compress=compress,
                             seed=None if seed is None else seed + chunk, device=device,
                             keep_trajectory=keep_trajectory, context=context)
                context = out.context()
                chunk += 1
            _, outputfiles, totalpackets, _ = self.search()
            print(f'Found {len(outputfiles)} files with {totalpackets} packets.')
            ntodo = npackets - totalpackets
        print(f'Model run completed in {time.time()-t0:.2f} sec.')

    def produce_image(self, format_, overwrite=False, distribute=None, **kwargs):
        from .ModelImage import ModelImage
        return ModelImage(self, format_, overwrite=overwrite, distribute=distribute, **kwargs)
