"""The C checker (kernels' roundings: deterministic cube / exp / log, tableau terms as fused
multiply-adds) against the NumPy oracle (the reference's roundings, pinned bit for bit to the
reference's rk5.py / state.py) on 20 000 seeded packets, all 1667 steps, 512 x 512 images of the float32
samples: step counts, per-pixel packet counts, brightness and final states.  CPU only, 2 GB, ~20 s:
    python tests/tools/fused_terms_vs_reference.py > profiles/r03_c_oracle_vs_reference_arithmetic.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import np_oracle as O
from oracle.c_oracle import COracle
from tests import helpers as H
co=COracle()
f=H.mercury_forces('Na',1.3)
n,endtime,step,edge=20000,50000.,30.,25.
X0=H.sample_x0(n,999,endtime)
nsteps,n_iter=O.n_output_steps(endtime,step)
t=time.time(); results,_,work=O.constant_step_driver(f,X0,endtime,step,edge); print('numpy driver',time.time()-t,'s work',work)
for q in ('radiance','column'):
    im=H.image_setup(f,q,dims=(512,512))
    desc=co.image_desc(im['M'],f.vrplanet,im['apix'],q,im['g_tables'],im['xedges'],im['zedges'],downcast=True)
    c=co.integrate_const(f,X0,step,n_iter,edge,img=desc,threads=co.max_threads())
    last=(results[:,7,:n_iter]>0).sum(axis=1)
    s=O.samples_from_results(results,compress=True,downcast=True)
    ref_img,ref_cnt,_,_=O.create_image(s['x'],s['y'],s['z'],s['vy'],s['frac'],f.vrplanet,im['M'],q,im['g_tables'],im['dims'],im['xrange'],im['zrange'],im['apix'],matmul=False)
    fin=results[np.arange(n),:,last]
    nz=ref_img>0
    print(q,'work equal',c['work']==work,'steps equal',np.array_equal(c['steps'],last),'binned',int(ref_cnt.sum()),'count pixels differing',int((c['counts']!=ref_cnt.astype(np.uint64)).sum()),
          'max rel image diff',float(np.max(np.abs(c['image'][nz]-ref_img[nz])/ref_img[nz])),'max rel state diff',float(np.nanmax(np.abs(c['final']-fin)/np.maximum(np.abs(fin),1e-300))))
