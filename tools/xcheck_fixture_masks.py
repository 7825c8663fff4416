import sys; sys.path.insert(0,'.')
import numpy as np, torch
from colormipsearch_amd import cds
W,H=1210,566
z=np.load('tests/golden/cds_fixtures.npz')
masks=np.stack([z[k] for k in ('em_12191','em_12191_FL','em_LPLC2')])
n=8192
lib=cds.Library(W,H,n); lib.generate_synthetic(0xC0FFEE,777,0,n,kind=0)
lib.upload(n-4, np.stack([z[k] for k in ('lm_BJD','lm_GMR','lm_VT016795','lm_VT033614')]))
sl=cds.SparseLibrary(W,H,n,70000*n//2); sl.append_from(lib)
for xy,mir,thr in ((2,True,20),(0,True,20),(2,False,60)):
    alg=cds.PixelMatchColorDepthSearchAlgorithm(masks,20,mir,None,0,False,thr,0.01,xy,cds.label_regions(W))
    a=alg.searchLibrary(lib); b=alg.searchSparseLibrary(sl)
    print(xy,mir,thr,[alg.sparseLayout(m)['slabs'] for m in range(3)], np.array_equal(a[0],b[0]) and np.array_equal(a[1],b[1]), int((a[0]>0).sum()), a[0][:, -4:].tolist())
