import sys, json, time
sys.path.insert(0, '.')
import numpy as np, torch
from colormipsearch_amd import cds
W,H=1210,566
ctx=cds.Context(0); st=torch.cuda.current_stream(); ctx.set_stream(st.cuda_stream)
ml=cds.Library(W,H,16,context=ctx); ml.generate_synthetic(0xBADC0DE,0,0,16,kind=2); masks=ml.download(0,16)
alg=cds.PixelMatchColorDepthSearchAlgorithm(masks,20,True,None,0,False,20,0.01,2,cds.label_regions(W),context=ctx)
n=262144
dense=cds.Library(W,H,16384,context=ctx); sl=cds.SparseLibrary(W,H,n,24000*n+4000000,context=ctx)
for f in range(0,n,16384):
    dense.generate_synthetic(0xC0FFEE,f,0,16384,kind=0); sl.append_from(dense,0,16384)
sc=torch.empty(n,dtype=torch.int32,device='cuda')
for m in range(16):
    a=cds.PixelMatchColorDepthSearchAlgorithm(masks[m],20,True,None,0,False,20,0.01,2,cds.label_regions(W),context=ctx)
    for _ in range(2): a.searchSparseLibraryDevice(sl,0,n,sc.data_ptr())
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(3): a.searchSparseLibraryDevice(sl,0,n,sc.data_ptr())
    e1.record(st); torch.cuda.synchronize()
    print(m, a.getQuerySize(), a.sparseLayout(), round(e0.elapsed_time(e1)/3,3), "ms")
