"""Offline: how often the cell-run gather of a Hilbert-ordered C4-patch cloud must fetch its four corner texels: one remembered cell (the kernel)
against two (LRU), runs of 8 / 16 points."""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/scripts/notebook')
import numpy as np, torch
import sim_cloud_tiles as S
import bench
from d3fields_amd import synth
w = bench.WORKLOADS["c4_patch"]; V,H,W = w["V"],w["H"],w["W"]; fh,fw = w["fhw"]
sc = synth.make_scene(V,H,W,"smooth")
pts = synth.random_cloud(1000000, seed=3).numpy()
lo = pts.min(0); ext = (pts.max(0)-lo).max(); inv = 511.0/ext
q = np.floor((pts-lo)*inv).astype(np.int64)
key = S.hilbert_key(q)
order = np.argsort(key, kind="stable")
p = pts[order]
K = sc["K"].numpy(); pose = sc["pose"].numpy()
cells=[]
for v in range(V):
    M = K[v] @ pose[v]
    xc = p@M[:,:3].T + M[:,3]
    u = xc[:,0]/xc[:,2]; ww = xc[:,1]/xc[:,2]
    gx = u/(W-1)*2-1; gy = ww/(H-1)*2-1
    ix = (gx+1)/2*(fw-1); iy=(gy+1)/2*(fh-1)
    cells.append((np.floor(iy).astype(np.int64)<<16) + np.floor(ix).astype(np.int64))
cells=np.stack(cells,1)   # [n,V]
n=len(p)
for Krun in (8,16):
    same1 = (cells[1:]==cells[:-1])
    start = (np.arange(1,n)%Krun==0)[:,None]
    f1 = 1-(same1 & ~start).mean()
    # LRU-2: hit if equals prev or prev-prev cell register content: simulate exactly per view
    miss=0
    for v in range(V):
        c=cells[:,v]; a=-1;b=-1; m=0
        for i in range(0, n if n<200000 else 200000):
            if i%Krun==0: a=-1;b=-1
            ci=c[i]
            if ci==a: pass
            elif ci==b: a,b=b,a
            else: m+=1; b=a; a=ci
        miss+=m
    tot = V*min(n,200000)
    print("K=%d: fetch fraction 1-cell %.3f, LRU-2 %.3f" % (Krun, f1, miss/tot))
