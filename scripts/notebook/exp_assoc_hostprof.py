import cProfile, pstats, sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from d3fields_amd import Fusion, synth
dev="cuda:0"
V,H,W=4,240,320
sc=synth.make_scene(V,H,W,"smooth")
K,pose,depth=sc["K"].numpy(),sc["pose"].numpy(),sc["depth"].numpy()
det=synth.multiview_segmentation(K,pose,depth,seed=3)
f=Fusion(num_cam=V,device=dev,mask_producer=lambda fu,q,t,b,**kw:{"mask_gs":det[0],"mask_label":det[1],"mask_conf":det[2]})
f.update({"color":np.zeros((V,H,W,3),np.uint8),"depth":depth,"pose":pose,"K":K,"dino_feats":np.zeros((V,4,4,4),np.float32)})
box=dict(synth.WORK_BOX)
f.text_queries_for_inst_mask_no_track(["mug","box","pen"],[0.3]*3,box)
torch.cuda.synchronize()
import time
t0=time.perf_counter()
f.text_queries_for_inst_mask_no_track(["mug","box","pen"],[0.3]*3,box)
torch.cuda.synchronize()
print("second call ms", 1e3*(time.perf_counter()-t0))
pr=cProfile.Profile(); pr.enable()
f.text_queries_for_inst_mask_no_track(["mug","box","pen"],[0.3]*3,box)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
