import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import torch, numpy as np
import time_ref_vs_port as T
from oracle import step as OS, encoder as E, loss as L
torch.set_num_threads(8)
B,G,N,D=8,24,2048,3
g=torch.Generator().manual_seed(0)
pts=OS.view_major(torch.rand(B,G,N,D,generator=g)-0.5).contiguous()
# reference pieces
R_utils=T.R_utils
for it in range(3):
    t0=time.time(); xt,yt=R_utils.group_points_3DV_2048(pts,64,64,SAMPLE_NUM=N); t1=time.time()
    _,xt2,yt2=OS.group_torch(pts,64,64,0.16); t2=time.time()
    print("group ref %.3f port %.3f"%(t1-t0,t2-t1))
ref=T.ref_step_fn(B,G,N,D); port=T.port_step_fn(B,G,N,D)
for it in range(4):
    t0=time.time(); ref(pts); t1=time.time(); port(pts); t2=time.time()
    print("step ref %.3f port %.3f"%(t1-t0,t2-t1))
