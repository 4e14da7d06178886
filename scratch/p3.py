import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from helpers import load_golden
from oracle import encoder as E, grouping as OG, step as OS
from oracle.weights import formula_state_dict
for tag,D,neg in (("d4",4,False),("d3",3,False),("d4_neg",4,True)):
    g=load_golden(f"c1_{tag}.npz")
    B,G,N,S,K,_=[int(v) for v in g["meta"]]
    _,xt,yt=OG.group_points(g["points"],S,K,0.06)
    M=G*B
    res={}
    for dt in (torch.float64, torch.float32):
        sd={k:(torch.as_tensor(v).to(dt) if np.asarray(v).dtype.kind=='f' else torch.as_tensor(v).clone()) for k,v in formula_state_dict(D,neg_gamma=neg).items()}
        opt=OS.AdamState(sd)
        xt_t=torch.from_numpy(xt).permute(0,3,1,2).to(dt); yt_t=torch.from_numpy(yt).view(M,1,S,3).transpose(1,3).to(dt)
        for it in range(3):
            OS.train_step(sd,opt,None,B,G,S,K,0.06,g["order"],epoch=0,grouped=(xt_t,yt_t))
        res[dt]={k:v.detach().double().numpy() for k,v in sd.items()}
    p0=formula_state_dict(D,neg_gamma=neg)
    for key in [k for k in g if k.startswith("param3/")]:
        k=key[7:]
        d_ref=g[key].astype(np.float64)-p0[k].astype(np.float64)
        d64=res[torch.float64][k].reshape(d_ref.shape)-p0[k]
        d32=res[torch.float32][k].reshape(d_ref.shape)-p0[k]
        n=np.linalg.norm
        print(tag,k,"gold-vs-64 %.3f  o32-vs-64 %.3f  gold-vs-o32 %.3f"%(n(d_ref-d64)/n(d64), n(d32-d64)/n(d64), n(d_ref-d32)/n(d32)))
