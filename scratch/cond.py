import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import encoder as E, grouping as OG
from oracle.weights import formula_state_dict
from helpers import load_golden, max_rel_rows, rel_err
g = load_golden("c1_d4.npz")
B,G,N,S,K,D = [int(v) for v in g["meta"]]
idx, xt, yt = OG.group_points(g["points"], S, K, 0.06)
M=G*B
xt_t = torch.from_numpy(xt).permute(0,3,1,2); yt_t = torch.from_numpy(yt).view(M,1,S,3).transpose(1,3)
def run(sdnp, dtype, perm=None):
    sd = {k:(torch.as_tensor(v).to(dtype) if v.dtype.kind=='f' else torch.as_tensor(v)) for k,v in sdnp.items()}
    x_in = xt_t.to(dtype)
    if perm is not None: x_in = x_in[:,:,:,perm]
    with torch.no_grad():
        out,inter = E.encoder_forward(sd, x_in, yt_t.to(dtype), G, training=True, return_intermediates=True)
    return out, inter
for wname, sdnp in [("formula", formula_state_dict(4))]:
    o64,i64 = run(sdnp, torch.float64)
    o32,i32 = run(sdnp, torch.float32)
    perm = torch.randperm(K)
    o32p,i32p = run(sdnp, torch.float32, perm)
    for n,a,b,c in zip(("x","code","x_nor","x_global"), o32,o64,o32p):
        print(wname, n, "fp32 vs fp64 maxrow", max_rel_rows(a.numpy(), b.numpy()), "global", rel_err(a.numpy(), b.numpy()), "| perm vs fp32", max_rel_rows(c.numpy(), a.numpy()))
    for n in i64:
        print("  inter", n, rel_err(i32[n].numpy(), i64[n].numpy()), float(i64[n].abs().mean()))
    print("golden vs fp64:", max_rel_rows(g["train_x"], o64[0].numpy()), "golden vs fp32", max_rel_rows(g["train_x"], o32[0].numpy()))
# default torch init
import torch.nn as nn
torch.manual_seed(1)
sd2 = {}
from oracle.weights import state_dict_shapes
for k,shape in state_dict_shapes(4):
    if k.endswith("num_batches_tracked"): sd2[k]=np.array(0)
    elif "running_mean" in k: sd2[k]=np.zeros(shape,np.float32)
    elif "running_var" in k: sd2[k]=np.ones(shape,np.float32)
    elif len(shape)==1 and any(s in k for s in (".1.",".4.",".7.")): sd2[k]=(np.ones if k.endswith("weight") else np.zeros)(shape,np.float32)
    elif k.endswith("weight"):
        w=torch.empty(shape); nn.init.kaiming_uniform_(w, a=5**0.5); sd2[k]=w.numpy()
    else:
        fan_in = {"net3DV_1.0":4,"net3DV_1.3":64,"net3DV_1.6":64,"net3DV_3.0":259,"net3DV_3.3":256,"net3DV_3.6":512,"netR_FC.0":1024,"netR_FC.3":1024}[k.rsplit('.',1)[0]]
        b=1/fan_in**0.5; sd2[k]=(torch.rand(shape)*2*b-b).numpy()
o64,i64 = run(sd2, torch.float64); o32,i32 = run(sd2, torch.float32); o32p,_=run(sd2, torch.float32, torch.randperm(K))
for n,a,b,c in zip(("x","code","x_nor","x_global"), o32,o64,o32p):
    print("default-init", n, "fp32 vs fp64 maxrow", max_rel_rows(a.numpy(), b.numpy()), "global", rel_err(a.numpy(), b.numpy()), "| perm vs fp32", max_rel_rows(c.numpy(), a.numpy()))
