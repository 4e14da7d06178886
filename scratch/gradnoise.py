import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import encoder as E, grouping as OG, loss as OL
from oracle.weights import formula_state_dict
from helpers import load_golden, rel_err
g = load_golden("c1_d4.npz")
B,G,N,S,K,D = [int(v) for v in g["meta"]]
idx, xt, yt = OG.group_points(g["points"], S, K, 0.06)
M=G*B
def run(dtype):
    sd = {k:(torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind=='f' else torch.as_tensor(v).clone()) for k,v in formula_state_dict(4).items()}
    keys=[k for k in sd if 'running' not in k and 'num_b' not in k]
    for k in keys: sd[k].requires_grad_(True)
    x,code,xn,xg = E.encoder_forward(sd, torch.from_numpy(xt).permute(0,3,1,2).to(dtype), torch.from_numpy(yt).view(M,1,S,3).transpose(1,3).to(dtype), G, True)
    loss = OL.global_contrast(G,xg,x,B)+OL.circle_contrast(G,x,B,g["order"])
    loss.backward()
    return {k:sd[k].grad for k in keys if sd[k].grad is not None}, float(loss)
g64,l64 = run(torch.float64); g32,l32=run(torch.float32)
print("loss", l64, l32, float(g["losses3"][0]))
for k in g64:
    a=g64[k].numpy(); b=g32[k].numpy()
    gold = g.get("grad/"+k)
    print(f"{k:22s} |g|={np.linalg.norm(a):9.3f} oracle32-vs-64 {rel_err(b,a):.2e}", (f"golden-vs-64 {rel_err(gold,a):.2e}" if gold is not None else ""), f"goldnorm-vs-64norm {abs(float(g['gradnorm/'+k])-np.linalg.norm(a))/np.linalg.norm(a):.2e}")
