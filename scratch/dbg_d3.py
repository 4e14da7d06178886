import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from types import SimpleNamespace
from helpers import load_golden, max_rel_rows, rel_err
from oracle import encoder as E, grouping as OG
from oracle.weights import formula_state_dict
from facl_amd.cn3d_model_conbag import PointNet_Plus
from facl_amd.utils_my import group_points_3DV
tag, D = sys.argv[1], int(sys.argv[2])
g = load_golden(f"c1_{tag}.npz")
B,G,N,S,K,_ = [int(v) for v in g["meta"]]
opt = SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64, sample_num_level2=64, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B, pooling="concatenation", SAMPLE_NUM=N)
net = PointNet_Plus(opt, gost=G); net.load_state_dict({k: torch.as_tensor(v) for k,v in formula_state_dict(D, neg_gamma=('neg' in tag)).items()}); net=net.cuda().train()
pts = torch.from_numpy(g["points"]).cuda()
xt, yt = group_points_3DV(pts, opt)
with torch.no_grad(): out = net(xt, yt, 1)
_, xt_o, yt_o = OG.group_points(g["points"], S, K, 0.06)
M=G*B
def ref(dtype):
    sd = {k:(torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind=='f' else torch.as_tensor(v).clone()) for k,v in formula_state_dict(D, neg_gamma=('neg' in tag)).items()}
    with torch.no_grad():
        return E.encoder_forward(sd, torch.from_numpy(xt_o).permute(0,3,1,2).to(dtype), torch.from_numpy(yt_o).view(M,1,S,3).transpose(1,3).to(dtype), G, True, return_intermediates=True)
o64,i64 = ref(torch.float64); o32,i32 = ref(torch.float32)
for n,a,b,c in zip(("x","code","x_nor","x_global"), out, o64, o32):
    print(n, "mine-vs-64 %.2e"%max_rel_rows(a.cpu().numpy(), b.numpy()), "t32-vs-64 %.2e"%max_rel_rows(c.numpy(), b.numpy()), "golden-vs-64 %.2e"%max_rel_rows(g["train_"+n], b.numpy()), "mine-vs-golden %.2e"%max_rel_rows(a.cpu().numpy(), g["train_"+n]))
