import sys, os; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from types import SimpleNamespace
from facl_amd.cn3d_model_conbag import PointNet_Plus
from facl_amd.train_common import ContrastiveStep, synthetic_batch
dev=torch.device('cuda:0')
B,T,N,D=32,24,2048,3
opt=SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64, sample_num_level2=64, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B, pooling="concatenation", SAMPLE_NUM=N)
torch.manual_seed(1)
net=PointNet_Plus(opt,gost=T).to(dev).train()
optim=torch.optim.Adam(net.parameters(), lr=3e-4, betas=(0.5,0.999), eps=1e-6)
step=ContrastiveStep(net,optim,opt,T)
x=synthetic_batch(B,T,N,D,dev)
for _ in range(3): step(x)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(3): step(x)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
