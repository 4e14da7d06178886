import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import utils_my
dev = "cuda:0"
torch.manual_seed(0)
for (M, N, D) in ((768, 2048, 3), (768, 2048, 4), (320, 512, 4)):
    pts = torch.rand(M, N, D, device=dev) - 0.5
    for _ in range(3): utils_my.knn_radius_group(pts, 64, 64, 0.16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): utils_my.knn_radius_group(pts, 64, 64, 0.16)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} N={N} D={D}: {e0.elapsed_time(e1)/20:.4f} ms  (FACL_GROUP_LDS={os.environ.get('FACL_GROUP_LDS','0')})")
