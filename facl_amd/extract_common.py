"""Shared body of the feature-extraction entries: counterpart of the loop of
training_code/extract_motion_feature.py:143-221 -- load a checkpoint into PointNet_Plus, eval() (BatchNorm folded from
the running statistics: the SA point-MLP then runs as fwd2 + fwd3 + pool with no statistics passes), group, forward,
``cat((x, x_global), 0)`` -> one (num_crop+1)*512 float32 vector per clip."""
import os

import numpy as np
import torch

from . import cn3d_model_conbag as MM
from .train_common import build_parser, synthetic_batch
from .utils_my import group_points_3DV, knn_radius_group


def save_single_feature(feature, save_path, name, num_crop=11):
    """extract_motion_feature.py:217-221: (num_crop*B, 512) rows [view-major x | x_global] -> per-clip vectors."""
    feature = feature.reshape(num_crop, -1, 512).transpose(1, 0, 2).reshape(-1, num_crop * 512)
    for batch_i in range(feature.shape[0]):
        np.save(os.path.join(save_path, name[batch_i] + '.npy'), feature[batch_i])
    return feature


def extract_batch(netR, out_points, opt, group_radius=None):
    """(B,G,N,D) clips -> (B, (G+1)*512) features, exactly the reference's per-batch body (:171-182)."""
    B, G, N, D = out_points.shape
    data1 = out_points.permute(1, 0, 2, 3).reshape(-1, N, D).float()
    if group_radius is None and opt.SAMPLE_NUM == 512:
        xt, yt = group_points_3DV(data1, opt)
    else:
        opt.INPUT_FEATURE_NUM = D
        xt, yt = knn_radius_group(data1, opt.sample_num_level1, opt.knn_K, 0.16 if group_radius is None else group_radius)
    x, _, _, x_global = netR(xt, yt)
    feat = torch.cat((x, x_global), dim=0)
    return feat.reshape(G + 1, B, 512).permute(1, 0, 2).reshape(B, (G + 1) * 512)


def run(default_branch, default_ckpt, args=None):
    p = build_parser(default_branch)
    p.add_argument('--checkpoint', type=str, default=default_ckpt, help='NEW: state_dict to load (reference: literal path)')
    p.add_argument('--save_path', type=str, default='', help='NEW: output folder for <clip>.npy (reference: literal path)')
    p.add_argument('--num_batches', type=int, default=2, help='NEW: synthetic batches to extract')
    opt = p.parse_args(args)
    device = torch.device("cuda", opt.main_gpu)
    torch.cuda.set_device(device)
    netR = MM.PointNet_Plus(opt, gost=opt.num_crop)
    netR.load_state_dict(torch.load(opt.checkpoint, map_location="cpu", weights_only=True))
    netR = netR.to(device).eval()
    if opt.save_path:
        os.makedirs(opt.save_path, exist_ok=True)
    gen = torch.Generator(device=device)
    gen.manual_seed(7)
    feats = []
    with torch.no_grad():
        for i in range(opt.num_batches):
            pts = synthetic_batch(opt.batchSize, opt.num_crop, opt.SAMPLE_NUM, opt.INPUT_FEATURE_NUM, device, gen)
            f = extract_batch(netR, pts, opt, opt.group_radius).cpu().numpy()
            feats.append(f)
            if opt.save_path:
                for b in range(f.shape[0]):
                    np.save(os.path.join(opt.save_path, 'synthetic_%04d_%03d.npy' % (i, b)), f[b])
    return np.concatenate(feats)
