"""Oracle: farthest-point sampling + FPS reorder (NumPy).

Restates /root/reference/training_code/cn3d_data_load.py:301-320
(= cn3D_data_set.py:675-694 = generate_data/generate_NTU.py:299-318) and the reorder
cn3D_data_set.py:665-672 / cn3d_data_load.py:287-298, with the random start index made an
explicit argument (the reference draws it with np.random.randint).
"""
import numpy as np


def farthest_point_sampling_fast(pc, sample_num, start_idx):
    """pc (N,3) float32|float64 -> (sample_num,1) int32, the reference's return shape.

    dist^2 is accumulated in pc's dtype as (dx*dx + dy*dy) + dz*dz (numpy ``sum(axis=1)`` of a
    3-wide row); argmax takes the LOWEST index among equal maxima (np.argmax)."""
    pc = np.asarray(pc)
    n = pc.shape[0]
    idx = np.zeros((sample_num, 1), dtype=np.int32)
    idx[0] = start_idx
    d = pc - pc[start_idx][None, :]
    min_dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    for j in range(1, sample_num):
        idx[j] = np.argmax(min_dist)
        if j < sample_num - 1:                      # cn3d_data_load.py:315
            d = pc - pc[idx[j, 0]][None, :]
            nd = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            min_dist = np.minimum(min_dist, nd)     # concatenate(...).min(axis=1), :317-318
    return idx


def fps_order(pc_xyz, sample_num, start_idx):
    """Permutation putting the FPS picks first, then the remaining rows in ascending order
    (np.setdiff1d sorts), cn3D_data_set.py:668-670."""
    picks = farthest_point_sampling_fast(pc_xyz, sample_num, start_idx).ravel()
    others = np.setdiff1d(np.arange(pc_xyz.shape[0]), picks)
    return np.concatenate((picks, others)).astype(np.int64)


def fps_sample_data(points_xyzc, sample_num_level1, start_idx):
    """points (b,N,D) -> reordered copy; start_idx (b,) ints.  cn3D_data_set.py:665-672."""
    out = np.array(points_xyzc, copy=True)
    for kk in range(out.shape[0]):
        order = fps_order(out[kk, :, 0:3], sample_num_level1, int(start_idx[kk]))
        out[kk] = out[kk, order]
    return out


def fps_sample_data_2level(points_xyzc, s1, s2, start1, start2):
    """Two-level variant, cn3d_data_load.py:287-298."""
    out = np.array(points_xyzc, copy=True)
    for kk in range(out.shape[0]):
        order = fps_order(out[kk, :, 0:3], s1, int(start1[kk]))
        out[kk] = out[kk, order]
        order2 = fps_order(out[kk, 0:s1, 0:3], s2, int(start2[kk]))
        out[kk, 0:s1] = out[kk, order2]
    return out
