"""View construction of one clip: NumPy restatement of the reference's per-sample augmentation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/views.npz: tools/make_goldens.py imports
`training_code/cn3D_data_set.py` (with an empty harness-side stand-in for its unused `import imageio`), calls the
reference class's own get_temporal_augment_data / get_data_train on synthetic clips under np.random.seed and stores
the float64 outputs; tests/test_oracle_golden.py holds this restatement to them bit for bit.

Follows cn3D_data_set.py:
  get_temporal_augment_data :654-663    jitter_point_cloud :767-778    reverse_transform :708-713
  rotate_trans :734-749                 get_data_train :285-350 (the 10-view layout of the motion / appearance streams)
The reference draws from NumPy's GLOBAL generator; here the generator is explicit (`rng`, a RandomState seeded like
the global one would be) and every draw happens in the reference's order, so the same seed gives the same views.
"""
import numpy as np

NUM_POINT = 512            # cn3D_data_set.py:24


def jitter_point_cloud(rng, batch_data, sigma=0.01, clip=0.05):
    """:767-778  float64 noise clip(sigma*randn, -clip, clip) + data (result float64)."""
    B, N, C = batch_data.shape
    jittered = np.clip(sigma * rng.randn(B, N, C), -1 * clip, clip)
    jittered += batch_data
    return jittered


def reverse_transform(rng, points):
    """:708-713  copy into float32, mirror x, jitter (float64 noise + float32 xyz, stored back into the float32 copy)."""
    rev = np.zeros(points.shape, dtype=np.float32)
    rev[:, :, :] = points[:, :, :]
    rev[:, :, 0] = -rev[:, :, 0]
    rev[:, :, 0:3] = jitter_point_cloud(rng, rev[:, :, 0:3])
    return rev


def rotate_trans(rng, points):
    """:734-749  copy into float32, rotate xyz about y by (rand-0.5)*0.8*pi: float32 (N,3) @ float64 (3,3) -> float32."""
    rot = np.zeros(points.shape, dtype=np.float32)
    rot[:, :, :] = points[:, :, :]
    for k in range(rot.shape[0]):
        angle = (rng.rand() - 0.5) * np.pi * 0.8
        Ry = np.array([[np.cos(angle), 0, np.sin(angle)],
                       [0, 1, 0],
                       [-np.sin(angle), 0, np.cos(angle)]])
        shape_pc = rot[k, :, 0:3]
        rot[k, :, 0:3] = np.dot(shape_pc.reshape((-1, 3)), Ry)
    return rot


def get_temporal_augment_data(rng, pointss, temporal_int):
    """:654-663  xyz + one temporal channel, rows whose temporal channel is non-zero, 512 draws with replacement."""
    points = pointss.copy()
    pt = np.concatenate((points[:, 0:3].copy(), points[:, temporal_int:temporal_int + 1].copy()), axis=1)
    pt = pt[np.where(pt[:, 3] != 0)]
    idex = rng.randint(0, pt.shape[0], 512)
    return pt[idex]


def get_data_train(rng, points, key_points, time_seg2, time_seg4, res_points_1, res_points_2, num_crop=10):
    """:285-350  (num_crop, NUM_POINT, 4) float64: raw, reversed, 2 key-point views, 2 rotated, 2 temporal, 2 low-res."""
    def draw(src):
        idex = rng.randint(0, src.shape[0], NUM_POINT)
        return src[idex].copy().reshape(1, NUM_POINT, 4)

    raw_p = draw(points)
    rev_p = draw(points)
    rev_p[:, :, :3] = jitter_point_cloud(rng, rev_p[:, :, :3])
    rev_p = reverse_transform(rng, rev_p)

    ke1_p = draw(key_points)
    ke1_p[:, :, :3] = jitter_point_cloud(rng, ke1_p[:, :, :3])
    ke2_p = draw(key_points)
    ke2_p[:, :, :3] = jitter_point_cloud(rng, ke2_p[:, :, :3])
    ke2_p = reverse_transform(rng, ke2_p)

    ro1_p = draw(points)
    ro1_p[:, :, :3] = jitter_point_cloud(rng, ro1_p[:, :, :3])
    ro1_p = rotate_trans(rng, ro1_p)
    ro2_p = draw(points)
    ro2_p[:, :, :3] = jitter_point_cloud(rng, ro2_p[:, :, :3])
    ro2_p = rotate_trans(rng, ro2_p)

    ti1_p = time_seg2.reshape(1, NUM_POINT, 4)
    ti2_p = time_seg4.reshape(1, NUM_POINT, 4)

    rs1_p = draw(res_points_1)
    rs2_p = draw(res_points_2)

    data_pairs = np.empty([num_crop, NUM_POINT, 4], dtype=float)
    for i, v in enumerate((raw_p, rev_p, ke1_p, ke2_p, ro1_p, ro2_p, ti1_p, ti2_p, rs1_p, rs2_p)):
        data_pairs[i:i + 1, :, :] = v
    return data_pairs


def get_item(rng, points, key_points, res_points_1, res_points_2):
    """__getitem__ :105-122 (branch '0'; the appearance branch differs in file paths only)."""
    time_seg2 = get_temporal_augment_data(rng, points, 4)
    time_seg4 = get_temporal_augment_data(rng, points, 7)
    return get_data_train(rng, points[:, :4], key_points[:, :4], time_seg2[:, :4], time_seg4[:, :4],
                          res_points_1[:, :4], res_points_2[:, :4], num_crop=10)


def collate_view_major(items):
    """The loop head cn3d_train_motion_GL.py:225-228: (B,G,N,D) -> permute(1,0,2,3).reshape(G*B,N,D) -> float32."""
    out = np.stack(items, 0)
    B, G, N, D = out.shape
    return out.transpose(1, 0, 2, 3).reshape(G * B, N, D).astype(np.float32)
