"""Motion-stream training entry -- counterpart of the reference's training_code/cn3d_train_motion_GL.py
(main at :74-343).  Same flags, same checkpoint name ``corr_GL_<epoch>.pth`` (:341).

    python -m facl_amd.cn3d_train_motion_GL --batchSize 32 --num_crop 24 --SAMPLE_NUM 2048 --nepoch 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m facl_amd.cn3d_train_motion_GL ...
"""
from .train_common import run


def main(args=None):
    return run(default_branch='0', ckpt_pattern='%s/corr_GL_%d.pth', args=args)


if __name__ == '__main__':
    main()
