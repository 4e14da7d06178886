"""Appearance-stream training entry -- counterpart of training_code/cn3d_train_apperance_GL.py, which
differs from the motion script in three lines (:135 branch_choose='1', :161 data root, :341 checkpoint
name ``corr_GL_appereance_<epoch>.pth``).  Same model, same kernels, same loss."""
from .train_common import run


def main(args=None):
    return run(default_branch='1', ckpt_pattern='%s/corr_GL_appereance_%d.pth', args=args)


if __name__ == '__main__':
    main()
