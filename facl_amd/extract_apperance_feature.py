"""Appearance-stream feature extraction -- counterpart of training_code/extract_apperance_feature.py (the motion
script with a different checkpoint / output path)."""
from .extract_common import run


def main(args=None):
    return run(default_branch='1', default_ckpt='../ntu/ntu60_new2/model/corr_GL_appereance_.pth', args=args)


if __name__ == '__main__':
    main()
