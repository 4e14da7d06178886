"""Motion-stream feature extraction -- counterpart of training_code/extract_motion_feature.py (main :34-221).

    python -m facl_amd.extract_motion_feature --checkpoint ck/corr_GL_0.pth --num_crop 10 --save_path feats/
"""
from .extract_common import run


def main(args=None):
    return run(default_branch='0', default_ckpt='../ntu/ntu60_new2/model/corr_GL_.pth', args=args)


if __name__ == '__main__':
    main()
