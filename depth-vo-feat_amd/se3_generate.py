"""Drop-in for the reference's ``pytorch_version/se3_generate.py`` on MI355X: ``generate_se3`` maps a batch of
se(3) vectors (wx,wy,wz,ux,uy,uz), shaped [N,6,1,1] like the Caffe blob, to [N,1,4,4] rigid transforms with
R = exp([w]x) and t = R u.  The reference does this with per-sample numpy loops on the CPU and a device round trip
(se3_generate.py:7-105); here forward and the reference's hand-written backward are two tiny HIP kernels."""
import torch

from dvf import lib as _L
from dvf.ops import PoseVec2MatFn


def generate_se3(input):
    n = input.size(0)
    mat34 = PoseVec2MatFn.apply(input.reshape(n, 6), _L.POSE_SE3)          # [N,3,4]
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], device=mat34.device).expand(n, 1, 4)
    return torch.cat((mat34, bottom), dim=1).view(n, 1, 4, 4)
