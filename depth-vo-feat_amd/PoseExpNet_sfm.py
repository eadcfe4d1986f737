"""Drop-in for the reference's ``pytorch_version/PoseExpNet_sfm.py`` (class ``PoseExpNet`` taking
``(target_image, ref_imgs)``) on MI355X: same constructor, return convention (4 masks in train mode, the finest
in eval mode) and ``state_dict`` keys; convolutions on the fp32-MFMA kernels, the channel concatenation of the
input frames done virtually inside the first convolution, the pose head's spatial mean in its own kernel."""
import torch
import torch.nn as nn

from dvf import lib as _L
from dvf.conv import FusedAct, FusedConv2d, FusedConvTranspose2d, SpatialMeanFn, xavier_init_

_ENC_PLANES = (16, 32, 64, 128, 256, 256, 256)      # reference PoseExpNet_sfm.py:27
_ENC_KERNEL = (7, 5, 3, 3, 3, 3, 3)
_DEC_PLANES = (256, 128, 64, 32, 16)                 # :39


class _PoseExpBase(nn.Module):
    """Shared body of the two PoseExpNet variants (they differ only in how frames / poses are packed)."""

    def _build(self, in_planes, n_pose_out, nb_masks, output_exp):
        cin = in_planes
        for i, (co, k) in enumerate(zip(_ENC_PLANES, _ENC_KERNEL), start=1):
            setattr(self, f"conv{i}", nn.Sequential(FusedConv2d(cin, co, k, 2, (k - 1) // 2, _L.ACT_RELU, fuse_bwd=True), FusedAct()))
            cin = co
        self.pose_pred = FusedConv2d(_ENC_PLANES[6], n_pose_out, 1, 1, 0)
        if output_exp:
            up_in = (_ENC_PLANES[4],) + _DEC_PLANES[:-1]
            for j, lvl in enumerate(range(5, 0, -1)):
                setattr(self, f"upconv{lvl}", nn.Sequential(
                    FusedConvTranspose2d(up_in[j], _DEC_PLANES[j], 4, 2, 1, _L.ACT_RELU, fuse_bwd=True), FusedAct()))
            for lvl, ci in zip((4, 3, 2, 1), _DEC_PLANES[1:]):
                setattr(self, f"predict_mask{lvl}", FusedConv2d(ci, nb_masks, 3, 1, 1, _L.ACT_SIGMOID_AFFINE))

    def init_weights(self):
        xavier_init_(self)

    def _run(self, frames):
        outs = []
        h = frames
        for i in range(1, 8):
            h = getattr(self, f"conv{i}")[0](*h) if isinstance(h, (list, tuple)) else getattr(self, f"conv{i}")[0](h)
            outs.append(h)
        pose = SpatialMeanFn.apply(self.pose_pred(h), 0.01)                 # mean(3).mean(2) * 0.01
        masks = [None, None, None, None]
        if self.output_exp:
            first = frames[0] if isinstance(frames, (list, tuple)) else frames
            refs = [outs[3], outs[2], outs[1], outs[0], first]
            h = outs[4]
            ups = []
            for j, lvl in enumerate(range(5, 0, -1)):
                h = getattr(self, f"upconv{lvl}")[0](h, out_hw=(refs[j].size(2), refs[j].size(3)))
                ups.append(h)
            for lvl in (4, 3, 2, 1):
                masks[lvl - 1] = getattr(self, f"predict_mask{lvl}")(ups[5 - lvl])   # sigmoid fused
        return masks, pose


class PoseExpNet(_PoseExpBase):

    def __init__(self, nb_ref_imgs=2, output_exp=True):
        super(PoseExpNet, self).__init__()
        self.nb_ref_imgs = nb_ref_imgs
        self.output_exp = output_exp
        self._build(3 * (1 + nb_ref_imgs), 6 * nb_ref_imgs, nb_ref_imgs, output_exp)

    def forward(self, target_image, ref_imgs):
        assert(len(ref_imgs) == self.nb_ref_imgs)
        frames = [target_image] + list(ref_imgs)
        if len(frames) > _L.MAX_SEGS:         # the first convolution walks up to 5 frames (nb_ref_imgs <= 4) in place
            raise ValueError(f"nb_ref_imgs = {self.nb_ref_imgs}: at most {_L.MAX_SEGS - 1} reference images are supported")
        masks, pose = self._run(frames)
        pose = pose.view(pose.size(0), self.nb_ref_imgs, 6)
        if self.training:
            return masks, pose
        return masks[0], pose
