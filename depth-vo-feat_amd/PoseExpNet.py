"""Drop-in for the reference's ``pytorch_version/PoseExpNet.py`` (class ``PoseExpNet`` taking one 6-channel
frame pair ``imgs[B,6,H,W]`` and returning one 6-DoF pose ``[B,6]``) on MI355X."""
from PoseExpNet_sfm import _PoseExpBase


class PoseExpNet(_PoseExpBase):

    def __init__(self, output_exp=True):
        super(PoseExpNet, self).__init__()
        self.nb_ref_imgs = 2
        self.output_exp = output_exp
        self._build(6, 6, self.nb_ref_imgs, output_exp)

    def forward(self, imgs):
        """``imgs`` is the reference's [B,6,H,W] pair; a (img_a, img_b) tuple of [B,3,H,W] tensors is accepted
        as well and concatenated virtually inside the first convolution."""
        masks, pose = self._run(imgs)
        pose = pose.view(pose.size(0), 6)
        if self.training:
            return masks, pose
        return masks[0], pose
