"""Drop-in for the reference's ``pytorch_version/feat_extractor.py`` on MI355X: same class, ``forward`` contract
([N,3,H,W] -> [N,32,H,W]) and ``state_dict`` keys.  Bottom-up 3x3 convolutions run on the fp32-MFMA kernels (the
``torch.cat((image, features))`` inputs are concatenated virtually), the image pyramid and the depthwise
transposed convolutions of the top-down path (with their residual adds fused) on dedicated kernels.  conv_1 / conv_2 of
every level feed exactly one convolution each, so their ReLU backward is fused into that consumer (``fuse_bwd``)."""
import torch.nn as nn

from dvf import lib as _L
from dvf.conv import DepthwiseUp2x, FusedAct, FusedConv2d, bilinear_half, xavier_init_

_LEVELS = (5, 4, 3, 2, 1)         # b5 = full resolution ... b1 = 1/16 (reference feat_extractor.py:18-36)


class FeatExtractor(nn.Module):

    def __init__(self):
        super(FeatExtractor, self).__init__()
        for lvl in _LEVELS:
            cin, stride = (3, 1) if lvl == 5 else (35, 2)
            setattr(self, f"conv_1_b{lvl}", nn.Sequential(FusedConv2d(cin, 32, 3, stride, 1, _L.ACT_RELU, fuse_bwd=True), FusedAct()))
            setattr(self, f"conv_2_b{lvl}", nn.Sequential(FusedConv2d(32, 32, 3, 1, 1, _L.ACT_RELU, fuse_bwd=True), FusedAct()))
            setattr(self, f"conv_3_b{lvl}", FusedConv2d(32, 32, 3, 1, 1))
        for name in ("conv_3_b1_up", "conv_3_bb2_up", "conv_3_bb3_up", "conv_3_bb4_up"):      # :38-41
            setattr(self, name, DepthwiseUp2x(32))

    def forward(self, imgs):
        pyramid = [imgs]
        for _ in range(3):                                                  # :44-46 (x0.5 bilinear, no gradient)
            pyramid.append(bilinear_half(pyramid[-1]))
        lateral = {}
        inputs = (imgs,)
        for j, lvl in enumerate(_LEVELS):
            h = getattr(self, f"conv_1_b{lvl}")[0](*inputs)                # (image, features) concatenated virtually
            h = getattr(self, f"conv_2_b{lvl}")[0](h)
            lateral[lvl] = getattr(self, f"conv_3_b{lvl}")(h)
            if lvl > 1:
                inputs = (pyramid[j], lateral[lvl])
        h = lateral[1]
        for name, lvl in (("conv_3_b1_up", 2), ("conv_3_bb2_up", 3), ("conv_3_bb3_up", 4), ("conv_3_bb4_up", 5)):
            h = getattr(self, name)(h, lateral[lvl])                         # lateral + up(h), :72-82
        return h

    def init_weights(self):
        xavier_init_(self)
