"""Drop-in for the reference's ``pytorch_version/DispNetS.py`` on MI355X.

Same class name, constructor arguments, ``forward`` / ``init_weights`` contract and ``state_dict`` keys
(``conv1.0.weight`` ... ``predict_disp1.0.bias``), so checkpoints interchange.  Every convolution runs on the
hand-written fp32-MFMA kernels of ``libdvf_hip.so`` with bias + ReLU / alpha*sigmoid+beta fused in the
epilogue; ``torch.cat`` is replaced by virtual concatenation inside the consuming kernel, ``crop_like`` by
never computing the cropped pixels, and the disparity up-sampling by a dedicated kernel.  Every ReLU layer's output is
consumed by convolutions only (the next layer, a skip into an iconv, a predict_disp head), so all of them are declared
``fuse_bwd``: the consumers' dgrad kernels apply relu' and sum the bias gradient (dvf/conv.py::ReluTag).
"""
import torch
import torch.nn as nn

from dvf import lib as _L
from dvf.conv import FusedAct, FusedConv2d, FusedConvTranspose2d, Upsample2xFn, xavier_init_

_ENC_PLANES = (32, 64, 128, 256, 512, 512, 512)     # reference DispNetS.py:50
_ENC_KERNEL = (7, 5, 3, 3, 3, 3, 3)                  # :51-57
_DEC_PLANES = (512, 512, 256, 128, 64, 32, 16)       # :59


def _pair(seq, *mods):
    """nn.Sequential whose odd slots are activation placeholders: keeps the reference's key numbering."""
    layers = []
    for m in mods:
        layers += [m, FusedAct()]
    return nn.Sequential(*layers)


class DispNetS(nn.Module):

    def __init__(self, alpha=10, beta=0.01):
        super(DispNetS, self).__init__()
        self.alpha = alpha
        self.beta = beta
        cin = 3
        for i, (co, k) in enumerate(zip(_ENC_PLANES, _ENC_KERNEL), start=1):
            p = (k - 1) // 2
            setattr(self, f"conv{i}", _pair(None, FusedConv2d(cin, co, k, 2, p, _L.ACT_RELU, fuse_bwd=True),
                                            FusedConv2d(co, co, k, 1, p, _L.ACT_RELU, fuse_bwd=True)))
            cin = co
        up_in = (_ENC_PLANES[6],) + _DEC_PLANES[:-1]
        skip = (_ENC_PLANES[5], _ENC_PLANES[4], _ENC_PLANES[3], _ENC_PLANES[2], 1 + _ENC_PLANES[1], 1 + _ENC_PLANES[0], 1)
        for j, lvl in enumerate(range(7, 0, -1)):
            setattr(self, f"upconv{lvl}", _pair(None, FusedConvTranspose2d(up_in[j], _DEC_PLANES[j], 3, 2, 1, _L.ACT_RELU,
                                                                         output_padding=1, fuse_bwd=True)))
            setattr(self, f"iconv{lvl}", _pair(None, FusedConv2d(_DEC_PLANES[j] + skip[j], _DEC_PLANES[j], 3, 1, 1,
                                                                _L.ACT_RELU, fuse_bwd=True)))
        for lvl, ci in zip((4, 3, 2, 1), _DEC_PLANES[3:]):
            setattr(self, f"predict_disp{lvl}", _pair(None, FusedConv2d(ci, 1, 3, 1, 1, _L.ACT_SIGMOID_AFFINE,
                                                                       alpha=alpha, beta=beta)))

    def init_weights(self):
        xavier_init_(self)

    def forward(self, x):
        feats = []
        h = x
        for i in range(1, 8):
            block = getattr(self, f"conv{i}")
            h = block[2](block[0](h))
            feats.append(h)
        skips = [feats[5], feats[4], feats[3], feats[2], feats[1], feats[0], x]
        disps = {}
        up_disp = None
        h = feats[6]
        for j, lvl in enumerate(range(7, 0, -1)):
            ref_hw = (skips[j].size(2), skips[j].size(3))
            up = getattr(self, f"upconv{lvl}")[0](h, out_hw=ref_hw)          # transposed conv + ReLU + crop_like
            parts = [up] if lvl == 1 else [up, skips[j]]
            if up_disp is not None:
                parts.append(up_disp)
            h = getattr(self, f"iconv{lvl}")[0](*parts)                       # virtual concat
            if lvl <= 4:
                d = getattr(self, f"predict_disp{lvl}")[0](h)                                                   # alpha * sigmoid(conv) + beta, fused
                disps[lvl] = d
                if lvl > 1:
                    nxt = skips[j + 1]
                    up_disp = Upsample2xFn.apply(d, (nxt.size(2), nxt.size(3)))
        return [disps[1], disps[2], disps[3], disps[4]]
