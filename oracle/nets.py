"""Oracle (test infrastructure): the three networks as pure functions of a state dict.

The layer tables restate the reference constructors; ``state_dict`` key names and tensor
shapes are the reference's (``conv1.0.weight``, ``upconv7.0.weight``, ``predict_disp1.0.bias``,
``pose_pred.weight`` ...), so a dict produced here loads into the reference modules and
vice versa.  Arithmetic is aten CPU conv / conv_transpose / interpolate, i.e. the same
third-party ops the reference calls.

  DispNetS        pytorch_version/DispNetS.py:42-132
  PoseExpNet_sfm  pytorch_version/PoseExpNet_sfm.py:20-95     (forward(target, ref_imgs))
  PoseExpNet      pytorch_version/PoseExpNet.py:20-91         (forward(imgs[B,6,H,W]))
  FeatExtractor   pytorch_version/feat_extractor.py:13-83
"""
import math
import zlib

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- layer tables
# (name, kind, cin, cout, k, stride, pad, output_padding); kind 'c' = Conv2d, 't' = ConvTranspose2d

_DISP_ENC = [32, 64, 128, 256, 512, 512, 512]          # DispNetS.py:50
_DISP_DEC = [512, 512, 256, 128, 64, 32, 16]           # DispNetS.py:59


def dispnet_layers():
    L = []
    cin = 3
    for i, (co, k) in enumerate(zip(_DISP_ENC, [7, 5, 3, 3, 3, 3, 3])):      # DispNetS.py:7-13,51-57
        L.append((f"conv{i+1}.0", "c", cin, co, k, 2, (k - 1) // 2, 0))
        L.append((f"conv{i+1}.2", "c", co, co, k, 1, (k - 1) // 2, 0))
        cin = co
    up_in = [512] + _DISP_DEC[:-1]
    for i, (ci, co) in enumerate(zip(up_in, _DISP_DEC)):                      # DispNetS.py:30-34,60-66
        L.append((f"upconv{7-i}.0", "t", ci, co, 3, 2, 1, 1))
    skip = [512, 512, 256, 128, 64 + 1, 32 + 1, 1]                            # DispNetS.py:68-74
    for i, co in enumerate(_DISP_DEC):
        L.append((f"iconv{7-i}.0", "c", co + skip[i], co, 3, 1, 1, 0))
    for lvl, ci in zip([4, 3, 2, 1], _DISP_DEC[3:]):                           # DispNetS.py:16-20,76-79
        L.append((f"predict_disp{lvl}.0", "c", ci, 1, 3, 1, 1, 0))
    return L


_POSE_ENC = [16, 32, 64, 128, 256, 256, 256]           # PoseExpNet_sfm.py:27 ; PoseExpNet.py:27
_POSE_DEC = [256, 128, 64, 32, 16]                     # PoseExpNet_sfm.py:39 ; PoseExpNet.py:39


def posenet_layers(cin0, n_pose_out, nb_masks, output_exp=True):
    L = []
    cin = cin0
    for i, (co, k) in enumerate(zip(_POSE_ENC, [7, 5, 3, 3, 3, 3, 3])):
        L.append((f"conv{i+1}.0", "c", cin, co, k, 2, (k - 1) // 2, 0))
        cin = co
    L.append(("pose_pred", "c", 256, n_pose_out, 1, 1, 0, 0))
    if output_exp:
        up_in = [256] + _POSE_DEC[:-1]
        for i, (ci, co) in enumerate(zip(up_in, _POSE_DEC)):
            L.append((f"upconv{5-i}.0", "t", ci, co, 4, 2, 1, 0))
        for lvl, ci in zip([4, 3, 2, 1], _POSE_DEC[1:]):
            L.append((f"predict_mask{lvl}", "c", ci, nb_masks, 3, 1, 1, 0))
    return L


def featnet_layers():
    L = []
    for lvl in (5, 4, 3, 2, 1):                                               # feat_extractor.py:18-36
        cin, s = (3, 1) if lvl == 5 else (35, 2)
        L.append((f"conv_1_b{lvl}.0", "c", cin, 32, 3, s, 1, 0))
        L.append((f"conv_2_b{lvl}.0", "c", 32, 32, 3, 1, 1, 0))
        L.append((f"conv_3_b{lvl}", "c", 32, 32, 3, 1, 1, 0))
    for n in ("conv_3_b1_up", "conv_3_bb2_up", "conv_3_bb3_up", "conv_3_bb4_up"):   # :38-41
        L.append((n, "dw", 32, 32, 4, 2, 1, 0))
    return L


def param_shapes(layers):
    shapes = {}
    for name, kind, ci, co, k, s, p, op in layers:
        if kind == "c":
            shapes[name + ".weight"] = (co, ci, k, k)
        elif kind == "t":
            shapes[name + ".weight"] = (ci, co, k, k)
        else:  # depthwise transposed, groups = channels
            shapes[name + ".weight"] = (ci, 1, k, k)
        shapes[name + ".bias"] = (co,)
    return shapes


def fill_params(layers, seed=0, dtype=torch.float32, bias_scale=0.05):
    """Deterministic per-parameter filler keyed by the parameter NAME (crc32 ^ seed): Xavier-uniform
    range for weights (same bound as init_weights, DispNetS.py:81-86) and small NON-zero biases so
    the bias paths are exercised.  Build-side recipe, independent of torch's global RNG stream."""
    out = {}
    for key, shp in param_shapes(layers).items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if key.endswith(".weight"):
            rf = shp[2] * shp[3]
            fan_in, fan_out = shp[1] * rf, shp[0] * rf
            a = math.sqrt(6.0 / (fan_in + fan_out))
            out[key] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * a).to(dtype)
        else:
            out[key] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bias_scale).to(dtype)
    return out


# ----------------------------------------------------------------------------- forwards

def _c(sd, n, x, stride=1, pad=1):
    return F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], stride=stride, padding=pad)


def _t(sd, n, x, stride, pad, opad, groups=1):
    return F.conv_transpose2d(x, sd[n + ".weight"], sd[n + ".bias"], stride=stride, padding=pad,
                              output_padding=opad, groups=groups)


def _crop(x, ref):
    return x[:, :, :ref.shape[2], :ref.shape[3]]


def dispnet_forward(sd, x, alpha=10.0, beta=0.01):
    """DispNetS.forward -> [disp1, disp2, disp3, disp4].  DispNetS.py:88-132."""
    enc = []
    h = x
    for i, k in enumerate([7, 5, 3, 3, 3, 3, 3]):
        p = (k - 1) // 2
        h = F.relu(_c(sd, f"conv{i+1}.0", h, 2, p))
        h = F.relu(_c(sd, f"conv{i+1}.2", h, 1, p))
        enc.append(h)
    skips = [enc[5], enc[4], enc[3], enc[2], enc[1], enc[0], x]
    disps = {}
    up_disp = None
    h = enc[6]
    for i in range(7):
        lvl = 7 - i
        ref = skips[i]
        u = _crop(F.relu(_t(sd, f"upconv{lvl}.0", h, 2, 1, 1)), ref)
        parts = [u] if lvl == 1 else [u, ref]
        if up_disp is not None:
            parts.append(_crop(up_disp, ref))
        h = F.relu(_c(sd, f"iconv{lvl}.0", torch.cat(parts, 1)))
        if lvl <= 4:
            d = alpha * torch.sigmoid(_c(sd, f"predict_disp{lvl}.0", h)) + beta
            disps[lvl] = d
            up_disp = F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=False)
    return [disps[1], disps[2], disps[3], disps[4]]


def posenet_forward(sd, x, nb_ref_imgs, output_exp=True, sfm=True, training=True):
    """x = cat(target, refs) on channels (sfm, PoseExpNet_sfm.py:58-62) or imgs[B,6,H,W] (PoseExpNet.py:58).
    Returns (masks, pose): pose [B,nb_ref,6] (sfm) / [B,6]; 4 masks in train mode, mask1 in eval."""
    outs = []
    h = x
    for i, k in enumerate([7, 5, 3, 3, 3, 3, 3]):
        h = F.relu(_c(sd, f"conv{i+1}.0", h, 2, (k - 1) // 2))
        outs.append(h)
    pose = F.conv2d(h, sd["pose_pred.weight"], sd["pose_pred.bias"]).mean(3).mean(2)
    pose = 0.01 * (pose.view(pose.shape[0], nb_ref_imgs, 6) if sfm else pose.view(pose.shape[0], 6))
    masks = [None] * 4
    if output_exp:
        refs = [outs[3], outs[2], outs[1], outs[0], x]
        h = outs[4]
        ups = []
        for i in range(5):
            h = _crop(F.relu(_t(sd, f"upconv{5-i}.0", h, 2, 1, 0)), refs[i])
            ups.append(h)
        for lvl in (4, 3, 2, 1):
            masks[lvl - 1] = torch.sigmoid(_c(sd, f"predict_mask{lvl}", ups[5 - lvl]))
    return (masks, pose) if training else (masks[0], pose)


def featnet_forward(sd, imgs):
    """FeatExtractor.forward -> [N,32,H,W].  feat_extractor.py:43-83."""
    pyr = [imgs]
    for _ in range(3):                                                         # :44-46
        pyr.append(F.interpolate(pyr[-1], scale_factor=0.5, mode="bilinear"))
    c3 = {}
    x = imgs
    for j, lvl in enumerate((5, 4, 3, 2, 1)):
        s = 1 if lvl == 5 else 2
        h = F.relu(_c(sd, f"conv_1_b{lvl}.0", x, s, 1))
        h = F.relu(_c(sd, f"conv_2_b{lvl}.0", h, 1, 1))
        c3[lvl] = _c(sd, f"conv_3_b{lvl}", h, 1, 1)
        if lvl > 1:
            x = torch.cat((pyr[j], c3[lvl]), dim=1)
    h = c3[1]
    for name, lvl in (("conv_3_b1_up", 2), ("conv_3_bb2_up", 3), ("conv_3_bb3_up", 4), ("conv_3_bb4_up", 5)):
        h = c3[lvl] + _t(sd, name, h, 2, 1, 0, groups=32)                     # :72-82
    return h
