"""Stereo-sequence dataset for unsupervise.py / unsupervise_dvo.py -- the reference's ``pytorch_version/un_dataset.py``
without its dead dependencies (``path``, and ``scipy.misc.imread/imresize`` which SciPy removed): PIL does the decoding
and the resize that scipy.misc delegated to it.

Same sample definition and return tuple (un_dataset.py:43-84): a line of the listing names left_1 left_2 right_1 right_2
and the ids of the raw K and of T_R2L; __getitem__ returns float32 tensors
    (img_R1, img_L2, img_R2 [3,H,W] in 0..255, intrinsics [3,3], inv(intrinsics) [3,3], raw_K, T_R2L)
The reference hard-codes /home/gaof/... (un_dataset.py:21-22); here the root and listing are arguments, with the same
layout underneath: <root>/train.txt, <root>/intrinsics/<drive>_cam.txt, <root>/train_K/<id>.npy, <root>/train_T_R2L/<id>.npy.

scipy.misc.imresize quirk kept on purpose: it was called on the float32 image, and for a non-uint8 array scipy's
``toimage`` first rescaled [min, max] of the image to [0, 255] (``bytescale``) before the bilinear resize to uint8."""
import os
import random

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

SKIPPED_DRIVE = "2011_09_26_drive_0060_sync"        # un_dataset.py:28 (a static scene)


def imread(path):
    """HxWx3 uint8, like scipy.misc.imread on an RGB file."""
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def bytescale(arr):
    """scipy.misc.bytescale with its defaults (cmin=min, cmax=max, low=0, high=255) as imresize applied it."""
    arr = np.asarray(arr)
    if arr.dtype == np.uint8:
        return arr
    cmin, cmax = float(arr.min()), float(arr.max())
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1.0
    scaled = (arr - cmin) * (255.0 / cscale)
    return (scaled.clip(0, 255) + 0.5).astype(np.uint8)


def imresize(arr, size):
    """scipy.misc.imresize(arr, (H, W)) with its default interp='bilinear': PIL resize of the byte-scaled image."""
    im = Image.fromarray(bytescale(arr))
    return np.asarray(im.resize((int(size[1]), int(size[0])), resample=Image.BILINEAR))


class dataset(data.Dataset):
    def __init__(self, transform=None, seed=9999, img_height=160, img_width=608, shuffle=True,
                 root="./data/kitti_eigen", listing=None, raw=False):
        """raw=True (this build's extension): __getitem__ returns the three frames as DECODED uint8 [H0,W0,3] tensors
        instead of resized float ones; ``to_batch_raw`` then performs the reference's imresize on the GPU
        (dvf.image_ops.gpu_imresize, bit-identical to the host path) -- use ``collate_raw`` in the DataLoader."""
        np.random.seed(seed)
        random.seed(seed)
        self.shuffle = shuffle
        self.transform = transform
        self.raw = raw
        self.height, self.width = img_height, img_width
        self.root = root
        self.listing = listing if listing is not None else os.path.join(root, "train.txt")
        self.generator()

    def generator(self):
        self.samples = []
        with open(self.listing) as f:
            for line in f:
                parts = line.split()
                if len(parts) < 6:
                    continue
                l1, l2, r1, r2, k, t = parts[:6]
                drive = l1.split("/")[-4]
                if drive == SKIPPED_DRIVE:
                    continue
                self.samples.append({
                    "left_1": l1, "left_2": l2, "right_1": r1, "right_2": r2,
                    "intrinsics": os.path.join(self.root, "intrinsics", drive + "_cam.txt"),
                    "raw_K": os.path.join(self.root, "train_K", k + ".npy"),
                    "T_R2L": os.path.join(self.root, "train_T_R2L", t + ".npy"),
                })

    def __getitem__(self, index):
        s = self.samples[index]
        if self.raw:
            frames = [torch.from_numpy(np.ascontiguousarray(imread(s[k]))) for k in ("right_1", "left_2", "right_2")]
            intrinsics = np.genfromtxt(s["intrinsics"]).astype(np.float32).reshape((3, 3))
            f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).type(torch.FloatTensor)
            return (frames[0], frames[1], frames[2], f32(intrinsics), f32(np.linalg.inv(intrinsics)),
                    f32(np.load(s["raw_K"], allow_pickle=False).astype(np.float32)),
                    f32(np.load(s["T_R2L"], allow_pickle=False).astype(np.float32)))
        imgs = [imread(s[k]).astype(np.float32) for k in ("right_1", "left_2", "right_2")]
        imgs = [imresize(im, (self.height, self.width)).astype(np.float32) for im in imgs]
        intrinsics = np.genfromtxt(s["intrinsics"]).astype(np.float32).reshape((3, 3))
        raw_K = np.load(s["raw_K"], allow_pickle=False).astype(np.float32)
        T_R2L = np.load(s["T_R2L"], allow_pickle=False).astype(np.float32)
        if self.transform is not None:
            imgs = [self.transform(im).numpy() for im in imgs]
        else:
            imgs = [np.transpose(im, (2, 0, 1)) for im in imgs]
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).type(torch.FloatTensor)
        return (f32(imgs[0]), f32(imgs[1]), f32(imgs[2]), f32(intrinsics), f32(np.linalg.inv(intrinsics)), f32(raw_K),
                f32(T_R2L))

    def __len__(self):
        return len(self.samples)


def se3_to_tr_euler(se3):
    """Stereo pose file convention -> pose_vec2mat convention.  The reference's dataset stores T_R2L as the se(3)
    vector (wx, wy, wz, ux, uy, uz) = (0, 0, 0, Tx, 0, 0) (data/dataset_builder.py:155; consumed as such by the
    Caffe-style chain, unsupervise_dvo.py:98-100 -> se3_generate.py: R = exp([w]x), t = R u).  inverse_warp.py's
    pose_vec2mat wants (tx, ty, tz, rx, ry, rz) with R = Rx Ry Rz (inverse_warp.py:77-114, :141-157).  Exact
    conversion through the rotation matrix (for the rectified stereo rig w = 0, so this is t = u, r = 0)."""
    w, u = se3[:, :3].double(), se3[:, 3:6].double()
    th = w.norm(dim=1, keepdim=True)
    k = torch.where(th > 1e-12, w / th.clamp_min(1e-300), torch.zeros_like(w))
    K = torch.zeros(se3.shape[0], 3, 3, dtype=torch.float64, device=se3.device)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    I = torch.eye(3, dtype=torch.float64, device=se3.device).expand_as(K)
    s, c = torch.sin(th)[..., None], torch.cos(th)[..., None]
    R = I + s * K + (1 - c) * (K @ K)                                   # Rodrigues
    t = (R @ u[..., None])[..., 0]
    ry = torch.asin(R[:, 0, 2].clamp(-1, 1))                            # R = Rx Ry Rz: R02 = sin(ry)
    rz = torch.atan2(-R[:, 0, 1], R[:, 0, 0])
    rx = torch.atan2(-R[:, 1, 2], R[:, 2, 2])
    return torch.cat((t, torch.stack((rx, ry, rz), dim=1)), dim=1).to(se3.dtype)


def to_batch(sample_batch, device, reference_stereo_pose=False):
    """Collated dataset output -> the batch dict of dvf/steps.py (device tensors).  The stereo pose is provided in BOTH
    conventions: ``T_R2L_se3`` as the files hold it (w, u) for unsupervise_dvo.py, and ``T_R2L`` for unsupervise.py /
    train.py, whose pose_vec2mat wants (t, r-euler).

    DELIBERATE DEPARTURE FROM THE REFERENCE (default): the reference feeds the file vector (0, 0, 0, Tx, 0, 0) UNCHANGED
    into pose_vec2mat (unsupervise.py:101) and into the stereo-pose target F.mse_loss(pose[:,1], T_R2L) (train.py:184,201),
    where it means "no translation, a rotation of Tx radians about x" (SURVEY preamble #6) -- not the stereo rig.  By
    default this build converts the vector to (Tx, 0, 0, 0, 0, 0); ``reference_stereo_pose=True`` (the entry scripts'
    ``--reference-stereo-pose``) passes the file vector through exactly as the reference does, for runs that must reproduce
    the reference's numbers on real data (pinned by tests/golden/photo_c3_32x104_rawpose.npz)."""
    r1, l2, r2, K, Kinv, raw_K, T = [x.to(device, non_blocking=True) for x in sample_batch]
    se3 = T.reshape(T.shape[0], -1)[:, :6].contiguous()
    tr = se3 if reference_stereo_pose else se3_to_tr_euler(se3).contiguous()
    return {"img_R1": r1.contiguous(), "img_L2": l2.contiguous(), "img_R2": r2.contiguous(), "K": K.contiguous(),
            "Kinv": Kinv.contiguous(), "raw_K": raw_K, "T_R2L_se3": se3, "T_R2L": tr}


def collate_raw(samples):
    """DataLoader collate_fn for ``dataset(raw=True)``: frames stay a list (native sizes may differ between drives),
    the small tensors are stacked."""
    cols = list(zip(*samples))
    return [list(cols[0]), list(cols[1]), list(cols[2])] + [torch.stack(c) for c in cols[3:]]


def to_batch_raw(raw_batch, device, size, reference_stereo_pose=False):
    """``collate_raw`` output -> the batch dict, with the reference's per-frame imresize done on the GPU."""
    from dvf.image_ops import gpu_imresize
    frames = [torch.stack([gpu_imresize(f.to(device, non_blocking=True), size) for f in col]) for col in raw_batch[:3]]
    return to_batch(frames + list(raw_batch[3:]), device, reference_stereo_pose)
