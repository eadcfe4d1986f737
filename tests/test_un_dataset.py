"""un_dataset.dataset on a synthetic KITTI-shaped directory.  The reference loader needs `path` and the
removed scipy.misc API and cannot be imported: its semantics (incl. the float -> bytescale quirk of imresize) are
restated; parity unpinned."""
import os

import numpy as np
import torch
from PIL import Image

import un_dataset


def _make_tree(tmp_path, n=3):
    root = tmp_path / "kitti_eigen"
    for d in ("intrinsics", "train_K", "train_T_R2L"):
        (root / d).mkdir(parents=True)
    rng = np.random.default_rng(5)
    lines = []
    for drive in ("2011_09_26_drive_0001_sync", un_dataset.SKIPPED_DRIVE):
        for cam in ("image_02", "image_03"):
            (tmp_path / "raw" / "2011_09_26" / drive / cam / "data").mkdir(parents=True)
        K = np.array([[700.0, 0, 600], [0, 710.0, 180], [0, 0, 1]])
        np.savetxt(root / "intrinsics" / (drive + "_cam.txt"), K.reshape(1, 9))
        for i in range(n):
            paths = []
            for cam, j in (("image_02", i), ("image_02", i + 1), ("image_03", i), ("image_03", i + 1)):
                p = tmp_path / "raw" / "2011_09_26" / drive / cam / "data" / f"{j:010d}.png"
                if not p.exists():
                    img = rng.integers(20, 200, size=(37, 123, 3), dtype=np.uint8)      # min > 0, max < 255: bytescale acts
                    Image.fromarray(img).save(p)
                paths.append(str(p))
            kid = f"{drive[-9:-5]}_{i}"
            np.save(root / "train_K" / (kid + ".npy"), K.astype(np.float64))
            # the reference's builder writes the stereo pose as an se(3) vector (w, u) = (0, 0, 0, Tx, 0, 0)
            # (data/dataset_builder.py:155)
            np.save(root / "train_T_R2L" / (kid + ".npy"), np.array([[0, 0, 0, -0.54, 0, 0]]))
            lines.append(" ".join(paths + [kid, kid]))
    (root / "train.txt").write_text("\n".join(lines) + "\n")
    return root


def test_dataset_samples_and_resize_semantics(tmp_path):
    root = _make_tree(tmp_path)
    ds = un_dataset.dataset(img_height=16, img_width=48, root=str(root))
    assert len(ds) == 3                                      # the static drive is skipped
    r1, l2, r2, K, Kinv, raw_K, T = ds[1]
    for t in (r1, l2, r2):
        assert t.dtype == torch.float32 and tuple(t.shape) == (3, 16, 48)
        assert float(t.min()) >= 0 and float(t.max()) <= 255
    assert torch.allclose(K @ Kinv, torch.eye(3), atol=1e-3)      # (fp32 inverse, as in the reference)
    assert tuple(raw_K.shape) == (3, 3) and tuple(T.shape) == (1, 6) and float(T[0, 3]) == np.float32(-0.54)
    # the resize is PIL's bilinear on the byte-scaled image
    src = un_dataset.imread(ds.samples[1]["right_1"]).astype(np.float32)
    lo, hi = src.min(), src.max()
    manual = np.asarray(Image.fromarray(((src - lo) * (255.0 / (hi - lo)) + 0.5).clip(0, 255.5).astype(np.uint8))
                        .resize((48, 16), resample=Image.BILINEAR)).astype(np.float32)
    assert np.array_equal(r1.numpy(), manual.transpose(2, 0, 1))
    # uint8 input is not rescaled
    assert np.array_equal(un_dataset.bytescale(np.arange(10, dtype=np.uint8)), np.arange(10, dtype=np.uint8))


def test_loader_and_batch_dict(tmp_path):
    root = _make_tree(tmp_path)
    ds = un_dataset.dataset(img_height=16, img_width=48, root=str(root))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)
    batch = un_dataset.to_batch(next(iter(loader)), "cpu")
    assert tuple(batch["img_R2"].shape) == (2, 3, 16, 48) and tuple(batch["K"].shape) == (2, 3, 3)
    assert tuple(batch["T_R2L"].shape) == (2, 6) and batch["img_L2"].is_contiguous()
    # the stereo baseline reaches each API in ITS convention: se(3) (w, u) as stored for unsupervise_dvo.py,
    # (t, r-euler) for pose_vec2mat (unsupervise.py / train.py) -- a 0.54 m translation along x in both, no rotation
    assert torch.equal(batch["T_R2L_se3"], torch.tensor([[0, 0, 0, -0.54, 0, 0]] * 2))
    assert torch.allclose(batch["T_R2L"], torch.tensor([[-0.54, 0, 0, 0, 0, 0]] * 2), atol=1e-7)
    # --reference-stereo-pose: the file vector reaches pose_vec2mat unchanged, as in the reference (unsupervise.py:101)
    ref_mode = un_dataset.to_batch(next(iter(loader)), "cpu", reference_stereo_pose=True)
    assert torch.equal(ref_mode["T_R2L"], torch.tensor([[0, 0, 0, -0.54, 0, 0]] * 2))
    assert torch.equal(ref_mode["T_R2L_se3"], ref_mode["T_R2L"])


def test_se3_to_tr_euler_is_the_same_rigid_motion():
    """General case (w != 0): exp-map pose [R | R u] (se3_generate.py:13-47) == pose_vec2mat of the converted vector."""
    from oracle import geometry as og
    g = torch.Generator().manual_seed(5)
    se3 = torch.randn(6, 6, generator=g) * 0.3
    se3[0, :3] = 0
    tr = un_dataset.se3_to_tr_euler(se3)
    want = og.se3_exp(se3.double())                          # [n, 3, 4]
    got = og.pose_vec2mat(tr.double(), "euler")
    assert torch.allclose(got, want, atol=1e-6)
    # and the synthetic stream carries the same pose under both keys
    from dvf.synthetic import synthetic_batch
    b = synthetic_batch(2, 8, 16)
    assert torch.allclose(un_dataset.se3_to_tr_euler(b["T_R2L_se3"]), b["T_R2L"], atol=1e-7)


def test_pil_resample_tables_reproduce_pil():
    """The fixed-point coefficient tables of dvf.image_ops (Pillow's precompute_coeffs / normalize_coeffs_8bpc) drive an
    integer emulation of the two resampling passes that must equal PIL's own BILINEAR resize exactly -- down- and
    up-scaling.  (The GPU kernels run the same arithmetic: tests/test_gpu_image_ops.py.)"""
    from dvf.image_ops import _coeffs
    rng = np.random.default_rng(0)
    for (ih, iw), (oh, ow) in (((94, 311), (64, 208)), ((37, 123), (64, 128))):
        img = rng.integers(3, 250, size=(ih, iw, 3), dtype=np.uint8)
        ref = un_dataset.imresize(img.astype(np.float32), (oh, ow)).astype(np.int64)
        bs = un_dataset.bytescale(img.astype(np.float32)).astype(np.int64)
        hb, hk, _ = _coeffs(iw, ow)
        vb, vk, _ = _coeffs(ih, oh)
        tmp = np.zeros((ih, ow, 3), dtype=np.int64)
        for ox in range(ow):
            x0, n = hb[ox]
            tmp[:, ox, :] = np.clip(((1 << 21) + np.tensordot(bs[:, x0:x0 + n, :], hk[ox, :n].astype(np.int64), axes=([1], [0]))) >> 22, 0, 255)
        out = np.zeros((oh, ow, 3), dtype=np.int64)
        for oy in range(oh):
            y0, n = vb[oy]
            out[oy] = np.clip(((1 << 21) + np.tensordot(tmp[y0:y0 + n], vk[oy, :n].astype(np.int64), axes=([0], [0]))) >> 22, 0, 255)
        assert np.array_equal(out, ref)


def test_raw_mode_and_collate(tmp_path):
    root = _make_tree(tmp_path)
    ds = un_dataset.dataset(img_height=16, img_width=48, root=str(root), raw=True)
    r1, l2, r2, K, Kinv, raw_K, T = ds[0]
    assert r1.dtype == torch.uint8 and tuple(r1.shape) == (37, 123, 3) and tuple(K.shape) == (3, 3)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0, collate_fn=un_dataset.collate_raw)
    b = next(iter(loader))
    assert isinstance(b[0], list) and len(b[0]) == 2 and tuple(b[3].shape) == (2, 3, 3) and tuple(b[6].shape) == (2, 1, 6)
