"""Drop-in for the reference's ``pytorch_version/dataset.py`` (``pose_framework_KITTI``): consecutive-frame pairs of
KITTI odometry sequences with the ground-truth RELATIVE pose of the pair, used by ``validate()`` (train.py:220-247).

Same class name, constructor arguments and sample layout as the reference (dataset.py:11-77): a 6-channel float tensor
(channels 0-2 = the LATER frame, 3-5 = the earlier one, both resized to (img_height, img_width)) and the 4x4 transform
inv(T_{i}) @ T_{i+1} built from ``poses/<seq>.txt``.  The reference reads files with the removed scipy.misc functions
and the ``path`` package; this build uses PIL with the same byte-scale + bilinear semantics as ``un_dataset.py``
(parity unpinned: no reference fixture exists for image decoding)."""
import glob
import os
import random

import numpy as np
import torch
import torch.utils.data as data

from un_dataset import imread, imresize


def read_scene_data(data_root, sequence_set, step=1):
    """(image file lists, [n,3,4] pose arrays) of the sequences whose directory name matches an entry of
    ``sequence_set`` under ``<root>/sequences`` (reference dataset.py:80-97)."""
    im_sequences, poses_sequences = [], []
    seq_dirs = set()
    for seq in sequence_set:
        seq_dirs |= set(d for d in glob.glob(os.path.join(data_root, "sequences", seq)) if os.path.isdir(d))
    for d in sorted(seq_dirs):
        name = os.path.basename(d)
        poses = np.genfromtxt(os.path.join(data_root, "poses", f"{name}.txt")).astype(np.float32).reshape(-1, 3, 4)
        imgs = sorted(glob.glob(os.path.join(d, "image_2", "*.png")))
        im_sequences.append(imgs)
        poses_sequences.append(poses)
    return im_sequences, poses_sequences


class pose_framework_KITTI(data.Dataset):
    def __init__(self, root, sequence_set, step=1, transform=None, seed=2018, img_height=160, img_width=608, shuffle=True):
        np.random.seed(seed)
        random.seed(seed)
        self.shuffle = shuffle
        self.root, self.transform = root, transform
        self.img_files, self.poses = read_scene_data(self.root, sequence_set, step)
        self.sequence_num = len(self.poses)
        self.height, self.width = img_height, img_width
        self.generator()

    def generator(self):
        samples = []
        self.gt_se3 = []
        for pose_list in self.poses:                                  # relative poses inv(T_{i-1}) @ T_i  (:22-33)
            prev, rel = np.eye(4), []
            for idx in range(pose_list.shape[0]):
                cur = np.eye(4)
                cur[:3] = pose_list[idx]
                rel.append(np.linalg.inv(prev).dot(cur))
                prev = cur
            self.gt_se3.append(rel)
        for img_list, pose_list in zip(self.img_files, self.gt_se3):
            for i in range(len(img_list) - 1):
                samples.append({"imgs": [img_list[i], img_list[i + 1]], "pose": pose_list[i + 1]})
        if self.shuffle:
            random.shuffle(samples)
        self.samples = samples

    def __getitem__(self, index):
        sample = self.samples[index]
        imgs = [imread(p).astype(np.float32) for p in sample["imgs"]]
        imgs = [imresize(im, (self.height, self.width)).astype(np.float32) for im in imgs]
        if self.transform is not None:
            imgs = [t.numpy() for t in self.transform(imgs)]
        else:
            imgs = [np.transpose(im, (2, 0, 1)) for im in imgs]
        img_data = np.zeros((6, self.height, self.width), dtype=np.float32)
        img_data[:3] = imgs[1]                                        # later frame first (:60-61, :74-75)
        img_data[3:] = imgs[0]
        return torch.from_numpy(img_data).type(torch.FloatTensor), torch.from_numpy(sample["pose"]).type(torch.FloatTensor)

    def __len__(self):
        return len(self.samples)
