"""Cityscapes video-clip I/O: the data format on the input side of the hot path.

Restates the reference's clip layout (semantic_segmentation/lib/datasets/cityscapes_vid.py:104-202):

    <root>/leftImg8bit/<split>/<city>/<city>_<seq>_<frame>_leftImg8bit.png            the labelled frame of each clip
    <root>/leftImg8bit_sequence/<split>/<city>/<city>_<seq>_<frame - i>_leftImg8bit.png   its preceding frames, i = 1 .. clip_length - 1
    <root>/gtFine/<split>/<city>/<city>_<seq>_<frame>_gtFine_labelIds.png             label ids of the labelled frame

A sample is ``(frames, target, meta)``: ``frames`` = clip_length images in CHRONOLOGICAL order (the labelled frame is the
LAST one, :196), ``target`` = the label map re-coded from Cityscapes ids to the 19 train ids (255 = ignore, :162-172),
``meta = {'relpath': '<city>/<file>'}``.  Normalisation constants are the dataset statistics the reference trains with
(:29-30).  No dataset exists offline, so ``write_synthetic_tree`` produces a miniature tree with the same layout for tests
and demos."""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

MEAN = (73.1584 / 255, 82.9090 / 255, 72.3924 / 255)   # cityscapes_vid.py:29
STD = (44.9149 / 255, 46.1529 / 255, 45.3192 / 255)    # cityscapes_vid.py:30

# (name, id, train_id, colour) of the Cityscapes label set (github.com/mcordts/cityscapesScripts, as used at :35-71)
LABELS = (
    ("unlabeled", 0, 255, (0, 0, 0)), ("ego vehicle", 1, 255, (0, 0, 0)), ("rectification border", 2, 255, (0, 0, 0)),
    ("out of roi", 3, 255, (0, 0, 0)), ("static", 4, 255, (0, 0, 0)), ("dynamic", 5, 255, (111, 74, 0)), ("ground", 6, 255, (81, 0, 81)),
    ("road", 7, 0, (128, 64, 128)), ("sidewalk", 8, 1, (244, 35, 232)), ("parking", 9, 255, (250, 170, 160)),
    ("rail track", 10, 255, (230, 150, 140)), ("building", 11, 2, (70, 70, 70)), ("wall", 12, 3, (102, 102, 156)),
    ("fence", 13, 4, (190, 153, 153)), ("guard rail", 14, 255, (180, 165, 180)), ("bridge", 15, 255, (150, 100, 100)),
    ("tunnel", 16, 255, (150, 120, 90)), ("pole", 17, 5, (153, 153, 153)), ("polegroup", 18, 255, (153, 153, 153)),
    ("traffic light", 19, 6, (250, 170, 30)), ("traffic sign", 20, 7, (220, 220, 0)), ("vegetation", 21, 8, (107, 142, 35)),
    ("terrain", 22, 9, (152, 251, 152)), ("sky", 23, 10, (70, 130, 180)), ("person", 24, 11, (220, 20, 60)), ("rider", 25, 12, (255, 0, 0)),
    ("car", 26, 13, (0, 0, 142)), ("truck", 27, 14, (0, 0, 70)), ("bus", 28, 15, (0, 60, 100)), ("caravan", 29, 255, (0, 0, 90)),
    ("trailer", 30, 255, (0, 0, 110)), ("train", 31, 16, (0, 80, 100)), ("motorcycle", 32, 17, (0, 0, 230)), ("bicycle", 33, 18, (119, 11, 32)),
    ("license plate", -1, 255, (0, 0, 142)),
)
ID_TO_TRAIN_ID = np.array([l[2] for l in LABELS])                                   # index = label id (-1 wraps to the last row, as in the reference)
TRAIN_ID_TO_ID = np.array([l[1] for l in LABELS if l[2] < 255])
TRAIN_ID_TO_COLOR = np.array([l[3] for l in LABELS if l[2] not in (-1, 255)] + [(0, 0, 0)])
FINE_CLASSES = [6, 7, 11, 12, 13, 14, 15, 16, 17, 18]                               # thin / small classes reported as "Fine mIoU" (:73)
NUM_CLASSES = 19


def train_id_names() -> List[str]:
    """Comma-joined label names per train id (index 19 = everything ignored); keys of the per-class IoU report."""
    names = [[] for _ in range(20)]
    for name, _, tid, _ in LABELS:
        names[19 if tid == 255 else tid].append(name)
    return [", ".join(n) for n in names]


def encode_target(target) -> np.ndarray:
    """Cityscapes label ids -> train ids (255 = ignore)."""
    return ID_TO_TRAIN_ID[np.array(target)]


def encode_target_test(target) -> np.ndarray:
    """train ids -> Cityscapes label ids (for test-server submissions)."""
    return TRAIN_ID_TO_ID[np.array(target)]


def decode_target(target) -> np.ndarray:
    """train ids -> RGB colours (ignore = black)."""
    target = np.array(target)
    target[target == 255] = 19
    return TRAIN_ID_TO_COLOR[target]


def normalize_transform(size: Optional[Tuple[int, int]] = None) -> Callable:
    """``transform(image, label) -> (float tensor (3,H,W) normalised with MEAN/STD, uint8 label tensor or None)``; the
    reference's validation pipeline (test_swiftnet.py:61-65: resize to (res, 2*res) -- bilinear for the image, nearest for
    the labels --, to [0,1] tensor, normalise) without torchvision."""
    from PIL import Image

    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)

    def transform(img, lbl):
        if size is not None:
            img = img.resize((size[1], size[0]), Image.BILINEAR)
            if lbl is not None:
                lbl = lbl.resize((size[1], size[0]), Image.NEAREST)
        x = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).to(torch.float32).div_(255.0)
        x = (x - mean) / std
        y = torch.from_numpy(np.array(lbl, dtype=np.uint8)) if lbl is not None else None
        return x, y

    return transform


def _load_rgb(path):
    from PIL import Image

    return Image.open(path).convert("RGB")


class CityscapesClips(torch.utils.data.Dataset):
    """Clip dataset over a Cityscapes tree (layout in the module docstring); same constructor arguments, sample structure,
    file naming and error behaviour as the reference's ``CityscapesVid`` (cityscapes_vid.py:104-202)."""

    mean, std = MEAN, STD
    fine_classes = FINE_CLASSES

    def __init__(self, root, split="train", target_type="semantic", transform=None, clip_length=20, has_labels=True):
        self.root = os.path.expanduser(root)
        self.mode = "gtFine"
        self.target_type = target_type
        self.images_dir = os.path.join(self.root, "leftImg8bit", split)
        self.vid_dir = os.path.join(self.root, "leftImg8bit_sequence", split)
        self.targets_dir = os.path.join(self.root, self.mode, split)
        self.extension = ".png"
        self.transform = transform
        assert 0 < clip_length <= 20, "Clip length must be between 1 and 20 frames"
        self.clip_length, self.interval, self.has_labels, self.split = clip_length, 1, has_labels, split
        if split not in ("train", "test", "val"):
            raise ValueError('Invalid split for mode! Please use split="train", split="test" or split="val"')
        if not (os.path.isdir(self.images_dir) and os.path.isdir(self.targets_dir) and os.path.isdir(self.vid_dir)):
            raise RuntimeError("Dataset not found or incomplete. Please make sure all required folders for the specified "
                               f'"split" and "mode" are available:\n images dir: {self.images_dir}\n video dir: {self.vid_dir}'
                               f"\n targets dir: {self.targets_dir}")
        self.images, self.relative_dirs, self.file_names, self.targets = [], [], [], []
        suffix = {"instance": "instanceIds.png", "semantic": "labelIds.png", "color": "color.png", "polygon": "polygons.json",
                  "depth": "depth.png"}[target_type]
        for city in os.listdir(self.images_dir):
            img_dir, target_dir = os.path.join(self.images_dir, city), os.path.join(self.targets_dir, city)
            for file_name in os.listdir(img_dir):
                self.relative_dirs.append(os.path.join(city, file_name))
                self.images.append(os.path.join(img_dir, file_name))
                self.file_names.append(file_name)
                self.targets.append(os.path.join(target_dir, f"{file_name.split('_leftImg8bit')[0]}_{self.mode}_{suffix}"))

    def __len__(self):
        return len(self.images)

    def clip_paths(self, index: int) -> List[str]:
        """Paths of the clip's frames in chronological order; the last one is the labelled frame (from ``leftImg8bit``),
        the others come from ``leftImg8bit_sequence`` with the frame number counted down by ``interval``."""
        rel = self.relative_dirs[index].replace("_leftImg8bit.png", "")
        parts = rel.split("_")
        prefix, frame = "_".join(parts[:-1]), int(parts[-1])
        paths = [self.images[index]]
        for i in range(1, self.clip_length):
            paths.append(os.path.join(self.vid_dir, f"{prefix}_{str(frame - i * self.interval).zfill(6)}_leftImg8bit{self.extension}"))
        return paths[::-1]

    def __getitem__(self, index):
        from PIL import Image

        paths = self.clip_paths(index)
        target = Image.open(self.targets[index]) if self.has_labels else None
        frames = []
        for k, path in enumerate(paths):
            image = _load_rgb(path)
            if k == len(paths) - 1:      # the labelled frame is transformed together with its label map
                if self.transform:
                    image, target = self.transform(image, target)
            elif self.transform:
                image, _ = self.transform(image, None)
            frames.append(image)
        if target is not None:
            target = encode_target(target)
        return frames, (0 if target is None else target), {"relpath": self.relative_dirs[index]}


def write_synthetic_tree(root: str, split: str = "val", cities: Sequence[str] = ("aachen", "bonn"), clips_per_city: int = 2,
                         clip_length: int = 4, size: Tuple[int, int] = (32, 64), seed: int = 0) -> List[str]:
    """A miniature Cityscapes tree with the reference's layout: seeded random RGB frames, label-id maps over the whole id
    range 0..33.  Returns the relative paths of the labelled frames."""
    from PIL import Image

    rng = np.random.default_rng(seed)
    rels = []
    for ci, city in enumerate(cities):
        for d in ("leftImg8bit", "leftImg8bit_sequence", "gtFine"):
            os.makedirs(os.path.join(root, d, split, city), exist_ok=True)
        for c in range(clips_per_city):
            seq, frame = f"{ci:06d}", 19 + 30 * c
            stem = f"{city}_{seq}_{frame:06d}"
            Image.fromarray(rng.integers(0, 256, size + (3,), dtype=np.uint8)).save(os.path.join(root, "leftImg8bit", split, city, stem + "_leftImg8bit.png"))
            Image.fromarray(rng.integers(0, 34, size, dtype=np.uint8)).save(os.path.join(root, "gtFine", split, city, stem + "_gtFine_labelIds.png"))
            for i in range(1, clip_length):
                Image.fromarray(rng.integers(0, 256, size + (3,), dtype=np.uint8)).save(
                    os.path.join(root, "leftImg8bit_sequence", split, city, f"{city}_{seq}_{frame - i:06d}_leftImg8bit.png"))
            rels.append(os.path.join(city, stem + "_leftImg8bit.png"))
    return rels
