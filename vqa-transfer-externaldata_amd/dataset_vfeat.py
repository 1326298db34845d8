"""Image + box provider of the region-feature extractor: the counterpart of vqa/datasets/dataset_vfeat.py:20-98
(and of the box helpers it uses, util/box_utils.py:41-71, 140-147, 210-236).

One example = one image resized to 540x540 RGB float32 in [0, 255] (PIL, as the reference) plus the DenseCap boxes
of that image -- stored per image as (x, y, w, h) in ORIGINAL pixels in `<densecap_dir>/<split>/
results_original_size.hdf5` under `<image_id>/boxes` -- scaled to the resized image, converted to (x1, y1, x2, y2)
and to the normalised, clipped [y1, x1, y2, x2] form tf.image.crop_and_resize takes; at most 50 boxes per image.
The HDF5 files are read with hdf5_io (no h5py)."""
from __future__ import annotations

import collections
import os

import numpy as np

from . import hdf5_io

IMAGE_WIDTH = 540
IMAGE_HEIGHT = 540
MAX_ROI_NUM = 50
DENSECAP_FILENAME = "results_original_size.hdf5"


# ---------------------------------------------------------------- util/box_utils.py
def xywh_to_x1y1x2y2(boxes):
    """(x, y, w, h) -> (x1, y1, x2, y2) = (x, y, x + w, y + h)   (util/box_utils.py:41-71)"""
    boxes = np.asarray(boxes)
    return np.stack([boxes[..., 0], boxes[..., 1], boxes[..., 0] + boxes[..., 2], boxes[..., 1] + boxes[..., 3]], axis=-1)


def scale_boxes_xywh(boxes, frac):
    """x, w scaled by frac[0] and y, h by frac[1] (a list), or everything by a scalar (util/box_utils.py:210-236)"""
    if isinstance(frac, (list, tuple)):
        assert len(frac) == 2, "only two dimension frac is possible for array input"
        new_boxes = np.array(boxes, copy=True)
        new_boxes[:, 0] *= frac[0]
        new_boxes[:, 1] *= frac[1]
        new_boxes[:, 2] *= frac[0]
        new_boxes[:, 3] *= frac[1]
        return new_boxes
    return np.asarray(boxes) * float(frac)


def normalize_boxes_x1y1x2y2(boxes, width, height):
    """(x1, y1, x2, y2) pixels -> [y1, x1, y2, x2] in [0, 1], clipped (util/box_utils.py:140-147)"""
    boxes = np.asarray(boxes, np.float32)
    new_boxes = np.stack([boxes[:, 1] / height, boxes[:, 0] / width, boxes[:, 3] / height, boxes[:, 2] / width], axis=1)
    return np.clip(new_boxes, 0, 1)


class Dataset(object):
    def __init__(self, image_paths, image_dir, densecap_dir, is_train=True, name="default", boxes=None):
        """boxes: optional {image_id: [n,4] xywh array} replacing the DenseCap HDF5 files (synthetic runs, tests)."""
        self.name = name
        self.image_paths = list(image_paths)
        self._ids = list(range(len(self.image_paths)))
        self.image_dir, self.densecap_dir, self.is_train = image_dir, densecap_dir, is_train
        self.width, self.height, self.max_roi_num = IMAGE_WIDTH, IMAGE_HEIGHT, MAX_ROI_NUM
        self._boxes = boxes
        self.densecap = {}
        if boxes is None:
            for split in ("train2014", "val2014", "test2015"):
                path = os.path.join(densecap_dir, split, DENSECAP_FILENAME)
                if os.path.exists(path):       # the reference opens all three; a partial download works split by split
                    self.densecap[split] = hdf5_io.File(path)

    @property
    def ids(self):
        return self._ids

    def __len__(self):
        return len(self._ids)

    def get_config(self):
        config = collections.namedtuple("dataset_config", [])
        config.image_width, config.image_height, config.max_roi_num = IMAGE_WIDTH, IMAGE_HEIGHT, MAX_ROI_NUM
        return config

    def _raw_boxes(self, split, image_id):
        if self._boxes is not None:
            return np.asarray(self._boxes[image_id])
        if split not in self.densecap:
            raise KeyError("no DenseCap file for split %r under %s" % (split, self.densecap_dir))
        return np.asarray(self.densecap[split][image_id]["boxes"])

    supports_image_out = True     # get_data(id, image_out=...) writes the pixels into a caller-owned [H,W,3] f32 slot

    def get_data(self, id, image_out=None):
        from PIL import Image
        image_path = self.image_paths[id]
        o_image = Image.open(os.path.join(self.image_dir, image_path))
        o_w, o_h = o_image.size
        rgb = o_image.resize([self.width, self.height]).convert("RGB")
        if image_out is None:
            image = np.array(rgb, dtype=np.float32)
        else:       # uint8 -> float32 straight into the batch buffer: no 3.5 MB temporary per image, no np.stack later
            np.copyto(image_out, np.asarray(rgb), casting="unsafe")
            image = image_out
        frac_x, frac_y = self.width / float(o_w), self.height / float(o_h)
        split = image_path.split("/")[0]
        image_id = image_path.replace("/", "-")
        raw = self._raw_boxes(split, image_id)[:MAX_ROI_NUM]
        box = xywh_to_x1y1x2y2(scale_boxes_xywh(raw.astype(np.float32), [frac_x, frac_y]))
        normal_box = normalize_boxes_x1y1x2y2(box, self.width, self.height)
        return {"image": image, "box": box.astype(np.float32), "normal_box": normal_box,
                "num_box": np.array(box.shape[0], dtype=np.int32), "image_id": image_id,
                "image_id_len": np.array(len(image_id), dtype=np.int32)}

    def get_data_shapes(self):
        return {"image": [self.height, self.width, 3], "box": [None, 4], "normal_box": [None, 4], "num_box": (),
                "image_id": [None], "image_id_len": ()}


def create_dataset(image_paths, image_dir, densecap_dir, is_train=False, boxes=None):
    return Dataset(image_paths, image_dir, densecap_dir, is_train=is_train, name="vfeat", boxes=boxes)
