"""ILSVRC-12 image preprocessing (reference utils/external/imagenet_preprocessing.py:40-260), split host / device:

  host    JPEG decode (Pillow = libjpeg, worker threads), the random crop window of
          tf.image.sample_distorted_bounding_box (training), size arithmetic -> one PfImageDesc per image
  device  pf_image_resize_bilinear: TF-1.x bilinear resize + flip + central-crop window + mean subtraction + cast,
          one launch per mini-batch over the packed uint8 images (pocketflow_amd/csrc/pf_image.hip)

so the float32 intermediates of the reference chain (decoded image -> resized image -> cropped image -> centred image)
never exist; only the decoded bytes cross PCIe (training: only the crop window).
"""
from __future__ import annotations

import io
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from pocketflow_amd import hip

_R_MEAN, _G_MEAN, _B_MEAN = 123.68, 116.78, 103.94
_CHANNEL_MEANS = [_R_MEAN, _G_MEAN, _B_MEAN]
_RESIZE_MIN = 256


def decode_jpeg(image_buffer: bytes) -> np.ndarray:
  """tf.image.decode_jpeg(channels=3): uint8 HWC RGB (grey-scale and CMYK files are converted, as TF does)."""
  from PIL import Image
  return np.asarray(Image.open(io.BytesIO(image_buffer)).convert('RGB'), dtype=np.uint8)


def _smallest_size_at_least(height: int, width: int, resize_min: int = _RESIZE_MIN) -> Tuple[int, int]:
  """New size with the shorter side == resize_min: float32 scale, float32 products truncated (:149-174)."""
  scale_ratio = np.float32(resize_min) / np.float32(min(height, width))
  return int(np.float32(height) * scale_ratio), int(np.float32(width) * scale_ratio)


def sample_distorted_bounding_box(rng: np.random.RandomState, height: int, width: int, bbox: Optional[np.ndarray],
                                  min_object_covered=0.1, aspect_ratio_range=(0.75, 1.33), area_range=(0.05, 1.0),
                                  max_attempts=100) -> Tuple[int, int, int, int]:
  """Crop window (y, x, h, w) with the constraints the reference passes to tf.image.sample_distorted_bounding_box
  (:62-70; use_image_if_no_bounding_boxes=True): aspect ratio w/h in [0.75, 1.33], area in [5 %, 100 %] of the image,
  at least 10 % of some annotated box (rows [ymin, xmin, ymax, xmax] in [0, 1]) covered; whole image after
  max_attempts failures."""
  boxes = np.asarray(bbox, dtype=np.float64).reshape(-1, 4) if bbox is not None and np.size(bbox) else \
      np.array([[0.0, 0.0, 1.0, 1.0]])
  by0, bx0, by1, bx1 = boxes[:, 0] * height, boxes[:, 1] * width, boxes[:, 2] * height, boxes[:, 3] * width
  box_area = np.maximum((by1 - by0) * (bx1 - bx0), 1e-12)
  img_area = float(width * height)
  for _ in range(max_attempts):
    ratio = rng.uniform(*aspect_ratio_range)
    h_min = int(np.rint(np.sqrt(area_range[0] * img_area / ratio)))
    h_max = int(np.rint(np.sqrt(area_range[1] * img_area / ratio)))
    if int(np.rint(h_max * ratio)) > width:
      h_max = int((width + 0.5 - 1e-7) / ratio)
    h_max = min(h_max, height)
    h_min = min(h_min, h_max)
    h = h_min + (rng.randint(0, h_max - h_min + 1) if h_max > h_min else 0)
    w = int(np.rint(h * ratio))
    if not (0 < w <= width and 0 < h <= height) or not (area_range[0] * img_area <= w * h <= area_range[1] * img_area):
      continue
    y, x = rng.randint(0, height - h + 1), rng.randint(0, width - w + 1)
    inter = np.clip(np.minimum(by1, y + h) - np.maximum(by0, y), 0, None) * \
        np.clip(np.minimum(bx1, x + w) - np.maximum(bx0, x), 0, None)
    if np.any(inter / box_area >= min_object_covered):
      return y, x, h, w
  return 0, 0, height, width


def describe_eval(height: int, width: int, output_height: int, output_width: int):
  """Descriptor fields of `_aspect_preserving_resize(256)` + `_central_crop` for a height x width source."""
  new_h, new_w = _smallest_size_at_least(height, width)
  return dict(h=height, w=width, scale_y=np.float32(height) / np.float32(new_h), scale_x=np.float32(width) / np.float32(new_w),
              off_y=(new_h - output_height) // 2, off_x=(new_w - output_width) // 2, flip=0)


def describe_train(crop_h: int, crop_w: int, output_height: int, output_width: int, flip: bool):
  """Descriptor fields of `random_flip_left_right` + `_resize_image` for an already cropped source."""
  return dict(h=crop_h, w=crop_w, scale_y=np.float32(crop_h) / np.float32(output_height),
              scale_x=np.float32(crop_w) / np.float32(output_width), off_y=0, off_x=0, flip=int(bool(flip)))


_STAGING = {}        # (device, slot) -> pinned uint8 buffer, grown geometrically (pinned allocations are expensive)
_UPLOADED = {}       # (device, slot) -> event recorded behind the last asynchronous upload from that buffer


def _staging_buffer(nbytes: int, device, pinned: bool, slot: int) -> torch.Tensor:
  if not pinned:
    return torch.empty(nbytes, dtype=torch.uint8)
  key = (str(device), slot)
  if key in _UPLOADED:
    _UPLOADED[key].synchronize()                     # the previous upload from this buffer has left the host memory
  buf = _STAGING.get(key)
  if buf is None or buf.numel() < nbytes:
    buf = _STAGING[key] = torch.empty(max(nbytes, 2 * (buf.numel() if buf is not None else 0)), dtype=torch.uint8, pin_memory=True)
  return buf[:nbytes]


_slot_counter = [0]


def preprocess_batch(images_u8: Sequence[np.ndarray], descs: Sequence[dict], output_height: int, output_width: int,
                     device, dtype=torch.float32, pinned: bool = True) -> torch.Tensor:
  """Pack the decoded (training: cropped) uint8 HWC images, upload, run the resize kernel.
  Returns [B, output_height, output_width, 3] `dtype` on `device` (NHWC, means subtracted)."""
  B = len(images_u8)
  table = np.zeros(B, dtype=hip.IMAGE_DESC_DTYPE)
  sizes = [int(im.shape[0]) * int(im.shape[1]) * 3 for im in images_u8]
  offsets = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])        # 16-byte aligned starts
  use_pin = pinned and torch.device(device).type == 'cuda'
  # two alternating pinned buffers: the asynchronous upload of batch i may still be in flight while batch i + 1 is packed
  _slot_counter[0] ^= 1
  staging = _staging_buffer(int(offsets[-1]), device, use_pin, _slot_counter[0])
  flat = staging.numpy()
  for i, (im, d) in enumerate(zip(images_u8, descs)):
    assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3 and im.shape[0] == d['h'] and im.shape[1] == d['w']
    flat[offsets[i]:offsets[i] + sizes[i]] = np.ascontiguousarray(im).reshape(-1)
    table[i] = (offsets[i], d['h'], d['w'], d['scale_y'], d['scale_x'], d['off_y'], d['off_x'], d['flip'], 0)
  src = staging.to(device, non_blocking=True)
  if use_pin:
    _UPLOADED[(str(device), _slot_counter[0])] = ev = torch.cuda.Event()
    ev.record()
  desc = torch.from_numpy(np.frombuffer(table.tobytes(), dtype=np.uint8).copy()).to(device, non_blocking=True)
  out = torch.empty((B, output_height, output_width, 3), dtype=dtype, device=device)
  hip.image_resize_bilinear(src, desc, out, _CHANNEL_MEANS)
  return out


def preprocess_image(image_buffer: bytes, bbox, output_height: int, output_width: int, num_channels: int = 3,
                     is_training: bool = False, rng: Optional[np.random.RandomState] = None):
  """Host half of the reference's `preprocess_image` for ONE image: returns (uint8 HWC image to upload, descriptor).
  Training draws the crop window and the flip from `rng`."""
  assert num_channels == 3
  img = decode_jpeg(image_buffer)
  if not is_training:
    return img, describe_eval(img.shape[0], img.shape[1], output_height, output_width)
  rng = rng if rng is not None else np.random
  y, x, h, w = sample_distorted_bounding_box(rng, img.shape[0], img.shape[1], bbox)
  flip = rng.uniform() < 0.5
  return img[y:y + h, x:x + w], describe_train(h, w, output_height, output_width, flip)
