"""CPU oracle for the ILSVRC-12 input pipeline  --  TEST INFRASTRUCTURE ONLY.

NumPy restatement of the reference's image preprocessing behind the JPEG decoder
(utils/external/imagenet_preprocessing.py:40-260, datasets/ilsvrc12_dataset.py:39-93):

  training  crop window from tf.image.sample_distorted_bounding_box -> random left/right flip ->
            resize_images(224 x 224, BILINEAR, align_corners=False) -> subtract (123.68, 116.78, 103.94)
  eval      aspect-preserving resize (shorter side 256, sizes truncated from float32 products) ->
            central 224 x 224 crop -> subtract the means

PARITY.  `preprocess_eval` / `preprocess_train` are pinned against the reference's own `preprocess_image`
executed over oracle/tf_stub.py (tests/golden/make_reference_image_golden.py -> tests/golden/reference_image.npz):
that pins the composition (size arithmetic, crop offsets, flip-before-resize, means).  The bilinear kernel itself is
TF's legacy (half_pixel_centers = false) kernel restated [3P] in two independent forms (loops in the stub, vectorised
here).  `sample_distorted_bounding_box` restates TF's sampler (sample_distorted_bounding_box_op.cc) [3P]; it draws
from its own generator, so only its constraints are testable.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
CHANNEL_MEANS = np.array([123.68, 116.78, 103.94], dtype=np.float32)   # imagenet_preprocessing.py:40-43
RESIZE_MIN = 256                                                      # :45


def resize_bilinear_legacy(img: np.ndarray, out_h: int, out_w: int, off_y: int = 0, off_x: int = 0,
                           win_h: int = None, win_w: int = None) -> np.ndarray:
  """TF-1.x `resize_images(BILINEAR, align_corners=False)` of an HWC image to out_h x out_w; optionally only the
  window [off_y : off_y + win_h, off_x : off_x + win_w] of the resized image is produced."""
  img = np.asarray(img).astype(F32)
  ih, iw = img.shape[:2]
  win_h = out_h if win_h is None else win_h
  win_w = out_w if win_w is None else win_w
  sy, sx = F32(ih) / F32(out_h), F32(iw) / F32(out_w)
  iy = (np.arange(win_h, dtype=np.int64) + off_y).astype(F32) * sy
  ix = (np.arange(win_w, dtype=np.int64) + off_x).astype(F32) * sx
  y0, x0 = np.floor(iy).astype(np.int64), np.floor(ix).astype(np.int64)
  y1 = np.minimum(np.ceil(iy).astype(np.int64), ih - 1)
  x1 = np.minimum(np.ceil(ix).astype(np.int64), iw - 1)
  ly = (iy - np.floor(iy)).astype(F32)[:, None, None]
  lx = (ix - np.floor(ix)).astype(F32)[None, :, None]
  tl, tr = img[y0][:, x0], img[y0][:, x1]
  bl, br = img[y1][:, x0], img[y1][:, x1]
  top = (tl + (tr - tl) * lx).astype(F32)
  bot = (bl + (br - bl) * lx).astype(F32)
  return (top + (bot - top) * ly).astype(F32)


def smallest_size_at_least(height: int, width: int, resize_min: int = RESIZE_MIN):
  """imagenet_preprocessing.py:149-174: float32 scale, products truncated to int32."""
  scale = F32(resize_min) / F32(min(height, width))
  return int(F32(height) * scale), int(F32(width) * scale)


def central_crop_offsets(height: int, width: int, crop_h: int, crop_w: int):
  """:83-103."""
  return (height - crop_h) // 2, (width - crop_w) // 2


def preprocess_eval(img_u8: np.ndarray, out_h: int = 224, out_w: int = 224) -> np.ndarray:
  nh, nw = smallest_size_at_least(img_u8.shape[0], img_u8.shape[1])
  top, left = central_crop_offsets(nh, nw, out_h, out_w)
  return resize_bilinear_legacy(img_u8, nh, nw, top, left, out_h, out_w) - CHANNEL_MEANS


def preprocess_train(img_u8: np.ndarray, window, flip: bool, out_h: int = 224, out_w: int = 224) -> np.ndarray:
  y, x, h, w = window
  crop = img_u8[y:y + h, x:x + w]
  if flip:
    crop = crop[:, ::-1]
  return resize_bilinear_legacy(crop, out_h, out_w) - CHANNEL_MEANS


def sample_distorted_bounding_box(rng: np.random.RandomState, height: int, width: int, bboxes=None,
                                  min_object_covered=0.1, aspect_ratio_range=(0.75, 1.33), area_range=(0.05, 1.0),
                                  max_attempts=100):
  """TF's sampler [3P]: up to max_attempts random crops with aspect ratio (w / h) and area fraction in range that
  cover >= min_object_covered of at least one box (the whole image when there is none:
  use_image_if_no_bounding_boxes=True); falls back to the whole image.  Returns (y, x, h, w)."""
  boxes = np.asarray(bboxes if bboxes is not None and len(bboxes) else [[0.0, 0.0, 1.0, 1.0]], dtype=np.float64)
  boxes_px = np.stack([boxes[:, 0] * height, boxes[:, 1] * width, boxes[:, 2] * height, boxes[:, 3] * width], axis=1)
  for _ in range(max_attempts):
    ratio = rng.uniform(aspect_ratio_range[0], aspect_ratio_range[1])
    min_area, max_area = area_range[0] * width * height, area_range[1] * width * height
    h_lo = int(np.rint(np.sqrt(min_area / ratio)))
    h_hi = int(np.rint(np.sqrt(max_area / ratio)))
    if int(np.rint(h_hi * ratio)) > width:
      h_hi = int((width + 0.5 - 1e-7) / ratio)
    h_hi = min(h_hi, height)
    h_lo = min(h_lo, h_hi)
    h = h_lo + (rng.randint(0, h_hi - h_lo + 1) if h_hi > h_lo else 0)
    w = int(np.rint(h * ratio))
    if w > width or h > height or w <= 0 or h <= 0:
      continue
    area = w * h
    if area < min_area or area > max_area:
      continue
    y = rng.randint(0, height - h + 1)
    x = rng.randint(0, width - w + 1)
    iy = np.clip(np.minimum(boxes_px[:, 2], y + h) - np.maximum(boxes_px[:, 0], y), 0, None)
    ix = np.clip(np.minimum(boxes_px[:, 3], x + w) - np.maximum(boxes_px[:, 1], x), 0, None)
    box_area = (boxes_px[:, 2] - boxes_px[:, 0]) * (boxes_px[:, 3] - boxes_px[:, 1])
    covered = np.where(box_area > 0, iy * ix / np.maximum(box_area, 1e-12), 0.0)
    if np.any(covered >= min_object_covered):
      return y, x, h, w
  return 0, 0, height, width
