"""ILSVRC-12 input pipeline: the image oracle against fixtures produced by EXECUTING the reference's
own preprocess_image (tests/golden/make_reference_image_golden.py), the host half of the product (descriptors, crop
sampler, TFRecord dataset) on CPU with the kernel emulated, and -- marked gpu -- the HIP resize kernel against the
oracle."""
import io
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
A = np.load(os.path.join(HERE, 'golden', 'reference_image.npz'))
M = json.load(open(os.path.join(HERE, 'golden', 'reference_image.json')))
NAMES = ['landscape', 'portrait', 'small', 'square']


def _decode(name):
  from pocketflow_amd.datasets.imagenet_preprocessing import decode_jpeg
  return decode_jpeg(A[name + '/jpeg'].tobytes())


# -- oracle vs the reference's code ---------------------------------------------------------------------------------------
def test_oracle_size_arithmetic_and_constants():
  from oracle import image_oracle as O
  assert O.CHANNEL_MEANS.tolist() == [np.float32(m) for m in M['means']] and O.RESIZE_MIN == M['resize_min']
  for c in M['cases']:
    if 'resized' in c:
      assert list(O.smallest_size_at_least(*c['size'])) == c['resized'], c


@pytest.mark.parametrize('name', NAMES)
def test_oracle_preprocessing_matches_the_reference_code(name):
  from oracle import image_oracle as O
  img = _decode(name)
  size = [c for c in M['cases'] if c.get('name') == name and 'eval_out' in c][0]['eval_out']
  assert np.array_equal(O.preprocess_eval(img, size, size), A[name + '/eval'])
  for c in [c for c in M['cases'] if c.get('name') == name and 'window' in c]:
    got = O.preprocess_train(img, c['window'], c['flip'], c['out'], c['out'])
    assert np.array_equal(got, A['%s/train%d' % (name, c['k'])]), c


def test_oracle_crop_sampler_constraints():
  from oracle import image_oracle as O
  rng = np.random.RandomState(0)
  boxes = np.array([[0.2, 0.3, 0.6, 0.9]])
  for _ in range(200):
    y, x, h, w = O.sample_distorted_bounding_box(rng, 300, 400, boxes)
    assert 0 <= y and y + h <= 300 and 0 <= x and x + w <= 400
    if (h, w) != (300, 400):
      assert 0.05 * 120000 <= h * w <= 120000 and 0.74 <= w / h <= 1.35
      iy = max(0, min(180, y + h) - max(60, y)); ix = max(0, min(360, x + w) - max(120, x))
      assert iy * ix / (120 * 240) >= 0.1


# -- product host half (kernel emulated) -----------------------------------------------------------------------------------
@pytest.fixture
def fake_image_kernel(monkeypatch):
  import pocketflow_amd.datasets.imagenet_preprocessing as P
  from fake_hip import FakeHipFull
  monkeypatch.setattr(P, 'hip', FakeHipFull())
  return P


@pytest.mark.parametrize('name', NAMES)
def test_product_descriptors_reproduce_the_reference_outputs(fake_image_kernel, name):
  P = fake_image_kernel
  img = _decode(name)
  size = [c for c in M['cases'] if c.get('name') == name and 'eval_out' in c][0]['eval_out']
  out = P.preprocess_batch([img], [P.describe_eval(img.shape[0], img.shape[1], size, size)], size, size, 'cpu')
  assert np.array_equal(out[0].numpy(), A[name + '/eval'])
  for c in [c for c in M['cases'] if c.get('name') == name and 'window' in c]:
    y, x, h, w = c['window']
    out = P.preprocess_batch([img[y:y + h, x:x + w]], [P.describe_train(h, w, c['out'], c['out'], c['flip'])], c['out'], c['out'], 'cpu')
    assert np.array_equal(out[0].numpy(), A['%s/train%d' % (name, c['k'])]), c


def test_product_batch_packs_images_of_different_sizes(fake_image_kernel):
  P = fake_image_kernel
  imgs = [_decode(n) for n in NAMES]
  descs = [P.describe_eval(i.shape[0], i.shape[1], 56, 56) for i in imgs]
  out = P.preprocess_batch(imgs, descs, 56, 56, 'cpu')
  from oracle import image_oracle as O
  for k, im in enumerate(imgs):
    assert np.array_equal(out[k].numpy(), O.preprocess_eval(im, 56, 56))


def test_product_crop_sampler_matches_oracle_draw_for_draw():
  from oracle import image_oracle as O
  from pocketflow_amd.datasets.imagenet_preprocessing import sample_distorted_bounding_box
  boxes = np.array([[0.1, 0.1, 0.5, 0.7], [0.6, 0.2, 0.9, 0.4]])
  r1, r2 = np.random.RandomState(5), np.random.RandomState(5)
  for hw in ((300, 400), (97, 131), (500, 120)):
    for _ in range(50):
      assert sample_distorted_bounding_box(r1, hw[0], hw[1], boxes) == O.sample_distorted_bounding_box(r2, hw[0], hw[1], boxes)
  assert sample_distorted_bounding_box(r1, 64, 64, None) == O.sample_distorted_bounding_box(r2, 64, 64, None)


# -- the HIP kernel ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_resize_kernel_matches_oracle_on_gpu(dtype):
  from oracle import image_oracle as O
  import pocketflow_amd.datasets.imagenet_preprocessing as P
  imgs = [_decode(n) for n in NAMES]
  # eval descriptors at the real output size and training descriptors (crop + flip) mixed in one batch
  descs, want, srcs = [], [], []
  for k, im in enumerate(imgs):
    descs.append(P.describe_eval(im.shape[0], im.shape[1], 224, 224)); srcs.append(im)
    want.append(O.preprocess_eval(im, 224, 224))
    h, w = im.shape[0] // 2, (2 * im.shape[1]) // 3
    y, x = im.shape[0] // 7, im.shape[1] // 5
    flip = bool(k % 2)
    descs.append(P.describe_train(h, w, 224, 224, flip)); srcs.append(im[y:y + h, x:x + w])
    want.append(O.preprocess_train(im, (y, x, h, w), flip, 224, 224))
  out = P.preprocess_batch(srcs, descs, 224, 224, 'cuda', dtype=dtype)
  torch.cuda.synchronize()
  got = out.float().cpu().numpy()
  want = np.stack(want)
  if dtype == torch.float32:
    assert np.array_equal(got, want)                         # same float32 operation order: bit-exact
  else:
    assert np.array_equal(got, torch.from_numpy(want).to(torch.bfloat16).float().numpy())
