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


# -- TFRecord dataset end to end (CPU, kernel emulated) ---------------------------------------------------------------------
def _write_shards(tmp_path, prefix, nb_files, per_file, hw=(48, 64), first_label=1):
  from PIL import Image
  from pocketflow_amd.datasets.tfrecord import make_example, write_records
  rng = np.random.RandomState(3)
  label, images = first_label, {}
  for k in range(nb_files):
    recs = []
    for _ in range(per_file):
      h, w = hw[0] + rng.randint(0, 9), hw[1] + rng.randint(0, 9)
      base = np.clip(rng.randint(0, 256, (1, 1, 3)) + 20 * np.sin(np.arange(h)[:, None, None] / 5.0) + np.zeros((h, w, 3)), 0, 255)
      buf = io.BytesIO()
      Image.fromarray(base.astype(np.uint8)).save(buf, format='JPEG', quality=95)
      recs.append(make_example({'image/encoded': buf.getvalue(), 'image/class/label': np.array([label], np.int64),
                                'image/class/text': b'n%08d' % label,
                                'image/object/bbox/xmin': np.array([0.1], np.float32), 'image/object/bbox/ymin': np.array([0.2], np.float32),
                                'image/object/bbox/xmax': np.array([0.9], np.float32), 'image/object/bbox/ymax': np.array([0.8], np.float32)}))
      images[label] = buf.getvalue()
      label += 1
    write_records(str(tmp_path / ('%s-%05d-of-%05d' % (prefix, k, nb_files))), recs)
  return images


def test_ilsvrc12_tfrecord_dataset_on_cpu(fake_image_kernel, tmp_path, monkeypatch):
  from oracle import image_oracle as O
  import pocketflow_amd.datasets.ilsvrc12_dataset as D
  from pocketflow_amd.flags import FLAGS
  train_imgs = _write_shards(tmp_path, 'train', 3, 7)
  val_imgs = _write_shards(tmp_path, 'validation', 2, 5, first_label=500)
  FLAGS.data_dir_local, FLAGS.batch_size, FLAGS.batch_size_eval, FLAGS.image_size = str(tmp_path), 4, 5, 32
  FLAGS.nb_classes, FLAGS.nb_smpls_val, FLAGS.buffer_size, FLAGS.nb_threads, FLAGS.prefetch_size = 1001, 6, 8, 2, 2
  # evaluation: file order, no shuffling, exact preprocessing of every image
  it = D.Ilsvrc12Dataset(is_train=False).build()
  seen = []
  for _ in range(2):
    images, labels = it.get_next()
    assert images.shape == (5, 32, 32, 3) and images.dtype == torch.float32 and labels.shape == (5, 1001)
    for k in range(5):
      lab = int(labels[k].argmax())
      assert float(labels[k].sum()) == 1.0
      seen.append(lab)
      want = O.preprocess_eval(np.asarray(__import__('PIL.Image', fromlist=['x']).open(io.BytesIO(val_imgs[lab])).convert('RGB')), 32, 32)
      assert np.array_equal(images[k].numpy(), want)
  assert seen == [500, 505, 501, 506, 502, 507, 503, 508, 504, 509]           # round-robin interleave of the two shards
  images, labels = it.get_next()                                             # repeat(): starts over
  assert int(labels[0].argmax()) == 500
  it.close()
  # training: every record of the split shows up, shuffled, augmented; the validation split is disjoint
  it_trn, it_val = D.Ilsvrc12Dataset(is_train=True).build(enbl_trn_val_split=True)
  val_labels = set()
  for _ in range(3):
    images, labels = it_val.get_next()
    val_labels.update(int(l.argmax()) for l in labels)
  assert len(val_labels) == 6
  trn_labels = []
  for _ in range(12):
    images, labels = it_trn.get_next()
    assert images.shape == (4, 32, 32, 3) and torch.isfinite(images).all()
    assert float(images.max()) <= 255 - 103.94 + 1e-3 and float(images.min()) >= -123.68 - 1e-3
    trn_labels += [int(l.argmax()) for l in labels]
  assert set(trn_labels) == set(train_imgs) - val_labels and len(set(trn_labels)) == 15
  assert trn_labels[:15] != sorted(trn_labels[:15])                          # shuffled
  # seeded: a second iterator reproduces the stream
  it2, _v = D.Ilsvrc12Dataset(is_train=True).build(enbl_trn_val_split=True)
  again = []
  for _ in range(3):
    again += [int(l.argmax()) for l in it2.get_next()[1]]
  assert again == trn_labels[:12]
  for i in (it_trn, it_val, it2, _v):
    i.close()
