"""Real-data CIFAR-10 reader (reference datasets/cifar10_dataset.py parse_fn) on small hand-made binary files,
and the tensor contract shared with the synthetic stream."""
import numpy as np
import pytest
import torch


def _write_bin(path, n, seed):
  rng = np.random.RandomState(seed)
  labels = rng.randint(0, 10, n).astype(np.uint8)
  images = rng.randint(0, 256, (n, 3, 32, 32)).astype(np.uint8)          # CHW on disk
  rec = np.concatenate([labels[:, None], images.reshape(n, -1)], axis=1)
  rec.tofile(path)
  return labels, images


def test_cifar10_binary_reader_and_parse(tmp_path):
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.datasets import cifar10_dataset as C
  l1, i1 = _write_bin(tmp_path / 'data_batch_1.bin', 40, 1)
  l2, i2 = _write_bin(tmp_path / 'data_batch_2.bin', 24, 2)
  lt, it = _write_bin(tmp_path / 'test_batch.bin', 20, 3)
  FLAGS.data_dir_local = str(tmp_path)
  FLAGS.batch_size, FLAGS.batch_size_eval, FLAGS.nb_classes = 16, 8, 10
  # evaluation subset: file order, no augmentation, exact standardisation
  ev = C.Cifar10Dataset(is_train=False).build()
  x, y = ev.get_next()
  assert x.shape == (8, 32, 32, 3) and x.dtype == torch.float32 and y.shape == (8, 10) and y.dtype == torch.float32
  ref = (it[:8].transpose(0, 2, 3, 1).astype(np.float32) - C._MEAN) / C._STD
  np.testing.assert_allclose(x.numpy(), ref, rtol=0, atol=1e-6)
  assert np.array_equal(y.numpy().argmax(1), lt[:8]) and np.all(y.numpy().sum(1) == 1)
  ev.get_next(); ev.get_next()                           # 24 > 20: repeat() wraps into the next epoch
  ev.reset()
  x2, _ = ev.get_next()
  assert torch.equal(x, x2)
  # training subset: every output is a flipped / shifted 32x32 window of the zero-padded standardised image
  tr = C.Cifar10Dataset(is_train=True).build()
  assert len(tr) == 64
  xs, ys = tr.get_next()
  allimg = np.concatenate([i1, i2]).transpose(0, 2, 3, 1).astype(np.float32)
  alll = np.concatenate([l1, l2])
  std = (allimg - C._MEAN) / C._STD
  pad = np.pad(std, ((0, 0), (4, 4), (4, 4), (0, 0)))
  for b in range(4):
    lab = int(ys[b].argmax())
    cands = np.where(alll == lab)[0]
    out = xs[b].numpy()
    found = False
    for c in cands:
      for oy in range(9):
        for ox in range(9):
          win = pad[c, oy:oy + 32, ox:ox + 32]
          if np.allclose(win, out, atol=1e-6) or np.allclose(win[:, ::-1], out, atol=1e-6):
            found = True
            break
        if found:
          break
      if found:
        break
    assert found, 'sample %d is not an augmented window of a record with its label' % b
  # seeded: a rebuilt iterator yields the same stream; train / validation split = skip / take
  tr2 = C.Cifar10Dataset(is_train=True).build()
  assert torch.equal(tr2.get_next()[0], xs)
  FLAGS.nb_smpls_val = 10
  trn, val = C.Cifar10Dataset(is_train=True).build(enbl_trn_val_split=True)
  assert len(val) == 10 and len(trn) == 54


def test_synthetic_fallback_has_the_same_contract(tmp_path):
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd.datasets import cifar10_dataset as C
  FLAGS.data_dir_local = None
  FLAGS.batch_size, FLAGS.nb_classes = 4, 10     # (nb_classes' default depends on which dataset module was imported last)
  it = C.Cifar10Dataset(is_train=True).build()
  x, y = it.get_next()
  assert x.shape == (4, 32, 32, 3) and x.dtype == torch.float32 and y.shape == (4, 10) and float(y.sum()) == 4.0
  FLAGS.data_disk = 'hdfs'
  with pytest.raises(ValueError, match='HDFS'):
    C.Cifar10Dataset(is_train=True)


def test_cifar10_parse_fn_matches_the_reference_code():
  """standardize / augment (and the label one-hot) against the reference's own parse_fn executed on hand-made records
  (tests/golden/make_reference_image_golden.py; crop offsets and flip injected): pins the CHW -> HWC transpose, the
  standardisation constants, zero padding AFTER the standardisation, the crop origin and the flip."""
  import json
  import os
  import numpy as np
  import torch
  from pocketflow_amd.datasets import cifar10_dataset as D
  here = os.path.dirname(os.path.abspath(__file__))
  A = np.load(os.path.join(here, 'golden', 'reference_image.npz'))
  M = json.load(open(os.path.join(here, 'golden', 'reference_image.json')))
  aug = [c for c in M['cases'] if 'cifar_augment' in c][0]['cifar_augment']
  records = A['cifar/records']
  path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'pf_cifar_records_test.bin')
  records.tofile(path)
  try:
    images, labels = D.read_records([path])
  finally:
    os.remove(path)
  assert labels.tolist() == [3, 0, 9, 7]
  x = D.standardize(torch.from_numpy(images))
  for i in range(4):
    assert np.array_equal(x[i].numpy(), A['cifar/eval%d' % i])
    onehot = np.zeros(10, np.float32)
    onehot[labels[i]] = 1
    assert np.array_equal(A['cifar/label%d' % i], onehot)
    for k, (oy, ox, flip) in enumerate(aug):
      got = D.augment(x[i:i + 1], torch.tensor([oy]), torch.tensor([ox]), torch.tensor([bool(flip)]))
      assert np.array_equal(got[0].numpy(), A['cifar/train%d_%d' % (i, k)]), (i, k)
  # a whole batch with per-image parameters at once
  oy = torch.tensor([a[0] for a in aug]); ox = torch.tensor([a[1] for a in aug]); fl = torch.tensor([bool(a[2]) for a in aug])
  got = D.augment(x, oy, ox, fl)
  for i in range(4):
    assert np.array_equal(got[i].numpy(), A['cifar/train%d_%d' % (i, i)])
