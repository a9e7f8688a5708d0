#!/usr/bin/env python
"""Headline benchmark: images/sec of the ResNet-50 8-bit UniformQuantLearner fine-tune step with
distillation (BASELINE.json configs[2]) on N MI355X GPUs of one node, synthetic 224x224x3 batches.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one batch per GPU: teacher forward, weight fake-quant
of all 54 kernels, student forward (BN+ReLU+activation fake-quant applied inside the 1x1 convolutions), CE + coupled L2 +
distillation loss, backward with straight-through estimators, [RCCL all-reduce], fused Adam.
Rank 0 prints ONE JSON line (contract in the task description).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

# MIOpen's one-off solver search otherwise times its naive reference convolutions (~60 s of warm-up)
for _k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD',
           'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
  os.environ.setdefault(_k, '0')

R50_FLOPS_PER_IMAGE_STEP_DST = 32.71e9    # fwd + bwd-data + bwd-filter + teacher fwd
MFMA_BF16_PEAK = 2.5e15                   # dense bf16, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12                         # B/s, MI355X_MICROARCH.md


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--config', default='c2', choices=sorted(CONFIGS),
                  help='which concrete run of SURVEY 8(d): c2 = BASELINE.json configs[2] (the headline line, default); '
                       'c2a32 = the same with the reference default of 32-bit activations; c1 = configs[1]; c3 = configs[3]; '
                       'c4 = configs[4] (each at its per-GPU batch on N GPUs)')
  ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the configuration\'s; north star: 256)')
  ap.add_argument('--resnet_size', type=int, default=None)
  ap.add_argument('--image_size', type=int, default=None)
  ap.add_argument('--dtype', default='bfloat16')
  ap.add_argument('--act_bits', type=int, default=None)
  ap.add_argument('--weight_bits', type=int, default=None)
  ap.add_argument('--roofline_kernel', default=None,
                  help='profiling region whose launches are timed with HIP events: conv1x1_fwd | conv1x1_wrw | '
                       'conv1x1_bwd_data | conv2d_fwd | bn_bwd_apply | bn_bwd_stats | bn_act_quant_apply | bn_stats')
  ap.add_argument('--no_cpu_baseline', action='store_true')
  ap.add_argument('--step_graph', type=int, default=None,
                  help='1 (default): the steady-state step is recorded in a hipGraph during the warm-up and replayed '
                       '(pocketflow_amd/step_graph.py; N > 1: two graphs around the gradient-exchange calls); 0: every step is issued launch by launch.  The throughput of a '
                       'GPU-bound step does not depend on it when the host is fast (C2: 10 307 vs 10 302 images/s) but a loaded host '
                       'cannot slow a recorded step down; a replayed graph cannot carry timing events on ROCm, so the last --event_steps '
                       'steps are issued launch by launch for the roofline figure')
  ap.add_argument('--unshared_steps', type=int, default=2,
                  help='extra launch-by-launch steps OUTSIDE the timed region with the teacher branch serialised with the student step: '
                       'the roofline region with the chip to itself (roofline.unshared); 0 = skip')
  ap.add_argument('--event_steps', type=int, default=3,
                  help='with --step_graph 1: this many of the K timed steps run launch by launch, their roofline-region launches '
                       'bracketed by HIP events (a replayed graph cannot carry timing events)')
  ap.add_argument('--no_reexec', action='store_true', help=argparse.SUPPRESS)       # accepted, ignored (older scripts)
  ap.add_argument('--no_prewarm', action='store_true', help=argparse.SUPPRESS)      # accepted, ignored (older scripts)
  ap.add_argument('--cpu_batch', type=int, default=32, help='batch of the CPU baseline sample (SURVEY 8d: 32)')
  ap.add_argument('--cpu_steps', type=int, default=5, help='timed CPU steps after the warm-up (SURVEY 8d: >= 5)')
  ap.add_argument('--cpu_budget_s', type=float, default=200.0, help='wall-clock bound of the CPU baseline sample')
  args = ap.parse_args()
  cfg = CONFIGS[args.config]
  for k in ('batch', 'resnet_size', 'image_size', 'act_bits', 'weight_bits', 'roofline_kernel'):
    if getattr(args, k) is None:
      setattr(args, k, cfg[k])
  return args


# The concrete runs of SURVEY.md section 8(d) ("Configs -> concrete runs"); every one prints the same JSON contract.
#   flops: fwd + bwd-data + bwd-filter of the student (+ the teacher's forward with distillation) per image, SURVEY App. C
CONFIGS = {
    'c2': dict(model='resnet', learner='uniform', resnet_size=50, image_size=224, batch=256, weight_bits=8, act_bits=8, dst=True,
               roofline_kernel='conv1x1_fwd', flops=32.71e9, nb_classes=1001,
               workload='ResNet-v2-{resnet_size}@ILSVRC-12-synthetic {image_size}x{image_size}x3, UniformQuantLearner w{weight_bits}/a{act_bits} + '
                        'distillation, Adam, batch {batch}/GPU (BASELINE.json configs[2])',
               metric='images/sec ResNet-50 INT8 quant-aware fine-tune (whole job; per GPU = value / n_gpus)'),
    'c2a32': dict(model='resnet', learner='uniform', resnet_size=50, image_size=224, batch=256, weight_bits=8, act_bits=32, dst=True,
                  roofline_kernel='conv1x1_fwd', flops=32.71e9, nb_classes=1001,
                  workload='ResNet-v2-{resnet_size}@ILSVRC-12-synthetic {image_size}x{image_size}x3, UniformQuantLearner w{weight_bits}/a{act_bits} '
                           '(the reference default uql_activation_bits) + distillation, Adam, batch {batch}/GPU (configs[2] variant)',
                  metric='images/sec ResNet-50 8-bit-weight quant-aware fine-tune, 32-bit activations (whole job)'),
    'c1': dict(model='resnet_cifar', learner='weight-sparse', resnet_size=20, image_size=32, batch=128, weight_bits=8, act_bits=32, dst=False,
               roofline_kernel='bn_bwd_apply', flops=3 * 82.15e6, nb_classes=10,
               workload='ResNet-v2-{resnet_size}@CIFAR-10-synthetic 32x32x3, WeightSparseLearner 50 % sparsity (uniform), Momentum, '
                        'batch {batch}/GPU (BASELINE.json configs[1])',
               metric='images/sec ResNet-20 weight-sparsification fine-tune (whole job)'),
    'c3': dict(model='mobilenet', learner='channel', resnet_size=0, image_size=224, batch=256, weight_bits=8, act_bits=32, dst=True,
               roofline_kernel='conv1x1_fwd', flops=4 * 1.137e9, nb_classes=1001,
               workload='MobileNet-v1@ILSVRC-12-synthetic {image_size}x{image_size}x3, ChannelPrunedLearner masked fine-tune at preserve '
                        'ratio 0.5 (seeded keep-masks; the LASSO selection is host work outside the step) + distillation, Adam, '
                        'batch {batch}/GPU (BASELINE.json configs[3])',
               metric='images/sec MobileNet-v1 channel-pruned fine-tune (whole job)'),
    'c4': dict(model='resnet', learner='non-uniform', resnet_size=50, image_size=224, batch=256, weight_bits=4, act_bits=32, dst=True,
               roofline_kernel='conv1x1_fwd', flops=32.71e9, nb_classes=1001,
               workload='ResNet-v2-{resnet_size}@ILSVRC-12-synthetic {image_size}x{image_size}x3, NonUniformQuantLearner {weight_bits}-bit codebooks + '
                        'distillation, Adam, batch {batch}/GPU (BASELINE.json configs[4])',
               metric='images/sec ResNet-50 4-bit non-uniform quant-aware fine-tune (whole job)'),
}


def set_flags(args, tmp, world):
  """Import the modules of the configuration (importing DEFINES their flags, as in the reference) and assign the flag
  values of the run.  Needs no GPU: tests/test_bench_cpu.py runs it in a fresh interpreter for every configuration.
  Returns (ModelHelper class, Learner class)."""
  from pocketflow_amd.flags import FLAGS
  import pocketflow_amd.learners.abstract_learner  # noqa: F401
  import pocketflow_amd.learners.learner_utils  # noqa: F401
  cfg = CONFIGS[args.config]
  kind = cfg['learner']
  FLAGS.enbl_multi_gpu = world > 1
  # importing the modules DEFINES their flags (as in the reference); values are assigned afterwards
  import pocketflow_amd.learners.distillation_helper  # noqa: F401
  if cfg['model'] == 'resnet':
    from pocketflow_amd.nets.resnet_at_ilsvrc12 import ModelHelper
  elif cfg['model'] == 'resnet_cifar':
    from pocketflow_amd.nets.resnet_at_cifar10 import ModelHelper
  else:
    from pocketflow_amd.nets.mobilenet_at_ilsvrc12 import ModelHelper
  if kind == 'uniform':
    from pocketflow_amd.learners.uniform_quantization.learner import UniformQuantLearner as Learner
  elif kind == 'non-uniform':
    from pocketflow_amd.learners.nonuniform_quantization.learner import NonUniformQuantLearner as Learner
  elif kind == 'weight-sparse':
    from pocketflow_amd.learners.weight_sparsification.learner import WeightSparseLearner as Learner
  else:
    from pocketflow_amd.learners.channel_pruning.learner import ChannelPrunedLearner as Learner
  FLAGS.nb_classes = cfg['nb_classes']
  if cfg['model'] != 'resnet_cifar':
    FLAGS.image_size = args.image_size        # (the CIFAR-10 pipeline has a fixed 32 x 32 input and no such flag)
  FLAGS.batch_size = args.batch
  FLAGS.compute_dtype = args.dtype
  FLAGS.enbl_dst = cfg['dst']
  FLAGS.dst_eval_teacher = False            # the teacher's one-off evaluation is not part of a step
  FLAGS.synthetic_pool = 8              # "a fixed pool of 8 distinct batches cycled"
  FLAGS.enbl_step_graph = bool(getattr(args, 'step_graph', None) == 1)     # (default: decided after the warm-up, see main)
  FLAGS.save_path = os.path.join(tmp, 'models', 'model.ckpt')
  FLAGS.save_path_dst = os.path.join(tmp, 'models_dst', 'model.ckpt')
  if cfg['model'] != 'mobilenet':
    FLAGS.resnet_size = args.resnet_size
  if kind == 'uniform':
    FLAGS.uql_weight_bits, FLAGS.uql_activation_bits = args.weight_bits, args.act_bits
    FLAGS.uql_save_quant_model_path = os.path.join(tmp, 'uql', 'model.ckpt')
  elif kind == 'non-uniform':
    FLAGS.nuql_weight_bits, FLAGS.nuql_activation_bits = args.weight_bits, args.act_bits
    FLAGS.nuql_save_quant_model_path = os.path.join(tmp, 'nuql', 'model.ckpt')
  elif kind == 'weight-sparse':
    FLAGS.ws_prune_ratio, FLAGS.ws_prune_ratio_prtl = 0.5, 'uniform'
    FLAGS.ws_save_path = os.path.join(tmp, 'ws', 'model.ckpt')
  else:
    FLAGS.cp_channel_pruned_path = os.path.join(tmp, 'models', 'pruned_model.ckpt')
    FLAGS.cp_best_path = os.path.join(tmp, 'models', 'best_model.ckpt')
    FLAGS.cp_original_path = os.path.join(tmp, 'models', 'original_model.ckpt')
    FLAGS.cp_lrn_rate_ft = 1e-4
  return ModelHelper, Learner


def build_learner(args, FLAGS, tmp, rank, world, barrier):
  """(learner, step function) of the configuration: the learner classes, flags and entry points are the ones
  nets/*_run.py would use; only the data are synthetic and the start checkpoint is seeded."""
  import numpy as np
  from pocketflow_amd.learners.learner_utils import create_synthetic_checkpoint
  cfg = CONFIGS[args.config]
  kind = cfg['learner']
  ModelHelper, Learner = set_flags(args, tmp, world)
  mh = ModelHelper()
  if rank == 0:
    create_synthetic_checkpoint(mh)
  barrier()
  learner = Learner(None, mh)
  if kind == 'non-uniform':
    learner.init_clusters()
  if kind == 'channel':
    # the masked fine-tune of cp learner.py:381-471 with seeded keep-masks at preserve ratio 0.5: every Conv2D but the first
    # (inputs) and the last (outputs) loses half of its input / output channels; depthwise kernels are never masked
    from pocketflow_amd.utils import checkpoint
    rng = np.random.RandomState(11)
    convs = [op for op in learner.graph.matmul_ops if op.var.kind == 'conv']
    vals = learner.graph.store.export_numpy()
    fake = {}
    for i, op in enumerate(convs):
      kh, kw, cin, cout = op.var.ref_shape
      keep_in = np.ones(cin, bool) if i == 0 else rng.rand(cin) < 0.5
      keep_out = np.ones(cout, bool) if i == len(convs) - 1 else rng.rand(cout) < 0.5
      keep_in[0] = keep_out[0] = True
      fake[op.name] = [keep_in.tolist(), keep_out.tolist()]
      m = np.zeros(op.var.ref_shape, np.float32)
      m[:, :, keep_in, :] = 1.0
      m[:, :, :, ~keep_out] = 0.0
      vals[op.var.name] = vals[op.var.name] * m
    pruned = None
    if rank == 0:
      pruned = checkpoint.save(vals, FLAGS.cp_channel_pruned_path, None)
    barrier()
    pruned = pruned or checkpoint.latest_checkpoint(os.path.dirname(FLAGS.cp_channel_pruned_path))
    learner.setup_finetune(pruned, finetune=True, fake_pruning_dict=fake)
  if kind == 'weight-sparse':
    # the step the second half of the run executes: masks at their final ratio (ws learner.py:296-312 reaches it at
    # 0.5 * nb_iters), gradient * mask + Momentum fused in the optimiser kernel
    learner.global_step = int(FLAGS.ws_iter_ratio_end * learner.nb_iters_train) + 1
    learner.prune_step()
  if world > 1 and kind != 'channel':
    bcast = getattr(learner, 'ops', {}).get('bcast') or getattr(learner, 'bcast_op', None)
    if bcast is not None:
      bcast()
  return learner, learner.train_step


# profiling region -> regular expressions of the kernels it launches (names as in the rocprofv3 summaries under profiles/).
# The fused 1x1 forward (region conv1x1_fwd: producer BN + ReLU + fake-quant prologue, residual / statistics epilogue) is
# dispatched per shape to three kernels: the resident-kernel variant (pf_conv_stream.hip), the direct-to-LDS staged
# variant with the in-LDS prologue pass (pf_igemm.hip, MODE 2) and the register-staged tiles (pf_conv.hip).
REGION_KERNELS = {'conv1x1_fwd': [r'^k_conv1x1_stream<\d+, true, ', r'^k_igemm<\d+, \d+, \d+, \d+, \d+, 2[,>]',
                                  r'^k_conv1x1_fwd<\d+, true, '],
                  'conv1x1_wrw': [r'^k_wrw2<', r'^k_conv1x1_wrw<', r'^k_wrw_tr<'],
                  'conv1x1_bwd_data': [r'^k_conv1x1_stream<\d+, false, ', r'^k_igemm<\d+, \d+, \d+, \d+, \d+, [01][,>]',
                                       r'^k_conv1x1_fwd<\d+, false, '],
                  'conv2d_fwd': [r'^k_igemm<\d+, \d+, \d+, \d+, \d+, 0[,>]'],
                  'bn_bwd_apply': [r'^k_bn_bwd_apply<'], 'bn_bwd_stats': [r'^k_bn_bwd_stats'],
                  'bn_act_quant_apply': [r'^k_bn_apply<'], 'bn_stats': [r'^k_bn_stats']}
PROFILE_TAG = 'r06'


def pmc_traffic_per_launch(region, tag=PROFILE_TAG):
  """HBM bytes per launch of the roofline region's kernels from the committed rocprofv3 PMC passes of THIS command
  (profiles/<tag>_pmc_FETCH_SIZE.csv / _WRITE_SIZE.csv; separate --pmc passes, values in KiB).  Per
  MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts wide coalesced reads at half their size on gfx950
  (x2 correction); WRITE_SIZE needs none (calibrated in round 1 on k_bn_apply, whose writes are exactly one
  tensor).  Returns None when the summaries are absent or the region's kernels cannot be told apart by name."""
  import csv
  import re
  pats = REGION_KERNELS.get(region)
  if pats is None or region in ('conv1x1_bwd_data', 'conv2d_fwd'):
    return None                                   # these share kernel names with other regions: not separable
  tot = 0.0
  for counter, mult in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
    path = os.path.join(ROOT, 'profiles', '%s_pmc_%s.csv' % (tag, counter))
    if not os.path.exists(path):
      return None
    disp, kib = 0, 0.0
    with open(path) as f:
      for r in csv.DictReader(f):
        if any(re.search(p, r['kernel']) for p in pats):
          disp += int(r['dispatches'])
          kib += float(r['mean_' + counter]) * int(r['dispatches'])
    if disp == 0:
      return None
    tot += mult * kib / disp * 1024.0
  return tot


def pmc_traffic_source(tag=PROFILE_TAG):
  """Where `roofline.traffic` comes from: it is NOT measured by this run (PMC collection needs rocprofv3 around the process) but
  read from the committed PMC passes of this command; profiles/<tag>_pmc_SOURCE.txt names the commit they were taken at."""
  path = os.path.join(ROOT, 'profiles', '%s_pmc_SOURCE.txt' % tag)
  if not os.path.exists(path):
    return None
  return open(path).read().strip()


def memory_snapshot(torch):
  """Device-memory picture of this process: a caching allocator that has to go to the driver inside the step
  (retries after a failed hipMalloc, segments allocated / released per step) is what made some bench processes host-bound."""
  st = torch.cuda.memory_stats()
  free, total = torch.cuda.mem_get_info()
  gb = 1.0 / (1 << 30)
  return {'device_total_gb': total * gb, 'device_free_gb': free * gb,
          'reserved_gb': st.get('reserved_bytes.all.current', 0) * gb, 'reserved_peak_gb': st.get('reserved_bytes.all.peak', 0) * gb,
          'allocated_peak_gb': st.get('allocated_bytes.all.peak', 0) * gb,
          'alloc_retries': st.get('num_alloc_retries', 0), 'ooms': st.get('num_ooms', 0),
          'segments_allocated': st.get('segment.all.allocated', 0), 'segments_freed': st.get('segment.all.freed', 0)}


def launch_probe(torch, n=1000):
  """Host cost of one launch and wall time per (empty) dispatch, in microseconds: ~5 / ~5 on a healthy process."""
  probe = torch.zeros(64, device='cuda')
  torch.cuda.synchronize()
  p0 = time.perf_counter()
  for _ in range(n):
    probe.add_(1.0)
  p1 = time.perf_counter()
  torch.cuda.synchronize()
  p2 = time.perf_counter()
  return {'host_us_per_launch': (p1 - p0) * 1e6 / n, 'us_per_dispatch': (p2 - p0) * 1e6 / n}


def main():
  args = parse_args()
  import torch
  import torch.distributed as dist
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
  from pocketflow_amd.flags import FLAGS
  from pocketflow_amd import profiling
  from pocketflow_amd.utils.multi_gpu_wrapper import MultiGpuWrapper as mgw

  # one scratch directory for ALL ranks (rank 0 writes the synthetic "pre-trained" checkpoint and the teacher's
  # renamed copy, the other ranks restore from them after a barrier -- exactly the reference's file-based hand-off)
  if world > 1:
    tag = '%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'run'))
    tmp = os.path.join(tempfile.gettempdir(), 'pf_bench_' + ''.join(c if c.isalnum() else '_' for c in tag))
  else:
    tmp = tempfile.mkdtemp(prefix='pf_bench_')
  if world > 1:
    import pocketflow_amd.learners.abstract_learner  # noqa: F401  (defines enbl_multi_gpu & co. before mgw reads flags)
    mgw.init()
    if rank == 0:
      shutil.rmtree(tmp, ignore_errors=True)
      os.makedirs(tmp, exist_ok=True)
    dist.barrier()
  torch.backends.cudnn.benchmark = os.environ.get('PF_CUDNN_BENCHMARK', '1') != '0'

  def barrier():
    if world > 1:
      dist.barrier()
  learner, train_step = build_learner(args, FLAGS, tmp, rank, world, barrier)
  cfg = CONFIGS[args.config]
  if os.environ.get('PF_BENCH_MAIN_PRIORITY', '0') == '1':
    # experiment: the student's queue at high priority (the teacher's and the backward-filter queue stay at normal priority)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.Stream(priority=-1))

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    train_step()
  # Host hygiene for the launch-by-launch mode: a full collection of Python's cyclic collector walks every tracked object of the
  # process (torch, numpy, the layer graph: 50-90 ms, seen as host stalls of that size every few steps in host_submit_ms_steps --
  # with a GPU step of 11-22 ms the launch queue runs dry).  Everything alive after the warm-up moves to the permanent generation;
  # the per-step garbage (autograd nodes, tensor wrappers) is still collected, cheaply.
  import gc
  gc.collect()
  gc.freeze()
  # --step_graph: the recording (three launch-by-launch steps, then one pass of the Python step under stream capture) belongs to
  # the warm-up whatever W is; a learner whose step cannot be recorded says so once and stays launch-by-launch
  sg = None
  auto_share = None
  recorded_choice = None
  # N > 1: launch-by-launch steps by default (round 6).  They keep the bucket all-reduces inside the backward pass, which the recorded
  # step gives up, and on the boxes of this round a launch-by-launch step was as fast as a replay (10 445 vs 10 395 images/s: the host
  # submits a step in ~11 ms); the two-graph recorded step has only ever run over gloo on a shared GPU.  PF_STEP_GRAPH_DIST=1 opts in
  # (kept only where its replays are not slower, decided by all ranks together, below).
  if args.step_graph is None and (world == 1 or os.environ.get('PF_STEP_GRAPH_DIST', '0') == '1'):
    # default: the step is recorded; how host-bound it is launch by launch (the host's share of two untimed steps from an empty launch
    # queue) is measured first and reported in the line (config.host_share_of_two_launch_by_launch_steps)
    torch.cuda.synchronize()
    a0 = time.perf_counter()
    train_step()
    train_step()
    a1 = time.perf_counter()
    torch.cuda.synchronize()
    auto_share = (a1 - a0) / max(time.perf_counter() - a0, 1e-9)
    FLAGS.enbl_step_graph = True          # (the measured host share is reported; the recording is the default)
  if FLAGS.enbl_step_graph:
    # N > 1: every rank records its own two graphs around the gradient-exchange calls (step_graph.CudaBackend.cut).  Every
    # rank makes the same number of train_step calls whether its recording succeeded or not (a call that records also executes
    # one step), and whether the job stays on the recorded step is decided together.
    from pocketflow_amd import step_graph
    sg = step_graph.of(learner)
    extra = 0
    while sg.state == 'warm' and extra < 8:
      train_step()
      extra += 1
    ready = sg.state == 'ready'
    if world > 1:
      flag = torch.tensor([1.0 if ready else 0.0], device='cuda')
      dist.all_reduce(flag, op=dist.ReduceOp.MIN)
      if ready and float(flag.item()) == 0.0:
        sg.suspend()                       # another rank could not record: this one goes back to launch-by-launch steps with it
        ready = False
    if not ready:
      sys.stderr.write('bench.py: step graph not recorded (%r): launch-by-launch steps\n' % (sg.error,))
      FLAGS.enbl_step_graph = False
      sg = None
    else:
      # torch.cuda.graph() empties the caching allocator before it records: the launch-by-launch steps of the timed region (the ones
      # whose roofline-region launches are bracketed by events) would pay hipMalloc for every activation again -- 100-280 ms in ONE
      # step, measured.  Re-grow the ordinary pool here, untimed: suspend, two launch-by-launch steps, resume, one replay.
      sg.suspend()
      train_step()
      train_step()
      sg.resume()
      train_step()
      train_step()
      # The job stays on the recorded step only where it is not slower than launch-by-launch steps ON THIS JOB.  N > 1: decided
      # together from the slowest rank's times.  (The recorded step trades the overlap of the exchange with the backward pass
      # for ~700 fewer host launches per step; which one wins depends on the host and the links.  With two ranks sharing ONE GPU
      # over gloo, the test hook, replays were measured 10-20x SLOWER: the exchange call waits 80-900 ms,
      # profiles/r05_recorded_step_two_ranks.txt.)  N = 1 (round 6): with the backward-filter launches on a second queue
      # (graph.WrwSide) a launch-by-launch step ran 22.4 ms where the replay of the same step -- the forks are edges of the
      # hipGraph -- stayed at 23.4 ms (profiles/r06_wrw_side_ab.txt); a slower host turns that around, hence measured, here.
      def step_seconds(n):
        sync()
        t = time.perf_counter()
        for _ in range(n):
          train_step()
        sync()
        return (time.perf_counter() - t) / n
      n_cal = 2 if world > 1 else 6
      t_rec = step_seconds(n_cal)
      sg.suspend()
      train_step()                       # (the hand-over between the modes is not part of either figure)
      t_lbl = step_seconds(n_cal)
      if world > 1:
        both = torch.tensor([t_rec, t_lbl], dtype=torch.float64, device='cuda')
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        t_rec, t_lbl = (float(v) for v in both.tolist())
      recorded_choice = {'graphs': len(sg.backend.graphs), 'exchange_calls_between_graphs': len(sg.backend.actions),
                         'replay_ms_per_step': t_rec * 1e3, 'launch_by_launch_ms_per_step': t_lbl * 1e3,
                         'kept': bool(t_rec <= (1.1 if world > 1 else 1.0) * t_lbl)}
      if args.step_graph == 1:
        recorded_choice['kept'] = True     # --step_graph 1: the recorded step, whatever the calibration says
      if recorded_choice['kept']:
        sg.resume()
        train_step()
        train_step()
      else:
        recorded_choice['replayed_steps'] = sg.n_replays
        FLAGS.enbl_step_graph = False
        sg = None
  # Self-diagnosis of a host-bound process.  Seven bench processes of round 2 ran at 150 ms instead of
  # 29 ms per step with the same kernels: the caching allocator was going to the driver for every tensor (torch.empty at
  # 185 us) because a reference cycle in the layer executor kept each step's activations alive until Python's cyclic
  # collector happened to run.  The cycle is gone (graph._BnLazy) and tests/ hold that property; what stays here is cheap
  # and outside the timed region: an empty-launch probe, the host's share of two untimed steps from an empty queue (0.45
  # for this workload, 1.0 when host-bound), the allocator's counters -- all part of the JSON line -- and, if the process
  # IS host-bound, one more untimed step under cProfile on stderr.
  torch.cuda.synchronize()
  probe_warm = launch_probe(torch)
  h0 = time.perf_counter()
  train_step()
  train_step()
  h1 = time.perf_counter()
  torch.cuda.synchronize()
  h2 = time.perf_counter()
  probe_warm['host_share_of_two_steps'] = (h1 - h0) / max(h2 - h0, 1e-9)
  mem_warm = memory_snapshot(torch)
  if world == 1 and (probe_warm['us_per_dispatch'] > 40.0 or probe_warm['host_share_of_two_steps'] > 0.9):   # (an extra step on one rank only would dead-lock the all-reduce)
    import cProfile
    import io
    import pstats
    prof = cProfile.Profile()
    prof.enable()
    train_step()
    prof.disable()
    torch.cuda.synchronize()
    buf = io.StringIO()
    pstats.Stats(prof, stream=buf).sort_stats('tottime').print_stats(14)
    text = 'bench.py: host-bound process (probe %r)\nmemory before the profiled step %r\nmemory after it %r\n%s' % (
        probe_warm, mem_warm, memory_snapshot(torch), buf.getvalue())
    sys.stderr.write(text)
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
      with open(os.path.join(out_dir, 'bench_host_bound_profile.txt'), 'w') as f:
        f.write(text)
  if sg is None and world == 1:
    # Launch-by-launch steps: the host runs ahead of the GPU (6 vs 11 ms per step on MobileNet, 11 vs 22 on ResNet-50), and the blocks
    # handed between the streams (record_stream: freed when the reader's event has passed) stay out of the pool for as long as the launch
    # queue is deep -- the caching allocator grows while the backlog builds up, one hipMalloc of 40-90 ms at a time (seen as
    # host stalls of that size and as `segments_allocated` moving INSIDE the timed region of round 6's first launch-by-launch lines:
    # 16.2 instead of 11.2 ms per step on MobileNet).  So the backlog is built up once, untimed: as many back-to-back steps as the timed
    # region has.
    for _ in range(args.steps):
      train_step()
    sync()
  n_event = args.steps if sg is None else max(0, min(args.event_steps, args.steps))
  # Recorded steps first, the launch-by-launch steps (roofline-region launches bracketed by events) LAST: the replays are submitted in
  # ~1 ms each, so the host's slower launch-by-launch submission (and the hand-over between the modes, 60-90 ms measured) runs while
  # the GPU still works through the replays.  The other way round the GPU sat idle through the first launch-by-launch step: 3 ms per
  # step over a 20-step region.
  profiling.include_side = True
  profiling.enable(args.roofline_kernel)
  if sg is not None:
    profiling.pause()
  sync()
  t0 = time.perf_counter()
  marks = []
  rec_marks = []                       # region launches recorded after each launch-by-launch step (recorded-step mode)
  for i in range(args.steps):
    if sg is not None and i == args.steps - n_event:
      sg.suspend()
      profiling.unpause()
    train_step()
    marks.append(time.perf_counter())
    if sg is not None and i >= args.steps - n_event:
      rec_marks.append(profiling.count(args.roofline_kernel))
  sync()
  dt = time.perf_counter() - t0
  host_ms = (marks[-1] - t0) * 1e3 / max(1, args.steps)        # host-side submission time per step (GPU-bound when << ms_per_step)
  host_steps = sorted((b - a) * 1e3 for a, b in zip([t0] + marks[:-1], marks))
  probe_after = launch_probe(torch)
  mem_after = memory_snapshot(torch)
  if os.environ.get('PF_BENCH_TRACE_STEPS') and rank == 0:    # host-side submission time of every step (diagnostics)
    sys.stderr.write('host ms/step: %s | tail sync %.1f ms\n' % (
        ' '.join('%.1f' % ((b - a) * 1e3) for a, b in zip([t0] + marks[:-1], marks)), (t0 + dt - marks[-1]) * 1e3))
  multi_gpu = None
  if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # data-parallel invariants, outside the timed region: after the broadcast and K all-reduced updates every rank must
    # hold bit-identical parameters (deterministic kernels + one summed gradient); how much of the exchange was launched
    # from inside backward; bytes on the links per step (utils/multi_gpu_wrapper.py:83-98 of the reference = Horovod)
    st = learner.graph.store
    sig = torch.stack([b.double().sum() for b in (st.w_master, st.o_master)] +
                      [b.double().abs().sum() for b in (st.w_master, st.o_master)]).to('cuda')
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(sigs, sig)
    red = getattr(st, 'reducer', None)
    el = 4 if (red is not None and red.reduce_dtype is not None) else st.w_grad.element_size()
    multi_gpu = {'backend': dist.get_backend(), 'params_identical_across_ranks': all(bool(torch.equal(sigs[0], x)) for x in sigs),
                 'buckets': len(red.buckets) if red is not None else None,
                 'buckets_launched_inside_backward': red.n_overlapped if red is not None else None,
                 'allreduce_bytes_per_step': int(st.w_size * el + st.o_size * 4),
                 'recorded_step': (dict(recorded_choice, replayed_steps=sg.n_replays) if sg is not None and recorded_choice is not None
                                   else recorded_choice)}
  # The region's launches of BOTH streams: the student's (main stream) and -- profiling.include_side -- the teacher's over the next
  # batch (second stream).  A launch that shares the chip with the other stream takes longer from start to end; that is what a
  # rocprofv3 kernel trace of this command shows for it too, so the average over both is the figure the trace's average agrees with.
  n_launch, ms, work = profiling.summary(args.roofline_kernel)
  n_m, ms_m, work_m = profiling.summary(args.roofline_kernel, side=False)
  n_s, ms_s, work_s = profiling.summary(args.roofline_kernel, side=True)
  by_stream = {'main_stream': {'launches': n_m, 'avg_launch_ms': (ms_m / n_m) if n_m else None},
               'teacher_stream': {'launches': n_s, 'avg_launch_ms': (ms_s / n_s) if n_s else None}}
  # The same region with the chip to itself (`roofline.unshared`, outside the timed region): two more launch-by-launch steps in which the
  # teacher branch is SERIALISED with the student's step (TeacherAhead.serialise: the side stream waits for the main stream before the
  # teacher is issued, the next step waits for the teacher as always) -- every launch of the region then runs alone, which is the figure
  # that is comparable round over round (round 3's bench had no teacher beside the region) and that measures the kernels rather than
  # the overlap.  `frac` above stays the shared figure: it is what a rocprofv3 trace of this command shows.
  unshared = None
  ahead_h = getattr(learner, '_teacher_ahead', None)
  if world == 1 and n_launch and ahead_h is not None and not getattr(ahead_h, 'handover_only', False) and args.unshared_steps > 0:
    sync()
    ahead_h.serialise = True
    train_step()                                       # (its teacher logits were issued un-serialised by the last timed step)
    sync()
    profiling.reset()
    for _ in range(args.unshared_steps):
      train_step()
    sync()
    ahead_h.serialise = False
    n_u, ms_u, work_u = profiling.summary(args.roofline_kernel)
    if n_u and ms_u > 0:
      unshared = {'launches': n_u, 'avg_launch_ms': ms_u / n_u, 'achieved': work_u / (ms_u * 1e-3) / 1e9, 'unit': 'GB/s',
                  'frac': work_u / (ms_u * 1e-3) / HBM_PEAK, 'steps': args.unshared_steps,
                  'how': 'extra launch-by-launch steps outside the timed region, teacher branch serialised with the student step'}

  if rank == 0:
    images = args.batch * world * args.steps
    value = images / dt
    per_gpu = value / world
    ach = (work / (ms * 1e-3)) if ms > 0 else 0.0
    # Headline: the roofline north_star names -- the step's dense-convolution FLOPs (student forward +
    # backward-data + backward-filter + teacher forward, 2 x MAC, conv + FC only) over the step time against the dense bf16 MFMA peak.
    # `hbm_region`: the HBM-bound region measured launch by launch with HIP events (the fused 1x1 forward of both networks), the figure
    # with the chip to itself (`unshared`: measures the kernels) first, the figure beside the teacher branch (`shared`: what a rocprofv3
    # trace of this command shows; it moves with the stream overlap, not with the kernels) second.
    shared = {'achieved': ach / 1e9, 'unit': 'GB/s', 'frac': ach / HBM_PEAK, 'launches': n_launch,
              'avg_launch_ms': (ms / n_launch) if n_launch else None, 'by_stream': by_stream,
              'sharing': ('the teacher branch (forward over the next batch, second stream) runs beside these launches: their durations '
                          'include the sharing' if getattr(learner, '_teacher_ahead', None) is not None or (sg is not None and sg.nxt is not None)
                          else 'none')}
    hbm_region = {'bound': 'hbm',
                  'kernel': 'fused 1x1 convolution forward, student + teacher (k_conv1x1_stream / k_igemm<..,2> / k_conv1x1_fwd)'
                            if args.roofline_kernel == 'conv1x1_fwd' else args.roofline_kernel,
                  'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                  'algorithmic_bytes_per_launch': (work / n_launch) if n_launch else None,
                  'traffic': pmc_traffic_per_launch(args.roofline_kernel), 'traffic_source': pmc_traffic_source(),
                  'unshared': unshared, 'shared': shared}
    mfma_ach = per_gpu * cfg['flops']
    roofline = {'bound': 'mfma',
                'kernel': 'whole step: every dense convolution / FC launch of the student (forward, backward-data, backward-filter) and of the '
                          'teacher (forward); %.2f GFLOP per image (SURVEY 8(d))' % (cfg['flops'] / 1e9),
                'achieved': mfma_ach / 1e12, 'peak': MFMA_BF16_PEAK / 1e12, 'unit': 'TFLOP/s', 'frac': mfma_ach / MFMA_BF16_PEAK,
                'traffic': None, 'step_mfma_frac': mfma_ach / MFMA_BF16_PEAK, 'hbm_region': hbm_region}
    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline and cfg['learner'] == 'uniform':   # the oracle timer restates the UQ step
      # the reference path restated on the host cores (oracle/learner_oracle.py), in a child process with a
      # hard wall-clock limit so that the default run always finishes within minutes
      import subprocess
      code = ('import json, sys; sys.path.insert(0, %r); from oracle.learner_oracle import time_cpu_baseline; '
              'print("CPU_BASELINE " + json.dumps(time_cpu_baseline(%d, %d, %d, %d, %d, %d, budget_s=%f)))'
              % (ROOT, args.resnet_size, args.image_size, args.cpu_batch, args.cpu_steps, args.weight_bits,
                 args.act_bits, args.cpu_budget_s))
      try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True,
                             timeout=args.cpu_budget_s + 60)
        for ln in out.stdout.splitlines():
          if ln.startswith('CPU_BASELINE '):
            cpu_baseline = json.loads(ln[len('CPU_BASELINE '):])
      except subprocess.TimeoutExpired:
        cpu_baseline = None
    line = {
        'metric': cfg['metric'],
        'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16' if args.dtype.startswith('bf') else 'f32', 'data': 'synthetic',
        'value_per_gpu': per_gpu, 'host_submit_ms_per_step': host_ms,
        'host_submit_ms_min_median_max': [host_steps[0], host_steps[len(host_steps) // 2], host_steps[-1]],
        'host_submit_ms_steps': [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)],
        'launch_probe': {'after_warmup': probe_warm, 'after_timed_region': probe_after, 'extra_untimed_steps': 2},
        'memory': {'after_warmup': mem_warm, 'after_timed_region': mem_after},
        'config': {'workload': cfg['workload'].format(**vars(args)), 'name': args.config,
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                   # learners/teacher_ahead.py (the default since round 4): the teacher's forward over batch k+1 on a second stream;
                   # the roofline region holds BOTH streams' launches (profiling.include_side), `roofline.unshared` the same
                   # launches with the chip to themselves
                   'teacher': 'next batch, side stream' if getattr(learner, '_teacher_ahead', None) is not None else 'in line',
                   'step_graph': ({'replayed_steps': args.steps - n_event, 'launch_by_launch_steps_with_events': n_event,
                                   'teacher_branch': sg.nxt is not None} if sg is not None else None),
                   # recorded step or launch-by-launch steps: measured on this job before the timed region (kept = the recorded step)
                   'step_mode_calibration': recorded_choice,
                   # graph.WrwSide (round 6): the backward-filter launches of a pass on a second stream, one join before the optimiser
                   'backward_filter_queue': ('second stream' if getattr(learner.graph.store, 'wrw_side', None) is not None else 'one queue'),
                   'host_share_of_two_launch_by_launch_steps': auto_share},
        'roofline': roofline, 'cpu_baseline': cpu_baseline, 'multi_gpu': multi_gpu}
    print(json.dumps(line))
  if world > 1:
    dist.barrier()
  if rank == 0:
    shutil.rmtree(tmp, ignore_errors=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
