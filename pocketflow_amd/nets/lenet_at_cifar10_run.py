"""Execution script for LeNet-like models on the CIFAR-10 dataset (reference nets/lenet_at_cifar10_run.py:27-69).

    python -m pocketflow_amd.nets.lenet_at_cifar10_run --learner uniform --uql_weight_bits 8 ...
    scripts/run_local.sh pocketflow_amd/nets/lenet_at_cifar10_run.py -n=8 --learner uniform ...
"""
from pocketflow_amd.nets.lenet_at_cifar10 import ModelHelper
from pocketflow_amd.nets.run_utils import run_main

if __name__ == '__main__':
  raise SystemExit(run_main(ModelHelper))
