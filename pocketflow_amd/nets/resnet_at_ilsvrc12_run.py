"""Execution script for ResNet models on the ILSVRC-12 dataset (reference nets/resnet_at_ilsvrc12_run.py:27-69).

    python -m pocketflow_amd.nets.resnet_at_ilsvrc12_run --learner uniform --uql_weight_bits 8 ...
    scripts/run_local.sh pocketflow_amd/nets/resnet_at_ilsvrc12_run.py -n=8 --learner uniform ...
"""
from pocketflow_amd.nets.resnet_at_ilsvrc12 import ModelHelper
from pocketflow_amd.nets.run_utils import run_main

if __name__ == '__main__':
  raise SystemExit(run_main(ModelHelper))
