"""pocketflow_amd -- MI355X-native compression-training hot path with PocketFlow's plugin surface.

Host side mirrors the reference's package layout (learners/, nets/, utils/, datasets/) so that a
`nets/*_run.py` entry script reads the same; device side is the gfx950 kernel library in csrc/
reached through the C ABI of include/pocketflow_hip.h (pocketflow_amd.hip).
"""
__version__ = '0.1.0'

import os as _os

# MIOpen (what torch falls back to for the few contractions that are not ours) otherwise considers its naive reference convolutions:
# their one-off solver search costs a minute of warm-up, and round 4 saw hipStreamEndCapture crash on a step recorded with them
# (pocketflow_amd/step_graph.py).  Read by MIOpen at its first convolution; a value set by the user wins.
for _k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
  _os.environ.setdefault(_k, '0')
