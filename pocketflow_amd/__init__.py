"""pocketflow_amd -- MI355X-native compression-training hot path with PocketFlow's plugin surface.

Host side mirrors the reference's package layout (learners/, nets/, utils/, datasets/) so that a
`nets/*_run.py` entry script reads the same; device side is the gfx950 kernel library in csrc/
reached through the C ABI of include/pocketflow_hip.h (pocketflow_amd.hip).
"""
__version__ = '0.1.0'
