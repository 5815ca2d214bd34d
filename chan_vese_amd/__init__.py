"""chan_vese_amd — MI355X-native Chan-Sandberg-Vese + Perona-Malik hot path.

The compute lives in chan_vese_amd/csrc (hand-written HIP for gfx950) behind the C ABI of
include/chanvese_hip.h; this package is the thin Python host over that ABI (ctypes).
There is no CPU fallback: importing `chan_vese_amd.capi` fails loudly if the HIP library
has not been built (`python -c 'import __graft_entry__ as g; g.build()'`).
"""
__version__ = "0.1.0"
