"""scenesplat_amd -- MI355X-native hot path of SceneSplat (PTv3 encoder + language head).

The compute path is the hand-written HIP library ``lib/libscenesplat_hip.so`` (C-ABI in
``include/scenesplat_hip.h``) driven from a PyTorch-ROCm host.  There is NO CPU fallback:
importing the native layer without the built library, or calling an op on a non-GPU tensor,
raises.  Build with ``python -m scenesplat_amd.build`` (hipcc, gfx950).
"""
__version__ = "0.1.0"
