"""ali_hip: MI355X-native kernels + host glue for the ALI/BiGAN training path.

``libali_hip.so`` (C ABI: include/ali_hip.h) holds the hand-written gfx950 kernels;
this package binds it with ctypes and schedules it behind the reference's
``image_scms`` module surface.  Nothing here falls back to eager PyTorch or to the
CPU oracle for CUDA tensors: without the library every call raises.
"""
from ._lib import AliHipUnavailable, LIB_PATH, load  # noqa: F401
from .dropout import injected_masks, manual_seed  # noqa: F401
