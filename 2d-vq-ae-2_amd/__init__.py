"""vqae_amd -- MI355X-native (gfx950) VQ-AE inference hot path, a drop-in for the conv-encoder ->
vector-quantise -> conv-decoder forward pass of sara-nl/2D-VQ-AE-2.

The directory is named `2d-vq-ae-2_amd` (not an importable identifier); import it as `vqae_amd`
through the loader module of the same name at the repository root.

Layout: csrc/ (HIP kernels + C ABI, built into libvqae_hip.so), _lib.py / ops.py (ctypes binding),
native.py (whole-model handle), layers/ + model.py (mirrors of the reference's module API),
extract_embeddings.py (whole-slide driver), dist.py (one-process-per-GPU sharding over RCCL).
"""
from . import _lib, ops, spec  # noqa: F401
from .native import NativeVQAE  # noqa: F401
from .spec import SPECS, VQAESpec  # noqa: F401
