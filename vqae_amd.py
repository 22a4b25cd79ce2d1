"""Loader: makes the package directory `2d-vq-ae-2_amd/` importable as `vqae_amd`
(`import vqae_amd`, `from vqae_amd.layers.vq import EMAVectorQuantizer`, and Hydra
`_target_: vqae_amd.layers.vq.EMAVectorQuantizer`)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "2d-vq-ae-2_amd")
_spec = importlib.util.spec_from_file_location("vqae_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vqae_amd"] = _mod
_spec.loader.exec_module(_mod)
