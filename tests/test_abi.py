"""CPU: libvqae_hip.so loads, exports every symbol include/vqae_hip.h declares, and its argument
validation (which runs before any HIP call) reports errors through the documented convention.
No compute is launched here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vqae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqae_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(amd):
    lib = amd._lib.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vqae_hip.h but not exported"
    assert set(names) == set(amd._lib.SYMBOLS), set(names) ^ set(amd._lib.SYMBOLS)


def test_build_info_and_sizes(amd):
    lib = amd._lib.lib()
    assert lib.vqae_build_info().startswith(b"gfx950")
    assert lib.vqae_conv_packed_floats(128, 128, 3) == 128 * 9 * 128
    assert lib.vqae_conv_packed_floats(16, 16, 1) == 128 * 16          # rows padded to 128
    assert lib.vqae_vq_workspace_bytes(1024, 256, 128) >= 256 * 128 * 4 + 2 * 1024 * 4


def test_error_convention_without_gpu(amd):
    L = amd._lib
    lib = L.lib()
    # null pointers -> VQAE_ERR_INVALID -> AssertionError (reference: assert, vq.py:98)
    with pytest.raises(AssertionError):
        L.check(lib.vqae_vq_forward_f32(None, None, 16, 4, 8, 1.0, None, 0, None, None, None, None, None))
    a = L.ConvArgs()
    a.batch, a.in_h, a.in_w, a.cin, a.cout, a.ksize, a.stride = 1, 8, 8, 3, 8, 3, 1
    one = ctypes.c_void_p(16)          # never dereferenced: validation fails first
    with pytest.raises(NotImplementedError):      # cin % 8 != 0 -> VQAE_ERR_UNSUPPORTED
        L.check(lib.vqae_conv2d_f32(ctypes.byref(a), one, one, None, None, one, None))
    assert b"cin" in lib.vqae_last_error()
    with pytest.raises(NotImplementedError):      # dim out of range (any 1 .. 4096 is taken, like the reference)
        L.check(lib.vqae_vq_forward_f32(one, one, 16, 4, 5000, 1.0, one, 0, None, None, None, one, None))
    with pytest.raises(NotImplementedError):      # n_codes out of range
        L.check(lib.vqae_vq_forward_f32(one, one, 16, 70000, 8, 1.0, one, 0, None, None, None, one, None))


def test_ops_refuse_cpu_tensors(amd):
    import torch
    with pytest.raises(amd._lib.VqaeHipError):
        amd.ops.vq_forward(torch.zeros(4, 8), torch.zeros(2, 8))


def test_state_dict_names_match_reference(amd, oracle):
    """The module mirrors expose exactly the reference's state-dict names/shapes."""
    from vqae_amd.model import VQAE
    for name in ("tiny", "tinyP", "tinyM"):
        m = VQAE.from_spec(amd.SPECS[name])
        sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        ref = oracle.param_shapes(oracle.SPECS[name])
        for k, shp in ref.items():
            assert sd[k] == tuple(shp), k
        extra = set(sd) - set(ref)
        assert all(k.endswith(("embed_avg", "cluster_size", "first_pass", "num_batches_tracked")) for k in extra), extra


def test_reference_error_types(amd):
    from vqae_amd.layers.conv_block import PreActFixupResBlock
    from vqae_amd.layers.vq import EMAVectorQuantizer
    import torch
    with pytest.raises(AssertionError):               # conv_block.py:148
        PreActFixupResBlock(8, 8, "sideways")
    vq = EMAVectorQuantizer(16, 8, 1.0, 0.99, 1e-5)
    with pytest.raises(AssertionError):               # vq.py:98
        vq(torch.zeros(4, 8))
    with pytest.raises(NotImplementedError):          # vq.py:100-104
        vq(torch.zeros(1, 4, 2, 2))


def test_lightning_checkpoint_import(amd, oracle, tmp_path):
    """A Lightning-style .ckpt ({'state_dict': ...}, reference naming) loads into the mirror model."""
    import torch
    from vqae_amd.model import VQAE, load_lightning_state_dict
    spec = oracle.SPECS["tiny"]
    p = oracle.make_params(spec, 0)
    vq = "encoder.vq_layers.0."
    full = dict(p)
    full[vq + "embed_avg"] = p[vq + "embed"].clone()
    full[vq + "cluster_size"] = torch.zeros(spec.num_embeddings)
    full[vq + "first_pass"] = torch.as_tensor(0)
    path = tmp_path / "epoch=3-step=100.ckpt"
    torch.save({"state_dict": full, "epoch": 3, "global_step": 100}, path)
    sd = load_lightning_state_dict(str(path))
    m = VQAE.from_spec(amd.SPECS["tiny"])
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    got = m.state_dict()
    for k, v in p.items():
        assert torch.equal(got[k], v), k


def test_mirrors_are_inference_only():
    """Inputs that require grad raise instead of returning values without a graph (no GPU needed: raised before any launch)."""
    import torch
    import vqae_amd
    from vqae_amd.model import VQAE
    m = VQAE.from_spec(vqae_amd.SPECS["tiny"])
    x = torch.zeros(1, 3, 32, 32, requires_grad=True)
    with pytest.raises(NotImplementedError):
        m(x)
    with pytest.raises(NotImplementedError):
        m.encoder(x)


def test_mirror_deepcopy_and_pickle_rebind_children(tmp_path):
    """A copied / unpickled VQAE owns its children (its encoder / decoder resolve their handle through the COPY, not the
    original) and carries no device snapshot; the original is untouched."""
    import copy
    import pickle
    import torch
    import vqae_amd
    from vqae_amd.model import VQAE
    m = VQAE.from_spec(vqae_amd.SPECS["tiny"])
    c = copy.deepcopy(m)
    assert c.encoder._owner() is c and c.decoder._owner() is c
    assert m.encoder._owner() is m and m.decoder._owner() is m
    assert c._native is None and c.encoder._native is None
    with torch.no_grad():
        c.encoder.in_stem.weight.add_(1.0)
    assert not torch.equal(c.encoder.in_stem.weight, m.encoder.in_stem.weight)
    r = pickle.loads(pickle.dumps(m))
    assert r.encoder._owner() is r and r.decoder._owner() is r
    for k, v in m.state_dict().items():
        assert torch.equal(r.state_dict()[k], v), k
    torch.save(m, tmp_path / "whole.pt")          # torch.save(model) pickles the module
