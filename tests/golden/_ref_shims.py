"""Import shims used ONLY by tests/golden/make_golden.py, in the build container, to import the
unmodified reference modules from /root/reference (which never travels to the GPU box).

The reference imports hydra / omegaconf / pytorch_lightning / torchvision at module top
(vq_ae/model.py:3-7, vq_ae/layers/conv_block.py:9,12, utils/conf_helpers.py:8-9); none is
installed and there is no network, so minimal stand-ins for the *glue* are registered in
sys.modules.  No arithmetic is shimmed: every conv / elu / cdist / upsample the reference
executes is torch's own.
"""
import collections
import collections.abc
import importlib
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _resolve(path: str):
    mod, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def instantiate(config=None, *args, **kwargs):
    """~hydra.utils.instantiate for the subset the reference uses: `_target_` class paths,
    kwargs overrides, `_recursive_` (nested `_target_` dicts are instantiated unless False)."""
    if config is None:
        return None
    conf = dict(config)
    conf.update(kwargs)
    target = conf.pop("_target_")
    recursive = conf.pop("_recursive_", True)
    conf.pop("_convert_", None)
    conf.pop("_partial_", None)
    if recursive:
        conf = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in conf.items()}
    fn = _resolve(target) if isinstance(target, str) else target
    return fn(*args, **conf)


def install():
    if not hasattr(collections, "Sequence"):          # conv_block.py:1 `from collections import Sequence`
        collections.Sequence = collections.abc.Sequence

    oc = types.ModuleType("omegaconf")

    class DictConfig(dict):
        pass

    class ListConfig(list):
        pass

    class OmegaConf:
        @staticmethod
        def register_new_resolver(*a, **k):
            return None

        @staticmethod
        def save(*a, **k):
            return None

    oc.DictConfig, oc.ListConfig, oc.OmegaConf, oc.MISSING = DictConfig, ListConfig, OmegaConf, "???"
    sys.modules["omegaconf"] = oc

    hy = types.ModuleType("hydra")
    hy.main = lambda *a, **k: (lambda f: f)
    hy.compose = lambda *a, **k: None
    hy.initialize_config_dir = lambda *a, **k: None
    hu = types.ModuleType("hydra.utils")
    hu.instantiate = instantiate
    hu.call = instantiate
    hy.utils = hu
    hc = types.ModuleType("hydra.core")
    hg = types.ModuleType("hydra.core.global_hydra")

    class GlobalHydra:
        @staticmethod
        def instance():
            return GlobalHydra()

        def clear(self):
            return None

    hg.GlobalHydra = GlobalHydra
    hc.global_hydra = hg
    sys.modules.update({"hydra": hy, "hydra.utils": hu, "hydra.core": hc, "hydra.core.global_hydra": hg})

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            return None

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Callback = _Dummy
    pl.Trainer = _Dummy
    pl.LightningDataModule = _Dummy
    plu = types.ModuleType("pytorch_lightning.utilities")
    ple = types.ModuleType("pytorch_lightning.utilities.exceptions")
    ple.MisconfigurationException = type("MisconfigurationException", (Exception,), {})
    plt = types.ModuleType("pytorch_lightning.utilities.types")
    plt.STEP_OUTPUT = object
    pl.utilities = plu
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": plu,
                        "pytorch_lightning.utilities.exceptions": ple,
                        "pytorch_lightning.utilities.types": plt})

    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    sys.modules.update({"torchvision": tv, "torchvision.utils": tvu})

    tq = types.ModuleType("tqdm")
    tq.tqdm = lambda it, *a, **k: it
    sys.modules.setdefault("tqdm", tq)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


# ---- hand-built config mirroring the YAML tree (Hydra's fork cannot be installed offline) ----
def _conv(kernel_size, stride=1, padding=0, bias=True, padding_mode="zeros", target="torch.nn.Conv2d"):
    # conf/model/layers/conv_block/conv_layer/conv2d.yaml
    return {"_target_": target, "in_channels": None, "out_channels": None, "kernel_size": kernel_size,
            "stride": stride, "padding": padding, "dilation": 1, "groups": 1, "bias": bias,
            "padding_mode": padding_mode}


def fixup_conf(n_layers):
    # conf/model/layers/conv_block/pre_activation_fixup.yaml
    proj = lambda: _conv(1, bias=False)                                        # proj2d.yaml
    down = lambda: _conv(2, stride=2, bias=False, padding_mode="circular")     # down2d.yaml
    same = lambda: _conv(3, padding=1, bias=False, padding_mode="circular")    # same2d.yaml
    out_z = lambda: _conv(3, padding=1, bias=False)                            # out2d.yaml (skip: zeros)
    upres = lambda: _conv(1, bias=False, target="vq_ae.layers.conv.ResizeConv2D")  # up2dresize.yaml
    return {
        "_target_": "vq_ae.layers.conv_block.PreActFixupResBlock", "_recursive_": False,
        "in_channels": None, "out_channels": None, "mode": None, "n_layers": n_layers,
        "bottleneck_divisor": 1,
        "activation": {"_target_": "torch.nn.ELU", "alpha": 1.0},
        "conv_conf": {
            "down": {"branch_conv1": proj(), "branch_conv2": down(), "branch_conv3": proj(), "skip_conv": down()},
            "up": {"branch_conv1": proj(), "branch_conv2": upres(), "branch_conv3": proj(), "skip_conv": upres()},
            "same": {"branch_conv1": proj(), "branch_conv2": same(), "branch_conv3": proj(), "skip_conv": proj()},
            "out": {"branch_conv1": proj(), "branch_conv2": same(), "branch_conv3": proj(), "skip_conv": out_z()},
        },
    }


def mbconv_conf(spec):
    # conf/model/layers/conv_block/mbconv.yaml (+ activation/silu.yaml, misc/se.yaml, misc/batchnorm2d.yaml)
    proj = lambda: _conv(1, bias=False)                                        # proj2d.yaml
    down = lambda: _conv(2, stride=2, bias=False, padding_mode="circular")     # down2d.yaml
    same = lambda: _conv(3, padding=1, bias=False, padding_mode="circular")    # same2d.yaml / out2d.yaml
    up = lambda: {"_target_": "torch.nn.ConvTranspose2d", "in_channels": None, "out_channels": None,   # up2d.yaml
                  "kernel_size": 2, "stride": 2, "padding": 0, "output_padding": 0, "groups": 1, "bias": False,
                  "dilation": 1, "padding_mode": "zeros"}
    return {
        "_target_": "vq_ae.layers.conv_block.MBConv", "_recursive_": False,
        "in_channels": None, "out_channels": None, "mode": None, "expand_ratio": spec.expand_ratio,
        "activation_conf": {"_target_": "torch.nn.SiLU"},
        "se_conf": {"_target_": "vq_ae.layers.misc.SELayer", "in_channels": None, "out_channels": None,
                    "bottleneck_divisor": spec.se_divisor},
        "batchnorm_conf": {"_target_": "torch.nn.BatchNorm2d", "num_features": None, "eps": 1e-05, "momentum": 0.1,
                           "affine": True, "track_running_stats": True},
        "conv_conf": {
            "down": {"branch_conv1": proj(), "branch_conv2": down(), "branch_conv3": proj(), "skip_conv": down()},
            "up": {"branch_conv1": proj(), "branch_conv2": up(), "branch_conv3": proj(), "skip_conv": up()},
            "same": {"branch_conv1": proj(), "branch_conv2": same(), "branch_conv3": proj(), "skip_conv": proj()},
            "out": {"branch_conv1": proj(), "branch_conv2": same(), "branch_conv3": proj(), "skip_conv": same()},
        },
    }


def vqae_conf(spec):
    """Nested dict equal to what Hydra composes from conf/model/vq_ae.yaml for `spec`."""
    fx = mbconv_conf(spec) if getattr(spec, "block", "fixup") == "mbconv" else fixup_conf(spec.n_layers)
    if spec.projection_dim > 0:
        vq = {"_target_": "vq_ae.layers.vq.ProjectedEMAVectorQuantizer2d", "num_embeddings": spec.num_embeddings,
              "embedding_dim": spec.channels, "commitment_cost": spec.commitment_cost, "decay": spec.decay,
              "laplace_alpha": spec.laplace_alpha, "projection_dim": spec.projection_dim}
    else:
        vq = {"_target_": "vq_ae.layers.vq.EMAVectorQuantizer", "num_embeddings": spec.num_embeddings,
              "embedding_dim": spec.channels, "commitment_cost": spec.commitment_cost, "decay": spec.decay,
              "laplace_alpha": spec.laplace_alpha}
    stem_in = _conv(3, padding=1)
    stem_in.update(in_channels=spec.in_channels, out_channels=spec.stem)
    stem_out = _conv(3, padding=1)
    stem_out.update(in_channels=spec.stem, out_channels=spec.in_channels)
    enc = {
        "_target_": "vq_ae.model.Encoder", "_recursive_": False,
        "stem_conf": stem_in,
        "down_block_conf": {"_target_": "vq_ae.layers.conv_block.DownBlock", "_recursive_": False,
                            "in_channels": None, "n_down": spec.n_down, "n_pre_layers": spec.n_pre,
                            "n_post_layers": spec.n_post, "conv_conf": fx},
        "n_pre_enc_layers": spec.n_enc,
        "vq_conf": {"_target_": "utils.conf_helpers.instantiate_dictified_listconf", "_recursive_": False,
                    "0": vq},
        "conv_block_conf": fx,
        "shortcut_block_conf": None,
    }
    dec = {
        "_target_": "vq_ae.model.Decoder", "_recursive_": False,
        "n_enc_layers": 1,
        "stem_conf": stem_out,
        "up_block_conf": {"_target_": "vq_ae.layers.conv_block.UpBlock", "_recursive_": False,
                          "out_channels": None, "n_up": spec.n_down, "n_pre_layers": spec.n_pre,
                          "n_post_layers": spec.n_post, "conv_conf": fx},
        "n_post_enc_layers": spec.n_enc,
        "conv_block_conf": fx,
        "shortcut_block_conf": None,
    }
    return {
        "optim_conf": {"_target_": "torch.optim.AdamW", "lr": 1e-4},
        "loss_f_conf": {"_target_": "torch.nn.HuberLoss", "reduction": "mean", "delta": 1.0},
        "encoder_conf": enc,
        "decoder_conf": dec,
    }


def build_reference_model(spec, params):
    """Instantiate the reference VQAE for `spec` and load `params` (state-dict names)."""
    install()
    from vq_ae.model import VQAE  # noqa: the reference, unmodified
    model = VQAE(**vqae_conf(spec))
    sd = model.state_dict()
    missing = [k for k in sd if k not in params
               and not k.endswith(("embed_avg", "cluster_size", "first_pass", "num_batches_tracked"))]
    assert not missing, missing
    extra = [k for k in params if k not in sd]
    assert not extra, extra
    full = {k: v.clone() for k, v in params.items()}
    vq = "encoder.vq_layers.0."
    full[vq + "embed_avg"] = params[vq + "embed"].clone()
    full[vq + "cluster_size"] = torch.zeros(spec.num_embeddings)
    full[vq + "first_pass"] = torch.as_tensor(0)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            full[k] = torch.as_tensor(1)
    for k, v in full.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    model.load_state_dict(full)
    model.eval()
    return model
