"""Model hyper-parameters of the single-level Fixup VQ-AE that every shipped reference config
composes (conf/model/vq_ae.yaml:21-43), and the reference's state-dict
naming."""
from dataclasses import asdict, dataclass


@dataclass(frozen=True)
class VQAESpec:
    in_channels: int = 3          # conf/model/vq_ae.yaml:23
    stem: int = 8                 # vq_ae.yaml:24
    n_down: int = 4               # vq_ae.yaml:26
    n_pre: int = 1                # vq_ae.yaml:27
    n_post: int = 4               # vq_ae.yaml:28
    n_enc: int = 50               # vq_ae.yaml:29,39
    num_embeddings: int = 256     # conf/model/layers/vq/ema_vq.yaml:2
    projection_dim: int = 0       # 0: EMAVectorQuantizer; >0: ProjectedEMAVectorQuantizer2d (vq.py:157)
    commitment_cost: float = 1.0  # ema_vq.yaml:4
    decay: float = 0.99           # ema_vq.yaml:5
    laplace_alpha: float = 1e-5   # ema_vq.yaml:6
    block: str = "fixup"          # "fixup" (pre_activation_fixup.yaml) | "mbconv" (conf/model/encoder/efficientnetv2.yaml:3-4)
    expand_ratio: int = 4         # mbconv.yaml:32
    se_divisor: int = 4           # layers/misc/se.yaml:5
    bn_eps: float = 1e-5          # layers/misc/batchnorm2d.yaml:4

    @property
    def channels(self) -> int:    # embedding_dim = stem * 2**n_down (vq_ae.yaml:32)
        return self.stem * 2 ** self.n_down

    @property
    def code_dim(self) -> int:
        return self.projection_dim if self.projection_dim > 0 else self.channels

    def to_dict(self):
        return asdict(self)


# SURVEY.md §8 configuration legend
SPECS = {
    "A": VQAESpec(stem=8, n_down=4, n_enc=50, num_embeddings=256, projection_dim=8),     # reference default, 512^2
    "B": VQAESpec(stem=16, n_down=3, n_enc=50, num_embeddings=256, projection_dim=0),    # BASELINE configs 1-2, 256^2
    "C": VQAESpec(stem=32, n_down=3, n_enc=50, num_embeddings=1024, projection_dim=0),   # BASELINE config 4
    # mid-size models that reach the production kernels of A/B/C on 128x128 inputs (per-block parity taps)
    "mid": VQAESpec(stem=32, n_down=2, n_pre=1, n_post=4, n_enc=3, num_embeddings=64, projection_dim=0),
    "mid16": VQAESpec(stem=16, n_down=2, n_pre=1, n_post=2, n_enc=2, num_embeddings=32, projection_dim=0),
    "midA": VQAESpec(stem=8, n_down=3, n_pre=1, n_post=2, n_enc=2, num_embeddings=64, projection_dim=8),
    "midC": VQAESpec(stem=64, n_down=2, n_pre=1, n_post=2, n_enc=2, num_embeddings=64, projection_dim=0),   # 64@128 -> 128@64 -> 256@32: cfg C's levels
    "midW": VQAESpec(stem=32, n_down=2, n_pre=1, n_post=1, n_enc=1, num_embeddings=32, projection_dim=0),    # on 256x256: 32@256 (column-blocked tiles) -> 64@128 -> 128@64
    "tiny": VQAESpec(stem=8, n_down=2, n_pre=1, n_post=1, n_enc=2, num_embeddings=16, projection_dim=0),
    "tinyP": VQAESpec(stem=8, n_down=2, n_pre=1, n_post=1, n_enc=2, num_embeddings=32, projection_dim=8),
    # MBConv / EfficientNetV2 variant
    "tinyM": VQAESpec(stem=8, n_down=2, n_pre=1, n_post=1, n_enc=2, num_embeddings=16, block="mbconv"),
    "BM": VQAESpec(stem=16, n_down=3, n_enc=50, num_embeddings=256, block="mbconv"),
}


def encoder_block_names(spec: VQAESpec):
    """[(state-dict prefix, mode, cin, cout)] in execution order (model.py:198-208; conv_block.py:35-47)."""
    out, c = [], spec.stem
    for lvl in range(spec.n_down):
        base, b = f"encoder.down_layers.0.layers.{lvl}.layers.", 0
        for _ in range(spec.n_pre):
            out.append((base + str(b), "same", c, c)); b += 1
        out.append((base + str(b), "down", c, 2 * c)); b += 1
        for _ in range(spec.n_post):
            out.append((base + str(b), "same", 2 * c, 2 * c)); b += 1
        c *= 2
    out += [(f"encoder.pre_enc_layers.0.{i}", "same", c, c) for i in range(spec.n_enc)]
    return out


def decoder_block_names(spec: VQAESpec):
    """(model.py:278-289; conv_block.py:72-88)."""
    c = spec.channels
    out = [(f"decoder.post_enc_layers.0.{i}", "same", c, c) for i in range(spec.n_enc)]
    for lvl in range(spec.n_down):
        base, b = f"decoder.up_layers.0.layers.{lvl}.layers.", 0
        for _ in range(spec.n_pre):
            out.append((base + str(b), "same", c, c)); b += 1
        out.append((base + str(b), "up", c, c // 2)); b += 1
        for _ in range(spec.n_post):
            out.append((base + str(b), "same", c // 2, c // 2)); b += 1
        c //= 2
    return out
