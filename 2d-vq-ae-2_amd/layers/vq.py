"""Drop-in mirrors of vq_ae.layers.vq.{EMAVectorQuantizer, ProjectedEMAVectorQuantizer2d}
(reference vq_ae/layers/vq.py:6-154, 157-192) backed by libvqae_hip.so.

Same constructor kwargs, buffer names (embed / embed_avg / cluster_size / first_pass, vq.py:27-34),
forward contract `(quantized, encoding_indices, loss)` ("don't change this order", vq.py:148-154) and
exception types.  Select through Hydra with
    _target_: vqae_amd.layers.vq.EMAVectorQuantizer
"""
import torch
from torch import nn

from .. import _lib as L
from .. import ops


def fused_all_reduce_stats(counts, dw):
    """`all_reduce(new_cluster_size)` + `all_reduce(dw)` of _update_ema (vq.py:57-58) as ONE all-reduce of the flat
    [K] + [K * D] buffer (latency-bound messages: 1 KB + 8..128 KB).  Element-wise SUM: fusing changes no element's
    reduction, only the number of collectives (tests/test_driver_cpu.py checks equality under 2 ranks)."""
    flat = torch.cat([counts.reshape(-1), dw.reshape(-1)])
    torch.distributed.all_reduce(flat)
    return flat[: counts.numel()].reshape_as(counts), flat[counts.numel():].reshape_as(dw)


class EMAVectorQuantizer(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float, decay: float,
                 laplace_alpha: float):
        super().__init__()
        embed = torch.randn(num_embeddings, embedding_dim)
        self.register_buffer("embed", embed)                       # e_i   (vq.py:27)
        self.register_buffer("embed_avg", embed.clone())           # m_i   (vq.py:28)
        self.register_buffer("cluster_size", torch.zeros(num_embeddings))  # N_i (vq.py:29)
        self.register_buffer("first_pass", torch.as_tensor(1))     # vq.py:33
        self.commitment_cost = commitment_cost
        self.decay = decay
        self.laplace_alpha = laplace_alpha
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings

    def embed_code(self, embed_idx):                               # vq.py:44-45
        return ops.embed_code(embed_idx, self.embed)

    # ---- training-mode bookkeeping (vq.py:47-94) ------------------------------------------------
    @torch.no_grad()
    def _update_ema(self, flat_input, encoding_indices):
        counts, dw = ops.vq_code_stats(flat_input, encoding_indices, self.num_embeddings)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            counts, dw = fused_all_reduce_stats(counts, dw)
        ops.vq_ema_update(self.embed, self.embed_avg, self.cluster_size, counts.contiguous(), dw.contiguous(),
                          self.decay, self.laplace_alpha)

    @torch.no_grad()
    def _init_ema(self, flat_input):
        mean = flat_input.mean(dim=0)
        std = flat_input.std(dim=0)
        cluster_size = flat_input.size(dim=0)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            ws = torch.distributed.get_world_size()
            both = torch.stack([mean, std])
            torch.distributed.all_reduce(both)                     # vq.py:82-87 (mean of per-rank stds, sic)
            mean, std = both[0] / ws, both[1] / ws
            cluster_size *= ws
        self.embed.mul_(std)
        self.embed.add_(mean)
        self.embed_avg.copy_(self.embed)
        self.cluster_size.data.add_(cluster_size / self.num_embeddings)
        self.first_pass.mul_(0)

    def forward(self, inputs):
        ndim = inputs.dim()
        assert ndim >= 3                                           # vq.py:98
        if inputs.shape[1] != self.embedding_dim:                  # vq.py:100-104
            raise NotImplementedError(
                'VQ dim != channel dim not supported;'
                f' found channel dim of {inputs.shape[1]}, expected {self.embedding_dim}')
        if ndim > 5:
            # the reference's distance exponent is p = inputs.dim() (vq.py:121-129); the kernels implement p = 3, 4, 5
            raise NotImplementedError(f'inputs of rank 3 .. 5 (p = 3, 4, 5) are implemented; got a {ndim}-D input')
        with torch.no_grad():
            x = inputs.detach().float()
            if ndim == 4:
                cl = ops.nchw_to_nhwc(x)
            else:                                                      # channel last (vq.py:107-113): [B, L, D] / [B, d, h, w, D]
                cl = x.permute(0, *range(2, ndim), 1).contiguous()
            D = cl.shape[-1]
            flat_input = cl.reshape(-1, D)
            if self.first_pass and self.training:
                self._init_ema(flat_input)
            q_flat, idx, loss, _ = ops.vq_forward(flat_input, self.embed, self.commitment_cost, p=ndim)
            if self.training:
                self._update_ema(flat_input, idx)
            q_cl = q_flat.reshape(cl.shape)                             # = inputs + (q - inputs), vq.py:146
            quantized = ops.nhwc_to_nchw(q_cl) if ndim == 4 else q_cl.permute(0, -1, *range(1, ndim - 1)).contiguous()
            encoding_indices = idx.reshape(cl.shape[:-1])
        return quantized, encoding_indices, loss


class ProjectedEMAVectorQuantizer2d(EMAVectorQuantizer):
    """proj_out(VQ(proj_in(x))) with 1x1 convs (vq.py:157-192)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float, decay: float,
                 laplace_alpha: float, projection_dim: int):
        super().__init__(num_embeddings, projection_dim, commitment_cost, decay, laplace_alpha)
        self.proj_in = nn.Conv2d(embedding_dim, projection_dim, kernel_size=1)     # parameter holders
        self.proj_out = nn.Conv2d(projection_dim, embedding_dim, kernel_size=1)
        self._packed = None

    def _weights(self):
        key = (self.proj_in.weight._version, self.proj_out.weight._version, self.proj_in.weight.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.pack_conv_weight(self.proj_in.weight.detach()),
                            ops.pack_conv_weight(self.proj_out.weight.detach()))
        return self._packed[1], self._packed[2]

    def forward(self, inputs):
        assert inputs.dim() == 4
        if self.embedding_dim == 8 and not self.training:
            # eval mode, the reference default projection_dim: one fused launch (csrc/vq_proj.hip)
            with torch.no_grad():
                x = ops.nchw_to_nhwc(inputs.detach().float())
                B, H, W, C = x.shape
                out, idx, loss, _, _ = ops.vq_projected(x.reshape(-1, C), self.proj_in.weight.detach(), self.proj_in.bias.detach(),
                                                        self.embed, self.proj_out.weight.detach(), self.proj_out.bias.detach(),
                                                        self.commitment_cost)
                return ops.nhwc_to_nchw(out.reshape(B, H, W, C)), idx.reshape(B, H, W), loss
        w_in, w_out = self._weights()
        with torch.no_grad():
            x = ops.nchw_to_nhwc(inputs.detach().float())
            z = ops.conv2d(x, w_in, self.embedding_dim, 1, bias_vec=self.proj_in.bias.detach())
            B, H, W, D = z.shape
            flat = z.reshape(-1, D)
            if self.first_pass and self.training:
                self._init_ema(flat)
            q_flat, idx, loss, _ = ops.vq_forward(flat, self.embed, self.commitment_cost)
            if self.training:
                self._update_ema(flat, idx)
            out = ops.conv2d(q_flat.reshape(B, H, W, D), w_out, self.proj_out.out_channels, 1,
                             bias_vec=self.proj_out.bias.detach())
            return ops.nhwc_to_nchw(out), idx.reshape(B, H, W), loss
