"""Debug helper (GPU box): run a small 16/32/64-channel model whose 64-channel blocks form a fused
conv2+conv3+next-conv1 chain and compare the pre-VQ activations with the CPU oracle element by element.
    VQAE_NO_TRUNK_FUSION=1 python tools/check_trunk_chain.py   # same with the 3-launch path
"""
import sys, os
sys.path.insert(0, '/root/repo')
import torch, vqae_amd
from oracle import vqae_oracle as O
spec = O.VQAESpec(stem=16, n_down=2, n_pre=0, n_post=2, n_enc=1, num_embeddings=16, projection_dim=0)   # channels 16,32,64 ; trunk 64ch chain of 3
vs = vqae_amd.VQAESpec(**spec.to_dict())
p = O.make_params(spec, 0)
x = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
p = O.calibrate_codebook(x, p, spec)
taps = {}
O.vqae_forward(x, p, spec, taps)
nat = vqae_amd.NativeVQAE(vs, p)
z = nat.encode_features(x.cuda()).cpu()     # NHWC [1,16,16,64]
ref = taps["z"].permute(0, 2, 3, 1)
err = (z - ref).abs()
print("max err", err.max().item(), "ref max", ref.abs().max().item())
bad = (err > 1e-3).nonzero()
print("n bad", bad.shape[0], "of", err.numel())
if bad.shape[0]:
    rows = (bad[:, 1] * 16 + bad[:, 2])
    print("bad pixel idx mod 8 histogram", torch.bincount(rows % 8, minlength=8).tolist())
    print("bad channel mod 4 hist", torch.bincount(bad[:, 3] % 4, minlength=4).tolist())
