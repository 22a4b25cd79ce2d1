"""How many code indices would flip between a direct fp32 conv2 and an F(4x4,3x3) fp32 conv2 in the trunk, over N patches?  (CPU experiment)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wino43_numerics import O, wino
from conftest import load_golden
torch.set_num_threads(8)
name, n = sys.argv[1], int(sys.argv[2])
g = load_golden(f"model_{name}")
spec = O.SPECS[name]
p = O.make_params(spec, 0)
p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
size = {"B": 256, "A": 512, "C": 256}[name]
direct = O.conv_circular3x3
vq = "encoder.vq_layers.0."
tot = {2: 0, 4: 0}; rows = 0
for b0 in range(0, n, 8):
    x = O.make_patches(8, size, 1 + b0 // 8)
    out = {}
    for mode in (0, 2, 4):
        def conv(xx, w, mode=mode):
            return wino(xx, w, mode) if (mode and w.shape[0] == spec.channels) else direct(xx, w)
        O.conv_circular3x3 = conv
        z = O.encoder_features(x, p, spec)
        O.conv_circular3x3 = direct
        if spec.projection_dim > 0:
            z = torch.nn.functional.conv2d(z, p[vq + "proj_in.weight"], p[vq + "proj_in.bias"])
        flat = z.permute(0, 2, 3, 1).reshape(-1, z.shape[1]).contiguous()
        out[mode] = np.asarray(O.vq_argmin_p4(flat, p[vq + "embed"])[0]).reshape(-1)
    rows += out[0].size
    for m in (2, 4): tot[m] += int((out[m] != out[0]).sum())
    print(f"cfg {name}: {rows} rows: flips vs direct  F(2,3) {tot[2]}  F(4,3) {tot[4]}", flush=True)
