"""CPU experiment: what would Winograd F(4x4,3x3) in fp32 for the trunk's conv2 do to parity?  The oracle's circular 3x3 conv is replaced
(C = 128 / 256 'same' blocks only) by a fp32 simulation of F(2x2,3x3) (what wino_trunk_kernel does today) or F(4x4,3x3); compared: indices
against the reference fixture, pre-VQ features against an fp64 evaluation."""
import sys, os, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vqae_oracle as O

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def wino(x, w, m):
    BT, G, AT = (BT4, G4, AT4) if m == 4 else (BT2, G2, AT2)
    t = m + 2
    dt = x.dtype
    U = (G @ w.double() @ G.T).to(dt)                                 # [O, C, t, t], rounded once
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode="circular")
    d = xp.unfold(2, t, m).unfold(3, t, m)                            # [B, C, th, tw, t, t]
    BTf, ATf = BT.to(dt), AT.to(dt)
    V = torch.einsum("ij,bcyxjk->bcyxik", BTf, d)
    V = torch.einsum("bcyxik,lk->bcyxil", V, BTf)
    B_, C, th, tw = V.shape[:4]
    Vp = V.permute(4, 5, 1, 0, 2, 3).reshape(t * t, C, -1)            # [pos, C, n]
    Up = U.permute(2, 3, 0, 1).reshape(t * t, U.shape[0], C)          # [pos, O, C]
    M = torch.bmm(Up, Vp).reshape(t, t, U.shape[0], B_, th, tw)
    Y = torch.einsum("ai,ijobyx->ajobyx", ATf, M)
    Y = torch.einsum("ajobyx,cj->acobyx", Y, ATf)                     # [m, m, O, B, th, tw]
    return Y.permute(3, 2, 4, 0, 5, 1).reshape(B_, U.shape[0], th * m, tw * m)


def main():
    from conftest import load_golden
    torch.set_num_threads(8)
    name = sys.argv[1] if len(sys.argv) > 1 else "B"
    g = load_golden(f"model_{name}")
    spec = O.SPECS[name]
    p = O.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    B = int(g["batch"])
    size = {"B": 256, "A": 512, "C": 256}[name]
    x = O.make_patches(B, size, 0)
    direct = O.conv_circular3x3
    ref_idx = g["idx"].astype(np.int64)

    p64 = {k: v.double() for k, v in p.items() if torch.is_tensor(v) and v.is_floating_point()}
    z_ex = O.encoder_features(x.double(), p64, spec)
    emb = p["encoder.vq_layers.0.embed"]
    vq = "encoder.vq_layers.0."
    def feats(mode):
        def conv(xx, w):
            if mode and w.shape[0] == spec.channels and xx.dtype == torch.float32:
                return wino(xx, w, mode)
            return direct(xx, w)
        O.conv_circular3x3 = conv
        try:
            return O.encoder_features(x, p, spec)
        finally:
            O.conv_circular3x3 = direct
    # sanity of the simulation itself in fp64
    xx = torch.randn(1, 8, 8, 8, dtype=torch.float64); ww = torch.randn(8, 8, 3, 3, dtype=torch.float64)
    for m in (2, 4):
        print("sim check fp64 m =", m, float((wino(xx, ww, m) - direct(xx, ww)).abs().max()))
    scale = float(z_ex.abs().max())
    for mode in (0, 2, 4):
        t0 = time.time()
        z = feats(mode)
        err = (z.double() - z_ex).abs()
        zz = z
        if spec.projection_dim > 0:
            zz = torch.nn.functional.conv2d(z, p[vq + "proj_in.weight"], p[vq + "proj_in.bias"])
        flat = zz.permute(0, 2, 3, 1).reshape(-1, zz.shape[1]).contiguous()
        idx, *_ = O.vq_argmin_p4(flat, emb)
        idx = np.asarray(idx).reshape(-1)
        mism = int((idx != ref_idx.reshape(-1)).sum())
        print(f"cfg {name} conv2 = {['direct', '', 'F(2,3)', '', 'F(4,3)'][mode]:7s}: feature err vs fp64 rms {float((err ** 2).mean().sqrt()) / scale:.3e} max {float(err.max()) / scale:.3e}"
              f"  idx mismatches vs fixture {mism} / {idx.size}   ({time.time() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
