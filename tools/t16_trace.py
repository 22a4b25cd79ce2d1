#!/usr/bin/env python3
"""Per-phase cycle breakdown of trunk16_kernel from the -DVQAE_T16_TRACE build (tools/build_trace.sh).
    VQAE_HIP_LIB=2d-vq-ae-2_amd/libvqae_hip_trace.so python tools/t16_trace.py --config B --dtype bf16 --batch 256"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    import vqae_amd
    from vqae_amd import _lib as L
    from vqae_amd.spec import encoder_block_names
    from oracle import vqae_oracle as O
    spec = vqae_amd.SPECS[a.config]
    nat = vqae_amd.NativeVQAE(spec, O.make_params(O.SPECS[a.config], 0), compute_dtype=a.dtype)
    names = encoder_block_names(spec)
    first = len(names) - spec.n_enc
    h = (512 if a.config == "A" else 256) >> spec.n_down
    c = names[first][2]
    x = torch.randn(a.batch, h, h, c, device="cuda") * 1.5
    nat.run_blocks("encoder", first, 4, x)                     # warm
    n_wg = a.batch * h * h // 128
    buf = torch.zeros(n_wg * 8 * 8, dtype=torch.int64, device="cuda")
    lib = ctypes.CDLL(L.LIB_PATH)
    assert lib.vqae_debug_t16_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    nat.run_blocks("encoder", first, 4, x)                     # the last launch's stamps remain (chain tail: NEXT = false) ...
    torch.cuda.synchronize()
    nat.run_blocks("encoder", first, 2, x)                     # ... so trace a 2-run: last stamps = block 2 (NEXT = false); rerun with count 3 below
    torch.cuda.synchronize()
    for count, label in ((2, "last block of a chain (no next conv1)"),):
        t = buf.cpu().numpy().reshape(n_wg, 8, 8)
        waves = c // 32
        t = t[:, :waves, :]
        d = np.diff(t[:, :, :6], axis=2)
        print(label)
        for i, nm in enumerate(["stage A rows (+barrier)", "conv2 K loop", "t2 -> LDS (+barrier) + x loads issued", "conv3", "epilogue x' store"]):
            print(f"  {nm:42s} median {np.median(d[:, :, i]):9.0f}  p90 {np.percentile(d[:, :, i], 90):9.0f} cycles")
        tot = t[:, :, 5] - t[:, :, 0]
        print(f"  total per wave: median {np.median(tot):.0f}; kernel span {(t[:, :, 5].max() - t[:, :, 0].min())} cycles for {n_wg} workgroups")
        starts = np.sort(t[:, 0, 0] - t[:, 0, 0].min())
        print("  workgroup start times (cycles) percentiles 25/50/75/100:", [int(np.percentile(starts, q)) for q in (25, 50, 75, 100)])


if __name__ == "__main__":
    main()
