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
    ap.add_argument("--dbg", type=int, default=0, help="experiment bits: 1 no staging loads, 2 no residual loads, 4 no stores")
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
    assert lib.vqae_debug_t16_dbg(a.dbg) == 0
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
        info = t[:, 0, 6]
        xcc, hw = (info >> 32) & 15, info & 0xFFFF
        cu = (hw >> 8) & 0xFF
        key = xcc * 256 + cu
        print("  distinct (xcc, cu/sh/se) keys:", len(np.unique(key)), "; workgroups per key min/max:", np.bincount(key.astype(np.int64))[np.unique(key)].min(), np.bincount(key.astype(np.int64))[np.unique(key)].max())
        # per CU: conv2 intervals of its workgroups -> fraction of conv2 time spent with the other slot also in conv2
        ov = tot_c2 = 0
        for k_ in np.unique(key)[:64]:
            ids = np.nonzero(key == k_)[0]
            iv = sorted((int(t[i, 0, 1]), int(t[i, 0, 2])) for i in ids)
            for a_ in range(len(iv)):
                tot_c2 += iv[a_][1] - iv[a_][0]
                for b_ in range(a_ + 1, len(iv)):
                    ov += 2 * max(0, min(iv[a_][1], iv[b_][1]) - max(iv[a_][0], iv[b_][0]))
        print(f"  conv2 time overlapped with the CU's other workgroup also in conv2: {ov / max(tot_c2, 1):.2f}")
        fr, du = [], []
        for k_ in np.unique(key):
            ids = np.nonzero(key == k_)[0]
            iv = [(int(t[i, 0, 1]), int(t[i, 0, 2])) for i in ids]
            for a_ in range(len(iv)):
                o = sum(max(0, min(iv[a_][1], iv[b_][1]) - max(iv[a_][0], iv[b_][0])) for b_ in range(len(iv)) if b_ != a_)
                fr.append(o / (iv[a_][1] - iv[a_][0])); du.append(iv[a_][1] - iv[a_][0])
        fr, du = np.array(fr), np.array(du)
        for lo, hi in ((0, 0.05), (0.05, 0.3), (0.3, 0.7), (0.7, 0.95), (0.95, 1.01)):
            m = (fr >= lo) & (fr < hi)
            if m.any():
                print(f"    conv2 overlapped {lo:.2f}-{hi:.2f}: n = {int(m.sum()):5d}, conv2 median {np.median(du[m]):8.0f} cycles")
        first = ids_first = np.nonzero(np.arange(n_wg) < 512)[0]
        starts = np.sort(t[:, 0, 0] - t[:, 0, 0].min())
        print("  workgroup start times (cycles) percentiles 25/50/75/100:", [int(np.percentile(starts, q)) for q in (25, 50, 75, 100)])


if __name__ == "__main__":
    main()
