#!/usr/bin/env python3
"""Micro-benchmark of the fused projected quantiser (vqae_vq_projected_f32, csrc/vq_proj.hip) at cfg A's shape: N rows of
C = 128 channels, K = 256 codes, projection_dim 8.  The fused kernel alone is timed with the library's HIP events
(vqae_prof_begin / _end, class 3), i.e. without tier 2 / loss / index-conversion launches.

    python tools/vq_proj_bench.py [--rows 262144] [--reps 20] [--dtype f32]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vqae_amd  # noqa: E402
from vqae_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    C, K, N = a.channels, a.codes, a.rows
    w_in = (torch.randn(8, C, generator=g) / C ** 0.5).cuda()
    b_in = (torch.randn(8, generator=g) * 0.1).cuda()
    w_out = (torch.randn(C, 8, generator=g) / 8 ** 0.5).cuda()
    b_out = (torch.randn(C, generator=g) * 0.1).cuda()
    embed = torch.randn(K, 8, generator=g).cuda()
    x = torch.randn(N, C, generator=g).cuda()
    dt = None if a.dtype == "f32" else a.dtype
    for _ in range(3):
        vqae_amd.ops.vq_projected(x, w_in, b_in, embed, w_out, b_out, dtype=dt, idx_dtype=torch.uint8)
    torch.cuda.synchronize()
    lib = L.lib()
    L.check(lib.vqae_prof_begin(3, 4 * a.reps))
    for _ in range(a.reps):
        out, idx, loss, _, _ = vqae_amd.ops.vq_projected(x, w_in, b_in, embed, w_out, b_out, dtype=dt, idx_dtype=torch.uint8)
    torch.cuda.synchronize()
    ms, n, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
    L.check(lib.vqae_prof_end(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work)))
    avg = ms.value / max(1, n.value)
    byts = N * (2.0 * C * 4 + 36.0)
    print(json.dumps({"kernel": "vq_proj fused", "rows": N, "C": C, "K": K, "dtype": a.dtype, "launches": n.value,
                      "avg_us": round(avg * 1e3, 2), "alg_bytes": byts, "GBps": round(byts / (avg * 1e-3) / 1e9, 1),
                      "hbm_frac_of_8TBps": round(byts / (avg * 1e-3) / 1e9 / 8000.0, 4),
                      "codes_used": int(torch.unique(idx).numel()), "loss": float(loss)}))


if __name__ == "__main__":
    main()
