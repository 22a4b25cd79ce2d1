#!/usr/bin/env python3
"""Micro-benchmark of vqae_vq_forward_f32 (plain EMAVectorQuantizer lookup) at a model's shape: N rows, D channels, K codes.
The search kernel class alone is timed with the library's HIP events (class 3).

    python tools/vq_bench.py [--rows 262144] [--dim 128] [--codes 256] [--reps 10]
"""
import argparse, ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vqae_amd  # noqa: E402
from vqae_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=262144); ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--codes", type=int, default=256); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
z = torch.randn(a.rows, a.dim, generator=g).cuda()
e = torch.randn(a.codes, a.dim, generator=g).cuda()
for _ in range(2):
    vqae_amd.ops.vq_forward(z, e, idx_dtype=torch.int32, want_q=False)
torch.cuda.synchronize()
lib = L.lib()
L.check(lib.vqae_prof_begin(3, 8 * a.reps))
for _ in range(a.reps):
    _, idx, loss, _ = vqae_amd.ops.vq_forward(z, e, idx_dtype=torch.int32, want_q=False)
torch.cuda.synchronize()
ms, n, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
L.check(lib.vqae_prof_end(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work)))
_, idx_m, _, _ = vqae_amd.ops.vq_forward(z, e, idx_dtype=torch.int32, want_q=False, want_margin=True)     # margin output: the exact scan
print(json.dumps({"rows": a.rows, "D": a.dim, "K": a.codes, "launches": n.value, "avg_us": round(ms.value / max(1, n.value) * 1e3, 1),
                  "same_as_exact_scan": bool(torch.equal(idx, idx_m)), "loss": float(loss)}))
