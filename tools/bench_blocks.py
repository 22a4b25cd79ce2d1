#!/usr/bin/env python3
"""Micro-benchmark of a run of residual blocks through the handle's own dispatch (vqae_run_blocks): e.g. the 50-block
trunk of cfg B / cfg C in one compute dtype, without the rest of the model.  Used for kernel iteration and for
rocprofv3 --pmc passes over one kernel family.

    python tools/bench_blocks.py --config B --dtype bf16 --batch 256 --side encoder --first 18 --count 50 --reps 5
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--side", default="encoder")
    ap.add_argument("--first", type=int, default=-1, help="first block (default: first trunk block)")
    ap.add_argument("--count", type=int, default=50)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--size", type=int, default=0, help="input resolution of the model (default 256, cfg A 512)")
    a = ap.parse_args()
    import vqae_amd
    from vqae_amd.spec import decoder_block_names, encoder_block_names
    from oracle import vqae_oracle as O
    spec = vqae_amd.SPECS[a.config]
    params = O.make_params(O.SPECS[a.config], 0)
    nat = vqae_amd.NativeVQAE(spec, params, compute_dtype=None if a.dtype == "f32" else a.dtype)
    names = encoder_block_names(spec) if a.side == "encoder" else decoder_block_names(spec)
    first = a.first if a.first >= 0 else (len(names) - spec.n_enc if a.side == "encoder" else 0)
    size = a.size or (512 if a.config == "A" else 256)
    # resolution at block `first`
    h = size
    for _, mode, _, _ in (names[:first] if a.side == "encoder" else []):
        h = h // 2 if mode == "down" else h
    if a.side == "decoder":
        h = size >> spec.n_down
        for _, mode, _, _ in names[:first]:
            h = h * 2 if mode == "up" else h
    cin = names[first][2]
    x = torch.randn(a.batch, h, h, cin, device="cuda") * 1.5
    y = nat.run_blocks(a.side, first, a.count, x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        y = nat.run_blocks(a.side, first, a.count, x)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    px = a.batch * h * h
    flops = sum(2.0 * px * (ci * max(ci, co) + 9 * max(ci, co) ** 2 + max(ci, co) * co) for _, m, ci, co in names[first:first + a.count] if m == "same")
    print(f"{a.config} {a.dtype} B={a.batch} blocks {names[first][0]} +{a.count} @ {h}x{h}x{cin}: best {best * 1e3:.3f} ms "
          f"= {best / a.count * 1e6:.1f} us/block, {flops / best / 1e12:.1f} TFLOP/s (same-blocks, direct); finite={bool(torch.isfinite(y).all())}")


if __name__ == "__main__":
    main()
