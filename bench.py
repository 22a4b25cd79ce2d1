#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native VQ-AE hot path on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W            (N = 1: run directly)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic patches already resident in HBM:
VQAE.forward = conv encoder -> p=4 vector quantiser -> conv decoder (reference vq_ae/model.py:41-48)
on BASELINE.json configs[1]: batch 256 of 256x256x3 fp32 patches -> 32x32 codes, 256-entry/128-dim
codebook (SURVEY.md §8 "cfg B"), per GPU.  With N > 1 the patch batch is sharded (weak scaling:
256 patches per rank) and each step ends with the path's only collective, an RCCL all-gather of the
uint8 code grids.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak (= fp32 vector peak)
PEAK_16BIT_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 matrix peak
PEAK_HBM_GBS = 8000.0

# BASELINE.json `configs` that a (--config, --dtype, --batch, --mode) combination reproduces on one GPU
BASELINE_CONFIGS = {("B", "f32", 256, "full"): 1, ("A", "bf16", 256, "full"): 2, ("C", "f16", 256, "full"): 3}
# short legs attached to the default (driver-run) line under "other_configs": (key, config, dtype, mode, prof class)
OTHER_LEGS = [("configs[2]", "A", "bf16", "full", 1),                 # 512x512 deeper encoder, bf16, batch 256
              ("configs[3]_per_gpu", "C", "f16", "full", 1),          # 1024-entry / 256-dim codebook, fp16, 256 per GPU
              ("configs[4]_encode_only_f16", "A", "f16", "encode", 1),  # the whole-slide extractor's encoder pass (cfg A, fp16)
              ("cfgB_bf16_full", "B", "bf16", "full", 1),
              ("cfgB_f16_encode", "B", "f16", "encode", 1),
              ("north_star_vq_argmin_D8", "A", "bf16", "full", 3),    # fused projected quantiser: HBM fraction
              ("north_star_conv1x1", "B", "f32", "full", 2)]          # stand-alone 1x1 conv: fp32 MFMA fraction


def host_cores():
    """CPU share of this process (the GPU box gives 16 cores per GPU; os.cpu_count() reports the host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("VQAE_CPU_THREADS", "16"))))


def log(msg):
    print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.perf_counter()


def kernel_source_hash(files):
    """sha256 over the HIP sources of the dominant kernel: PMC traffic figures are only valid for the code they were
    measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "2d-vq-ae-2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(key, kernel=None):
    """HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes (FETCH_SIZE doubled per
    MI355X_MICROARCH.md, + WRITE_SIZE), as recorded in profiles/r0N_pmc_traffic.json by tools/pmc_traffic.py.  Each
    entry carries the hash of the kernel sources it was measured on; a figure for other code is refused (null)."""
    try:
        d = None
        for rnd in ("r03", "r02"):                  # newest round first; an entry is only valid for the sources it names
            path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
            if os.path.exists(path) and key in json.load(open(path)):
                d = json.load(open(path))[key]
                if d["source_hash"] == kernel_source_hash(d["sources"]):
                    break
        if d is None:
            raise KeyError(key)
        if d["source_hash"] != kernel_source_hash(d["sources"]):
            return None, f"stale: measured on sources {d['source_hash']}, kernel has changed since"
        if kernel and not d["kernel"].startswith(kernel):
            return None, f"not measured for {kernel} (the entry is for {d['kernel']})"
        return d["bytes_per_launch"], f"rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, sources {d['source_hash']}"
    except Exception as e:                      # no profile for this configuration
        return None, f"not measured ({type(e).__name__})"


def synth_patches_u8(batch, size, batch_no, device):
    """uint8 ~ U{0..255} NHWC from PCG64(1000 + batch_no)."""
    rng = np.random.Generator(np.random.PCG64(1000 + batch_no))
    return torch.from_numpy(rng.integers(0, 256, size=(batch, size, size, 3), dtype=np.uint8)).to(device)


def normalise(u8):
    """albumentations Normalize + ToTensorV2 -> NCHW fp32 (camelyon16_transforms.yaml:15-23)."""
    mean = torch.tensor([0.7279, 0.5955, 0.7762], device=u8.device) * 255.0
    inv = 1.0 / (torch.tensor([0.2419, 0.3083, 0.1741], device=u8.device) * 255.0)
    return ((u8.float() - mean) * inv).permute(0, 3, 1, 2).contiguous()


def synth_weights(spec_name):
    """Procedural non-zero weights.  bench.py may use the oracle's generator: it is data, not compute."""
    from oracle import vqae_oracle as O
    return O.make_params(O.SPECS[spec_name], 0)


def cpu_baseline(spec_name, size, params, embed, sample_batch, iters):
    """The CPU oracle (PyTorch-CPU fp32 restatement of the reference path) timed on this box's host
    cores, on a bounded sample of the same workload."""
    from oracle import vqae_oracle as O
    spec = O.SPECS[spec_name]
    p = dict(params)
    p["encoder.vq_layers.0.embed"] = embed.cpu()
    cores = host_cores()
    torch.set_num_threads(cores)
    x = O.make_patches(sample_batch, size, 0)
    taps = {}
    O.vqae_forward(x, p, spec, taps)                 # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        taps = {}
        out, _ = O.vqae_forward(x, p, spec, taps)
    dt = (time.perf_counter() - t0) / iters
    return dict(value=sample_batch / dt, unit="patches/s", cores=cores, kind="port",
                sample=f"{iters} x full VQAE.forward of {sample_batch} patches ({size}x{size}x3 fp32), "
                       f"PyTorch-CPU oracle, {cores} threads"), x, out, taps["idx"]


def run_leg(args, ctx, headline):
    """One measurement: W warm-up + K timed steps of (args.config, args.dtype, args.mode) at args.batch patches per GPU,
    bracketed by barrier + synchronize; returns the result dict on rank 0 (None elsewhere)."""
    import vqae_amd
    from vqae_amd import _lib as L
    world, rank, dev, use_dist, dist = ctx["world"], ctx["rank"], ctx["dev"], ctx["use_dist"], ctx["dist"]
    size = 512 if args.config == "A" else 256
    spec = vqae_amd.SPECS[args.config]
    params = synth_weights(args.config)
    log("weights generated")
    nat = vqae_amd.NativeVQAE(spec, params, compute_dtype=args.dtype)
    nat.reserve(args.batch, size, size)
    log("native handle created, workspace reserved")

    # codebook ~ N(mu_z, sigma_z) of a calibration batch (mirrors _init_ema, vq.py:76-94)
    calib = normalise(synth_patches_u8(min(args.batch, 8), size, 99, dev))
    embed = nat.calibrate_codebook(calib, params["encoder.vq_layers.0.embed"])

    x = normalise(synth_patches_u8(args.batch, size, 1000 * rank + 1, dev))     # resident in HBM
    B = args.batch
    zh = size // nat.factor
    idx_dtype = torch.uint8 if spec.num_embeddings <= 256 else torch.int32
    gathered = torch.empty((world * B, zh, zh), dtype=idx_dtype, device=dev) if use_dist else None

    def step():
        if args.mode == "full":
            out, idx, loss = nat.forward(x, "NCHW", idx_dtype=idx_dtype)
        else:
            out = None
            _, idx, loss = nat.encode(x, "NCHW", idx_dtype=idx_dtype, want_q=False)
        if use_dist:
            dist.all_gather_into_tensor(gathered, idx)       # reassemble the per-slide code grids
        return out, idx, loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    log("codebook calibrated, inputs resident")
    for _ in range(args.warmup):
        out, idx, loss = step()
    barrier()
    log("warm-up done")

    # timed region; the dominant kernel class is timed with HIP events on the launch stream
    lib = L.lib()
    n_launch_max = args.steps * 256
    prof_on = rank == 0 and args.prof_class > 0
    if prof_on:
        L.check(lib.vqae_prof_begin(args.prof_class, n_launch_max))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, idx, loss = step()
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")
    k_ms, k_n, k_work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
    if prof_on:
        L.check(lib.vqae_prof_end(ctypes.byref(k_ms), ctypes.byref(k_n), ctypes.byref(k_work)))

    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank == 0:          # the gathered grids hold every rank's tiles in rank order
            assert torch.equal(gathered[:B], idx)

    # ---- the same steps with the batch coming from pinned host memory each step (PCIe-inclusive rate; never `value`) ----
    h2d_value = None
    if rank == 0 and world == 1:
        host = x.cpu().pin_memory()
        xd = torch.empty_like(x)
        n_h2d = max(2, min(args.steps, 5))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_h2d):
            xd.copy_(host, non_blocking=True)
            if args.mode == "full":
                nat.forward(xd, "NCHW", idx_dtype=idx_dtype)
            else:
                nat.encode(xd, "NCHW", idx_dtype=idx_dtype, want_q=False)
        torch.cuda.synchronize()
        h2d_value = B * n_h2d / (time.perf_counter() - t1)

    res = None
    if rank == 0:
        total_patches = world * B * args.steps
        value = total_patches / dt
        flops_patch = nat.flops_per_patch(size, size, True, args.mode == "full")
        bidx = BASELINE_CONFIGS.get((args.config, args.dtype, B, args.mode))
        what = (f"BASELINE configs[{bidx}]" if bidx is not None else "variant (not a BASELINE.json config)")
        in_dt = "fp32" if args.dtype == "f32" else f"fp32 patches, {args.dtype} autocast convolutions"
        res = {
            "metric": "patches/sec (VQAE.forward: encoder -> VQ -> decoder)" if args.mode == "full"
                      else "patches/sec (Encoder.forward: encoder -> VQ indices)",
            "value": round(value, 2), "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{what} (cfg {args.config}): batch {B}/GPU of {size}x{size}x3 {in_dt} "
                                   f"-> {zh}x{zh} codes, K={spec.num_embeddings}, D={spec.code_dim}, "
                                   f"{args.mode} forward", "global_batch": world * B,
                       "parallelism": f"patch-sharded x{world}" + (", all-gather of code grids" if world > 1 else "")},
            "conv_tflops_direct_equivalent": round(value * flops_patch / 1e12, 2),
            "recon_mse_vs_input": float(((out - x) ** 2).mean()) if out is not None else None,
            "vq_loss": float(loss),
            "value_with_h2d": round(h2d_value, 2) if h2d_value else None,
        }
        # ---- roofline of the dominant kernel ---------------------------------------------------
        if prof_on and k_n.value > 0:
            avg_ms = k_ms.value / k_n.value
            C = spec.channels
            M = B * zh * zh
            alg = k_work.value / k_n.value                  # algorithmic work per launch, summed by the library
            if args.prof_class == 1 and args.dtype == "f32":
                # fp32 trunk Fixup block kernel (csrc/conv_wino.hip at C = 128 / 32-wide grid, else conv_mfma.hip TAIL):
                # conv2 3x3 circular + fused conv3 and next-block conv1 tails, M = B*32*32 pixels per launch.
                #   direct form:   2*M*C*(9C + C + C) flop
                #   Winograd form: 2*M*C*(4C + C + C) flop EXECUTED on the matrix pipe (F(2x2,3x3): 16 multiplies per 4
                #                  outputs) -- `achieved`/`frac` price the executed MFMA work, i.e. real pipe utilisation;
                #                  `direct_equivalent_*` price the same launch as a direct conv (may exceed 1.0 of peak).
                direct = 2.0 * M * C * (9 * C + C + C)
                #   F(4x4,3x3) form (C = 128, round 3): 2*M*C*(2.25C + C + C) flop executed (36 multiplies per 16 outputs): fewer
                #                  executed flops in less time -- the MFMA-only fraction FALLS while the launch gets faster; the
                #                  roof that binds is MFMA + VALU issue (fp32 MFMA does not co-execute on gfx950)
                wino = C in (256, 128, 64, 32) and not os.environ.get("VQAE_NO_WINOGRAD")
                w43 = wino and C in (128, 256) and zh % 8 == 0 and os.environ.get("VQAE_WINO43", "1") != "0"
                traffic, tnote = pmc_traffic(f"{args.config}_{args.dtype}_B{B}", "wino43_trunk_kernel" if w43 else None)
                res["roofline"] = {"kernel": "wino43_trunk_kernel: trunk Fixup block, conv2 3x3 as Winograd F(4x4,3x3) + fused conv3 / next-conv1 tails" if w43
                                             else "wino_trunk_kernel: trunk Fixup block, conv2 3x3 as Winograd F(2x2,3x3) + fused conv3 / "
                                             "next-conv1 tails" if wino else "conv_mfma_kernel TAIL: trunk Fixup block, direct conv2 + fused tails",
                                   "bound": "mfma", "achieved": round(alg / (avg_ms * 1e-3) / 1e12, 2),
                                   "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "traffic": traffic, "traffic_note": tnote,
                                   "launches": k_n.value, "avg_ms": round(avg_ms, 4), "alg_flops_per_launch": alg,
                                   "winograd": ("F(4x4,3x3)" if w43 else "F(2x2,3x3)") if wino else False,
                                   "f23_equivalent_tflops": round(2.0 * M * C * 6 * C / (avg_ms * 1e-3) / 1e12, 2),
                                   "direct_equivalent_tflops": round(direct / (avg_ms * 1e-3) / 1e12, 2),
                                   "direct_equivalent_frac": round(direct / (avg_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
            elif args.prof_class == 1:
                # 16-bit trunk Fixup block kernel (csrc/trunk16.hip): direct conv2 on v_mfma_f32_32x32x16_{bf16,f16} + fused
                # conv3 / next-conv1; algorithmic = executed flops 2*M*C*(9C + C + C); algorithmic HBM bytes per launch
                # M*C*(2 [t1 in] + 4 + 4 [x in/out] + 2 [t1' out]) -- reported as hbm_* beside the MFMA fraction: the
                # kernel sits between the two roofs.
                traffic, tnote = pmc_traffic(f"{args.config}_{args.dtype}_B{B}")
                hbm_bytes = M * C * 12.0
                res["roofline"] = {"kernel": "trunk16_kernel: trunk Fixup block, direct conv2 3x3 + fused conv3 / next-conv1, "
                                             f"{args.dtype} MFMA, 16-bit t1 in HBM",
                                   "bound": "mfma", "achieved": round(alg / (avg_ms * 1e-3) / 1e12, 2),
                                   "peak": PEAK_16BIT_MFMA_TFLOPS, "unit": "TFLOP/s", "traffic": traffic, "traffic_note": tnote,
                                   "launches": k_n.value, "avg_ms": round(avg_ms, 4), "alg_flops_per_launch": alg,
                                   "alg_hbm_bytes_per_launch": hbm_bytes,
                                   "hbm_achieved_gbs": round(hbm_bytes / (avg_ms * 1e-3) / 1e9, 1),
                                   "hbm_frac": round(hbm_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            elif args.prof_class == 2:
                res["roofline"] = {"kernel": "fixup_conv1p_kernel<128> (stand-alone 1x1 conv1 at the head of a block chain, persistent form; the "
                                             "other 1x1 convs run inside the fused trunk kernel)", "bound": "mfma",
                                   "achieved": round(alg / (avg_ms * 1e-3) / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "traffic": None, "launches": k_n.value,
                                   "avg_ms": round(avg_ms, 4), "alg_flops_per_launch": alg}
            elif spec.projection_dim == 8:
                # fused projected quantiser (csrc/vq_proj.hip): one pass over the C-channel activation.  Algorithmic HBM
                # bytes per row: C*4 read (x) + C*4 written (proj_out(q)) + 32 (z, kept for tier 2 / the loss) + 4 (index).
                byts = M * (2.0 * C * 4 + 36.0)
                res["roofline"] = {"kernel": "vq_proj16_kernel / vq_proj_fused_kernel (proj_in + p=4 argmin + lookup + proj_out in one pass, projection_dim 8)",
                                   "bound": "hbm", "achieved": round(byts / (avg_ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS,
                                   "unit": "GB/s", "traffic": None, "launches": k_n.value, "avg_ms": round(avg_ms, 4),
                                   "alg_bytes_per_launch": byts, "valu_ops_per_launch": alg,
                                   "valu_frac_of_78.6Tops": round(alg / (avg_ms * 1e-3) / 78.6e12, 4)}
            else:                          # VQ tier 1: algorithmic HBM bytes = N*D*4 read + N*4 idx; VALU-bound
                byts = M * spec.code_dim * 4.0 + M * 4.0
                res["roofline"] = {"kernel": "plain p=4 lookup: vqf_main_kernel (f16 MFMA filter + exact survivors) at 128 / 256 channels, else vq_tier1_kernel", "bound": "hbm",
                                   "achieved": round(byts / (avg_ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS,
                                   "unit": "GB/s", "traffic": None, "launches": k_n.value,
                                   "avg_ms": round(avg_ms, 4), "alg_bytes_per_launch": byts,
                                   "valu_ops_per_launch": alg,
                                   "valu_frac_of_78.6Tops": round(alg / (avg_ms * 1e-3) / 78.6e12, 4)}
            res["roofline"]["frac"] = round(res["roofline"]["achieved"] / res["roofline"]["peak"], 4)
        # ---- CPU baseline beside it (rank 0, N = 1 only) -----------------------------------------
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f32":
            sb = 16 if size == 256 else 4                  # ~10-20 s of CPU work on the box's 16-core share
            log("cpu baseline ...")
            cb, cx, cout, cidx = cpu_baseline(args.config, size, params, embed, sb, 4)
            log("cpu baseline done")
            res["cpu_baseline"] = cb
            g_out, g_idx, _ = nat.forward(cx.to(dev), "NCHW")
            res["parity_on_cpu_sample"] = {
                "idx_agreement": float((g_idx.cpu() == cidx).float().mean()),
                "recon_mse_vs_cpu": float(((g_out.cpu() - cout) ** 2).mean()),
            }

    del x, out, idx
    nat.close()
    return res if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="B", choices=["A", "B", "C", "BM"])
    ap.add_argument("--batch", type=int, default=256, help="patches per GPU per step")
    ap.add_argument("--mode", default="full", choices=["full", "encode"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"],
                    help="compute dtype of the convolutions (16-bit = torch.autocast semantics); headline: f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-class", type=int, default=1, help="kernel class timed for `roofline` (1 = trunk 3x3 conv)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs of the other BASELINE configs that the default N = 1 line carries")
    ap.add_argument("--no-slide-leg", action="store_true", help="skip the whole-slide extraction leg of other_configs")
    ap.add_argument("--other-steps", type=int, default=5)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start one fresh process per GPU through torch.distributed.run as children
        # (nothing has touched the GPU in this process) and leave with their exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun even a single rank goes through RCCL
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev)      # nccl == RCCL on ROCm

    import vqae_amd
    from vqae_amd import _lib as L

    ctx = dict(world=world, rank=rank, dev=dev, use_dist=use_dist, dist=dist if use_dist else None)
    res = run_leg(args, ctx, headline=True)

    # ---- the other BASELINE.json configs that fit one GPU, as short legs of the same (driver-timed) process -------
    # `metric` / `value` / `config` stay the headline's; the legs land under "other_configs".  N = 1 and the default
    # headline only; each leg frees its handle and workspace before the next one starts.
    if rank == 0 and world == 1 and not args.no_other_configs and \
            (args.config, args.dtype, args.batch, args.mode, args.prof_class) == ("B", "f32", 256, "full", 1):
        other = {}
        for key, cfgname, dt, mode, pclass in OTHER_LEGS:
            leg = argparse.Namespace(**vars(args))
            leg.config, leg.dtype, leg.mode, leg.prof_class = cfgname, dt, mode, pclass
            leg.steps, leg.warmup, leg.no_cpu_baseline = args.other_steps, 2, True
            log(f"other_configs leg {key}: cfg {cfgname} {dt} {mode}")
            try:
                r = run_leg(leg, ctx, headline=False)
                other[key] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config",
                                                "roofline", "value_with_h2d") if k in r}
            except Exception as e:          # a failed leg must not take the headline line with it
                other[key] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        # BASELINE configs[4] on this one GPU, end to end: 20 000 uint8 512 x 512 tiles of one synthetic slide -> ring loader (8 worker
        # processes, CPU only) -> cfg A encoder + VQ under f16 autocast -> device-side stitching -> HDF5 (tools/bench_slide.py; the
        # 100 000-tile record is profiles/r03_slide.json: at 20 000 tiles the fixed start-up weighs 5x more)
        if not args.no_slide_leg:
            free_b, total_b = torch.cuda.mem_get_info()
            log(f"other_configs leg configs[4]_slide_pipeline_1gpu (device memory free {free_b / 2**30:.1f} of {total_b / 2**30:.1f} GiB, "
                f"torch reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB)")
            try:
                # a child process (this one stays alive and idle): inside this process the ring loader's host-to-device copies complete
                # 10-40x later than in a fresh one (16 - 75 ms per 79 MB batch instead of 2; measured, cause not found), which would
                # make the leg a measurement of that, not of the pipeline
                import subprocess
                cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_slide.py"), "--rows", "100", "--cols", "200", "--batch", "100",
                       "--workers", "8", "--prefetch", "2", "--dtype", "f16", "--loader", "ring"]
                cp = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                if cp.returncode != 0 or not lines:
                    raise RuntimeError(f"bench_slide.py exited {cp.returncode}: {cp.stderr[-300:]}")
                rec = json.loads(lines[-1])
                other["configs[4]_slide_pipeline_1gpu"] = {"value": rec["patches_per_s"], "unit": "patches/s", "seconds": rec["seconds"],
                                                           "patches": rec["patches"], "dtype": "f16",
                                                           "encoder_only_patches_per_s": rec["encoder_only_patches_per_s"],
                                                           "config": {"workload": rec["workload"]}, "stages": rec["stages"]}
            except Exception as e:
                other["configs[4]_slide_pipeline_1gpu"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        res["other_configs"] = other
    if rank == 0:
        print(json.dumps(res), flush=True)

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
