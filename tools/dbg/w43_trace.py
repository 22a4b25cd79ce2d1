#!/usr/bin/env python3
"""Per-phase cycle breakdown of wino43_trunk_kernel<128, 2> from the -DW43_TRACE variant build:
    tools/dbg/build_variant.sh trace conv_wino43.hip "-DW43_TRACE"
    VQAE_HIP_LIB=$PWD/2d-vq-ae-2_amd/build/var/libvqae_trace.so python3 tools/dbg/w43_trace.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    import vqae_amd
    from vqae_amd import _lib as L
    from vqae_amd.spec import encoder_block_names
    from oracle import vqae_oracle as O
    spec = vqae_amd.SPECS["B"]
    nat = vqae_amd.NativeVQAE(spec, O.make_params(O.SPECS["B"], 0))
    names = encoder_block_names(spec)
    first = len(names) - spec.n_enc
    B, h, c = 256, 32, names[first][2]
    x = torch.randn(B, h, h, c, device="cuda") * 1.5
    nat.run_blocks("encoder", first, 4, x)
    n_wg = B * h * h // 256
    buf = torch.zeros(n_wg * 8 * 64, dtype=torch.int64, device="cuda")
    lib = ctypes.CDLL(L.LIB_PATH)
    assert lib.vqae_debug_w43_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    nat.run_blocks("encoder", first, 4, x)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(n_wg, 8, 64)[:, :4, :]
    print(f"{n_wg} workgroups; kernel span {int(t[:, :, 46].max() - t[:, :, 0].min())} cycles (s_memtime ticks at 100 MHz x ?)")
    def med(a): return f"median {np.median(a):8.0f} p10 {np.percentile(a, 10):8.0f} p90 {np.percentile(a, 90):8.0f}"
    print("per pass (all waves):")
    for xi in range(6):
        b = 5 * xi
        prev = t[:, :, b] if xi == 0 else t[:, :, b]          # slot 5 xi = end of the previous pass (0: kernel start)
        print(f"  xi {xi}: transform {med(t[:, :, b + 1] - prev)} | barrier {med(t[:, :, b + 2] - t[:, :, b + 1])} | gemm+nu fold {med(t[:, :, b + 3] - t[:, :, b + 2])}"
              f" | xi fold {med(t[:, :, b + 4] - t[:, :, b + 3])} | barrier {med(t[:, :, b + 5] - t[:, :, b + 4])}")
    for hf in range(2):
        b = 31 + 8 * hf
        start = t[:, :, 30] if hf == 0 else t[:, :, 38]
        print(f"  half {hf}: t2+res {med(t[:, :, b] - start)} | barrier {med(t[:, :, b + 1] - t[:, :, b])} | conv3 {med(t[:, :, b + 2] - t[:, :, b + 1])} | acc->T {med(t[:, :, b + 3] - t[:, :, b + 2])}"
              f" | epilogue {med(t[:, :, b + 4] - t[:, :, b + 3])} | conv1n(+barrier) {med(t[:, :, b + 5] - t[:, :, b + 4])} | y2 {med(t[:, :, b + 6] - t[:, :, b + 5])} | barrier {med(t[:, :, b + 7] - t[:, :, b + 6])}")
    tot = t[:, :, 46] - t[:, :, 0]
    print(f"  total per wave: {med(tot)}; main phase {med(t[:, :, 30] - t[:, :, 0])}; tails {med(t[:, :, 46] - t[:, :, 30])}")
    # co-residence: workgroups by (xcc, cu) key; GEMM intervals of wave 0; overlap with the CU's other workgroups
    info = t[:, 0, 63]
    key = ((info >> 32) & 15) * 65536 + ((info & 0xFFFF) >> 8)
    t0 = t[:, :, 0].min()
    ov = gtot = 0
    shown = 0
    for k_ in np.unique(key):
        ids = np.nonzero(key == k_)[0]
        iv = []
        for i in ids:
            for xi in range(6):
                iv.append((int(t[i, 0, 2 + 5 * xi]), int(t[i, 0, 3 + 5 * xi]), i))
        for a_ in range(len(iv)):
            gtot += iv[a_][1] - iv[a_][0]
            for b_ in range(len(iv)):
                if iv[b_][2] != iv[a_][2]:
                    ov += max(0, min(iv[a_][1], iv[b_][1]) - max(iv[a_][0], iv[b_][0]))
        if shown < 3:
            shown += 1
            print(f"  CU key {k_:#x}: workgroups {list(ids)}")
            for i in ids:
                print("    wg %4d: start %7d | gemm intervals " % (i, t[i, 0, 0] - t0) + " ".join(f"[{t[i, 0, 2 + 5 * xi] - t0}-{t[i, 0, 3 + 5 * xi] - t0}]" for xi in range(6))
                      + f" | tails {t[i, 0, 30] - t0}-{t[i, 0, 46] - t0}")
    print(f"  fraction of GEMM time during which another workgroup of the CU is also in a GEMM phase: {ov / max(gtot, 1):.2f}")
    first_gen = np.arange(n_wg) < 512
    print(f"  first generation total {np.median(tot[first_gen]):.0f}, second {np.median(tot[~first_gen]):.0f}; start of second generation (median) {np.median(t[~first_gen][:, 0, 0]) - t[:, :, 0].min():.0f}")


if __name__ == "__main__":
    main()
