"""Which state of a long-lived process slows the ring loader's H2D copies?  Runs tools/bench_slide.run_slide in-process after optional
preambles: prof (the library's 2560 timing events), pinhost (a big torch pinned allocation + copies), legs (a few handles + forwards)."""
import argparse, importlib.util, json, os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import vqae_amd
from vqae_amd import _lib as L
what = sys.argv[1:] 
lib = L.lib()
if "prof" in what:
    L.check(lib.vqae_prof_begin(1, 1280))
    ms, n, w = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
    L.check(lib.vqae_prof_end(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w)))
if "pinhost" in what:
    x = torch.empty(256, 3, 512, 512, device="cuda")
    host = x.cpu().pin_memory()
    xd = torch.empty_like(x)
    for _ in range(3):
        xd.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    del x, host, xd
if "pinalloc" in what:                      # allocation only, kept alive
    keep = torch.empty(256, 3, 512, 512).pin_memory()
if "pincopy_keep" in what:                  # allocation + copies, kept alive
    keep = torch.empty(256, 3, 512, 512).pin_memory()
    xd = torch.empty(256, 3, 512, 512, device="cuda")
    for _ in range(3):
        xd.copy_(keep, non_blocking=True)
    torch.cuda.synchronize()
if "pincopy_small" in what:                 # a small pinned copy only
    keep = torch.empty(1024).pin_memory()
    xd = torch.empty(1024, device="cuda")
    xd.copy_(keep, non_blocking=True)
    torch.cuda.synchronize()
if "d2h" in what:                           # a big device-to-host copy into pageable memory
    x = torch.empty(256, 3, 512, 512, device="cuda")
    h = x.cpu()
    del x, h
if "hostcache" in what:
    try:
        torch._C._host_emptyCache()
    except Exception as e:
        print("no _host_emptyCache:", e)
if "events" in what:
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(6000)]
    for e in evs: e.record()
    torch.cuda.synchronize()
if "emptycache" in what:
    torch.cuda.empty_cache()
sp = importlib.util.spec_from_file_location("bench_slide", os.path.join(ROOT, "tools", "bench_slide.py"))
bs = importlib.util.module_from_spec(sp); sp.loader.exec_module(bs)
rec = bs.run_slide(argparse.Namespace(rows=100, cols=200, batch=100, workers=8, prefetch=2, dtype="f16", loader="ring", encode_batch="auto", out=None))
print(what, "->", rec["patches_per_s"], "patches/s,", rec["seconds"], "s; ring_backpressure", round(rec["stages"]["host_s"]["ring_backpressure"], 2))
