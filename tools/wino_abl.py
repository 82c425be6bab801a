#!/usr/bin/env python3
"""Timing-only runs of the LDS-staged Winograd kernel for ablation builds (results are wrong by construction)."""
import ctypes as C, sys, os, torch
so = sys.argv[1]
lib = C.CDLL(so)
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
fp = C.c_void_p
lib.yh_conv_wino_fwd_act.argtypes = [fp, C.c_int, fp, fp, fp, C.c_int, fp, fp, C.c_int, fp] + [C.c_int] * 5 + [fp]
lib.yh_conv_wino_lds_blocks.argtypes = [C.c_int] * 3
out = []
for (B, H, W, Cin, Cout) in [(64, 20, 20, 128, 128), (64, 80, 80, 64, 64), (64, 40, 40, 64, 64), (64, 20, 20, 256, 256)]:
    x = torch.randn(B, H, W, Cin, device=dev)
    U = torch.randn(16 * Cin * Cout, device=dev)
    y = torch.empty(B, H, W, Cout, device=dev)
    nb = lib.yh_conv_wino_lds_blocks(B, H, W)
    p = torch.empty(nb * 2 * Cout, device=dev)
    f = lambda: lib.yh_conv_wino_fwd_act(x.data_ptr(), Cin, None, None, U.data_ptr(), Cout, None, y.data_ptr(), Cout, p.data_ptr(), B, H, W, Cin, Cout, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{H}x{W} {Cin}->{Cout}: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us")
print(os.path.basename(so), " | ".join(out), flush=True)
