#!/usr/bin/env python3
"""Per-layer time of the latency-oriented inference convolution (conv_lat.hip) at batch 1 against the gather-GEMM with in-launch
split-K, on the layer shapes of a 640x640 image; checks both against an fp64 reference.

    python tools/lat_bench.py [lib.so]
"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(80, 64, 64, 3, 1), (80, 32, 32, 3, 1), (40, 128, 128, 3, 1), (40, 64, 64, 3, 1), (20, 256, 256, 3, 1), (20, 128, 128, 3, 1),
          (80, 64, 128, 3, 2), (40, 128, 256, 3, 2), (20, 512, 256, 1, 1), (20, 256, 18, 1, 1), (40, 256, 64, 1, 1), (80, 128, 32, 1, 1)]


def main():
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib() if len(sys.argv) < 2 else None
    if lib is None:
        lib = C.CDLL(sys.argv[1])
        for name, (res, args) in L._SIGS.items():
            if hasattr(lib, name):
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
    dev = "cuda"
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn, iters=50):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    for (H, Cin, Cout, k, s) in SHAPES:
        torch.manual_seed(H + Cin)
        x = torch.randn(1, H, H, Cin, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
        bias = torch.randn(Cout, device=dev)
        ldw = (Cout + 3) // 4 * 4
        Ho = (H + 2 * (k // 2) - k) // s + 1
        y0, y1 = torch.empty(1, Ho, Ho, Cout, device=dev), torch.empty(1, Ho, Ho, Cout, device=dev)
        wq = torch.zeros(k * k * Cin * ldw, device=dev)
        tab = torch.tensor([w.data_ptr(), wq.data_ptr()], dtype=torch.int64).view(torch.uint8)
        tab = torch.cat([tab, torch.tensor([Cout, Cin, k * k, ldw], dtype=torch.int32).view(torch.uint8)]).to(dev)
        L.check(lib.yh_lat_pack_multi(tab.data_ptr(), 1, st))
        wf = torch.empty(k * k * Cin * ldw, device=dev)
        wb = torch.empty(k * k * Cout * ((Cin + 3) // 4 * 4), device=dev)
        L.check(lib.yh_pack_weights(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), Cout, Cin, k, Cin, ldw, (Cin + 3) // 4 * 4, st))
        nws = lib.yh_conv_fwd_fused_ws(1, H, H, Cin, Cout, k, s)
        ws = torch.zeros(max(nws, 1), device=dev)
        f_lat = lambda: L.check(lib.yh_conv_lat_fwd_fused(x.data_ptr(), Cin, wq.data_ptr(), ldw, bias.data_ptr(), None, 0, y1.data_ptr(), Cout,
                                                          1, H, H, Cin, Cout, k, s, 1, 0, st))
        f_gem = lambda: L.check(lib.yh_conv_fwd_fused_splitk(x.data_ptr(), Cin, wf.data_ptr(), ldw, bias.data_ptr(), None, 0, y0.data_ptr(), Cout,
                                                             ws.data_ptr(), nws, 1, H, H, Cin, Cout, k, s, 1, 0, st))
        f_lat(); f_gem()
        torch.cuda.synchronize()
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), stride=s, padding=k // 2)
        ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1)
        e1, e0 = ((y1 - ref).abs().max() / ref.abs().max()).item(), ((y0 - ref).abs().max() / ref.abs().max()).item()
        print(f"{H}x{H} {Cin}->{Cout} k{k} s{s}: lat {timed(f_lat):6.1f} us (err {e1:.1e})  gather+splitK {timed(f_gem):6.1f} us (err {e0:.1e})", flush=True)


if __name__ == "__main__":
    main()
