#!/usr/bin/env python3
"""The 16 -> 16 3x3 stride-1 layer at 160x160, batch 64, on the direct narrow kernel (fp32 and bf16 storage), forward (GPU box).
    python tools/narrow16_bench.py"""
import os, struct, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, H, W, Cin, Cout = 64, 160, 160, 16, 16
    iters = int(os.environ.get("ITERS", "20"))
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / 12.0
    x = torch.randn(B, H, W, Cin, device="cuda")
    wf = torch.empty(9 * Cin * Cout, device="cuda")
    L.check(lib.yh_pack_weights(w.data_ptr(), wf.data_ptr(), None, Cout, Cin, 3, Cin, Cout, 0, st), "pack")
    y = torch.empty(B, H, W, Cout, device="cuda")
    nblk = lib.yh_conv_narrow_blocks(B, H, W, Cin, 1)
    part = torch.zeros(nblk * 2 * Cout, device="cuda")

    def f32():
        L.check(lib.yh_conv_narrow(x.data_ptr(), Cin, wf.data_ptr(), Cout, None, y.data_ptr(), Cout, part.data_ptr(), B, H, W, Cin, Cout, 1, 0, 0, st), "f32")
    for _ in range(3):
        f32()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f32()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"fp32 16->16 at 160x160: {us:7.1f} us  {(x.numel() + y.numel()) * 4 / us / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
