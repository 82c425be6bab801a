#!/usr/bin/env python3
"""The first layer (3(4) -> 16, 3x3 stride 2, 640x640, batch 64) on the direct narrow kernel, fp32 vs bf16 storage (GPU box).
    python tools/narrow_fwd_bench.py [ldx_bf16]"""
import os, struct, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from yolo_from_scratch_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, H, W, Cin, Cout = 64, 640, 640, 4, 16
    ldx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = 20
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / 6.0
    # fp32
    x = torch.randn(B, H, W, Cin, device="cuda")
    wf = torch.empty(9 * Cin * Cout, device="cuda")
    L.check(lib.yh_pack_weights(w.data_ptr(), wf.data_ptr(), None, Cout, Cin, 3, Cin, Cout, 0, st), "pack")
    y = torch.empty(B, H // 2, W // 2, Cout, device="cuda")
    nblk = lib.yh_conv_narrow_blocks(B, H, W, Cin, 2)
    part = torch.zeros(nblk * 2 * Cout, device="cuda")

    def f32():
        L.check(lib.yh_conv_narrow(x.data_ptr(), Cin, wf.data_ptr(), Cout, None, y.data_ptr(), Cout, part.data_ptr(), B, H, W, Cin, Cout, 2, 0, 0, st), "f32")
    xb = torch.zeros(B, H, W, ldx, device="cuda", dtype=torch.bfloat16)
    xb[..., :Cin] = x.to(torch.bfloat16)
    kpad = 8
    wfb = torch.zeros(9 * kpad * Cout, dtype=torch.bfloat16, device="cuda")
    rec = struct.pack("<QQQiiiiiiii", w.data_ptr(), wfb.data_ptr(), 0, Cout, Cin, 9, kpad, Cout, Cin, 0, Cout)
    tab = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    L.check(lib.yh_bf16_pack_multi(tab.data_ptr(), 1, st), "packb")
    yb = torch.empty(B, H // 2, W // 2, Cout, device="cuda", dtype=torch.bfloat16)

    def b16():
        L.check(lib.yh_bf16_conv_narrow(xb.data_ptr(), ldx, wfb.data_ptr(), Cout, kpad, None, yb.data_ptr(), Cout, part.data_ptr(), B, H, W, Cin, Cout, 2, 0, 0, st), "bf16")
    for name, fn, nbytes in (("fp32", f32, x.numel() * 4 + y.numel() * 4), ("bf16", b16, B * H * W * Cin * 2 + yb.numel() * 2)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"{name}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
    print("max |bf16 - fp32| =", float((yb.float() - y).abs().max()))


if __name__ == "__main__":
    main()
