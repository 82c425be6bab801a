#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at the benchmark shape (GPU box only).

    python tools/layer_bench.py [--batch 64] [--img 640] [--nc 1] [--iters 5]

Walks the traced training plan of the model and times, with HIP events on the launch stream, the
forward, backward-data and backward-weight launch of every convolution; prints ms and TFLOP/s per
layer and the totals.  Used to steer kernel tuning; not part of the product path.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--nc", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    args = ap.parse_args()
    import yolo_from_scratch_amd as y
    from yolo_from_scratch_amd import _lib as L
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = y.YOLO(num_classes=args.nc, img_size=args.img).to(dev).train()
    x = torch.rand(args.batch, 3, args.img, args.img, device=dev)
    tr = y.HipTrainer(model, dtype=args.dtype)
    tg = [t.to(dev) for t in y.synthetic_targets(args.batch, args.nc, args.img)]
    tr.step(x, tg)                       # populate every buffer
    plan = model._plan_for(x)
    st = torch.cuda.current_stream(dev).cuda_stream

    def time_ops(arr, n, kinds):
        """ms per op index for ops whose kind is in `kinds` (mean over iters)."""
        res = {}
        for k in range(n):
            if arr[k].kind not in kinds:
                continue
            one = ctypes.cast(ctypes.byref(arr, k * ctypes.sizeof(L.YhOp)), ctypes.POINTER(L.YhOp))
            L.run_ops(one, 1, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                L.run_ops(one, 1, st)
            e1.record()
            torch.cuda.synchronize()
            res[k] = e0.elapsed_time(e1) / args.iters
        return res

    fa, fn = plan.fwd_ops
    ba, bn = plan.bwd_ops
    tf = time_ops(fa, fn, {L.OP_CONV_FWD, L.OP_CONV_S2_FWD, L.OP_CONV_WINO_FWD, L.OP_CONV_PW_FWD, L.OP_CONV_PW_FWD2, L.OP_BF16_CONV_FWD,
                           L.OP_CONV_NARROW, L.OP_BF16_CONV_NARROW})
    DG = {L.OP_CONV_BWD_DATA, L.OP_CONV_WINO_BWD_DATA, L.OP_CONV_BWD_DATA_PAIR, L.OP_CONV_PW_BWD_DATA, L.OP_CONV_BWD_DATA_S2M,
          L.OP_BF16_CONV_BWD_DATA, L.OP_CONV_NARROW, L.OP_CONV_NARROW_DGRAD_S2, L.OP_BF16_CONV_NARROW, L.OP_BF16_CONV_NARROW_DGRAD_S2}
    tb = time_ops(ba, bn, DG | {L.OP_CONV_BWD_WEIGHT, L.OP_CONV_WINO_BWD_WEIGHT, L.OP_CONV_PW_BWD_WEIGHT, L.OP_BF16_CONV_BWD_WEIGHT,
                            L.OP_CONV_NARROW_BWD_WEIGHT, L.OP_BF16_CONV_NARROW_BWD_WEIGHT})
    other_f = time_ops(fa, fn, {L.OP_BN_SILU_FWD, L.OP_BN_FINALIZE, L.OP_PACK_WEIGHTS, L.OP_MAXPOOL5_FWD, L.OP_BF16_BN_SILU_FWD,
                                L.OP_BF16_MAXPOOL5_FWD, L.OP_BF16_PACK_MULTI})
    other_b = time_ops(ba, bn, {L.OP_BN_SILU_BWD_REDUCE, L.OP_BN_SILU_BWD_APPLY, L.OP_COLSUM, L.OP_MAXPOOL5_BWD,
                                L.OP_BF16_BN_SILU_BWD_REDUCE, L.OP_BF16_BN_SILU_BWD_APPLY, L.OP_BF16_COLSUM, L.OP_BF16_MAXPOOL5_BWD})

    def desc(o):
        i = o.i
        if o.kind == L.OP_CONV_PW_FWD2:       # two sibling convs in one launch: listed as one conv with N = cout1 + cout2
            return (i[3], i[4], i[5], i[6], i[7] + i[9], 1, 1)
        if o.kind == L.OP_CONV_PW_BWD_DATA:
            return (i[5], i[6], i[7], i[8], i[0], 1, 1)
        if o.kind in (L.OP_CONV_NARROW, L.OP_BF16_CONV_NARROW):        # i: ldx, ldw, ldy, B, H, W, Cin, Cout, s, flip, acc (backward-data: channels swapped)
            return (i[3], i[4], i[5], i[7], i[6], 3, 1) if i[9] else (i[3], i[4], i[5], i[6], i[7], 3, i[8])
        if o.kind == L.OP_BF16_CONV_BWD_DATA and i[11] > 0:       # fused sibling pair: listed under the first conv's shape
            return (i[3], i[4], i[5], i[6], i[11], 1, 1)
        if o.kind in (L.OP_CONV_FWD, L.OP_CONV_S2_FWD, L.OP_CONV_WINO_FWD, L.OP_CONV_PW_FWD, L.OP_CONV_BWD_DATA, L.OP_CONV_BWD_DATA_S2M,
                      L.OP_BF16_CONV_FWD, L.OP_BF16_CONV_BWD_DATA, L.OP_CONV_NARROW_DGRAD_S2, L.OP_BF16_CONV_NARROW_DGRAD_S2):
            return (i[3], i[4], i[5], i[6], i[7], i[8], i[9])
        if o.kind == L.OP_CONV_WINO_BWD_DATA:
            return (i[3], i[4], i[5], i[6], i[7], 3, 1)
        if o.kind == L.OP_CONV_BWD_DATA_PAIR:      # two sibling 1x1 convs: listed under the first one's shape
            return (i[5], i[6], i[7], i[8], i[0], 1, 1)
        return (i[2], i[3], i[4], i[5], i[7], i[8], i[9])

    rows = {}
    for k, ms in tf.items():
        rows.setdefault(desc(fa[k]), [0.0, 0.0, 0.0, 0])[0] += ms
        rows[desc(fa[k])][3] += 1
    for k, ms in tb.items():
        d = desc(ba[k])
        rows.setdefault(d, [0.0, 0.0, 0.0, 0])[1 if ba[k].kind in DG else 2] += ms
    esz = 2 if args.dtype == "bf16" else 4
    print(f"{'B,H,W,Cin,Cout,k,s':32s} {'n':>2s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>6s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s} | in+out MB, fwd TB/s")
    tot = [0.0, 0.0, 0.0, 0.0]
    for d, (f, dg, wg, cnt) in sorted(rows.items(), key=lambda kv: -(kv[1][0] + kv[1][1] + kv[1][2])):
        B, H, W, Cin, Cout, k, s = d
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        gf = 2.0 * B * Ho * Wo * Cin * Cout * k * k * cnt / 1e9
        tfs = lambda ms: gf / ms if ms > 0 else 0.0
        mb = cnt * B * (H * W * Cin + Ho * Wo * Cout) * esz / 1e6
        print(f"{str(d):32s} {cnt:2d} {gf:8.2f} | {f:8.3f} {tfs(f):6.1f} | {dg:8.3f} {tfs(dg):6.1f} | {wg:8.3f} {tfs(wg):6.1f} | {mb:7.0f} {mb / 1e3 / f if f > 0 else 0:5.2f}")
        tot[0] += f; tot[1] += dg; tot[2] += wg; tot[3] += gf
    print(f"{'TOTAL':32s}    {tot[3]:8.2f} | {tot[0]:8.3f} {tot[3] / tot[0]:6.1f} | {tot[1]:8.3f} {tot[3] / max(tot[1], 1e-9):6.1f} | {tot[2]:8.3f} {tot[3] / tot[2]:6.1f}")
    names = {L.OP_BN_SILU_FWD: "bn_silu_fwd", L.OP_BN_FINALIZE: "bn_finalize", L.OP_PACK_WEIGHTS: "pack_weights",
             L.OP_MAXPOOL5_FWD: "maxpool_fwd", L.OP_BN_SILU_BWD_REDUCE: "bn_bwd_reduce", L.OP_BN_SILU_BWD_APPLY: "bn_bwd_apply",
             L.OP_COLSUM: "colsum", L.OP_MAXPOOL5_BWD: "maxpool_bwd", L.OP_BF16_BN_SILU_FWD: "bn_silu_fwd", L.OP_BF16_MAXPOOL5_FWD: "maxpool_fwd",
             L.OP_BF16_PACK_MULTI: "pack_weights", L.OP_BF16_BN_SILU_BWD_REDUCE: "bn_bwd_reduce", L.OP_BF16_BN_SILU_BWD_APPLY: "bn_bwd_apply",
             L.OP_BF16_COLSUM: "colsum", L.OP_BF16_MAXPOOL5_BWD: "maxpool_bwd"}
    agg = {}
    for arr, res in ((fa, other_f), (ba, other_b)):
        for k, ms in res.items():
            agg[names[arr[k].kind]] = agg.get(names[arr[k].kind], 0.0) + ms
    print("other ops (ms/step):", {k: round(v, 3) for k, v in agg.items()})


if __name__ == "__main__":
    main()
