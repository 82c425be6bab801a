#!/usr/bin/env python3
"""Headline benchmark: images/s of one full training step (NCHW->NHWC, forward, fused loss, backward,
[gradient all-reduce], global-norm clip 10.0, Adam) on synthetic 640x640 batches, bs=64 per GPU, fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]           (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every forward-conv
launch (the gather-GEMM kernel) on the stream it runs on, in instrumented steps right after the timed
region; `cpu_baseline` times the CPU oracle (a port of the reference step on stock torch CPU ops) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time
from datetime import timedelta

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NC, IMG, BATCH = 1, 640, 64          # BASELINE.json configs[1]
# informational runs of the other configs' shapes in fp32 (never the default, never the reported metric):
#   YH_BENCH_SHAPE=80,640,64  or  80,1280,16
if os.environ.get("YH_BENCH_SHAPE"):
    NC, IMG, BATCH = (int(v) for v in os.environ["YH_BENCH_SHAPE"].split(","))
# YH_BENCH_DTYPE=bf16: the bf16 path of configs 3-4 (informational too: the headline line is fp32, the reference's arithmetic)
DTYPE = os.environ.get("YH_BENCH_DTYPE", "f32")
# YH_BENCH_SIZE=n|s|m|l|x: the reference's other model sizes (train.py:1346-1352), informational as well
SIZE = os.environ.get("YH_BENCH_SIZE", "s")
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md, HBM3E spec (6290 measured with a float4 copy)


def plan_conv_flops(plan):
    from yolo_from_scratch_amd.graph import ConvRec
    fl = fw = 0
    for r in plan.recs:
        if isinstance(r, ConvRec):
            f = 2 * r.weight.shape[1] * r.cout * r.k * r.k * r.Ho * r.Wo
            fl += f
            fw += f if r.wino_f else 0
    return fl, fw   # per image: all forward convs, and the part on the Winograd kernel


def plan_conv_bytes(plan):
    """Algorithmic activation bytes of the forward convolutions per image: every conv reads its input view once
    and writes its output once, in the plan's storage type (head outputs fp32)."""
    from yolo_from_scratch_amd.graph import ConvRec
    nb = 0
    for r in plan.recs:
        if isinstance(r, ConvRec):
            esz = 2 if plan.bf16 else 4
            nb += r.x.H * r.x.W * r.weight.shape[1] * esz + r.Ho * r.Wo * r.cout * (esz if r.bn is not None else 4)
    return nb


def time_forward_convs(trainer, plan, imgs, targets, steps):
    """HIP-event time of every forward-convolution launch (direct gather-GEMM and Winograd kernels) and,
    separately, of every op class, per step."""
    from yolo_from_scratch_amd import _lib as L
    import ctypes
    dev = trainer.device
    arr, n = plan.fwd_ops
    st = torch.cuda.current_stream(dev).cuda_stream
    conv_ms, n_launch = 0.0, 0
    per_kind = {}
    L.set_overlap(0, dev.index)        # every launch on the timed stream while instrumenting
    for _ in range(steps):
        trainer.model._load_input(plan, imgs)
        evs = []
        for k in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.run_ops(ctypes.cast(ctypes.byref(arr, k * ctypes.sizeof(L.YhOp)), ctypes.POINTER(L.YhOp)), 1, st)
            e1.record()
            evs.append((arr[k].kind, e0, e1))
        torch.cuda.synchronize(dev)
        for kind, e0, e1 in evs:
            ms = e0.elapsed_time(e1)
            per_kind[kind] = per_kind.get(kind, 0.0) + ms
            if kind in (L.OP_CONV_FWD, L.OP_CONV_S2_FWD, L.OP_CONV_WINO_FWD, L.OP_CONV_PW_FWD, L.OP_CONV_PW_FWD2, L.OP_BF16_CONV_FWD, L.OP_CONV_NARROW,
                        L.OP_BF16_CONV_NARROW):
                conv_ms += ms
                n_launch += 1
        plan.generation += 1
    # the whole forward op list in ONE yh_run call (no per-op events between the launches): serial, then with the lanes on
    whole = {}
    for name, ov in (("serial", 0), ("lanes", 1)):
        L.set_overlap(ov, dev.index)
        tot = 0.0
        for _ in range(steps):
            trainer.model._load_input(plan, imgs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.run_forward(st)
            e1.record()
            torch.cuda.synchronize(dev)
            tot += e0.elapsed_time(e1)
        whole[name] = tot / steps
    L.set_overlap(1, dev.index)
    per = {k: v / steps for k, v in per_kind.items()}
    per["whole_forward_serial"], per["whole_forward_lanes"] = whole["serial"], whole["lanes"]
    return conv_ms / steps, n_launch // steps, per


def host_cores():
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


PROFILE_ROUND = "r04"


def stored_traffic(nc, img, batch, dtype, size):
    """HBM bytes of the forward-convolution launches from the committed rocprofv3 PMC profile -- quoted only when the profile
    was taken on exactly these kernels (tools/provenance.py: hash of csrc/ + the C header); otherwise null."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from provenance import csrc_hash
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_step_profile_{dtype}.json")
    if not os.path.exists(path) or (nc, img, batch, size) != ((1, 640, 64, "s") if dtype == "f32" else (80, 640, 64, "s")):
        return None, f"no committed PMC profile for this workload ({os.path.basename(path)})"
    doc = json.load(open(path))
    if doc.get("stamp", {}).get("csrc_sha16") != csrc_hash():
        return None, f"{os.path.basename(path)} was taken on other kernels (csrc hash {doc.get('stamp', {}).get('csrc_sha16')} != {csrc_hash()}): not quoted"
    return doc["groups"]["fwd_conv"]["hbm_bytes_per_step"], (
        f"HBM-side bytes per step over the forward-convolution launches (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/{os.path.basename(path)}, "
        f"tools/step_profile.py, stamp {doc['stamp']})")


def step_conv_flops(plan):
    """Algorithmic convolution FLOPs of one training step per image: forward + backward-weight of every convolution, backward-data of
    every convolution whose input needs a gradient (not the first layer)."""
    from yolo_from_scratch_amd import graph as G
    tot = 0
    for r in plan.recs:
        if isinstance(r, G.ConvRec):
            f = 2 * r.weight.shape[1] * r.cout * r.k * r.k * r.Ho * r.Wo
            tot += 2 * f + (f if r.need_dx else 0)
    return tot


def cpu_baseline(batch=8, steps=3):
    """The CPU oracle's training step (same math on stock torch CPU ops) on a bounded sample."""
    from oracle import yolo_oracle as orc
    import yolo_from_scratch_amd as y
    torch.set_num_threads(host_cores())       # oversubscribing the quota is several times slower
    torch.manual_seed(0)
    m = y.YOLO(num_classes=NC, img_size=IMG, width_mult=y.YOLO_SIZES[SIZE][0], depth_mult=y.YOLO_SIZES[SIZE][1])
    P = {k: v.clone() for k, v in m.state_dict().items()}
    params = [P[n].requires_grad_(True) for n, _ in m.named_parameters()]
    opt = torch.optim.Adam(params, lr=1e-3)
    x = torch.rand(batch, 3, IMG, IMG, generator=torch.Generator().manual_seed(1000))
    tg = y.synthetic_targets(batch, NC, IMG, 8, 2000)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = orc.loss_multiscale(orc.forward(P, x, NC, True), tg, orc.anchors_of(P), NC)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(batch / t, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {steps} full training steps at batch {batch} (of {BATCH}) after 1 warm-up, same model/inputs"}


def self_launch(args):
    """`python bench.py --gpus N` as a bare command: start N fresh rank processes through torch.distributed.run (one per
    GPU, rendezvous on 127.0.0.1) and relay their output; this parent never initialises the GPU and never execs."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


FORCE_COLLECTIVES = [False]     # --collectives-at-world-1: rehearse the RCCL leg with ONE rank (tests/test_gpu_dp.py)


def timed_training(y, dist, dev, rank, world, nc, img, batch, dtype, size, steps, warmup):
    """W untimed + exactly K timed steps bracketed by barrier + synchronize, max over ranks.  Returns the line's base fields
    plus the live objects the roofline instrumentation needs."""
    torch.manual_seed(0)                                   # identical replicas
    model = y.YOLO(num_classes=nc, img_size=img, width_mult=y.YOLO_SIZES[size][0], depth_mult=y.YOLO_SIZES[size][1]).to(dev)
    trainer = y.HipTrainer(model, lr=1e-3, max_norm=10.0, dtype=dtype, collectives_at_world_1=FORCE_COLLECTIVES[0])
    imgs = torch.rand(batch, 3, img, img, generator=torch.Generator().manual_seed(1000 + rank)).to(dev)
    targets = [t.to(dev) for t in y.synthetic_targets(batch, nc, img, 8, 2000 + rank)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        trainer.step(imgs, targets)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step(imgs, targets)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    base = {"value": round(batch * world * steps / elapsed, 2), "unit": "images/s", "ms_per_step": round(1e3 * elapsed / steps, 3),
            "steps": steps, "warmup": warmup, "loss": [round(v, 6) for v in trainer.loss_out[:4].tolist()]}
    return base, model, trainer, imgs, targets


def roofline_bf16(model, trainer, imgs, targets, nc, img, batch, size):
    """bf16: every layer of size 's' is HBM-bound, so the roofline is bytes: algorithmic activation bytes of the
    forward convolutions (each input view read once, each output written once) / HIP-event time of those launches."""
    from yolo_from_scratch_amd import _lib as L
    plan = model._plan_for(imgs)
    conv_ms, n_launch, per_kind = time_forward_convs(trainer, plan, imgs, targets, 3)
    nbytes = plan_conv_bytes(plan) * batch
    flops = plan_conv_flops(plan)[0] * batch
    ach = nbytes / (conv_ms * 1e-3) / 1e9
    bn_ms = per_kind.get(L.OP_BN_FINALIZE, 0.0) + per_kind.get(L.OP_BF16_BN_SILU_FWD, 0.0)
    traffic, note = stored_traffic(nc, img, batch, "bf16", size)
    roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_note": note,
            "kernel": "forward convolutions: bf16_fstream_kernel (stride-1 layers as a flat pixel stream: halo ring in LDS, B fragments in "
                      "registers) + bf16_gemm_kernel (stride-2 and wide-K layers: gather implicit GEMM), both on v_mfma_f32_32x32x16_bf16; the "
                      "first layer and stem[3] on direct v_mfma_f32_16x16x32_bf16 kernels (narrow_first_bf16_kernel, narrow_s2_16x32_bf16_kernel)",
            "kernel_ms_per_step": round(conv_ms, 3), "launches_per_step": n_launch,
            "algorithmic_gbytes_per_step": round(nbytes / 1e9, 3),
            "algorithmic_gflop_per_step": round(flops / 1e9, 2),
            "conv_tflops": round(flops / (conv_ms * 1e-3) / 1e12, 1),
            "conv_bn_silu_forward_ms_per_step": round(conv_ms + bn_ms, 3),
            "whole_forward_ms_serial": round(per_kind["whole_forward_serial"], 3),
            "whole_forward_ms_with_lanes": round(per_kind["whole_forward_lanes"], 3)}
    return roof, {str(k): round(v, 3) for k, v in sorted(per_kind.items(), key=lambda kv: str(kv[0]))}


def roofline_f32(model, trainer, imgs, targets, nc, img, batch, size):
    from yolo_from_scratch_amd import _lib as L
    plan = model._plan_for(imgs)
    conv_ms, n_launch, per_kind = time_forward_convs(trainer, plan, imgs, targets, 3)
    flops, flops_wino = (v * batch for v in plan_conv_flops(plan))
    ach = flops / (conv_ms * 1e-3) / 1e12
    # the Winograd kernel executes 4/9 of the direct form's multiplies: the rate the MFMA pipe actually runs at
    executed = flops - flops_wino * (1.0 - 4.0 / 9.0)
    traffic, traffic_note = stored_traffic(nc, img, batch, "f32", size)
    cbs_ms = conv_ms + per_kind.get(L.OP_BN_FINALIZE, 0.0) + per_kind.get(L.OP_BN_SILU_FWD, 0.0)
    roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
            "traffic_note": traffic_note + "; algorithmic 8.41e9 (8 sibling pairs share one launch: 62 convs = 54 launches)",
            "executed_mfma_tflops": round(executed / (conv_ms * 1e-3) / 1e12, 2),
            "executed_mfma_frac": round(executed / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "kernel": "forward convolutions: wino_lds_kernel (3x3 stride-1 layers, Winograd F(2x2,3x3) with the input patch staged "
                      "through LDS: executes 4/9 of the algorithmic multiplies) + pw_tile_kernel / pw_stream_kernel (1x1) + "
                      "s2_lds_kernel (3x3 stride-2 layers, patch staged through LDS) + gather_gemm_kernel (wide 1x1) + narrow_conv_kernel "
                      "(16-channel 3x3 layers, stem[3], stem[0]); since round 4 "
                      "most of these kernels also apply their input's BatchNorm + SiLU while staging it (89 % of the normalised elements "
                      "are never materialised), so their time includes what the separate bn_silu_fwd pass used to cost",
            "narrow_ms_per_step": round(per_kind.get(L.OP_CONV_NARROW, 0.0), 3),
            "gather_gemm_ms_per_step": round(per_kind.get(L.OP_CONV_FWD, 0.0), 3),
            "s2_lds_ms_per_step": round(per_kind.get(L.OP_CONV_S2_FWD, 0.0), 3),
            "wino_ms_per_step": round(per_kind.get(L.OP_CONV_WINO_FWD, 0.0), 3),
            "pw_gemm_ms_per_step": round(per_kind.get(L.OP_CONV_PW_FWD, 0.0) + per_kind.get(L.OP_CONV_PW_FWD2, 0.0), 3),
            "wino_algorithmic_gflop_per_step": round(flops_wino / 1e9, 2),
            "launches_per_step": n_launch,
            # north_star's target quantity: the whole Conv+BN+SiLU forward against the same peak.  Two measurements:
            # (a) sum of the per-op event times of conv + statistics finalize + normalise/SiLU ops (each op bracketed by
            # its own event pair, which adds a few us per op); (b) the entire forward op list timed as one call, serial
            # (everything: also weight packs, pools, head convs) -- no per-op events
            "conv_bn_silu_forward_ms_per_step": round(cbs_ms, 3),
            "conv_bn_silu_forward_frac": round(flops / (cbs_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "whole_forward_ms_serial": round(per_kind["whole_forward_serial"], 3),
            "whole_forward_ms_with_lanes": round(per_kind["whole_forward_lanes"], 3),
            "whole_forward_frac_serial": round(flops / (per_kind["whole_forward_serial"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "whole_forward_frac_with_lanes": round(flops / (per_kind["whole_forward_lanes"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "kernel_ms_per_step": round(conv_ms, 3), "algorithmic_gflop_per_step": round(flops / 1e9, 2)}
    return roof, {str(k): round(v, 3) for k, v in sorted(per_kind.items(), key=lambda kv: str(kv[0]))}


def detecting_model(y, nc, img=640):
    """BASELINE config 5's model: untrained weights give no detections, so the objectness is spread (prior 0.2, head weights
    x60), the BatchNorm running statistics are warmed, and the confidence threshold is set for ~3000 candidates into NMS."""
    torch.manual_seed(0)
    m = y.YOLO(num_classes=nc, img_size=img)
    m.initialize_detection_biases(prior=0.2)
    with torch.no_grad():
        for hd in (m.head_p3, m.head_p4, m.head_p5):
            hd[-1].weight.mul_(60.0)
    m = m.cuda().train()
    with torch.no_grad():
        for _ in range(3):
            m(torch.rand(4, 3, img, img, device="cuda"))
    m.eval()
    probe = torch.rand(1, 3, img, img, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        obj = torch.cat([torch.sigmoid(p[..., 4]).flatten() for p in m(probe.cuda())])
    return m, float(torch.sort(obj, descending=True).values[3000])


def infer_latency(y, nc=1, iters=100):
    """BASELINE config 5: bs=1 640x640, H2D image copy, image load kernel, BN-folded fused forward, candidates, global NMS,
    result table, D2H, python list -- eager launches vs ONE captured hipGraph.  Two input forms: the reference's
    (train.py:1137 builds a float NCHW tensor on the host: 4.9 MB over PCIe) and this package's predict() (the letterboxed
    uint8 HWC bytes go over, 1.2 MB, and /255 runs on the device with the same rounding)."""
    m, thr = detecting_model(y, nc)
    img = torch.rand(1, 3, 640, 640, generator=torch.Generator().manual_seed(7))
    img_u8 = (img[0].permute(1, 2, 0) * 255).to(torch.uint8).contiguous().pin_memory()
    img = img.pin_memory()
    out = {"workload": f"nc={nc} 640x640 bs=1 eval, BN folded, ~3000 candidates into NMS, pinned host image in (fp32 NCHW; "
                       "*_u8: uint8 HWC as predict() sends it), python list of (x1,y1,x2,y2,conf,cls) out", "unit": "ms", "iters": iters}
    for name, use_graph, src in (("eager", False, img), ("hipgraph", True, img), ("eager_u8", False, img_u8), ("hipgraph_u8", True, img_u8)):
        ses = y.InferenceSession(m, conf_threshold=thr, iou_threshold=0.4, use_graph=use_graph)
        for _ in range(10):
            dets = ses.run(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            dets = ses.run(src)                        # includes the D2H fetch (host sync) like predict()
        dt = (time.perf_counter() - t0) / iters
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ses.run(src, fetch=False)
        e1.record()
        torch.cuda.synchronize()
        out[name] = {"end_to_end_ms": round(dt * 1e3, 3), "device_ms": round(e0.elapsed_time(e1) / iters, 3),
                     "candidates": int(ses.det.count.item()), "kept": len(dets)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bf16 nc=80 (config 3) and bs=1 inference (config 5) legs")
    ap.add_argument("--collectives-at-world-1", action="store_true",
                    help="with ONE rank under torch.distributed.run: still broadcast and all-reduce the gradient buckets over RCCL")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank path with several ranks sharing one GPU)")
    args = ap.parse_args()

    FORCE_COLLECTIVES[0] = bool(args.collectives_at_world_1)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)          # before anything touches the GPU in this process
    try:
        return run(args)
    except Exception as e:                # a failed collective / timed-out rank: report and exit non-zero (never re-exec)
        from yolo_from_scratch_amd.training import CollectiveError
        if isinstance(e, CollectiveError) or "NCCL" in str(e) or "ProcessGroup" in type(e).__name__:
            print(f"bench.py: rank {os.environ.get('RANK', '0')}: distributed failure: {e}", file=sys.stderr, flush=True)
            sys.exit(3)
        raise


DIST_TIMEOUT_S = 300


def run(args):

    import torch.distributed as dist
    import yolo_from_scratch_amd as y
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_pg = args.gpus > 1 or "WORLD_SIZE" in os.environ        # under torch.distributed.run even ONE rank goes through RCCL
    if use_pg:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count()
        if args.backend == "nccl":
            if local >= ndev:
                raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible")
            torch.cuda.set_device(local)
            # a finite collective timeout: a dead rank ends the job (non-zero exit of every rank through the watchdog / through
            # GradBuckets.wait -> CollectiveError) instead of hanging the other seven forever
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=timedelta(seconds=DIST_TIMEOUT_S))
        else:
            local = local % max(ndev, 1)
            dist.init_process_group(args.backend, timeout=timedelta(seconds=DIST_TIMEOUT_S))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    headline = (NC, IMG, BATCH, DTYPE, SIZE) == (1, 640, 64, "f32", "s")
    base, model, trainer, imgs, targets = timed_training(y, dist, dev, rank, world, NC, IMG, BATCH, DTYPE, SIZE, args.steps, args.warmup)
    result = {
        "metric": "images/sec training step, 640x640 bs=64/GPU" if headline else
                  f"images/sec training step, {IMG}x{IMG} bs={BATCH}/GPU nc={NC} {DTYPE} size {SIZE} (informational shape)",
        "value": base["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": base["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": f"nc={NC} {IMG}x{IMG} bs={BATCH}/GPU full training step (fwd+loss+bwd+clip10+Adam), model size {SIZE}, "
                               + ("fp32 MFMA" if DTYPE == "f32" else "bf16 MFMA convolutions, bf16 activations, fp32 master weights / statistics / loss"),
                   "global_batch": BATCH * world, "parallelism": f"dp{world}"},
        "loss": base["loss"],
    }
    if rank == 0 and not args.no_roofline:
        roof, by_op = (roofline_bf16 if DTYPE == "bf16" else roofline_f32)(model, trainer, imgs, targets, NC, IMG, BATCH, SIZE)
        if DTYPE == "f32":
            # the metric is the whole training step: 3 x the forward-convolution FLOPs minus the first layer's backward-data
            # (SURVEY 8d), all algorithmic, against the fp32 MFMA peak
            step_flops = step_conv_flops(model._plan_for(imgs)) * BATCH
            roof["step_algorithmic_gflop"] = round(step_flops / 1e9, 1)
            roof["step_frac"] = round(step_flops / (base["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
        result["roofline"], result["forward_ms_by_op"] = roof, by_op
    if world > 1:
        dist.barrier()
    del trainer, model, imgs, targets
    torch.cuda.empty_cache()

    # ---- extra keys of the same line (headline run only): BASELINE config 3's per-GPU workload in bf16, config 5's latency
    if headline and not args.no_extras:
        b, model, trainer, imgs, targets = timed_training(y, dist, dev, rank, world, 80, 640, 64, "bf16", "s", 25, 5)
        extra = {"workload": "nc=80 640x640 bs=64/GPU full training step, bf16 MFMA convolutions + bf16 activations, fp32 master "
                             "weights / statistics / loss (BASELINE config 3 per GPU)", "dtype": "bf16", "n_gpus": world, **b}
        if rank == 0 and not args.no_roofline:
            extra["roofline"], extra["forward_ms_by_op"] = roofline_bf16(model, trainer, imgs, targets, 80, 640, 64, "s")
        result["bf16_nc80"] = extra
        if world > 1:
            dist.barrier()
        del trainer, model, imgs, targets
        torch.cuda.empty_cache()
        if rank == 0 and world == 1:
            result["infer"] = infer_latency(y, 1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
        if (IMG, SIZE) == (640, "s"):       # BASELINE configs[0]: the reference's own CPU-runnable case, batch 2
            b2 = cpu_baseline(batch=2, steps=5)
            result["cpu_baseline"]["config1_batch2"] = {"value": b2["value"], "unit": b2["unit"], "sample": b2["sample"]}
            if headline:                    # the metric's own batch ("B=64 if memory/time allow"): ~1.5 s per step on 16 cores
                b64 = cpu_baseline(batch=BATCH, steps=3)
                result["cpu_baseline"]["batch64"] = {"value": b64["value"], "unit": b64["unit"], "sample": b64["sample"]}
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
