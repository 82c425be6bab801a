"""Probe: does replaying forward + loss + backward of one training step from a hipGraph beat the eager launches?
usage (GPU box): python tools/graph_probe.py [f32|bf16]   -- prints ms per step eager vs replay (Adam stays eager in both)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_from_scratch_amd as y
from yolo_from_scratch_amd.training import run_loss_kernel, _anchors18, _stream, check_targets

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, IMG, NC = 64, 640, 1
model = y.YOLO(num_classes=NC, img_size=IMG).to(dev)
tr = y.HipTrainer(model, lr=1e-3, max_norm=10.0, dtype=dtype)
imgs = torch.rand(B, 3, IMG, IMG).to(dev)
targets = [t.to(dev) for t in y.synthetic_targets(B, NC, IMG, 8, 2000)]
for _ in range(3):
    tr.step(imgs, targets)
torch.cuda.synchronize()


def body():
    st = _stream(dev)
    plan = model._plan_for(imgs)
    heads = [v for v, _ in plan.outputs]
    model._load_input(plan, imgs)
    plan.run_forward(st)
    run_loss_kernel([v.buf.data for v in heads], targets, [v.buf.grad for v in heads], _anchors18(model.anchors),
                    [v.H for v in heads], plan.B, NC, None, None, tr.loss_out, tr._loss_ws, st, dpred_bf16=plan.bf16,
                    dpred_ld=[v.ldg for v in heads] if plan.bf16 else None)
    begin = 0
    for end, rng in tr._segments[1]:
        plan.run_backward(st, begin, end)
        begin = end


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        tr.apply_update()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body(); tr.apply_update()
    torch.cuda.synchronize()
    e = timeit(body)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    r = timeit(g.replay)
    e2 = timeit(body)
print(f"{dtype}: eager {e:.3f} / {e2:.3f} ms, graph replay {r:.3f} ms per step; loss {tr.loss_out[:4].tolist()}")
