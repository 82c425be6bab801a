import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, numpy as np
import yolo_from_scratch_amd as y
from oracle import yolo_oracle as orc
from test_oracle_pinned import loss_inputs
g, nc, preds, targets = loss_inputs("nc1")
anchors = [torch.tensor(a, dtype=torch.float32).cuda() for a in orc.DEFAULT_ANCHORS]
pg = [p.cuda() for p in preds]; tg = [t.cuda() for t in targets]
ref = None; bad = 0
for it in range(300):
    out = [v.item() for v in y.yolo_loss_multiscale(pg, tg, anchors, nc)]
    if ref is None: ref = out
    if out != ref:
        bad += 1
        if bad < 5: print(it, out, ref)
print("bad", bad, "of 300; golden", g["nc1/scalars"], "got", ref)
# with gradient path
bad = 0; gref = None
for it in range(100):
    pp = [p.clone().requires_grad_(True) for p in pg]
    tot = y.yolo_loss_multiscale(pp, tg, anchors, nc)[0]
    tot.backward()
    s = [float(p.grad.double().abs().sum()) for p in pp] + [tot.item()]
    if gref is None: gref = s
    if s != gref:
        bad += 1
        if bad < 5: print("grad", it, s, gref)
print("grad bad", bad)
