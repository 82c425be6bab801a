"""ONE rank over RCCL (torch's "nccl" backend) on the box's one GPU, launched by tests/test_gpu_dp.py through
torch.distributed.run: the process-group creation bench.py uses at N > 1 (device_id, HSA_ENABLE_IPC_MODE_LEGACY=0), the
broadcast of the flat parameter buffer, four asynchronous all-reduces of gradient-bucket slices interleaved with the backward
segments of yh_run, wait, clip, Adam -- compared bitwise with the non-distributed step in the same process."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(y, dev, dtype, distributed):
    nc, S, B = 2, 160, 2
    torch.manual_seed(0)
    m = y.YOLO(num_classes=nc, img_size=S).to(dev)
    tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0, n_buckets=4, dtype=dtype, collectives_at_world_1=distributed)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1000)).to(dev)
    tg = [t.to(dev) for t in y.synthetic_targets(B, nc, S, 6, 2000)]
    losses = []
    for _ in range(3):
        losses.append(tr.step(x, tg)[:4].cpu().clone())
    torch.cuda.synchronize()
    segs = tr._segments[1]
    return {"p": tr.flat_p.cpu().clone(), "g": tr.flat_g.cpu().clone(), "losses": torch.stack(losses),
            "n_reduces": sum(1 for _, r in segs if r is not None), "active": tr.buckets.active}


def main():
    out = sys.argv[1]
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import yolo_from_scratch_amd as y
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    for dtype in ("f32", "bf16"):
        res[dtype] = {"dist": run(y, dev, dtype, True), "plain": run(y, dev, dtype, False)}
    torch.save(res, os.path.join(out, "rccl1.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
