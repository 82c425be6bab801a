"""One rank of the 2-rank HipTrainer rehearsal (launched by tests/test_gpu_dp.py through torch.distributed.run).

Every rank drives the HIP kernels on the ONE visible GPU (cuda:0) with its own shard; gradients are exchanged over
gloo (the collective backend is plumbing: on a multi-GPU node the same code path runs over RCCL, bench.py).  Rank r
writes what it saw to <out>/rank<r>.pt; the parent test compares with the CPU oracle run on the two shards."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    os.makedirs(out, exist_ok=True)
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
    big = len(sys.argv) > 4 and sys.argv[4] == "big"          # profiling runs: a step long enough to see the overlap
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    import yolo_from_scratch_amd as y
    dev = torch.device("cuda:0")
    nc, S, B = (1, 640, 16) if big else (1, 160, 2)
    torch.manual_seed(0 if rank == 0 else 12345)       # rank 1 starts from DIFFERENT weights: the broadcast must fix that
    m = y.YOLO(num_classes=nc, img_size=S).to(dev)
    kw = {} if dtype == "f32" else {"dtype": dtype}
    tr = y.HipTrainer(m, lr=1e-3, max_norm=10.0, n_buckets=4, **kw)
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1000 + rank)).to(dev)
    tg = [t.to(dev) for t in y.synthetic_targets(B, nc, S, 6, 2000 + rank)]
    p0 = tr.flat_p.clone()
    losses, norms, ps = [], [], []
    for _ in range(steps):
        losses.append(tr.step(x, tg)[:4].cpu().clone())
        norms.append(float(tr.norm.item()))
        ps.append(tr.flat_p.cpu().clone())
    torch.cuda.synchronize()
    segs = tr._segments[1]
    torch.save({"rank": rank, "world": tr.world, "p0": p0.cpu(), "p": tr.flat_p.cpu(), "p_steps": torch.stack(ps), "g": tr.flat_g.cpu(),
                "losses": torch.stack(losses), "norms": norms, "n_segments": len(segs),
                "n_reduces": sum(1 for _, r in segs if r is not None),
                "bn_rm": m.stem[1].running_mean.cpu().clone()}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
