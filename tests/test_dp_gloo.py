"""world_size=2 CPU tests (gloo) of the data-parallel host logic: bucket planning over the flat gradient
buffer, overlapped all-reduce, and DDP-style mean-of-replica-gradients semantics, checked against the CPU
oracle run on the two shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yolo_from_scratch_amd.training import GradBuckets
        torch.manual_seed(100 + rank)
        sizes = [12, 300, 7, 64, 1000, 33, 5, 256]
        spans, off = [], 0
        for n in sizes:
            spans.append((off, (n + 3) // 4 * 4))
            off += (n + 3) // 4 * 4
        flat = torch.randn(off)
        local = flat.clone()
        ready = [80, 70, 60, 50, 40, 30, 20, 10]          # later parameters finish earlier in the backward
        gb = GradBuckets(flat, None, n_buckets=3)
        segs = gb.plan_segments(spans, ready, 90)
        # every element is covered exactly once, op indices are non-decreasing and end at n_ops
        cover = torch.zeros(off)
        last = 0
        for end, rng in segs:
            assert end >= last
            last = end
            if rng is not None:
                cover[rng[0]:rng[1]] += 1
                for (o, n), r in zip(spans, ready):
                    if rng[0] <= o < rng[1]:
                        assert r <= end      # a bucket is reduced only after all of its gradients are final
        assert last == 90 and bool((cover == 1).all())
        for end, rng in segs:
            gb.launch(rng)
        gb.wait()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered)
        q.put((rank, float((flat - want).abs().max()), len(segs)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, nseg in res:
        assert err < 1e-6 and nseg >= 3


def _oracle_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import yolo_from_scratch_amd as y
        from yolo_from_scratch_amd.training import GradBuckets
        from oracle import yolo_oracle as orc
        torch.set_num_threads(2)
        nc, S, B = 1, 64, 2
        torch.manual_seed(0)
        m = y.YOLO(num_classes=nc, img_size=S)
        P = {k: v.clone() for k, v in m.state_dict().items()}
        names = [n for n, _ in m.named_parameters()]
        for n in names:
            P[n].requires_grad_(True)
        x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1000 + rank))
        tg = y.synthetic_targets(B, nc, S, 4, 2000 + rank)
        orc.loss_multiscale(orc.forward(P, x, nc, True), tg, orc.anchors_of(P), nc)[0].backward()
        spans, off = [], 0
        for n in names:
            k = (P[n].numel() + 3) // 4 * 4
            spans.append((off, k))
            off += k
        flat = torch.zeros(off)
        for (o, k), n in zip(spans, names):
            flat[o:o + P[n].numel()] = P[n].grad.reshape(-1)
        mine = flat.clone()
        gb = GradBuckets(flat, None, n_buckets=4)
        for end, rng in gb.plan_segments(spans, list(range(len(names), 0, -1)), len(names) + 1):
            gb.launch(rng)
        gb.wait()
        flat /= world                                     # the trainer folds this factor into clip+Adam
        others = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(others, mine)
        mean = sum(others) / world
        total, coef = orc.clip_coef([flat], 10.0)
        q.put((rank, float((flat - mean).abs().max()), total, coef))
    finally:
        dist.destroy_process_group()


def test_dp_mean_of_replica_gradients_matches_oracle_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_oracle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] < 1e-7 for r in res)
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]      # both ranks clip the averaged gradient identically


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yolo_from_scratch_amd.training import GradBuckets
        errs = []
        for nb in (1, 4, 8):
            torch.manual_seed(7 * nb + rank)
            sizes = [9, 400, 31, 64, 2048, 33, 5, 256, 12, 800, 100, 3]
            spans, off = [], 0
            for n in sizes:
                spans.append((off, (n + 3) // 4 * 4))
                off += (n + 3) // 4 * 4
            flat = torch.randn(off)
            local = flat.clone()
            ready = list(range(10 * len(sizes), 0, -10))
            gb = GradBuckets(flat, None, n_buckets=nb)
            segs = gb.plan_segments(spans, ready, 10 * len(sizes) + 5)
            cover = torch.zeros(off)
            for end, rng in segs:
                if rng is not None:
                    cover[rng[0]:rng[1]] += 1
                gb.launch(rng)
            gb.wait(timeout_s=60.0)
            assert bool((cover == 1).all()) and segs[-1][0] == 10 * len(sizes) + 5
            assert sum(1 for _, r in segs if r is not None) <= nb
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            errs.append(float((flat - sum(gathered)).abs().max()))
        q.put((rank, max(errs)))
    finally:
        dist.destroy_process_group()


def test_bucket_plans_world4_gloo():
    """plan_segments with 1, 4 and 8 buckets on a world of FOUR ranks: every gradient element reduced exactly once, never before
    its gradient is final, at most n_buckets collectives, sums equal to the all-gathered reference."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(err < 1e-5 for _, err in res)


def _dead_rank_worker(rank, world, port, q):
    from datetime import timedelta
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=20))
    from yolo_from_scratch_amd.training import GradBuckets, CollectiveError
    def leave():                                       # flush the result queue, then die without any distributed teardown
        q.close()
        q.join_thread()
        os._exit(0)
    if rank == 1:
        q.put((rank, "gone"))
        leave()                                       # dies before the collective: rank 0 must not hang in wait()
    flat = torch.randn(1024)
    gb = GradBuckets(flat, None, n_buckets=2)
    try:
        for end, rng in gb.plan_segments([(0, 512), (512, 512)], [2, 1], 3):
            gb.launch(rng)
        gb.wait(timeout_s=5.0)
        q.put((rank, "no error"))
    except CollectiveError as e:
        q.put((rank, "CollectiveError"))
    leave()


def test_dead_rank_surfaces_as_collective_error():
    """SURVEY section 5 (RCCL async error -> abort): a rank that disappears must not hang the others in GradBuckets.wait()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dead_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res[1] == "gone" and res[0] == "CollectiveError", res
