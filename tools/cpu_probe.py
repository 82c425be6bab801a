import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
import bench
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    t0 = time.time()
    r = bench.cpu_baseline(batch=8, steps=2)
    print(nt, r["value"], "img/s", round(time.time() - t0, 1), "s")
