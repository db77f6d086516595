"""What does the closing `dist.barrier()` + synchronize of bench.py's timed region cost under a ONE-rank RCCL job?
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 tools/diag/barrier_cost.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
x = torch.randn(4096, 4096, device="cuda")


def work():
    for _ in range(20):
        torch.mm(x, x)


for variant in ("barrier()", "barrier(device_ids=[0])", "all_reduce(1 float)"):
    def bar():
        if variant == "barrier()":
            dist.barrier()
        elif variant.startswith("barrier(d"):
            dist.barrier(device_ids=[0])
        else:
            t = torch.zeros(1, device="cuda")
            dist.all_reduce(t)
        torch.cuda.synchronize()
    bar()
    rows = []
    for _ in range(5):
        work()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bar()
        idle = time.perf_counter() - t0
        bar()
        t0 = time.perf_counter()
        work()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        work()
        bar()
        t2 = time.perf_counter()
        rows.append((idle * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print(variant, "| idle barrier ms, work+sync ms, work+barrier+sync ms:", ["%.3f %.3f %.3f" % r for r in rows])
dist.destroy_process_group()
