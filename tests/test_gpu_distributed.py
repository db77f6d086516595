"""RCCL (backend "nccl" on ROCm) + DistributedDataParallel around the HIP-backed canonicalizers, on the one GPU a test box has.

north_star's data-parallel loop is the reference's ``strategy="ddp"`` (examples/images/classification/train_utils.py:85-91,
step semantics model.py:59-127; point clouds: examples/pointcloud/classification/model.py:77-134).  With a single rank the
all-reduce is the identity, so a DDP-wrapped model must take EXACTLY the steps of the bare model; what the test proves is the
plumbing an N-rank job relies on:

  * ``init_process_group("nccl")`` comes up in this image (``HSA_ENABLE_IPC_MODE_LEGACY=0``, 127.0.0.1 rendezvous),
  * DDP's reducer (bucket copies + all-reduce on RCCL's own stream, autograd hooks) orders correctly against kernels that
    are launched through ctypes on torch's current stream -- a missing dependency would show up as torn gradients,
  * every parameter of the canonicalizer receives a gradient (DDP raises on unused parameters otherwise).

Each case runs in a spawned process (a process group cannot be re-initialised inside the pytest process) and returns
plain Python lists.
"""
import os
import socket
import types

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(kind, dev):
    import equiadapt_amd as ea
    from equiadapt_amd import training as tr

    torch.manual_seed(11)
    if kind == "images":
        # the headline canonicalizer (bench.py: ESCNN-shaped C8 network, 32 channels, k5, 3 layers, crop 0.8 -> resize 96: the FFT
        # training path) around a small prediction network
        net = ea.ESCNNEquivariantNetwork((3, 96, 96), out_channels=32, kernel_size=5, group_type="rotation", num_rotations=8, num_layers=3)
        hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=96)
        can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 224, 224))
        pred = torch.nn.Sequential(torch.nn.AvgPool2d(8), torch.nn.Flatten(), torch.nn.Linear(3 * 28 * 28, 32), torch.nn.ReLU(),
                                   torch.nn.Linear(32, 10))
        g = torch.Generator().manual_seed(5)
        xs = [torch.randn(8, 3, 224, 224, generator=g).to(dev) for _ in range(2)]
        ys = [torch.randint(0, 10, (8,), generator=g).to(dev) for _ in range(2)]
        model = tr.CanonicalizedClassifier(can, pred, tr.LossWeights(task_weight=1.0, prior_weight=100.0))
        opt_fn = lambda m: tr.configure_optimizer(m, 1e-3, 1e-3, kind="sgd", max_epochs=200)[0]  # noqa: E731
    else:
        hp = types.SimpleNamespace(n_knn=20, pooling="mean")
        can = ea.EquivariantPointcloudCanonicalization(ea.VNSmall(hp), hp)
        pred = torch.nn.Sequential(torch.nn.Conv1d(3, 32, 1), torch.nn.ReLU(), torch.nn.AdaptiveMaxPool1d(1), torch.nn.Flatten(),
                                   torch.nn.Linear(32, 40))
        g = torch.Generator().manual_seed(6)
        xs = [torch.randn(16, 3, 1024, generator=g).to(dev) for _ in range(2)]
        ys = [torch.randint(0, 40, (16,), generator=g).to(dev) for _ in range(2)]
        model = tr.CanonicalizedClassifier(can, pred, tr.LossWeights(task_weight=1.0, prior_weight=100.0))
        opt_fn = lambda m: tr.configure_pointcloud_optimizer(m, 1e-3, 1e-3, "SGD", "cosine", 250)[0]  # noqa: E731
    return model.to(dev).train(), opt_fn, xs, ys


def _run_steps(model, step_model, opt, xs, ys):
    """Two optimisation steps; -> (per-step losses, per-step gradients, final parameters + buffers) as CPU tensors."""
    from equiadapt_amd import training as tr

    torch.manual_seed(77)      # the dropout seeds of the fused blocks come from torch's CPU generator
    torch.cuda.manual_seed(77)
    losses, grads = [], []
    for x, y in zip(xs, ys):
        out = tr.train_step(step_model, opt, x, y)
        losses.append(out["loss"].detach().cpu())
        grads.append([None if p.grad is None else p.grad.detach().cpu().clone() for p in model.parameters()])
    state = [t.detach().cpu().clone() for t in list(model.parameters()) + list(model.buffers())]
    return losses, grads, state


def _worker(kind, port, q):
    import copy

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = {"ok": False}
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        from equiadapt_amd import _lib
        from equiadapt_amd import training as tr

        _lib.load()
        torch.backends.cudnn.deterministic = True
        # a real collective first: RCCL itself is up (sum over one rank = identity)
        t = torch.arange(1024, device=dev, dtype=torch.float32)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        res["allreduce_ok"] = bool(torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32)))
        res["backend"] = dist.get_backend()

        model, opt_fn, xs, ys = _build(kind, dev)
        bare = copy.deepcopy(model)
        bare2 = copy.deepcopy(model)
        ddp = tr.wrap_ddp(model, dev, force=True)
        res["wrapped"] = type(ddp).__name__
        a = _run_steps(bare, bare, opt_fn(bare), xs, ys)
        a2 = _run_steps(bare2, bare2, opt_fn(bare2), xs, ys)
        b = _run_steps(model, ddp, opt_fn(model), xs, ys)
        torch.cuda.synchronize()

        def same(u, v):
            return all((p is None and r is None) or (p is not None and r is not None and torch.equal(p, r)) for p, r in zip(u, v))

        res["self_deterministic"] = bool(all(torch.equal(l1, l2) for l1, l2 in zip(a[0], a2[0])) and all(same(g1, g2) for g1, g2 in zip(a[1], a2[1]))
                                         and same(a[2], a2[2]))
        res["loss_equal"] = bool(all(torch.equal(l1, l2) for l1, l2 in zip(a[0], b[0])))
        res["grads_equal"] = bool(all(same(g1, g2) for g1, g2 in zip(a[1], b[1])))
        res["state_equal"] = bool(same(a[2], b[2]))
        res["max_grad_diff"] = max(((p - r).abs().max().item() for g1, g2 in zip(a[1], b[1]) for p, r in zip(g1, g2) if p is not None), default=0.0)
        res["all_params_have_grad"] = bool(all(p is not None for p in b[1][-1]))
        res["grad_nonzero"] = bool(sum(p.abs().sum().item() for p in b[1][-1]) > 0)
        res["finite"] = bool(all(torch.isfinite(l).item() for l in b[0]))
        res["n_params"] = sum(p.numel() for p in model.parameters())
        res["ok"] = True
    except Exception as exc:  # noqa: BLE001 -- travels back to the test as text
        import traceback

        res["error"] = f"{exc!r}\n{traceback.format_exc()}"
    finally:
        try:
            import torch.distributed as dist

            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        q.put(res)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["images", "pointcloud"])
def test_nccl_world1_ddp_wrapped_step_equals_the_bare_step(kind):
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker, args=(kind, _free_port(), q))
    p.start()
    res = q.get()
    p.join(120)
    assert res["ok"], res.get("error")
    assert p.exitcode == 0
    assert res["backend"] == "nccl" and res["allreduce_ok"]
    assert res["wrapped"] == "DistributedDataParallel"
    assert res["finite"] and res["all_params_have_grad"] and res["grad_nonzero"]
    # the kernels are deterministic (no atomics in the gradients): two bare runs agree bit for bit ...
    assert res["self_deterministic"], res
    # ... and so must the run whose gradients went through DDP's buckets and the RCCL all-reduce
    assert res["loss_equal"] and res["grads_equal"] and res["state_equal"], res


# ----------------------------------------------------------------------------------------------------------------------
# world size 2 on the ONE GPU of the test box: two processes, both on cuda:0, gloo as the collective backend (RCCL refuses two
# ranks per device; gloo all-reduces CUDA tensors through the host).  Everything else is the N-rank job: wrap_ddp (bucket views),
# the ctypes-launched kernels writing gradients on torch's current stream, DDP's hooks averaging them over DIFFERENT shards.
# ----------------------------------------------------------------------------------------------------------------------
def _shards(kind, xs, ys, split):
    """Step i's batch cut into the two ranks' shards; `split` = size of rank 0's."""
    return [[(x[:split], y[:split]), (x[split:], y[split:])] for x, y in zip(xs, ys)]


def _by_value(o):
    """Tensors -> numpy arrays (pickled by value): torch's queue reduction shares CPU tensors through file descriptors the parent
    can no longer open once the worker has exited."""
    if torch.is_tensor(o):
        return o.detach().cpu().numpy()
    if isinstance(o, dict):
        return {k: _by_value(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_by_value(v) for v in o]
    return o


def _tensors(o):
    import numpy as np

    if isinstance(o, np.ndarray):
        return torch.from_numpy(o)
    if isinstance(o, dict):
        return {k: _tensors(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_tensors(v) for v in o]
    return o


def _seed_for(step, rank):
    return 1000 * (step + 1) + rank    # the fused blocks draw their dropout seeds from torch's CPU generator at forward time


def _worker2(kind, split, rank, port, q):
    import copy

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = {"ok": False, "rank": rank}
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        from equiadapt_amd import _lib
        from equiadapt_amd import training as tr

        _lib.load()
        torch.backends.cudnn.deterministic = True
        model, opt_fn, xs, ys = _build(kind, dev)          # same seeds on both ranks: identical initial replicas, identical full batches
        initial = copy.deepcopy(model)
        shards = _shards(kind, xs, ys, split)
        ddp = tr.wrap_ddp(model, dev)
        res["wrapped"] = type(ddp).__name__
        opt = opt_fn(model)
        losses, grads, metrics = [], [], []
        for i, per_rank in enumerate(shards):
            x, y = per_rank[rank]
            torch.manual_seed(_seed_for(i, rank))
            out = tr.train_step(ddp, opt, x.contiguous(), y.contiguous())
            losses.append(float(out["loss"].detach()))
            grads.append([p.grad.detach().cpu().clone() for p in model.parameters()])
            metrics.append(tr.reduce_metrics({k: v for k, v in out.items() if v.dim() == 0}))
        torch.cuda.synchronize()
        res.update(losses=losses, grads=grads, metrics=metrics, params=[p.detach().cpu().clone() for p in model.parameters()],
                   n_local=[int(s[rank][0].shape[0]) for s in shards])
        if rank == 0:
            # the single-process equivalent of the two-rank step.  DDP averages the per-rank gradients of per-rank MEAN losses, and
            # batch-norm statistics stay per replica (the reference uses plain DDP, no SyncBN: train_utils.py:89-91), so the
            # equivalent is "each shard forwarded on its own, loss = (L_0 + L_1) / 2" -- for unequal shards that is NOT the mean
            # over the concatenated batch, and neither is it in the reference.
            emul = initial
            eopt = opt_fn(emul)
            e_grads, e_losses = [], []
            for i, per_rank in enumerate(shards):
                eopt.zero_grad(set_to_none=True)
                ls = []
                for r, (x, y) in enumerate(per_rank):
                    torch.manual_seed(_seed_for(i, r))
                    o = emul(x.contiguous(), y.contiguous())
                    (o["loss"] / 2).backward()
                    ls.append(float(o["loss"].detach()))
                e_grads.append([p.grad.detach().cpu().clone() for p in emul.parameters()])
                e_losses.append(ls)
                eopt.step()
            torch.cuda.synchronize()
            res.update(e_grads=e_grads, e_losses=e_losses, e_params=[p.detach().cpu().clone() for p in emul.parameters()])
        dist.barrier()
        res["ok"] = True
    except Exception as exc:  # noqa: BLE001
        import traceback

        res["error"] = f"{exc!r}\n{traceback.format_exc()}"
    finally:
        try:
            import torch.distributed as dist

            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        q.put(_by_value(res))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,split", [("images", 5), ("images", 4), ("pointcloud", 10), ("pointcloud", 8)])
def test_gloo_world2_on_one_gpu_real_canonicalizers_average_different_shards(kind, split):
    """Reference: examples/images/classification/train_utils.py:89-91 (strategy="ddp"), model.py:59-127; point clouds
    examples/pointcloud/classification/train_utils.py:50-51.  Two ranks, different (and unequal: 5 + 3, 10 + 6) shards, two
    optimisation steps of the headline canonicalizer / the VNSmall canonicalizer around a small predictor."""
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        q = ctx.SimpleQueue()
        port = _free_port()
        procs = [ctx.Process(target=_worker2, args=(kind, split, r, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = [_tensors(q.get()), _tensors(q.get())]
        for p in procs:
            p.join(300)
        res = {r["rank"]: r for r in got}
        # the rendezvous port is picked by binding and releasing it: another process can take it in between.  Two retries for
        # failures of the process-group plumbing only (never for a numerical mismatch: those are asserted below, on the results)
        infra = [str(res[r].get("error", "")) for r in (0, 1) if not res[r]["ok"]]
        if attempt < 2 and infra and all(any(k in e for k in ("ddress already in use", "Connection", "connect", "timed out", "Timeout", "store")) for e in infra):
            continue
        break
    for r in (0, 1):
        assert res[r]["ok"], res[r].get("error")
        assert res[r]["wrapped"] == "DistributedDataParallel"
    assert all(p.exitcode == 0 for p in procs)
    total = 8 if kind == "images" else 16
    assert res[0]["n_local"] == [split, split] and res[1]["n_local"] == [total - split, total - split]
    # (i) the replicas stay identical: same averaged gradients after every step, same parameters at the end -- bit for bit
    for g0, g1 in zip(res[0]["grads"], res[1]["grads"]):
        assert all(torch.equal(a, b) for a, b in zip(g0, g1))
    assert all(torch.equal(a, b) for a, b in zip(res[0]["params"], res[1]["params"]))
    assert any(a.abs().sum() > 0 for a in res[0]["grads"][-1])
    # ... although the ranks saw different data (their local losses differ)
    assert res[0]["losses"] != res[1]["losses"]
    # (ii) equal to the single-process evaluation of the same two shards within fp32 reduction noise (the only differences: gloo
    # sums g_0 + g_1 then scales, autograd accumulates g_0 / 2 + g_1 / 2)
    for step, (g, e) in enumerate(zip(res[0]["grads"], res[0]["e_grads"])):
        for a, b in zip(g, e):
            scale = max(b.abs().max().item(), 1e-12)
            assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-9, (kind, step, (a - b).abs().max().item(), scale)
    for a, b in zip(res[0]["params"], res[0]["e_params"]):
        assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1.0), (a - b).abs().max().item()
    for step in range(2):
        assert res[0]["e_losses"][step] == pytest.approx([res[0]["losses"][step], res[1]["losses"][step]], rel=1e-5)
    # (iii) reduce_metrics = the mean over ranks (the reference's sync_dist=True), the same on both ranks
    for step in range(2):
        m0, m1 = res[0]["metrics"][step], res[1]["metrics"][step]
        assert m0 == m1
        assert m0["loss"] == pytest.approx((res[0]["losses"][step] + res[1]["losses"][step]) / 2, rel=1e-6)


@pytest.mark.timeout(900)
def test_bench_gpus_4_forward_over_gloo_on_one_gpu_prints_one_line():
    """The driver's scaling run is `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` on an 8-GPU node; what a
    one-GPU test box can exercise of it is everything except RCCL's transport: `bench.py --gpus 4 --mode forward` self-spawns four
    ranks (LOCAL_RANK 0..3 folded onto cuda:0, gloo: RCCL refuses two ranks per device), every rank loads the library, builds its
    own canonicalizer and batches (per-rank seeds), passes the barriers on both sides of the timed region, and rank 0 prints ONE
    JSON line with n_gpus = 4, one per-rank time for each of the four ranks, the global batch, and a green parity record for its
    own batch (reference: examples/images/classification/train_utils.py:89-91: one process per GPU)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EQA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--mode", "forward", "--steps", "3", "--warmup", "1",
                          "--batch", "64", "--check-images", "16"], capture_output=True, text=True, env=env, timeout=850)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["backend"] == "gloo" and line["ranks_share_gpu"] is True and line["rccl_ranks"] == 0
    assert len(line["per_rank_ms_per_step"]) == 4 and all(t > 0 for t in line["per_rank_ms_per_step"])
    assert line["config"]["global_batch"] == 256 and line["config"]["batch_per_gpu"] == 64
    assert line["parity_ok"] is True and line["self_check"]["ok"] is True
    assert line["value"] > 0 and abs(line["value"] - 256 * 1e3 / line["ms_per_step"]) <= 1e-6 * line["value"]
