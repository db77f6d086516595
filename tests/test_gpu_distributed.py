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
