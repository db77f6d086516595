"""CPU, world_size 2 over gloo: the N>1 logic (sharding, DDP gradient averaging, metric reduction, max-over-ranks
timing) that bench.py / training.py rely on.  The HIP kernels themselves have no CPU path, so the canonicalizer here
is the IdentityCanonicalization (pure bookkeeping) around a small prediction network."""
import os
import socket

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


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import equiadapt_amd as ea
        from equiadapt_amd import training as tr

        torch.manual_seed(0)  # identical initial weights on every rank
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
                                  torch.nn.Flatten(), torch.nn.Linear(4, 5))
        model = tr.CanonicalizedClassifier(ea.IdentityCanonicalization(), net,
                                           tr.LossWeights(task_weight=1.0, prior_weight=1.0))
        ddp = tr.wrap_ddp(model)
        opt, _ = tr.configure_optimizer(model, 1e-2, 1e-3, kind="sgd", max_epochs=30)
        g = torch.Generator().manual_seed(123)
        X, Y = torch.randn(8, 3, 8, 8, generator=g), torch.randint(0, 5, (8,), generator=g)
        lo, hi = tr.shard_range(8, rank, world)
        out = tr.train_step(ddp, opt, X[lo:hi], Y[lo:hi])
        metrics = tr.reduce_metrics({"loss": out["loss"], "acc": out["acc"]})
        flat = torch.cat([p.detach().flatten() for p in model.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        # max-over-ranks timing as bench.py does it
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            # plain lists: tensors would travel through shared memory owned by a process that is about to exit
            results.put({"params": [g_.tolist() for g_ in gathered], "metrics": metrics, "tmax": t.item(),
                         "shards": [tr.shard_range(8, r, world) for r in range(world)]})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_gloo_training_step_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    results = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    got = results.get()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    got["params"] = [torch.tensor(v) for v in got["params"]]
    assert got["shards"] == [(0, 4), (4, 8)]
    assert got["tmax"] == 2.0
    assert torch.equal(got["params"][0], got["params"][1]), "replicas diverged after the all-reduced step"

    # single process on the full batch: DDP averages per-rank mean losses == full-batch mean (equal shard sizes)
    import equiadapt_amd as ea
    from equiadapt_amd import training as tr

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
                              torch.nn.Flatten(), torch.nn.Linear(4, 5))
    model = tr.CanonicalizedClassifier(ea.IdentityCanonicalization(), net, tr.LossWeights(1.0, 1.0))
    opt, sched = tr.configure_optimizer(model, 1e-2, 1e-3, kind="sgd", max_epochs=30)
    assert sched.milestones == {10: 1, 15: 1}
    g = torch.Generator().manual_seed(123)
    X, Y = torch.randn(8, 3, 8, 8, generator=g), torch.randint(0, 5, (8,), generator=g)
    out = tr.train_step(model, opt, X, Y)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(flat, got["params"][0], atol=1e-6, rtol=1e-5)
    assert abs(out["loss"].item() - got["metrics"]["loss"]) < 1e-5


def test_loss_composition_follows_the_reference():
    from equiadapt_amd import training as tr

    class FakeOpt(torch.nn.Module):
        def forward(self, x):
            return x

        def get_optimization_specific_loss(self):
            return torch.tensor(3.0)

        def get_prior_regularization_loss(self):
            return torch.tensor(5.0)

        def get_identity_metric(self):
            return torch.tensor(0.25)

    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 2))
    x, y = torch.zeros(2, 3, 2, 2), torch.zeros(2, dtype=torch.long)
    m = tr.CanonicalizedClassifier(FakeOpt(), net, tr.LossWeights(task_weight=0.0, prior_weight=2.0, group_contrast_weight=0.5))
    out = m(x, y)
    # 2 x 0.5 x 3 (the reference adds the contrast term twice) + 2 x 5
    assert out["loss"].item() == pytest.approx(13.0)
    m.weights.reference_double_contrast = False
    assert m(x, y)["loss"].item() == pytest.approx(11.5)
    assert tr.shard_range(10, 0, 4) == (0, 3) and tr.shard_range(10, 3, 4) == (8, 10)


@pytest.mark.timeout(300)
def test_bench_gpus_2_self_spawns_two_ranks_dry_run():
    """`python bench.py --gpus 2` from a plain shell (no launcher environment) must start 2 ranks itself and report n_gpus = 2.
    --dry-run keeps it to the launcher logic (CPU, gloo, empty step): rendezvous, barrier, max-over-ranks timing, one JSON line."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, env=env, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["dry_run"] is True and line["backend"] == "gloo"
    assert line["steps"] == 3 and line["warmup"] == 1 and len(line["per_rank_ms_per_step"]) == 2
    # N = 1 stays a single process without a process group
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "0", "--dry-run"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["backend"] is None


def test_optimizer_selection_rules_follow_the_reference():
    """examples/images/classification/model.py:184-239 and examples/pointcloud/classification/model.py:245-300."""
    import equiadapt_amd as ea
    from equiadapt_amd import training as tr

    assert tr.select_image_optimizer_kind("resnet50", "cifar10") == "sgd"
    assert tr.select_image_optimizer_kind("resnet50", "rotated_mnist") == "adamw"
    assert tr.select_image_optimizer_kind("vit", "cifar10") == "adamw"
    net = torch.nn.Linear(4, 2)
    m = tr.CanonicalizedClassifier(ea.IdentityCanonicalization(), net)
    opt, sched = tr.configure_optimizer(m, 1e-3, 2e-3, kind=None, max_epochs=200, prediction_network_architecture="resnet50",
                                        dataset_name="cifar10")
    assert isinstance(opt, torch.optim.SGD) and opt.param_groups[0]["momentum"] == 0.9 and opt.param_groups[0]["weight_decay"] == 5e-4
    assert [g["lr"] for g in opt.param_groups] == [1e-3, 2e-3] and sched.milestones == {33: 1, 66: 1, 100: 1}
    opt, sched = tr.configure_optimizer(m, 1e-3, 2e-3, kind=None, prediction_network_architecture="vit", dataset_name="cifar10")
    assert isinstance(opt, torch.optim.AdamW) and sched is None
    opt, sched = tr.configure_pointcloud_optimizer(m, 1e-3, 1e-3, "SGD", "cosine", 250)
    assert isinstance(opt, torch.optim.SGD) and opt.param_groups[0]["lr"] == pytest.approx(0.1) and opt.param_groups[0]["weight_decay"] == 1e-4
    assert isinstance(sched, torch.optim.lr_scheduler.CosineAnnealingLR) and sched.T_max == 250 and sched.eta_min == 1e-3
    opt, sched = tr.configure_pointcloud_optimizer(m, 1e-3, 1e-3, "SGD", "step")
    assert isinstance(sched, torch.optim.lr_scheduler.StepLR) and sched.step_size == 20 and sched.gamma == 0.7
    opt, sched = tr.configure_pointcloud_optimizer(m, 1e-3, 1e-3, "Adam")
    assert isinstance(opt, torch.optim.Adam) and sched is None and opt.param_groups[0]["weight_decay"] == 1e-4
    with pytest.raises(NotImplementedError):
        tr.configure_pointcloud_optimizer(m, 1e-3, 1e-3, "SGD", "linear")
    with pytest.raises(NotImplementedError):
        tr.configure_pointcloud_optimizer(m, 1e-3, 1e-3, "RMSprop")
