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
