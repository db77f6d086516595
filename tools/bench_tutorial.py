"""The two training loops whose tqdm read-outs are the only throughput numbers in the reference repository
(tutorials/images/understanding_discrete_canonicalization.ipynb):

  prior      cell 17 + 21: GroupEquivariantImageCanonicalization + ESCNNEquivariantNetwork(16 ch, k = 9, 3 layers, C4), crop 0.9,
             resize 64, images 3 x 64 x 64, B = 512, Adam(lr 0.002) on the network, loss = 100 x prior regularisation;
             per step: canonicalize(image) -> loss -> backward -> step -> get_identity_metric().  3.63-3.68 it/s = 1,860-1,885 img/s
  optimized  cell 26 + 30: OptimizedGroupEquivariantImageCanonicalization + ConvNetwork(16 ch, k = 5, 3 layers, 128), C4,
             artifact_err_wt = 1000, loss = 100 x prior + 0.1 x optimisation-specific loss.  3.75-3.84 it/s = 1,920-1,965 img/s
(hardware unstated; synthetic CIFAR-shaped data here: randn in place of the normalised images).

    python tools/bench_tutorial.py [--leg prior|optimized|both] [--steps 20] [--warmup 5] [--batch 512]
Imported by bench.py (`tutorial` object of its JSON line).
"""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(leg: str, dev):
    import equiadapt_amd as ea

    torch.manual_seed(2)
    if leg == "prior":
        hp = types.SimpleNamespace(input_crop_ratio=0.9, beta=1.0, resize_shape=64)
        net = ea.ESCNNEquivariantNetwork((3, 64, 64), out_channels=16, kernel_size=9, num_layers=3, group_type="rotation", num_rotations=4)
        can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64))
    else:
        hp = types.SimpleNamespace(group_type="rotation", num_rotations=4, input_crop_ratio=0.9, beta=1.0, resize_shape=64,
                                   artifact_err_wt=1000, learn_ref_vec=False)
        net = ea.ConvNetwork((3, 64, 64), out_channels=16, kernel_size=5, num_layers=3, out_vector_size=128)
        can = ea.OptimizedGroupEquivariantImageCanonicalization(net, hp, (3, 64, 64))
    can = can.to(dev).train()
    opt = torch.optim.Adam(can.canonicalization_network.parameters(), lr=0.002)
    return can, opt


def make_step(leg: str, can, opt, xs):
    it = [0]

    def step():
        x = xs[it[0] % len(xs)]
        it[0] += 1
        opt.zero_grad()
        can.canonicalize(x)
        loss = 100 * can.get_prior_regularization_loss()
        if leg == "optimized":
            loss = loss + 0.1 * can.get_optimization_specific_loss()
        loss.backward()
        opt.step()
        return loss, can.get_identity_metric()
    return step


def run_leg(leg: str, dev, batch: int, steps: int, warmup: int, barrier=None):
    can, opt = build(leg, dev)
    g = torch.Generator().manual_seed(300)
    xs = [torch.randn(batch, 3, 64, 64, generator=g).to(dev) for _ in range(3)]
    step = make_step(leg, can, opt, xs)
    for _ in range(warmup):
        step()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, metric = step()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = float(loss.detach())
    assert loss == loss, "NaN loss"
    ref = {"prior": (1860, 1885), "optimized": (1920, 1965)}[leg]
    return {"images_s": batch * steps / dt, "ms_per_step": dt / steps * 1e3, "it_s": steps / dt, "batch": batch, "steps": steps,
            "final_loss": loss, "identity_metric": float(metric),
            "reference_tutorial_images_s": list(ref), "reference_hardware": "unstated GPU (notebook output)",
            "vs_reference_tutorial": batch * steps / dt / ref[1]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", default="both", choices=["prior", "optimized", "both"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for leg in (["prior", "optimized"] if args.leg == "both" else [args.leg]):
        out[leg] = run_leg(leg, dev, args.batch, args.steps, args.warmup)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
