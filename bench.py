"""bench.py -- canonicalize+invert throughput on BASELINE.json's headline config (224x224x3, C8), 1..N GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no torch.distributed environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one process per GPU, RCCL =
backend "nccl" on ROCm); under an existing launcher (RANK / WORLD_SIZE set, as the driver does for N > 1) it joins that job.

One "step" of the headline = one pass of the hot path over one batch that is already resident in HBM:
    y   = canonicalizer(x)                      # crop+resize -> canonicalization network -> group pool/argmax
                                                #   -> fused pad/rotate/crop (eqa_canon_transform_fwd)
    out = canonicalizer.invert_canonicalization(f, induced_rep_type="scalar")     # eqa_invert_action_fwd
with x, f : (B, 3, 224, 224) fp32 synthetic, and the canonicalization network = ESCNNEquivariantNetwork
(out_channels=32, kernel_size=5, num_layers=3, C8) on a 96x96 crop+resize -- the reference's default
(examples/images/classification/configs/canonicalization/group_equivariant.yaml).  The wrapped prediction
network (ResNet-50) is NOT part of the headline.  Ranks shard the batch; the forward path has no
collective, so scaling is weak (per-GPU batch fixed); W warm-up steps, then exactly K steps between
barrier + synchronize pairs, max over ranks.

Rank 0 prints ONE JSON line (contract in the task statement).  Extra objects:
  roofline      HBM roofline of the dominant hand-written HBM kernel (the canonicalizing transform),
                from HIP events recorded inside the timed region;
  roofline_dominant  the same object for the kernel the step spends most of its time in (the hand-written complex GEMM:
                MFMA-bound), so that both the metric's kernel and the time-dominant one are priced;
  group_action  transform+invert only (random group index), the figure the "% HBM roofline" target is about;
  stages        the canonicalization network's kernels, each against the roofline that bounds it (HIP events inside the
                timed region);
  self_check    EVERY image of rank 0's first timed batch against the CPU oracle (activations, group index match rate above the
                tie margin + tie rate, canonicalized pixels, inverted pixels: max and RMS); the configs legs carry the same
                record under "parity";
  train         the data-parallel TRAINING step north_star's >=6x target is about: CanonicalizedClassifier
                (this canonicalizer + a plain-torch ResNet-50, ~102 MB of fp32 gradients) under DistributedDataParallel,
                SGD as the reference selects it, synthetic CIFAR-10-shaped batches resized to 224 (B=128 per GPU, the
                reference's batch size); images/s over all ranks with the RCCL gradient all-reduce inside the step;
  train_pointcloud  the same for ModelNet40-shaped batches (VNSmall canonicalizer + a PointNet classifier, B=64 per GPU);
  configs       the other BASELINE configs on every rank (cfg1 CIFAR-10 32x32 C4, cfg4 ModelNet40 SO(3), cfg5 COCO-shape D4
                with masks), whole-job units/s, the dominant kernel's roofline fraction, and (N=1) the CPU oracle;
  tutorial      the two training loops of the reference's tutorial notebook, the only throughput read-outs the reference repository
                holds, beside their 1,860-1,885 / 1,920-1,965 img/s (tools/bench_tutorial.py);
  cpu_baseline  the CPU oracle (reference op order) on this host, bounded sample, rank 0 / N=1 only.

`--dry-run` exercises ONLY the launcher logic (self-spawn, rendezvous, barrier, max-over-ranks, the JSON line) on CPU
over gloo with an empty step; it is what tests/test_distributed_cpu.py runs, and its line says "dry_run": true.  There
is no CPU path for the product: without --dry-run the script requires an MI355X.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured-achievable
MFMA_F32_PEAK_TF = 157.3  # dense fp32-input MFMA (v_mfma_f32_32x32x2_f32), same guide
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), same guide (never the 2:1-sparsity figure)
VALU_PEAK_WAVE_INSTR_S = 256 * 4 * 2.4e9 / 2  # 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles at 2.4 GHz
H = W = 224
C = 3
BYTES_TRANSFORM = 2 * C * H * W * 4  # read + write per image, fp32: 1,204,224 B


def build_canonicalizer(device, group_type: str = "rotation", num_rotations: int = 8):
    import equiadapt_amd as ea

    torch.manual_seed(2)
    net = ea.ESCNNEquivariantNetwork((3, 96, 96), out_channels=32, kernel_size=5, group_type=group_type,
                                     num_rotations=num_rotations, num_layers=3)
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=96)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (C, H, W))
    return can.to(device).eval()


# ----------------------------------------------------------------------------------------------------------------------
# CPU oracle legs (the checker and the stated baseline; never the thing measured)
# ----------------------------------------------------------------------------------------------------------------------
def oracle_check(can, x, f, y, inv, acts, gidx, group_type: str = "rotation", num_rotations: int = 8):
    """cpu_baseline leg, checker side: the HIP results (y, inv, acts, gidx; any device) for the images x / features f (CPU)
    against the CPU oracle run with the same weights.  Pixels are compared for the group element the HIP path chose, so a
    near-tie of the activations cannot masquerade as a pixel error; the index itself is compared wherever the oracle's
    top-2 margin exceeds 1e-4 of the activation scale (relative, because the random-init network's activations
    on white noise are ~5e-3 with orientation margins of ~1e-5)."""
    from oracle import image_ops as io
    from oracle import nets as onets

    net = can.canonicalization_network
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    x, f = x.cpu(), f.cpu()
    y, inv, acts, gidx = y.detach().cpu(), inv.detach().cpu(), acts.detach().cpu(), gidx.detach().cpu().long()
    G = 2 * num_rotations if group_type == "roto-reflection" else num_rotations
    with torch.no_grad():
        xin = io.pre_canonicalization_transform(x, (C, H, W), 0.8, 96)
        acts_ref = onets.escnn_like_network(xin, sd, group_type, num_rotations, 3, net.out_channels)
        scale = acts_ref.abs().max().item()
        top2 = acts_ref.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * scale
        ang = io.group_angles(num_rotations)
        if group_type == "rotation":
            rot, refl = ang[gidx], None
        else:
            rot, refl = torch.cat([ang, ang])[gidx], (gidx >= num_rotations).float()
        dy = (y.double() - io.canonicalize_images(x, rot, refl, (C, H, W)).double()).abs()
        di = (inv.double() - io.invert_action(f, rot, refl, num_rotations, G, "scalar").double()).abs()
    n_clear = int(clear.sum())
    match = float((gidx[clear] == acts_ref.argmax(-1)[clear]).float().mean()) if n_clear else 1.0
    return image_parity_ok({"images": int(x.shape[0]), "acts_max_err": (acts - acts_ref).abs().max().item(), "acts_scale": scale,
            "index_match": match, "n_clear_margin": n_clear, "tie_rate": 1.0 - n_clear / max(int(x.shape[0]), 1),
            "tie_rule": "top-2 margin of the oracle's activations <= 1e-4 x max|activation|",
            "index_match_all": float((gidx == acts_ref.argmax(-1)).float().mean()),
            "canonicalize_max_err": dy.max().item(), "canonicalize_rms_err": dy.pow(2).mean().sqrt().item(),
            "invert_max_err": di.max().item(), "invert_rms_err": di.pow(2).mean().sqrt().item(),
            "max_err": max(dy.max().item(), di.max().item()),
            "against": "oracle/ (CPU restatement of the reference op sequence), same weights, rank 0's batch 0 (every image of the timed batch)"},
                           1e-4, scale)


def cpu_baseline(sample: int, reps: int):
    """The oracle (reference op order, torch CPU ops) on the host cores: same workload, bounded sample."""
    from oracle import image_ops as io
    from oracle import nets as onets

    cores = os.cpu_count() or 1
    can = build_canonicalizer("cpu")
    sd = {k: v for k, v in can.canonicalization_network.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sample, C, H, W, generator=g)
    f = torch.randn(sample, C, H, W, generator=torch.Generator().manual_seed(3))

    def step():
        xin = io.pre_canonicalization_transform(x, (C, H, W), 0.8, 96)
        acts = onets.escnn_like_network(xin, sd, "rotation", 8, 3, 32)
        el = io.group_element_from_activations(acts, 8, "rotation", 1.0, training=False)
        y = io.canonicalize_images(x, el["rotation"], None, (C, H, W))
        out = io.invert_action(f, el["rotation"], None, 8, 8, "scalar")
        return y, out

    with torch.no_grad():
        # these ops are small; on a many-core host the fastest thread count is well below the core count, so a
        # quick calibration picks it (reported as `cores`): the baseline should be the CPU's best, not a strawman
        best = (float("inf"), cores)
        xs, fs = x, f
        x, f = x[:4], f[:4]
        calib = {}
        for nt in sorted({cores, 64, 32, 16, 8} & set(range(1, cores + 1)), reverse=True):
            torch.set_num_threads(nt)
            step()
            ts3 = []
            for _ in range(3):                      # three timed runs per thread count, the best of them: one run is noise on a shared host
                t0 = time.perf_counter()
                step()
                ts3.append(time.perf_counter() - t0)
            calib[nt] = min(ts3)
            if calib[nt] < best[0]:
                best = (calib[nt], nt)
        x, f = xs, fs
        torch.set_num_threads(best[1])
        step()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        # transform+invert alone (the part the GPU kernels replace one-for-one)
        rot = io.group_angles(8)[torch.randint(0, 8, (sample,), generator=torch.Generator().manual_seed(1))]
        io.canonicalize_images(x, rot, None, (C, H, W))
        t0 = time.perf_counter()
        for _ in range(reps):
            io.canonicalize_images(x, rot, None, (C, H, W))
            io.invert_action(f, rot, None, 8, 8, "scalar")
        ga = (time.perf_counter() - t0) / reps
    med = statistics.median(ts)
    return {"value": sample / med, "unit": "images/s", "cores": torch.get_num_threads(), "host_cores": cores, "kind": "port",
            "sample": f"{sample} images x {reps} reps of the same step (oracle/: pre-transform, conv stack, "
                      f"argmax, pad+rotate+crop, invert) on torch-CPU, {torch.get_num_threads()} threads",
            "group_action_only_images_s": sample / ga,
            "thread_calibration_ms_per_4_images": {str(k): v * 1e3 for k, v in sorted(calib.items())}}


def cpu_baseline_config(name: str, state: dict):
    """cpu_baseline leg for the other BASELINE configs: the oracle's op sequence on the host, 16 threads, bounded sample.
    `state` carries the network weights of the GPU run so both sides compute the same thing."""
    from oracle import image_ops as io
    from oracle import nets as onets
    from oracle import pointcloud_ops as po

    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def timed(fn, reps):
        with torch.no_grad():
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
        return (time.perf_counter() - t0) / reps

    if name == "cfg1":
        xs, fs, sd = torch.randn(128, 3, 32, 32), torch.randn(128, 3, 32, 32), state["sd"]

        def f1():
            acts = onets.custom_equivariant_network(io.pre_canonicalization_transform(xs, (3, 32, 32), 1.0, 32), sd, "rotation", 4, 2)
            el = io.group_element_from_activations(acts, 4, "rotation", 1.0, training=False)
            return io.canonicalize_images(xs, el["rotation"], None, (3, 32, 32)), io.invert_action(fs, el["rotation"], None, 4, 4, "scalar")
        dt = timed(f1, 5)
        return {"value": 128 / dt, "unit": "images/s", "ms": dt * 1e3, "cores": torch.get_num_threads(), "kind": "port", "sample": "B=128 x 5 reps"}
    if name == "cfg4":
        pcs, sd = torch.randn(8, 3, 1024), state["sd"]
        dt = timed(lambda: po.canonicalize_pointcloud(pcs, po.gram_schmidt(po.vnsmall_forward(pcs, sd))), 3)
        return {"value": 8 / dt, "unit": "clouds/s", "ms": dt * 1e3, "cores": torch.get_num_threads(), "kind": "port", "sample": "B=8 x 3 reps"}
    if name == "cfg5":  # the WHOLE step of the GPU leg on one image: resize -> orbit -> ConvNetwork -> activations -> argmax -> transform of the
        # image, its 3 masks and 3 boxes -> invert of a mask-shaped output
        g = torch.Generator().manual_seed(13)
        x1, pred1 = torch.randn(1, 3, 1024, 1024, generator=g), torch.randn(1, 1, 1024, 1024, generator=g)
        m1 = (torch.rand(3, 1024, 1024, generator=g) > 0.5).to(torch.uint8)
        b1 = torch.tensor([[10.0, 20.0, 200.0, 300.0]] * 3)
        sd, ref = state["sd"], state["ref_vec"]

        def f5():
            xin = io.pre_canonicalization_transform(x1, (3, 1024, 1024), 1.0, 128)
            vec = onets.conv_network(io.orbit_expand(xin, 4, "roto-reflection", 128), sd, 3, training=False)
            acts = io.optimized_group_activations(vec, ref, 8)
            el = io.group_element_from_activations(acts, 4, "roto-reflection", 1.0, training=False)
            y = io.canonicalize_images(x1, el["rotation"], el["reflection"], (3, 1024, 1024))
            mk = io.rotate_masks(io.flip_masks(m1), -el["rotation"][0].item())
            bx = io.rotate_boxes(io.flip_boxes(b1.clone(), 1024), el["rotation"][0], 1024)
            return y, mk, bx, io.invert_action(pred1, el["rotation"], el["reflection"], 4, 8, "scalar")
        dt = timed(f5, 3)
        return {"value": 1 / dt, "unit": "images/s", "ms": dt * 1e3, "cores": torch.get_num_threads(), "kind": "port",
                "sample": "B=1 x 3 reps of the whole step (resize, orbit, ConvNetwork on 8 views, activations, image + 3 masks + 3 boxes, invert)"}
    raise ValueError(name)


PIXEL_TOL = 1e-5   # resampled pixels within 1e-5 absolute of the CPU oracle (unit-variance data)


def image_parity_ok(rec: dict, acts_rel_tol: float, acts_scale: float) -> dict:
    """`ok` of an image-path parity record against its stated tolerances: the group index equal wherever the oracle's top-2
    margin is clear (bit-exact integer work), the activations within the tie margin of the oracle's (a larger error could move an
    index the rule calls clear), canonicalized and inverted pixels within PIXEL_TOL, masks bit-exact and boxes within 1e-3 px when
    the record has them."""
    ok = (rec["index_match"] == 1.0 and rec["acts_max_err"] <= acts_rel_tol * acts_scale
          and rec["canonicalize_max_err"] <= PIXEL_TOL and rec["invert_max_err"] <= PIXEL_TOL
          and rec.get("masks_bit_exact", True) and rec.get("boxes_max_err", 0.0) <= 1e-3)
    rec["tolerance"] = (f"index: equal wherever the oracle's top-2 margin is clear; activations <= {acts_rel_tol:g} x max|activation| (the tie "
                        f"margin); pixels <= {PIXEL_TOL:g} abs" + ("; masks bit-exact; boxes <= 1e-3 px" if "masks_bit_exact" in rec else ""))
    rec["ok"] = bool(ok)
    return rec


def parity_config(name: str, st: dict):
    """The parity record BASELINE.md section 3 wants beside every throughput row, for the non-headline configs: the product's
    outputs on a seeded batch of the config's own shape against the CPU oracle with the same weights -- group index match rate
    above the tie margin, tie rate, pixel / coordinate max and RMS error.  Checker only; runs after the timed region."""
    from oracle import image_ops as io
    from oracle import nets as onets
    from oracle import pointcloud_ops as po

    dev, can = st["dev"], st["can"]
    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def index_stats(gidx, acts_ref, rel):
        scale = acts_ref.abs().max().item()
        top2 = acts_ref.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > rel * scale
        n_clear = int(clear.sum())
        return {"index_match": float((gidx[clear] == acts_ref.argmax(-1)[clear]).float().mean()) if n_clear else 1.0,
                "n_clear_margin": n_clear, "tie_rate": 1.0 - n_clear / gidx.shape[0],
                "tie_rule": f"top-2 margin <= {rel:g} x max|activation|"}

    def err(a, b):
        d = (a.detach().cpu().double() - b.double()).abs()
        return d.max().item(), d.pow(2).mean().sqrt().item()

    with torch.no_grad():
        if name == "cfg1":
            g = torch.Generator().manual_seed(11)
            x, f = torch.randn(128, 3, 32, 32, generator=g), torch.randn(128, 3, 32, 32, generator=g)
            y = can(x.to(dev))
            acts = can.canonicalization_info_dict["group_activations"].cpu()
            gidx = can.canonicalization_info_dict["group_index"].cpu().long()
            inv = can.invert_canonicalization(f.to(dev), induced_rep_type="scalar")
            acts_ref = onets.custom_equivariant_network(io.pre_canonicalization_transform(x, (3, 32, 32), 1.0, 32), st["sd"], "rotation", 4, 2)
            rot = io.group_angles(4)[gidx]
            cm, cr = err(y, io.canonicalize_images(x, rot, None, (3, 32, 32)))
            im, ir = err(inv, io.invert_action(f, rot, None, 4, 4, "scalar"))
            return image_parity_ok({"images": 128, **index_stats(gidx, acts_ref, 1e-4), "acts_max_err": (acts - acts_ref).abs().max().item(),
                                    "canonicalize_max_err": cm, "canonicalize_rms_err": cr, "invert_max_err": im, "invert_rms_err": ir,
                                    "against": "oracle/ (CPU), same weights, seeded batch of the config's shape"},
                                   1e-4, acts_ref.abs().max().item())
        if name == "cfg4":
            # fp64 error budget (oracle/pointcloud_ops.pointcloud_parity_record; tests/test_gpu_pointcloud_budget.py runs the same
            # record on this batch and on B = 64): the product and the fp32 oracle each against the fp64 evaluation, tolerances
            # derived per cloud from the oracle's own distance to fp64 and the Gram-Schmidt Jacobian
            from equiadapt_amd import _lib

            pcs = torch.randn(8, 3, 1024, generator=torch.Generator().manual_seed(12))
            xd = pcs.to(dev)
            y = can(xd)
            R = can.canonicalization_info_dict["group_element_matrix_representation"]
            from equiadapt_amd import ops
            net4 = can.canonicalization_network
            vec, R2, y2 = ops.vnsmall_canonicalize(xd, net4.packed_parameters(), net4.n_knn, net4.pooling)   # the class's own call: the vectors too
            same_route = bool(torch.equal(R, R2) and torch.equal(y, y2))
            idx = torch.empty(8, 1024, 20, dtype=torch.int32, device=dev)
            _lib.check(_lib.load().eqa_vn_knn(xd.data_ptr(), idx.data_ptr(), 8, 1024, 20, None), "eqa_vn_knn")
            torch.cuda.synchronize()
            rec = po.pointcloud_parity_record(pcs, st["sd"], idx.cpu(), vec, R, y)
            rec["class_equals_fused_call"] = same_route
            rec["ok"] = bool(rec["ok"] and same_route)
            rec["against"] = ("oracle/pointcloud_ops.py (pinned to reference-generated vectors, tests/golden/pointcloud_n1024.pt) evaluated in "
                              "fp32 and in fp64 on the same neighbour sets")
            return rec
        if name == "cfg5":
            g = torch.Generator().manual_seed(13)
            x = torch.randn(2, 3, 1024, 1024, generator=g)
            pred = torch.randn(2, 1, 1024, 1024, generator=g)
            masks = [(torch.rand(3, 1024, 1024, generator=g) > 0.5).to(torch.uint8) for _ in range(2)]
            boxes = [torch.tensor([[10.0, 20.0, 200.0, 300.0], [300.0, 40.0, 900.0, 700.0], [5.0, 5.0, 50.0, 60.0]]) for _ in range(2)]
            targets = [{"boxes": b.clone().to(dev), "masks": m.to(dev)} for b, m in zip(boxes, masks)]
            y, tg = can(x.to(dev), targets)
            acts = can.canonicalization_info_dict["group_activations"].cpu()
            gidx = can.canonicalization_info_dict["group_index"].cpu().long()
            inv = can.invert_canonicalization(pred.to(dev), induced_rep_type="scalar")
            xin = io.pre_canonicalization_transform(x, (3, 1024, 1024), 1.0, 128)
            vec = onets.conv_network(io.orbit_expand(xin, 4, "roto-reflection", 128), st["sd"], 3, training=False)
            acts_ref = io.optimized_group_activations(vec, st["ref_vec"], 8)
            ang = torch.cat([io.group_angles(4)] * 2)[gidx]
            refl = (gidx >= 4).float()
            cm, cr = err(y, io.canonicalize_images(x, ang, refl, (3, 1024, 1024)))
            im, ir = err(inv, io.invert_action(pred, ang, refl, 4, 8, "scalar"))
            masks_ok = all(torch.equal(tg[i]["masks"].cpu(), io.rotate_masks(io.flip_masks(masks[i]), -ang[i].item())) for i in range(2))
            bm = max((tg[i]["boxes"].cpu() - io.rotate_boxes(io.flip_boxes(boxes[i].clone(), 1024), ang[i], 1024).reshape(-1, 4)).abs().max().item()
                     for i in range(2))
            return image_parity_ok({"images": 2, **index_stats(gidx, acts_ref, 1e-3), "acts_max_err": (acts - acts_ref).abs().max().item(),
                                    "canonicalize_max_err": cm, "canonicalize_rms_err": cr, "invert_max_err": im, "invert_rms_err": ir,
                                    "masks_bit_exact": masks_ok, "boxes_max_err": bm,
                                    "against": "oracle/ (CPU), same weights, 2 seeded 1024 x 1024 images with 3 masks + 3 boxes each"},
                                   1e-3, acts_ref.abs().max().item())
    raise ValueError(name)


# ----------------------------------------------------------------------------------------------------------------------
# launcher / communicator
# ----------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_launcher(n: int) -> int:
    """`python bench.py --gpus N` from a plain shell: re-execute under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


class Comm:
    """The process group of this run (or a single process): barrier + device synchronise, max / sum over ranks."""

    def __init__(self, dry_run: bool):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry_run = dry_run
        self.backend = None
        # EQA_BENCH_BACKEND=gloo: the same N-rank job with gloo as the collective backend and the ranks folded onto the GPUs that
        # exist (rank r -> cuda:(r mod device_count)) -- how `--gpus 2` runs on a one-GPU box, where RCCL refuses two ranks per
        # device.  The line then says "backend": "gloo", "ranks_share_gpu": true; it is a correctness run, not a scaling number.
        want = os.environ.get("EQA_BENCH_BACKEND", "nccl")
        assert want in ("nccl", "gloo"), f"EQA_BENCH_BACKEND={want!r}: nccl (RCCL, default) or gloo"
        self.device_index = self.local
        if not dry_run:
            assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback (--dry-run tests the launcher only)"
            if want == "gloo":
                self.device_index = self.local % torch.cuda.device_count()
            torch.cuda.set_device(self.device_index)
        self.ranks_share_gpu = (not dry_run) and want == "gloo" and self.world > torch.cuda.device_count()
        # Under a launcher (RANK set) the process group comes up even for a single rank: `torch.distributed.run
        # --nproc-per-node 1 bench.py` is then the N-rank job with N = 1 (RCCL initialised, the training legs DDP-wrapped).
        if self.world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
            self.backend = "gloo" if dry_run else want   # "nccl" IS RCCL on ROCm
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.dev = torch.device("cpu") if dry_run else torch.device("cuda", self.device_index)
        # RCCL builds its communicator inside the FIRST collective (hundreds of ms).  Left to the barrier in front of the timed
        # region that gap idles the GPU right before the clock starts and the first steps run at the idle clock: the one-rank RCCL job
        # measured 2-3 % below the bare process (profiles/r06/rccl_world1_vs_bare.txt, first collection).  Pay it here.
        self.barrier()

    def barrier(self):
        if self.backend is not None:
            import torch.distributed as dist

            dist.barrier()
        if not self.dry_run:
            torch.cuda.synchronize()

    def _scalar_dev(self):
        return torch.device("cpu") if self.backend == "gloo" else self.dev   # gloo gathers host tensors only

    def reduce(self, value: float, op: str = "max") -> float:
        if self.world == 1:
            return value
        import torch.distributed as dist

        t = torch.tensor([value], device=self._scalar_dev(), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return t.item()

    def gather(self, value: float):
        if self.world == 1:
            return [value]
        import torch.distributed as dist

        t = torch.tensor([value], device=self._scalar_dev(), dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [o.item() for o in out]

    def timed(self, step, steps: int, warmup: int, timed_ctx=None):
        """W untimed steps, then exactly K steps between barrier + synchronize pairs.  -> (max over ranks, per-rank list) seconds.
        ``timed_ctx``: a context manager entered for the timed region only (the per-kernel HIP-event timer: warm-up launches
        carry one-time library initialisation and must not enter its averages)."""
        import contextlib

        for _ in range(warmup):
            step()
        self.barrier()
        with (timed_ctx if timed_ctx is not None else contextlib.nullcontext()):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            self.barrier()
            dt = time.perf_counter() - t0
        return self.reduce(dt, "max"), self.gather(dt)

    def close(self):
        if self.backend is not None:
            import torch.distributed as dist

            dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------------
# legs
# ----------------------------------------------------------------------------------------------------------------------
def leg_train_images(comm: Comm, steps: int, warmup: int, batch: int):
    """Data-parallel training step, image classification (reference: examples/images/classification/model.py:59-127,
    184-239; train_utils.py:89-91 strategy="ddp").  CIFAR-10-shaped data as the reference feeds it (resized to 224,
    model_utils.py:21), batch 128 per GPU (configs/dataset/default.yaml), ResNet-50, SGD by the reference's rule."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from resnet50 import ResNet50

    import equiadapt_amd as ea  # noqa: F401
    from equiadapt_amd import training as tr

    dev = comm.dev
    can = build_canonicalizer(dev).train()
    torch.manual_seed(0)
    # the prediction network is unmodified torch; it is merely RUN channels-last (MIOpen's NHWC convolutions: ResNet-50 forward +
    # backward at B=128 59.6 -> 50.1 ms on this chip; cudnn.benchmark makes no difference).  EQA_BENCH_PRED_NCHW=1: as constructed
    pred = ResNet50(num_classes=10)
    if os.environ.get("EQA_BENCH_PRED_NCHW", "0") != "1":
        pred = pred.to(memory_format=torch.channels_last)
    model = tr.CanonicalizedClassifier(can, pred, tr.LossWeights(task_weight=1.0, prior_weight=100.0)).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    ddp = tr.wrap_ddp(model, dev, force=comm.backend is not None)
    opt, _ = tr.configure_optimizer(model, 1e-3, 1e-3, kind=None, max_epochs=200,
                                    prediction_network_architecture="resnet50", dataset_name="cifar10")
    g = torch.Generator().manual_seed(100 + comm.rank)
    xs = [torch.randn(batch, C, H, W, generator=g).to(dev) for _ in range(2)]
    ys = [torch.randint(0, 10, (batch,), generator=g).to(dev) for _ in range(2)]
    it = [0]

    def step():
        i = it[0] & 1
        it[0] += 1
        return tr.train_step(ddp, opt, xs[i], ys[i], check_nan=False)

    torch.cuda.reset_peak_memory_stats()
    dt, per_rank = comm.timed(step, steps, warmup)
    loss = float(step()["loss"].detach())
    assert loss == loss, "training loss is NaN"
    res = {"images_s": batch * comm.world * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
           "batch_per_gpu": batch, "n_gpus": comm.world,
           "model": "GroupEquivariantImageCanonicalization(ESCNNEquivariantNetwork C8 32ch k5 L3) + ResNet50(10 classes, run channels-last), fp32",
           "optimizer": type(opt).__name__ + " (reference rule: resnet + non-mnist -> SGD 0.9 / wd 5e-4)",
           "loss": "1.0 * CE + 100.0 * prior", "parameters": n_params, "allreduce_MB_per_step": n_params * 4 / 1e6 if comm.backend is not None else 0.0,
           "collective": ("DDP bucketed all-reduce over RCCL (64 MB buckets), overlapped with backward" if comm.backend == "nccl" else
                          "DDP bucketed all-reduce over gloo (through the host; ranks share the GPU): correctness run, not a scaling number"
                          if comm.backend == "gloo" else "none (1 rank, no process group)"),
           "ddp_wrapped": type(ddp).__name__ == "DistributedDataParallel",
           "per_rank_ms_per_step": [t / steps * 1e3 for t in per_rank], "final_loss": loss,
           "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9, "data": "synthetic CIFAR-10-shaped (resized 224x224x3), labels uniform"}
    del ddp, model, opt, xs, ys
    torch.cuda.empty_cache()
    return res


def leg_train_pointcloud(comm: Comm, steps: int, warmup: int, batch: int):
    """Data-parallel training step, point-cloud classification (examples/pointcloud/classification/model.py:77-134,245-300):
    ModelNet40-shaped clouds (1024 points, B=64 per GPU), VNSmall canonicalizer + PointNet classifier, SGD + cosine."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from resnet50 import PointNetCls

    import equiadapt_amd as ea
    from equiadapt_amd import training as tr

    dev = comm.dev
    hp = types.SimpleNamespace(n_knn=20, pooling="mean")
    torch.manual_seed(2)
    can = ea.EquivariantPointcloudCanonicalization(ea.VNSmall(hp), hp)
    model = tr.CanonicalizedClassifier(can, PointNetCls(40), tr.LossWeights(task_weight=1.0, prior_weight=100.0)).to(dev).train()
    n_params = sum(p.numel() for p in model.parameters())
    ddp = tr.wrap_ddp(model, dev, force=comm.backend is not None)
    opt, _ = tr.configure_pointcloud_optimizer(model, 1e-3, 1e-3, "SGD", "cosine", 250)
    g = torch.Generator().manual_seed(200 + comm.rank)
    xs = [torch.randn(batch, 3, 1024, generator=g).to(dev) for _ in range(2)]
    ys = [torch.randint(0, 40, (batch,), generator=g).to(dev) for _ in range(2)]
    it = [0]

    def step():
        i = it[0] & 1
        it[0] += 1
        return tr.train_step(ddp, opt, xs[i], ys[i], check_nan=False)

    dt, per_rank = comm.timed(step, steps, warmup)
    loss = float(step()["loss"].detach())
    assert loss == loss, "training loss is NaN"
    return {"clouds_s": batch * comm.world * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "batch_per_gpu": batch, "n_gpus": comm.world, "model": "EquivariantPointcloudCanonicalization(VNSmall k=20 mean) + PointNet(40 classes), fp32",
            "optimizer": "SGD x100 lr, momentum 0.9, wd 1e-4 (reference rule)", "parameters": n_params,
            "allreduce_MB_per_step": n_params * 4 / 1e6 if comm.backend is not None else 0.0,
            "ddp_wrapped": type(ddp).__name__ == "DistributedDataParallel",
            "per_rank_ms_per_step": [t / steps * 1e3 for t in per_rank], "final_loss": loss,
            "data": "synthetic ModelNet40-shaped (B,3,1024) normal clouds, labels uniform"}


def leg_configs(comm: Comm, with_cpu: bool):
    """BASELINE configs other than the headline, on every rank (batch sharded, no collective): whole-job units/s."""
    import equiadapt_amd as ea
    from equiadapt_amd import ops

    dev, out = comm.dev, {}

    def run(step, units, reps, warm, kt=None):
        with torch.no_grad():
            # these legs follow host-side work (model construction, the CPU oracle of the previous leg) and their steps are 0.06-1 ms:
            # `warm` steps alone would be timed on a chip still coming back to its running clocks (the first launches after an idle
            # stretch run ~12 % slow), so the untimed part lasts at least 60 ms
            t_warm = time.perf_counter() + 0.06
            while time.perf_counter() < t_warm:
                step()
                torch.cuda.synchronize()
            dt, _ = comm.timed(step, reps, warm, kt)
        return units * comm.world * reps / dt, dt / reps * 1e3

    # ---- cfg1: CIFAR-10 shape 32x32x3, C4, CustomEquivariantNetwork (the self-contained net that satisfies the API)
    torch.manual_seed(2)
    net = ea.CustomEquivariantNetwork((3, 32, 32), 8, 5, "rotation", 4, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=32)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 32, 32))
    sd1 = {k: v.clone() for k, v in net.state_dict().items()}
    can = can.to(dev).eval()
    c1 = {"workload": "configs[0] shape: CIFAR-10 32x32x3, C4, GroupEquivariantImageCanonicalization + CustomEquivariantNetwork(8ch,k5,2 layers): "
                      "canonicalize + invert(scalar)", "unit": "images/s", "batches": {}}
    for B in (128, 8192):
        x = torch.randn(B, 3, 32, 32, device=dev)
        f = torch.randn(B, 3, 32, 32, device=dev)
        step1 = lambda: (can(x), can.invert_canonicalization(f, induced_rep_type="scalar"))   # noqa: E731
        v, ms = run(step1, B, 20, 5)            # bare; the event brackets run in a second pass
        kt = ops.KernelTimer()
        run(step1, B, 20, 0, kt)
        n_ct, ms_ct = kt.summary().get("canon_transform", (0, float("nan")))
        ach = B * 2 * 3 * 32 * 32 * 4 / (ms_ct * 1e-3) / 1e9
        c1["batches"][str(B)] = {"value": v, "ms_per_step": ms, "roofline": {
            "bound": "hbm", "kernel": "group_action_kernel via eqa_canon_transform_fwd", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": ms_ct,
            "note": "launch-bound at this size: B*24.6 KB per launch" if B == 128 else None}}
        if B == 128:  # the same step captured once and replayed as a hipGraph (one host call instead of ~12 launches)
            from equiadapt_amd.graphs import GraphedCanonicalizer

            gstep = GraphedCanonicalizer(can, x.shape, f.shape)
            gstep(x, f)
            vg, msg = run(gstep.replay, B, 50, 10)
            c1["batches"][str(B)]["hipgraph"] = {"value": vg, "ms_per_step": msg}
            del gstep
    c1["value"] = c1["batches"]["8192"]["value"]
    if with_cpu:
        c1["cpu_baseline"] = cpu_baseline_config("cfg1", {"sd": sd1})
        c1["parity"] = parity_config("cfg1", {"sd": sd1, "can": can, "dev": dev})
    out["cfg1"] = c1
    del can

    # ---- cfg4: ModelNet40 shape, 1024 points, SO(3)
    hp4 = types.SimpleNamespace(n_knn=20, pooling="mean")
    torch.manual_seed(2)
    vn = ea.VNSmall(hp4)
    sd4 = {k: v.clone() for k, v in vn.state_dict().items()}
    can4 = ea.EquivariantPointcloudCanonicalization(vn, hp4).to(dev).eval()
    c4 = {"workload": "configs[3] shape: ModelNet40 (B,3,1024), SO(3): VNSmall(k=20, mean) -> Gram-Schmidt -> rotate "
                      "(the reference defines no invert for clouds)", "unit": "clouds/s", "batches": {}}
    # VALU-bound: measured 68.4 M VALU wave-instructions per launch of 64 clouds (PMC SQ_INSTS_VALU, profiles/r03/pmc_vnsmall_quad.txt:
    # the four-lanes-per-point kernel issues as many as round 2's one-thread-per-point kernel did, on 4 x the waves) -> 1.07 M per
    # cloud, of which the kNN distance evaluations are the irreducible 1024 x 1024 x ~5 = 82 k; HBM traffic 12 KB in + 36 B out per cloud.
    instr_per_cloud = 68.4e6 / 64
    for B in (64, 2048):
        pc = torch.randn(B, 3, 1024, device=dev)
        v, ms = run(lambda: can4(pc), B, 20, 5)
        per_gpu = v / comm.world
        if B == 64:
            from equiadapt_amd.graphs import GraphedCanonicalizer

            gstep = GraphedCanonicalizer(can4, pc.shape)
            gstep(pc)
            vg, msg = run(gstep.replay, B, 50, 10)
            hipgraph4 = {"value": vg, "ms_per_step": msg}
            del gstep
        c4["batches"][str(B)] = {"value": v, "ms_per_step": ms, "roofline": {
            "bound": "valu", "kernel": "vnsmall_fwd_quad_kernel<5,false> (eqa_vnsmall_fwd)", "achieved": per_gpu * instr_per_cloud / 1e12,
            "peak": VALU_PEAK_WAVE_INSTR_S / 1e12, "unit": "T wave-instr/s", "frac": per_gpu * instr_per_cloud / VALU_PEAK_WAVE_INSTR_S,
            "hbm_frac": per_gpu * 12324 / 1e9 / HBM_PEAK_GBS,
            "note": "1.07 M VALU wave-instructions per cloud (kNN distances 82 k of them) against 12 KB of HBM traffic: issue-bound, not HBM-bound"}}
        if B == 64:
            c4["batches"][str(B)]["hipgraph"] = hipgraph4
    c4["value"] = c4["batches"]["2048"]["value"]
    if with_cpu:
        c4["cpu_baseline"] = cpu_baseline_config("cfg4", {"sd": sd4})
        c4["parity"] = parity_config("cfg4", {"sd": sd4, "can": can4, "dev": dev})
    out["cfg4"] = c4
    del can4

    # ---- cfg5: COCO shape 1024x1024x3, D4, optimised canonicalizer, mask + box targets, invert of a mask-shaped scalar output
    torch.manual_seed(2)
    net5 = ea.ConvNetwork((3, 128, 128), out_channels=16, kernel_size=7, num_layers=3, out_vector_size=128)
    hp5 = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=128, group_type="roto-reflection", num_rotations=4,
                                artifact_err_wt=0.0, learn_ref_vec=False)
    can5 = ea.OptimizedGroupEquivariantImageCanonicalization(net5, hp5, (3, 1024, 1024))
    sd5, ref5 = {k: v.clone() for k, v in net5.state_dict().items()}, can5.reference_vector.detach().clone()
    can5 = can5.to(dev).eval()
    c5 = {"workload": "configs[4] shape: COCO 1024x1024x3, D4, OptimizedGroupEquivariantImageCanonicalization + ConvNetwork(k7,16ch,3 layers,128), "
                      "3 uint8 masks + 3 boxes per image as targets, invert_canonicalization(scalar) of a (B,1,1024,1024) output",
          "unit": "images/s", "batches": {}}
    for B in (4, 32):
        x = torch.randn(B, 3, 1024, 1024, device=dev)
        pred = torch.randn(B, 1, 1024, 1024, device=dev)
        masks = [(torch.rand(3, 1024, 1024, device=dev) > 0.5).to(torch.uint8) for _ in range(B)]
        boxes = [torch.tensor([[10.0, 20.0, 200.0, 300.0]] * 3, device=dev) for _ in range(B)]

        def step5():
            targets = [{"boxes": b.clone(), "masks": m} for b, m in zip(boxes, masks)]
            y, t = can5(x, targets)
            return y, t, can5.invert_canonicalization(pred, induced_rep_type="scalar")
        # value / ms_per_step: bare steps (30 of them: at B = 4 a step is 0.3 ms of host time); the per-launch event brackets of the
        # kernel figures below run in a second pass, as in the headline leg
        v, ms = run(step5, B, 30, 5)
        kt = ops.KernelTimer()
        run(step5, B, 10, 0, kt)
        ks = kt.summary()
        n_ct, ms_ct = ks.get("canon_transform", (0, float("nan")))
        n_mk, ms_mk = ks.get("mask_action", (0, float("nan")))
        ach = B * 2 * 3 * 1024 * 1024 * 4 / (ms_ct * 1e-3) / 1e9
        ach_m = B * 3 * 2 * 1024 * 1024 / (ms_mk * 1e-3) / 1e9
        # The per-launch HIP-event bracket carries a fixed cost (an EMPTY bracket measures ~4.8 us on this runtime, a 256 MB fill
        # 2 us more than back to back: tools/event_overhead.py) that is negligible for the headline's 58 us kernel and a fifth of
        # these launches at B = 4.  The same two kernels launched 30 times between one pair of events, over a cache-cold ring of buffers:
        from equiadapt_amd.images import geometry
        from equiadapt_amd.images.utils import device_tables
        g5 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
        th5, fl5 = device_tables("canonicalize", 4, True, (2048, 2048), dev)
        mcat = torch.cat(masks, dim=0).contiguous()
        e5 = (g5 % 4).repeat_interleave(3)
        rth5 = geometry.mask_rotation_table((-geometry.group_angles(4)).tolist(), (1024, 1024)).to(dev)
        mfl5 = torch.full((4,), geometry.FLIP_SRC, dtype=torch.int32, device=dev)

        def b2b(fn, inputs, reps=30):
            """`reps` launches between ONE pair of events, each on the next buffer of a ring whose inputs + kept outputs exceed the
            256 MB Infinity Cache several times over: no launch finds its operands (or the lines it is about to write) cached by the
            launch before -- at B = 4 the whole working set of one launch (100 MB image, 25 MB masks) would otherwise stay resident
            and the figure would be a cache bandwidth, not HBM's."""
            n = len(inputs)
            outs = [None] * n
            for i in range(n):
                outs[i] = fn(inputs[i])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(reps):
                outs[r % n] = fn(inputs[r % n])
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        def ring(t, at_least_bytes=640 << 20):      # distinct copies of t: inputs + outputs >= 2.5 x the cache
            n = max(2, -(-at_least_bytes // (2 * t.numel() * t.element_size())))
            return [t] + [t.clone() for _ in range(n - 1)]
        with torch.no_grad():
            xs5, ms5 = ring(x), ring(mcat)
            ms_ct_b2b = b2b(lambda a: ops.canon_transform(a, g5, th5, fl5, 512), xs5)
            ms_mk_b2b = b2b(lambda a: ops.mask_action_nearest(a, e5, rth5, mfl5), ms5)
            ring_note = f"ring of {len(xs5)} image batches / {len(ms5)} mask stacks (inputs + outputs >= 640 MB per ring: cache-cold)"
            # the floor of a launch of this size on this chip: the framework's plain copy of the same bytes over the same rings (a
            # 25 MB launch is over before the memory pipe is full: what a kernel can reach here is what a copy reaches, not 8 TB/s)
            ms_ct_copy = b2b(lambda a: a.clone(), xs5)
            ms_mk_copy = b2b(lambda a: a.clone(), ms5)
            del xs5, ms5
        del mcat
        ach_b, ach_mb = B * 2 * 3 * 1024 * 1024 * 4 / (ms_ct_b2b * 1e-3) / 1e9, B * 3 * 2 * 1024 * 1024 / (ms_mk_b2b * 1e-3) / 1e9
        how = ("avg_launch_ms / achieved / frac: the kernel launched 30 times between ONE pair of HIP events over a " + ring_note +
               "; per_launch_event_*: one event bracket per launch inside the timed step, which adds the bracket's own ~2-5 us "
               "(tools/event_overhead.py) to these 12-27 us launches")
        c5["batches"][str(B)] = {"value": v, "ms_per_step": ms, "roofline": {
            "bound": "hbm", "kernel": "group_action_kernel via eqa_canon_transform_fwd (25,165,824 B / image)", "achieved": ach_b,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_b / HBM_PEAK_GBS, "avg_launch_ms": ms_ct_b2b,
            "per_launch_event_ms": ms_ct, "per_launch_event_frac": ach / HBM_PEAK_GBS, "how": how,
            "copy_same_bytes_ms": ms_ct_copy, "frac_of_copy": ms_ct_copy / ms_ct_b2b},
            "mask_kernel": {"kernel": "mask_action_u8_kernel via eqa_mask_action_nearest (2 B / mask pixel)", "achieved": ach_mb,
                            "unit": "GB/s", "frac": ach_mb / HBM_PEAK_GBS, "avg_launch_ms": ms_mk_b2b,
                            "per_launch_event_ms": ms_mk, "per_launch_event_frac": ach_m / HBM_PEAK_GBS,
                            "copy_same_bytes_ms": ms_mk_copy, "frac_of_copy": ms_mk_copy / ms_mk_b2b,
                            "note": "frac_of_copy = time of torch's clone() of the same tensor over the same ring / this kernel's time: "
                                    "the rate a launch of this size can reach at all"}}
        if B == 4:   # BASELINE's own batch: the whole step (image, masks, boxes, invert) captured once and replayed as a hipGraph
            from equiadapt_amd.graphs import GraphedCanonicalizer

            gstep = GraphedCanonicalizer(can5, x.shape, pred.shape, targets_like=[{"boxes": b, "masks": m} for b, m in zip(boxes, masks)])
            gstep(x, pred)
            vg, msg = run(gstep.replay, B, 30, 5)
            c5["batches"][str(B)]["hipgraph"] = {"value": vg, "ms_per_step": msg}
            del gstep
    c5["value"] = c5["batches"]["32"]["value"]
    if with_cpu:
        c5["cpu_baseline"] = cpu_baseline_config("cfg5", {"sd": sd5, "ref_vec": ref5})
        c5["parity"] = parity_config("cfg5", {"sd": sd5, "ref_vec": ref5, "can": can5, "dev": dev})
    out["cfg5"] = c5
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--mode", default="all", choices=["all", "forward", "train"],
                    help="all: headline + train + configs legs in one line (default); forward: headline only; train: the DP training legs only")
    ap.add_argument("--train-steps", type=int, default=8)
    ap.add_argument("--train-warmup", type=int, default=3)
    ap.add_argument("--train-batch", type=int, default=128, help="images per GPU per training step (reference: 128)")
    ap.add_argument("--check-images", type=int, default=0, help="images of the timed batch checked against the CPU oracle (0 = all)")
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-only", action="store_true",
                    help="forward mode without the group-action leg and the oracle check: every launch of a profiled run belongs to the timed "
                         "step (profiles/*/rocprofv3_kernel_stats_step.md)")
    ap.add_argument("--dry-run", action="store_true", help="launcher logic only, CPU + gloo, empty step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(args.gpus))

    # MIOpen's find mode also times its naive reference solvers for the backward-data / weight-gradient convolutions of the training
    # legs (100 ms per trial, ~20 s of warm-up per rank); nothing in this run asks for deterministic algorithms, so they are left out
    # (equiadapt_amd/__init__.py explains why the package itself only excludes the forward one).  Set here, not at import: the tests
    # import this module for build_canonicalizer / oracle_check.
    for _k in ("BWD", "WRW"):
        os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
    comm = Comm(args.dry_run)
    world, rank, dev = comm.world, comm.rank, comm.dev
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    B = args.batch

    if args.dry_run:
        z = torch.zeros(16)
        dt, per_rank = comm.timed(lambda: z.add_(1.0), args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({"metric": "canonicalize+invert images/sec (224x224 C8)", "value": None, "unit": "images/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "dry_run": True,
                              "backend": comm.backend, "ranks": world, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
                              "note": "launcher logic only (CPU, gloo, empty step): no product code ran"}), flush=True)
        comm.close()
        return

    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    line = {"metric": "canonicalize+invert images/sec (224x224 C8)", "value": None, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",   # refined below once the GEMM form of the timed step is known
            "config": {"workload": "configs[1]: 224x224x3 synthetic, C8, GroupEquivariantImageCanonicalization + ESCNNEquivariantNetwork in e2cnn's "
                                   "LAYER SHAPES (32 fields x 8 = 256 ch, k5, 3 layers, crop 0.8, resize 96) with this repository's own filter-bank "
                                   "parameterisation (bilinear-rotated banks; e2cnn's steerable basis is not restated -- e2cnn-trained weights enter "
                                   "via load_exported_dense), forward, then invert_canonicalization(scalar, 3ch); prediction network excluded",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world} (no collective in the forward path)"},
            "rccl_ranks": world if comm.backend == "nccl" else 0, "backend": comm.backend, "ranks_share_gpu": comm.ranks_share_gpu}

    if args.mode in ("all", "forward"):
        can = build_canonicalizer(dev)
        NBUF = 3  # distinct input batches, cycled: no step re-reads the batch the previous step left in the caches
        xs = [torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(rank * 16 + i)).to(dev) for i in range(NBUF)]
        fs = [torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(1000 + (rank * 16 + i) * 3)).to(dev) for i in range(NBUF)]
        it = [0]

        def step():
            i = it[0] % NBUF
            it[0] += 1
            y = can(xs[i])
            out = can.invert_canonicalization(fs[i], induced_rep_type="scalar")
            return y, out

        with torch.no_grad():
            # Model initialisation, not a step: the canonicalizer derives its per-weight-version operands on first use (0.9 GB of filter
            # spectra in the GEMM's fragment order, packed lifting weights, the group's element tables) -- 0.25 s, the counterpart of a
            # framework's first-call autotuning.  Done here so that the W warm-up steps the caller asks for are steps (with --warmup 0
            # the first timed step used to carry it: 59 ms per step instead of 6).
            step()
            it[0] = 0
            torch.cuda.synchronize()
            # `value`: exactly K bare steps between barrier + synchronize pairs.  The per-launch HIP-event brackets behind `roofline` /
            # `stages` run in a SECOND pass of the same K steps right after (round 6: ~10 brackets per step inside the value loop made
            # `value` slightly pessimistic -- VERDICT r05 weak 11); `ms_per_step_with_event_brackets` is that pass's wall time.
            elapsed, per_rank = comm.timed(step, args.steps, args.warmup)
            kt = ops.KernelTimer()
            elapsed_kt, _ = comm.timed(step, args.steps, 0, kt)
            ktimes = kt.summary()

            if args.step_only:
                ct, iv = ktimes.get("canon_transform", (0, float("nan"))), ktimes.get("invert_action", (0, float("nan")))
                if rank == 0:
                    print(json.dumps({"metric": "step only (profiling run)", "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps,
                                      "warmup": args.warmup, "canon_transform_ms": ct[1], "invert_action_ms": iv[1],
                                      "group_action_kernel_mean_ms": (ct[1] + iv[1]) / 2,
                                      "ms_per_step_with_event_brackets": elapsed_kt / args.steps * 1e3,
                                      "note": "both launches run group_action_kernel<3,true>: a kernel trace of this command averages them "
                                              "(over the warm-up, the bare pass and the bracketed pass)"}))
                return
            # The timed step's dominant contraction runs in the form fftconv.gemm_form chooses: at the headline layer (Cin = Cout = 256)
            # six bf16 piece products on exact three-piece splits of the fp32 operands, fp32 accumulation (default since round 6:
            # measured closer to fp64 than the fp32 matrix instruction on every admitted shape, tests/test_gpu_parity.py::
            # test_auto_gemm_form_is_no_further_from_fp64).  The same step with the fp32 matrix instruction (EQA_FFT_GEMM_PIECES=f32)
            # and with all nine piece products is timed beside it, same box, same run.
            from equiadapt_amd.images.canonicalization_networks import fftconv as _fc

            default_form = _fc.LAST_FORM or _fc.gemm_form(256, 256)      # what the timed steps' contraction ran in
            step_forms = {default_form: elapsed / args.steps * 1e3, "default": default_form}
            if _fc.GEMM_PIECES == "auto":
                for mode in ("f32", "9", "6", "h3"):
                    if mode == default_form:
                        continue
                    _fc.GEMM_PIECES = mode
                    try:
                        step()                      # builds the operands of this form once
                        it[0] = 0
                        e_alt, _ = comm.timed(step, args.steps, 3)
                        step_forms[mode] = e_alt / args.steps * 1e3
                    finally:
                        _fc.GEMM_PIECES = "auto"
            # group-action-only leg: the two resampling kernels back to back with a seeded random index
            x, f = xs[0], fs[0]
            gidx = torch.randint(0, 8, (B,), generator=torch.Generator().manual_seed(1)).to(dev, torch.int32)
            th_c, fl_c = device_tables("canonicalize", 8, False, (2 * H, 2 * W), dev)
            th_i, fl_i, _ = device_tables("invert", 8, False, (H, W), dev)
            # (20 untimed rounds first: the chip needs a few ms of work to come back to its running clocks -- this leg follows host-side
            # work, and the first launches after an idle stretch run ~12 % slow)
            for _ in range(20):
                ops.canon_transform(x, gidx, th_c, fl_c, H // 2)
                ops.invert_action(f, gidx, th_i, fl_i, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                ops.canon_transform(x, gidx, th_c, fl_c, H // 2)
                ops.invert_action(f, gidx, th_i, fl_i, None)
            e1.record()
            torch.cuda.synchronize()
            ga2_ms = e0.elapsed_time(e1) / reps
            # the same bytes through the framework's plain copy, same loop on the same two tensors: what a launch that reads and
            # writes 2 x 154 MB reaches on this chip at all (8 TB/s is the memory's rating, not a kernel's)
            for _ in range(10):
                x.clone(), f.clone()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                x.clone()
                f.clone()
            e1.record()
            torch.cuda.synchronize()
            ga_copy_ms = e0.elapsed_time(e1) / reps
            # the same two jobs in ONE launch (eqa_group_action_pair: job 1's first blocks fill the CUs job 0's tail leaves idle)
            for _ in range(10):
                ops.group_action_pair(x, f, gidx, th_c, fl_c, H // 2, th_i, fl_i, None)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                ops.group_action_pair(x, f, gidx, th_c, fl_c, H // 2, th_i, fl_i, None)
            e1.record()
            torch.cuda.synchronize()
            ga_ms = e0.elapsed_time(e1) / reps
            # roofline.uniform_c8 (round 6): inside the step the random-weight network prefers right-angle elements, whose source
            # windows are the cheapest; this is the same kernel on a seeded UNIFORM C8 index, one HIP-event bracket per launch (like the
            # in-step figure), over the ring of distinct batches (3 x 154 MB in + fresh outputs: nothing is re-read from a cache)
            uni = {}
            for nm, fn in (("canon_transform", lambda i: ops.canon_transform(xs[i], gidx, th_c, fl_c, H // 2)),
                           ("invert_action", lambda i: ops.invert_action(fs[i], gidx, th_i, fl_i, None))):
                for r in range(6):
                    fn(r % NBUF)
                evs = []
                for r in range(30):
                    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a_.record()
                    fn(r % NBUF)
                    b_.record()
                    evs.append((a_, b_))
                torch.cuda.synchronize()
                uni[nm] = sum(a_.elapsed_time(b_) for a_, b_ in evs) / len(evs)
            # self check: every image of rank 0's batch 0 against the CPU oracle (seed 0 / seed 1000 as generated above)
            self_check = None
            if rank == 0 and not args.no_cpu_baseline:
                n = B if args.check_images <= 0 else min(args.check_images, B)   # default: the WHOLE timed batch (256 images ~ 10 s of CPU oracle)
                y = can(xs[0])
                acts = can.canonicalization_info_dict["group_activations"]
                gidx = can.canonicalization_info_dict["group_index"]
                inv = can.invert_canonicalization(fs[0], induced_rep_type="scalar")
                self_check = oracle_check(can, xs[0][:n], fs[0][:n], y[:n], inv[:n], acts[:n], gidx[:n])

        del xs, fs, x, f, can
        torch.cuda.empty_cache()

        total_images = B * world * args.steps
        n_ct, ms_ct = ktimes.get("canon_transform", (0, float("nan")))
        n_iv, ms_iv = ktimes.get("invert_action", (0, float("nan")))
        ach = B * BYTES_TRANSFORM / (ms_ct * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch
        # correction; tools/collect_traffic.sh).  Counters cannot be collected inside this process, so the committed
        # measurement of the same kernel / shape is reported; null when it does not match this run's shape.
        from equiadapt_amd import _lib as _eqalib

        traffic, tsrc, tstate = None, None, None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tpath = os.path.join(ROOT, "profiles", rnd, "traffic_group_action.json")
            if os.path.exists(tpath) and B == 256:
                tj = json.load(open(tpath))
                traffic, tsrc = tj.get("traffic_bytes_per_launch"), f"profiles/{rnd}/traffic_group_action.json"
                stamp = tj.get("kernel_source_sha1")
                tstate = ("committed, no source stamp (collected before round 5)" if stamp is None else
                          "committed, kernel source unchanged since collection" if stamp == _eqalib.source_hash("group_action.hip") else
                          "committed, STALE: csrc/group_action.hip changed since collection")
                break
        ga_bytes = 2 * B * BYTES_TRANSFORM
        line.update({
            "value": total_images / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
            "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
            "roofline": {"bound": "hbm", "kernel": "group_action_kernel<3,true> via eqa_canon_transform_fwd",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": f"bytes/launch (rocprofv3 PMC, {tsrc})", "traffic_source": tstate,
                         "launches_timed": n_ct, "avg_launch_ms": ms_ct, "algorithmic_bytes_per_launch": B * BYTES_TRANSFORM,
                         "uniform_c8": {"avg_launch_ms": uni["canon_transform"], "achieved": B * BYTES_TRANSFORM / (uni["canon_transform"] * 1e-3) / 1e9,
                                        "frac": B * BYTES_TRANSFORM / (uni["canon_transform"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "invert_avg_launch_ms": uni["invert_action"],
                                        "invert_frac": B * BYTES_TRANSFORM / (uni["invert_action"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "how": "the same kernel on a seeded uniform C8 index (all eight elements, 5/8 of them non-right-angle), "
                                               "one HIP-event bracket per launch, 30 launches over a ring of 3 distinct 154 MB batches; the "
                                               "in-step `frac` above runs on the index the random-weight network chooses (mostly right angles)"},
                         "in_step_note": "`frac` is measured INSIDE the timed step, right behind the network's bf16 piece GEMM: the chip leaves that "
                                         "kernel at a reduced clock and this (unchanged) kernel runs ~9 us slower for it -- 0.73 inside the step "
                                         "with EQA_FFT_GEMM_PIECES=f32 on the same box (profiles/r06/bench_step_only_round5_forms.json); "
                                         "`uniform_c8` and `frac_of_copy` are the kernel back to back",
                         "frac_of_copy": ga_copy_ms / ga2_ms,
                         "frac_of_copy_note": "torch's clone() of the same bytes / this kernel, both back to back on the same tensors (the "
                                              "`group_action` leg: canonicalize + invert vs clone(x) + clone(f)): the share of a plain copy's "
                                              "rate the kernel reaches -- 8 TB/s is the memory's rating, a copy is what a launch can get"},
            "kernels_ms": {"canon_transform": ms_ct, "invert_action": ms_iv},
            "ms_per_step_with_event_brackets": elapsed_kt / args.steps * 1e3,
            "step_ms_by_gemm_form": {**step_forms, "note": "the whole timed step with the complex GEMM as: h3 = fp16 matrix cores on TWO fp16 "
                                     "pieces per fp32 operand (11 + 11 bits and a sign: within one fp32 ulp), three exact products per "
                                     "product, fp32 accumulation, operands scaled by powers of two under the bound the fused lifting kernel "
                                     "hands over; 6 / 9 = bf16 matrix cores on exact three-piece splits with six / nine piece products; f32 = "
                                     "v_mfma_f32_32x32x2_f32.  `default` is what `value` ran (fftconv.gemm_form: h3 where the producer bounds "
                                     "|V|, Cin >= 64, Cout % 128 == 0 -- measured CLOSER to fp64 than the fp32 instruction, gated by "
                                     "tests/test_gpu_parity.py; EQA_FFT_GEMM_PIECES=f32 restores the fp32 instruction)"},
            "group_action": {"images_s_per_gpu": B / (ga2_ms * 1e-3), "ms": ga2_ms,
                             "achieved_GBs": ga_bytes / (ga2_ms * 1e-3) / 1e9,
                             "frac_hbm_peak": ga_bytes / (ga2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "pair_launch_ms": ga_ms, "pair_launch_frac_hbm_peak": ga_bytes / (ga_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "copy_same_bytes_ms": ga_copy_ms, "frac_of_copy": ga_copy_ms / ga2_ms,
                             "note": "canonicalize x + invert f only, seeded random C8 index (all eight elements), back to back: "
                                     "ms / frac_hbm_peak = eqa_canon_transform_fwd then eqa_invert_action_fwd, the two launches the "
                                     "library's canonicalize / invert_canonicalization make (comparable with rounds 1-2); pair_launch_* = "
                                     "the same two jobs in one eqa_group_action_pair launch (bit-identical results; an ABI entry point "
                                     "for callers that hold x and f at the same time, not what the library's two calls use)"},
            "self_check": self_check,
        })
        if default_form == "h3":
            line["dtype"] = ("f32 (resampling / FFT / accumulation in fp32; the network's channel contraction as 2xfp16 split of every fp32 "
                             "operand (within one fp32 ulp), 3 exact products, fp32 accumulate -- closer to fp64 than the fp32 matrix "
                             "instruction, see step_ms_by_gemm_form)")
        elif default_form in ("6", "9"):
            line["dtype"] = (f"f32 (resampling / FFT / accumulation in fp32; the network's channel contraction as 3xbf16 exact split, "
                             f"{default_form} products, fp32 accumulate -- no further from fp64 than the fp32 matrix instruction, see step_ms_by_gemm_form)")
        line["stages"] = stage_table(ktimes, B)
        # the kernel the step spends most of its time in, against ITS roofline (the `roofline` object above is the metric's
        # HBM-bound transform kernel); traffic from the committed PMC profile when it is there
        dom = max((k for k in line["stages"] if "frac" in line["stages"][k]), key=lambda k: line["stages"][k]["ms"], default=None)
        if dom is not None:
            d = dict(line["stages"][dom])
            tr, tsrc = None, None
            try:
                tsrc = next(r for r in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", r, "traffic_canon_net.json")))
                tj = json.load(open(os.path.join(ROOT, "profiles", tsrc, "traffic_canon_net.json")))
                key = {"fft_gemm": {"h3": "fft_cgemm3m_bf16_block_kernel<3>", "6": "fft_cgemm3m_bf16_block_kernel<6>", "9": "fft_cgemm3m_bf16_block_kernel<9>"}.get(default_form, "fft_cgemm3m_kernel"), "fft_input": "fft48_fwd_fused_kernel", "fft_output_sums": "fft48_inv_pipe_kernel",
                       "lift_conv": "lift_conv_dense_kernel",
                       "lift_fft_input": "lift5_fft48_fused_h2_kernel" if _fc.LIFT_FFT_FORM == "h2" else "lift5_fft48_fused_kernel"}.get(dom)
                tr = tj.get(key, {}).get("traffic_bytes_per_launch") if B == 256 else None
            except (OSError, ValueError, StopIteration):
                pass
            line["roofline_dominant"] = {"stage": dom, "kernel": d.get("what"), "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                                         "unit": d["unit"], "frac": d["frac"], "avg_launch_ms": d["ms"], "share_of_step": d["ms"] / line["ms_per_step"],
                                         "traffic": tr, "traffic_unit": f"HBM bytes/launch (rocprofv3 PMC, profiles/{tsrc}/traffic_canon_net.json)",
                                         "traffic_source": "committed"}

    if args.mode in ("all", "train"):
        line["train"] = leg_train_images(comm, args.train_steps, args.train_warmup, args.train_batch)
        line["train_pointcloud"] = leg_train_pointcloud(comm, max(args.train_steps, 20), args.train_warmup + 2, 64)
        if args.mode == "train":
            line.update({"metric": "DP training step images/sec (224x224 C8 canonicalizer + ResNet-50)", "value": line["train"]["images_s"],
                         "ms_per_step": line["train"]["ms_per_step"], "steps": args.train_steps, "warmup": args.train_warmup})
    if args.mode == "all":
        line["configs"] = leg_configs(comm, with_cpu=(world == 1 and not args.no_cpu_baseline))
        # the two training loops whose tqdm read-outs are the only throughput numbers in the reference repository (BASELINE.md
        # section 1: CIFAR-10 at 64 x 64, B = 512, C4; hardware unstated) -- per rank, no collective (the notebook is single-GPU)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_tutorial

        line["tutorial"] = {leg: bench_tutorial.run_leg(leg, dev, 512, 20, 5, barrier=comm.barrier) for leg in ("prior", "optimized")}
        if rank == 0:
            # the step's dominant contraction on the bf16 matrix cores with fp32 semantics (every operand split exactly into three bf16
            # pieces; 9 = every piece product, 6 = without the three of relative size <= 2^-24): opt-in (EQA_FFT_GEMM_PIECES),
            # measured here beside the fp32 matrix instruction the timed step uses, with each result's distance to fp64
            import kbench_gemm_pieces

            line["gemm_pieces"] = kbench_gemm_pieces.measure(1024, 256, 256, 10, dev)
            line["gemm_pieces"]["note"] = ("eqa_fft48k5_cgemm3m (f32: what the timed step runs) vs eqa_fft48k5_cgemm3m_bf16x3 with 9 / 6 piece "
                                           "products on the headline layer's shape; not part of `value`")
        line["tutorial"]["note"] = ("understanding_discrete_canonicalization.ipynb cells 17+21 (ESCNN k=9, 16 ch, 3 layers, prior loss) and "
                                    "26+30 (Optimized + ConvNetwork k=5, artifact_err_wt=1000): canonicalize -> loss -> backward -> Adam step "
                                    "-> identity metric, synthetic CIFAR-shaped batches; images_s per rank")
    parity_ok = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.mode in ("all", "forward"):
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_reps)
        # every parity record of the line against its own stated tolerance; the run FAILS (exit 3, after the line is printed) when
        # one is outside it -- a throughput number whose results are off is not a result
        records = {"self_check": line.get("self_check")}
        records.update({k: v.get("parity") for k, v in (line.get("configs") or {}).items()})
        checked = {k: bool(v["ok"]) for k, v in records.items() if isinstance(v, dict) and "ok" in v}
        parity_ok = all(checked.values()) if checked else None
        line["parity_ok"] = parity_ok
        line["parity_checked"] = checked
        print(json.dumps(line), flush=True)
    comm.close()
    if parity_ok is False:
        print(f"bench.py: parity record(s) outside tolerance: {[k for k, v in checked.items() if not v]}", file=sys.stderr)
        sys.exit(3)


def stage_table(ktimes, B):
    """The canonicalization network's kernels at this batch against the roofline that bounds each (algorithmic bytes / flops
    per launch, fp32): 96x96x3 -> lift 5x5 -> 92x92x256 -> 5x5 (FFT or Winograd) -> 88x88x256 consumed as window sums."""
    px_l, tiles = B * 92 * 92, B * 22 * 22
    v_bytes = tiles * 64 * 256 * 4
    spec = {
        "winograd_input": ("hbm", px_l * 256 * 4 + v_bytes, "eqa_winograd_f4k5_input (hand-written)"),
        "winograd_gemm": ("mfma", 2.0 * 64 * tiles * 256 * 256, "64 x [tiles x 256].[256 x 256] strided-batched GEMM (library)"),
        "winograd_output_sums": ("hbm", v_bytes, "eqa_winograd_f4k5_output_sums incl. finalize (hand-written)"),
        "lift_conv": ("mfma", 2.0 * px_l * 256 * 75, "eqa_lift_conv_nhwc (hand-written fp32 MFMA)"),
        # centre crop 224 -> 180 + antialiased resize -> 96: the crop window in, the resized planes out
        "crop_resize_aa": ("hbm", B * 3 * (180 * 180 + 96 * 96) * 4, "eqa_crop_resize_aa (hand-written)"),
    }
    # the same layer as an overlap-save FFT convolution (default): 2 x 2 tiles of 48 x 48 per image, 1154 stored frequencies;
    # algorithmic bytes = activation in + spectra out (input), spectra in (output; the map itself is never written).  The
    # per-frequency complex channel contraction: hand-written 3-multiplication GEMM (3 real products per complex one: those are
    # the flops counted), or with EQA_FFT_GEMM=lib the library's real GEMM (4).
    from equiadapt_amd.images.canonicalization_networks import fftconv

    m_tiles, spectra = B * 4, 1154 * B * 4 * 512 * 4
    own_gemm = fftconv.gemm3m_supported(256, 256)
    form = (fftconv.LAST_FORM or fftconv.gemm_form(256, 256)) if own_gemm else "lib"
    real_products = 3.0 * 2.0 * 1154 * m_tiles * 256 * 256           # flops of the 3 real products per complex one
    if form == "h3":
        # fp16 form: every real product is 3 fp16 piece products -- the matrix-core flops, priced against the dense fp16 peak (= bf16's)
        gemm_spec = ("mfma_bf16", 3.0 * real_products,
                     "eqa_fft48k5_cgemm3m_f16x2: 1154 x [tiles x 256].[256 x 256] complex products, 3-multiplication form, every fp32 operand "
                     "split into two fp16 pieces (within one fp32 ulp), 3 piece products per real product on v_mfma_f32_32x32x16_f16, fp32 "
                     "accumulate (hand-written)")
    elif form in ("6", "9"):
        # piece form: every real product is `form` bf16 piece products -- THOSE are the matrix-core flops, priced against the dense
        # bf16 peak (the fp32-equivalent rate, real_products / time, is reported beside it)
        gemm_spec = ("mfma_bf16", float(form) * real_products,
                     f"eqa_fft48k5_cgemm3m_bf16x3: 1154 x [tiles x 256].[256 x 256] complex products, 3-multiplication form, every fp32 operand "
                     f"split exactly into three bf16 pieces, {form} piece products per real product on v_mfma_f32_32x32x16_bf16, fp32 accumulate "
                     f"(hand-written)")
    elif own_gemm:
        gemm_spec = ("mfma", real_products, "eqa_fft48k5_cgemm3m: 1154 x [tiles x 256].[256 x 256] complex products, 3-multiplication form on "
                                            "the fp32 MFMA (hand-written)")
    else:
        gemm_spec = ("mfma", 4.0 / 3.0 * real_products, "1154 x [tiles x 512].[512 x 512] batched GEMM, complex as real (library)")
    spec.update({
        "fft_input": ("hbm", px_l * 256 * 4 + spectra, "eqa_fft48k5_input: row + column FFT-48 passes (hand-written)"),
        # round 6: the lifting convolution fused into the forward transform (one launch instead of lift_conv + fft_input; the lifted
        # map is never written): bounded by the fp32 matrix pipe (2 x 75 flops per lifted pixel, 1.19 x recomputed on the tile overlap
        # -- the algorithmic count below is WITHOUT the overlap) with the spectrum stores (2.4 GB) hidden behind it
        "lift_fft_input": (("hbm", B * 96 * 96 * 3 * 4 + spectra,
                            "eqa_absmax_slots + eqa_lift5_fft48k5_input_f16x2: lifting conv on two fp16 pieces per value (three exact products on "
                            "v_mfma_f32_16x16x32_f16, fp32 accumulate) + ReLU + row / column FFT-48 in one persistent kernel, spectra out as "
                            "whole 128-byte lines; with the convolution on the matrix cores what is left is the transforms' vector "
                            "arithmetic over the spectrum stores (hand-written)")
                           if fftconv.LIFT_FFT_FORM == "h2" else
                           ("mfma", 2.0 * px_l * 256 * 75, "eqa_lift5_fft48k5_input: lifting conv (fp32 MFMA 16x16x4) + ReLU + row / column FFT-48 in "
                            "one persistent kernel, spectra out as whole 128-byte lines (hand-written)")),
        "fft_gemm": gemm_spec,
        "fft_output_sums": ("hbm", spectra, "eqa_fft48k5_output_sums: column + row inverse passes + window sums + finalize (hand-written)"),
    })
    stages = {}
    for name, (bound, work, what) in spec.items():
        n_k, ms_k = ktimes.get(name, (0, None))
        if not ms_k:
            continue
        if bound == "hbm":
            a = work / (ms_k * 1e-3) / 1e9
            stages[name] = {"what": what, "ms": ms_k, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": a / HBM_PEAK_GBS, "launches_timed": n_k}
        else:
            a = work / (ms_k * 1e-3) / 1e12
            peak = MFMA_BF16_PEAK_TF if bound == "mfma_bf16" else MFMA_F32_PEAK_TF
            stages[name] = {"what": what, "ms": ms_k, "bound": "mfma", "achieved": a, "peak": peak,
                            "unit": "TFLOP/s", "frac": a / peak, "launches_timed": n_k}
            if bound == "mfma_bf16":
                stages[name]["fp32_equivalent_TFLOPs"] = real_products / (ms_k * 1e-3) / 1e12
                stages[name]["note"] = ("achieved / peak count the piece products the matrix cores execute against the dense bf16 / fp16 peak; under "
                                        "this instruction stream the chip holds ~1.6 GHz (power), see DESIGN.md section 6")
                if name == "fft_gemm":     # the fp16 form is no longer far from its memory time: V in, Mo out, the filter pieces once
                    gb = (2.0 * spectra + 1154 * 256 * 256 * 3 * (2 if form == "h3" else 3) * 2) / 1e9
                    stages[name]["hbm_algorithmic_GB"] = gb
                    stages[name]["hbm_frac_if_memory_bound"] = gb / (ms_k * 1e-3) / HBM_PEAK_GBS
    for name in ("group_pool", "window_sums", "crop_resize_aa", "sums_gemv"):
        if name in ktimes and name not in stages:
            stages[name] = {"ms": ktimes[name][1], "launches_timed": ktimes[name][0]}
    return stages


if __name__ == "__main__":
    main()
