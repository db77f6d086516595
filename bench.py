"""bench.py -- canonicalize+invert throughput on BASELINE.json's headline config (224x224x3, C8).

One "step" = one pass of the hot path over one batch that is already resident in HBM:
    y   = canonicalizer(x)                      # crop+resize -> canonicalization network -> group pool/argmax
                                                #   -> fused pad/rotate/crop (eqa_canon_transform_fwd)
    out = canonicalizer.invert_canonicalization(f, induced_rep_type="scalar")     # eqa_invert_action_fwd
with x, f : (B, 3, 224, 224) fp32 synthetic, and the canonicalization network = ESCNNEquivariantNetwork
(out_channels=32, kernel_size=5, num_layers=3, C8) on a 96x96 crop+resize -- the reference's default
(examples/images/classification/configs/canonicalization/group_equivariant.yaml).  The wrapped prediction
network (ResNet-50) is NOT part of the measured path.  Ranks shard the batch; the forward path has no
collective, so scaling is weak (per-GPU batch fixed).

Prints ONE JSON line on rank 0 (contract in the task statement).  Extra objects:
  roofline      HBM roofline of the dominant hand-written kernel (the canonicalizing transform),
                from HIP events recorded inside the timed region;
  group_action  transform+invert only (random group index), the figure the "% HBM roofline" target is about;
  stages        the canonicalization network's kernels (97 % of the step), each against the roofline that bounds it: the
                hand-written FFT (or Winograd) transforms (HBM), the library batched fp32 GEMM between them and the
                hand-written lifting convolution (fp32 MFMA), HIP events inside the timed region;
  cpu_baseline  the CPU oracle (reference op order) on this host, bounded sample, rank 0 / N=1 only.
"""
import argparse
import json
import os
import statistics
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured-achievable
MFMA_F32_PEAK_TF = 157.3  # dense fp32-input MFMA (v_mfma_f32_32x32x2_f32), same guide
H = W = 224
C = 3
BYTES_TRANSFORM = 2 * C * H * W * 4  # read + write per image, fp32: 1,204,224 B


def build_canonicalizer(device):
    import equiadapt_amd as ea

    torch.manual_seed(2)
    net = ea.ESCNNEquivariantNetwork((3, 96, 96), out_channels=32, kernel_size=5, group_type="rotation",
                                     num_rotations=8, num_layers=3)
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=96)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (C, H, W))
    return can.to(device).eval()


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
        for nt in sorted({cores, 64, 32, 16, 8} & set(range(1, cores + 1)), reverse=True):
            torch.set_num_threads(nt)
            step()
            t0 = time.perf_counter()
            step()
            dt = time.perf_counter() - t0
            if dt < best[0]:
                best = (dt, nt)
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
            "group_action_only_images_s": sample / ga}


def cpu_baseline_config(name: str, state: dict):
    """cpu_baseline leg for the other BASELINE configs (used by tools/bench_configs.py): the oracle's op sequence on the host,
    16 threads, bounded sample.  `state` carries the network weights of the GPU run so both sides compute the same thing."""
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
        return {"images_s": 128 / dt, "ms": dt * 1e3, "sample": "B=128, 16 threads"}
    if name == "cfg4":
        pcs, sd = torch.randn(4, 3, 1024), state["sd"]
        dt = timed(lambda: po.canonicalize_pointcloud(pcs, po.gram_schmidt(po.vnsmall_forward(pcs, sd))), 2)
        return {"clouds_s": 4 / dt, "ms": dt * 1e3, "sample": "B=4, 16 threads"}
    if name == "cfg5":  # transform + invert only (the orbit / network part is a few ms either way)
        x1, ang, refl = torch.randn(1, 3, 1024, 1024), torch.tensor([90.0]), torch.tensor([1.0])
        dt = timed(lambda: (io.canonicalize_images(x1, ang, refl, (3, 1024, 1024)), io.invert_action(x1[:, :1], ang, refl, 4, 8, "scalar")), 2)
        return {"images_s": 1 / dt, "ms": dt * 1e3, "sample": "B=1, 16 threads"}
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL on ROCm
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    can = build_canonicalizer(dev)
    B = args.batch
    x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(rank)).to(dev)
    f = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(1000 + rank)).to(dev)

    def step():
        y = can(x)
        out = can.invert_canonicalization(f, induced_rep_type="scalar")
        return y, out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        with ops.KernelTimer() as kt:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            elapsed = time.perf_counter() - t0
        ktimes = kt.summary()

        # group-action-only leg: the two resampling kernels back to back with a seeded random index
        gidx = torch.randint(0, 8, (B,), generator=torch.Generator().manual_seed(1)).to(dev, torch.int32)
        th_c, fl_c = device_tables("canonicalize", 8, False, (2 * H, 2 * W), dev)
        th_i, fl_i, _ = device_tables("invert", 8, False, (H, W), dev)
        for _ in range(3):
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
        ga_ms = e0.elapsed_time(e1) / reps

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

    if rank == 0:
        total_images = B * world * args.steps
        n_ct, ms_ct = ktimes.get("canon_transform", (0, float("nan")))
        n_iv, ms_iv = ktimes.get("invert_action", (0, float("nan")))
        n_gp, ms_gp = ktimes.get("group_pool", (0, None))
        n_ws, ms_ws = ktimes.get("window_sums", (0, None))
        ach = B * BYTES_TRANSFORM / (ms_ct * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch
        # correction; tools/collect_traffic.sh).  Counters cannot be collected inside this process, so the committed
        # measurement of the same kernel / shape is reported; null when it does not match this run's shape.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01", "traffic_group_action.json")
        if os.path.exists(tpath) and B == 256:
            traffic = json.load(open(tpath)).get("traffic_bytes_per_launch")
        ga_bytes = 2 * B * BYTES_TRANSFORM
        line = {
            "metric": "canonicalize+invert images/sec (224x224 C8)",
            "value": total_images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 224x224x3 synthetic, C8, GroupEquivariantImageCanonicalization + "
                                   "ESCNNEquivariantNetwork(32ch,k5,3 layers, crop 0.8, resize 96) forward, "
                                   "then invert_canonicalization(scalar, 3ch); prediction network excluded",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world} (no collective)"},
            "roofline": {"bound": "hbm", "kernel": "group_action_kernel<3,true> via eqa_canon_transform_fwd",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/r01/traffic_group_action.json)",
                         "launches_timed": n_ct, "avg_launch_ms": ms_ct,
                         "algorithmic_bytes_per_launch": B * BYTES_TRANSFORM},
            "kernels_ms": {"canon_transform": ms_ct, "invert_action": ms_iv, "group_pool": ms_gp, "window_sums": ms_ws},
            "group_action": {"images_s_per_gpu": B / (ga_ms * 1e-3), "ms": ga_ms,
                             "achieved_GBs": ga_bytes / (ga_ms * 1e-3) / 1e9,
                             "frac_hbm_peak": ga_bytes / (ga_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "eqa_canon_transform_fwd + eqa_invert_action_fwd only, seeded random index"},
        }
        # canonicalization-network stages at this batch: 96x96x3 -> lift 5x5 -> 92x92x256 -> Winograd F(4x4,5x5) -> 88x88x256
        # (consumed as window sums); algorithmic bytes / flops per launch, fp32
        px_l, px_w, tiles = B * 92 * 92, B * 88 * 88, B * 22 * 22
        v_bytes = tiles * 64 * 256 * 4
        spec = {
            "winograd_input": ("hbm", px_l * 256 * 4 + v_bytes, "eqa_winograd_f4k5_input (hand-written)"),
            "winograd_gemm": ("mfma", 2.0 * 64 * tiles * 256 * 256, "64 x [tiles x 256].[256 x 256] strided-batched GEMM (library)"),
            "winograd_output_sums": ("hbm", v_bytes, "eqa_winograd_f4k5_output_sums incl. finalize (hand-written)"),
            "lift_conv": ("mfma", 2.0 * px_l * 256 * 75, "eqa_lift_conv_nhwc (hand-written fp32 MFMA)"),
        }
        # the same layer as an overlap-save FFT convolution (default): 2 x 2 tiles of 48 x 48 per image, 1154 stored frequencies;
        # algorithmic bytes = activation in + spectra out (input), spectra in (output; the map itself is never written)
        m_tiles, spectra = B * 4, 1154 * B * 4 * 512 * 4
        spec.update({
            "fft_input": ("hbm", px_l * 256 * 4 + spectra, "eqa_fft48k5_input: row + column FFT-48 passes (hand-written)"),
            "fft_gemm": ("mfma", 2.0 * 1154 * m_tiles * 512 * 512, "1154 x [tiles x 512].[512 x 512] batched GEMM, complex as real (library)"),
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
                stages[name] = {"what": what, "ms": ms_k, "bound": "mfma", "achieved": a, "peak": MFMA_F32_PEAK_TF,
                                "unit": "TFLOP/s", "frac": a / MFMA_F32_PEAK_TF, "launches_timed": n_k}
        line["stages"] = stages
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_reps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
