"""The canonicalizing transform on config 5's frames (3 x 1024 x 1024, D4, pad 512) against a plain copy of the same bytes:
per group element and per batch size, 30 launches between one pair of events, on a ring of buffers larger than the
Infinity Cache.  (bench.py's cfg5 leg reports 0.57-0.59 of the HBM peak at B = 32 where the 224 x 224 metric kernel reaches 0.66.)

    python tools/kbench_large.py            (EQA_LIB=build_variants/libeqa_<variant>.so for an A/B)
"""
import sys

import torch

sys.path.insert(0, ".")
from equiadapt_amd import ops                                      # noqa: E402
from equiadapt_amd.images.utils import device_tables               # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=30):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    th, fl = device_tables("canonicalize", 4, True, (2048, 2048), dev)
    if "nohint" in sys.argv:          # without the right-angle window hint (LDS sized for 47-row windows)
        th = th.clone()
    for B in ((32,) if "quick" in sys.argv else (8, 16, 24, 32, 64)):
        ring = max(2, int(1.2e9 // (B * 3 * 1024 * 1024 * 4)))
        xs = [torch.randn(B, 3, 1024, 1024, device=dev) for _ in range(ring)]
        ys = [torch.empty_like(xs[0]) for _ in range(ring)]
        nbytes = 2 * xs[0].numel() * 4
        it = [0]

        def copy():
            i = it[0] = (it[0] + 1) % ring
            ys[i].copy_(xs[i])

        ms = timeit(copy)
        line = f"B={B:3d} ring {ring}: copy_ {ms * 1e3:7.1f} us {nbytes / ms / 1e6:7.0f} GB/s |"
        cases = [("rand", torch.randint(0, 8, (B,), device=dev, dtype=torch.int32))]
        if B in (32,):
            cases += [(f"e{e}", torch.full((B,), e, device=dev, dtype=torch.int32)) for e in range(8)]
            cases += [("rand again", cases[0][1]), ("rand, other draw", torch.randint(0, 8, (B,), device=dev, dtype=torch.int32))]
        for name, g in cases:
            def run():
                i = it[0] = (it[0] + 1) % ring
                ops.canon_transform(xs[i], g, th, fl, 512, out=ys[i]) if HAS_OUT else ops.canon_transform(xs[i], g, th, fl, 512)

            ms = timeit(run)
            line += f" {name} {ms * 1e3:6.1f} us {nbytes / ms / 1e6 / 8000:.3f}"
        print(line, flush=True)
        del xs, ys
    if "quick" in sys.argv:
        return
    # the same bytes as 224 x 224 frames (the metric's shape): B = 672 images = 32 x 1024^2 worth
    th8, fl8 = device_tables("canonicalize", 8, False, (448, 448), dev)
    B = 672
    xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(3)]
    g = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
    it = [0]

    def run224():
        i = it[0] = (it[0] + 1) % 3
        ops.canon_transform(xs[i], g, th8, fl8, 112)

    ms = timeit(run224)
    print(f"224 x 224, B = {B} (as many bytes as 32 x 1024^2 .. x 0.64): {ms * 1e3:.1f} us {2 * xs[0].numel() * 4 / ms / 1e6 / 8000:.3f}")


import inspect                                                     # noqa: E402

HAS_OUT = "out" in inspect.signature(ops.canon_transform).parameters
if __name__ == "__main__":
    main()
