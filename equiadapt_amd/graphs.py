"""Replayable hipGraph of one inference step of a canonicalizer (canonicalize, optionally invert).

The kernels of libeqa_hip.so are stream-ordered and never synchronise the host, so a whole step -- ~12 launches for the
CIFAR-shaped configuration, ~20 for the 224 x 224 one -- can be captured once (``torch.cuda.CUDAGraph`` is a hipGraph on
ROCm) and replayed with a single host call.  That matters where the step is launch-bound: at B = 128 of the 32 x 32 C4
configuration the eager step is 0.26 ms of which ~0.1 ms is kernels.  Nothing in the reference corresponds to this (it
runs eager PyTorch); the results are bit-identical to the eager path (tests/test_gpu_parity.py).

    step = GraphedCanonicalizer(canonicalizer, x_shape=(128, 3, 32, 32), invert_shape=(128, 3, 32, 32))
    y, group_index, inv = step(x, f)        # views of static buffers: overwritten by the next call

With ``targets_like`` (a list of {"boxes": (n, 4) tensor, "masks": (n, H, W) uint8 tensor} per sample, as the reference's
segmentation loop passes them, discrete_group.py:217-236) the boxes / masks are canonicalized inside the same graph:
``step(x, f, targets)`` copies them into static buffers and ``step.targets`` holds the transformed ones.

Inference only (``torch.no_grad``); the canonicalizer must be in eval() mode and its input shape fixed.
"""
from typing import Optional, Sequence, Tuple

import torch


class GraphedCanonicalizer:
    def __init__(self, canonicalizer: torch.nn.Module, x_shape: Sequence[int], invert_shape: Optional[Sequence[int]] = None,
                 induced_rep_type: str = "scalar", device: Optional[torch.device] = None, warmup: int = 3,
                 targets_like: Optional[Sequence[dict]] = None):
        if canonicalizer.training:
            raise RuntimeError("GraphedCanonicalizer captures an inference step: call .eval() on the canonicalizer first")
        self.can = canonicalizer
        dev = torch.device(device) if device is not None else next(canonicalizer.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(f"the canonicalizer is on {dev}: a hipGraph needs a ROCm device (no CPU fallback)")
        self.x = torch.zeros(tuple(x_shape), dtype=torch.float32, device=dev)
        self.f = torch.zeros(tuple(invert_shape), dtype=torch.float32, device=dev) if invert_shape is not None else None
        self.rep = induced_rep_type
        self.targets_in = None
        self.targets = None
        if targets_like is not None:
            self.targets_in = [{k: torch.zeros_like(v, device=dev).copy_(v) for k, v in t.items()} for t in targets_like]
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.device(dev):
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.stream):
                for _ in range(max(warmup, 1)):       # table uploads, workspace allocation, library algorithm search
                    self._step()
                self.stream.synchronize()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.y, self.group_index, self.inv = self._step()
            torch.cuda.current_stream(dev).wait_stream(self.stream)

    def _step(self) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
        if self.targets_in is not None:
            # (the reference flips the caller's boxes in place: hand the canonicalizer copies, the static inputs stay as given)
            y, self.targets = self.can(self.x, [{k: (v.clone() if k == "boxes" else v) for k, v in t.items()} for t in self.targets_in])
        else:
            y = self.can(self.x)
        info = getattr(self.can, "canonicalization_info_dict", {})
        idx = info.get("group_index")
        if idx is None and isinstance(info.get("group_element"), dict):
            idx = info["group_element"].get("rotation")
        inv = self.can.invert_canonicalization(self.f, induced_rep_type=self.rep) if self.f is not None else None
        return y, idx, inv

    def __call__(self, x: torch.Tensor, f: Optional[torch.Tensor] = None, targets: Optional[Sequence[dict]] = None):
        """Copies the inputs into the captured buffers, replays, returns (canonicalized x, group element, inverted f); the
        canonicalized targets are in ``.targets``."""
        if targets is not None:
            if self.targets_in is None or len(targets) != len(self.targets_in):
                raise ValueError("captured without targets (or for another number of samples): pass targets_like at construction")
            for dst, src in zip(self.targets_in, targets):
                for k in dst:
                    if dst[k].shape != src[k].shape:
                        raise ValueError(f"target '{k}': captured for {tuple(dst[k].shape)}, got {tuple(src[k].shape)}")
                    dst[k].copy_(src[k], non_blocking=True)
        if x.shape != self.x.shape or (f is not None and (self.f is None or f.shape != self.f.shape)):
            raise ValueError(f"captured for x {tuple(self.x.shape)} / f {None if self.f is None else tuple(self.f.shape)}; "
                             "capture another graph for another shape")
        self.x.copy_(x, non_blocking=True)
        if self.f is not None and f is not None:
            self.f.copy_(f, non_blocking=True)
        self.graph.replay()
        return self.y, self.group_index, self.inv

    def replay(self) -> None:
        """Replays on whatever the static inputs (`.x`, `.f`) hold."""
        self.graph.replay()
