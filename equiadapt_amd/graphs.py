"""Replayable hipGraph of one inference step of a canonicalizer (canonicalize, optionally invert).

The kernels of libeqa_hip.so are stream-ordered and never synchronise the host, so a whole step -- ~12 launches for the
CIFAR-shaped configuration, ~20 for the 224 x 224 one -- can be captured once (``torch.cuda.CUDAGraph`` is a hipGraph on
ROCm) and replayed with a single host call.  That matters where the step is launch-bound: at B = 128 of the 32 x 32 C4
configuration the eager step is 0.26 ms of which ~0.1 ms is kernels.  Nothing in the reference corresponds to this (it
runs eager PyTorch); the results are bit-identical to the eager path (tests/test_gpu_parity.py).

    step = GraphedCanonicalizer(canonicalizer, x_shape=(128, 3, 32, 32), invert_shape=(128, 3, 32, 32))
    y, group_index, inv = step(x, f)        # views of static buffers: overwritten by the next call

Inference only (``torch.no_grad``); the canonicalizer must be in eval() mode and its input shape fixed.
"""
from typing import Optional, Sequence, Tuple

import torch


class GraphedCanonicalizer:
    def __init__(self, canonicalizer: torch.nn.Module, x_shape: Sequence[int], invert_shape: Optional[Sequence[int]] = None,
                 induced_rep_type: str = "scalar", device: Optional[torch.device] = None, warmup: int = 3):
        if canonicalizer.training:
            raise RuntimeError("GraphedCanonicalizer captures an inference step: call .eval() on the canonicalizer first")
        self.can = canonicalizer
        dev = torch.device(device) if device is not None else next(canonicalizer.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(f"the canonicalizer is on {dev}: a hipGraph needs a ROCm device (no CPU fallback)")
        self.x = torch.zeros(tuple(x_shape), dtype=torch.float32, device=dev)
        self.f = torch.zeros(tuple(invert_shape), dtype=torch.float32, device=dev) if invert_shape is not None else None
        self.rep = induced_rep_type
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
        y = self.can(self.x)
        info = getattr(self.can, "canonicalization_info_dict", {})
        idx = info.get("group_index")
        if idx is None and isinstance(info.get("group_element"), dict):
            idx = info["group_element"].get("rotation")
        inv = self.can.invert_canonicalization(self.f, induced_rep_type=self.rep) if self.f is not None else None
        return y, idx, inv

    def __call__(self, x: torch.Tensor, f: Optional[torch.Tensor] = None):
        """Copies the inputs into the captured buffers, replays, returns (canonicalized x, group element, inverted f)."""
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
