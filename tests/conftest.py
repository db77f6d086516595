import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built library is not in git: (re)build it when it is missing or older than its sources (hipcc cross-compiles
    gfx950 without a GPU, ~20 s), so that the suite runs from a fresh checkout.  Without hipcc the tests that need the
    library fail loudly by themselves."""
    try:
        from equiadapt_amd import _lib

        _lib.build()
    except Exception as exc:  # noqa: BLE001 -- reported, not fatal here
        print(f"[conftest] could not build libeqa_hip.so: {exc}")


def load_golden(name: str) -> dict:
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ----------------------------------------------------------------------------------------------------------------------------------
# Guard bands (round 4): `EQA_GUARD=1 python -m pytest tests -m gpu` runs the whole GPU suite with every device
# tensor that torch.empty / empty_like / zeros / zeros_like / ones / full hand out -- every output and workspace the package allocates
# for its kernels, and most test inputs -- placed between two poisoned bands of EQA_GUARD_KB (default 64) KiB.  After each test
# the bands of every buffer allocated during it are compared with the poison pattern: a kernel writing before or past a buffer
# it was given fails that test and names the buffer.  The hand-rolled producer / consumer pipelines, the inline-asm LDS-DMA with
# manual wait counts and the 32-bit buffer offsets of the kernels get an out-of-bounds WRITE detector that way (reads are not
# detected; a wild write beyond the band would be a fault, which the suite would also report).  Log: profiles/r04/guard_bands.txt.
# ----------------------------------------------------------------------------------------------------------------------------------
_GUARD_ON = os.environ.get("EQA_GUARD", "0") == "1"
_GUARD_BYTES = int(os.environ.get("EQA_GUARD_KB", "64")) * 1024
_GUARD_PATTERN = 0xA5
_guard_live = []      # (raw uint8 buffer, payload bytes, description) of the running test
_guard_stats = {"buffers": 0, "bytes": 0, "tests": 0, "violations": 0}


def _guard_install():
    import math

    real = {n: getattr(torch, n) for n in ("empty", "empty_like", "zeros", "zeros_like", "ones", "full")}

    def is_cuda(device):
        if device is None:
            return False
        if torch.device(device).type != "cuda":
            return False
        # (inside a hipGraph capture the band fills would be captured, not executed: buffers of a capture are left alone)
        return not torch.cuda.is_current_stream_capturing()

    def guarded(shape, dtype, device, memory_format, fill, what):
        dtype = dtype or torch.get_default_dtype()
        shape = tuple(int(s) for s in shape)
        n = math.prod(shape) if shape else 1
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pay = (nbytes + 255) // 256 * 256
        raw = real["empty"](pay + 2 * _GUARD_BYTES, dtype=torch.uint8, device=device)
        raw[:_GUARD_BYTES] = _GUARD_PATTERN
        raw[_GUARD_BYTES + nbytes:] = _GUARD_PATTERN         # (the alignment slack behind the payload is guard as well)
        flat = raw[_GUARD_BYTES:_GUARD_BYTES + nbytes].view(dtype)
        if memory_format == torch.channels_last and len(shape) == 4:
            N, C, H, W = shape
            t = flat.view(N, H, W, C).permute(0, 3, 1, 2)
        elif memory_format == torch.channels_last_3d and len(shape) == 5:
            N, C, D, H, W = shape
            t = flat.view(N, D, H, W, C).permute(0, 4, 1, 2, 3)
        else:
            t = flat.view(shape)
        if fill is not None:
            t.fill_(fill)
        _guard_live.append((raw, nbytes, f"{what}{shape} {dtype}"))
        _guard_stats["buffers"] += 1
        _guard_stats["bytes"] += nbytes
        return t

    def norm_size(args, kwargs):
        if "size" in kwargs:
            return tuple(kwargs.pop("size"))
        if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
            return tuple(args[0])
        return tuple(args)

    def plain(kwargs):
        return not (set(kwargs) - {"dtype", "device", "memory_format", "requires_grad"}) and not kwargs.get("requires_grad", False)

    def make_sized(name, fill):
        def fn(*args, **kwargs):
            if is_cuda(kwargs.get("device")) and plain(kwargs) and all(isinstance(a, (int, tuple, list, torch.Size)) for a in args):
                kw = dict(kwargs)
                size = norm_size(args, kw)
                if all(isinstance(s, int) for s in size):
                    return guarded(size, kw.get("dtype"), kw["device"], kw.get("memory_format"), fill, name)
            return real[name](*args, **kwargs)
        return fn

    def make_like(name, fill):
        def fn(t, **kwargs):
            dev = kwargs.get("device", t.device)
            if isinstance(t, torch.Tensor) and is_cuda(dev) and plain(kwargs) and t.layout == torch.strided:
                mf = kwargs.get("memory_format", torch.preserve_format)
                if mf == torch.preserve_format:
                    mf = (torch.channels_last if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)
                          else torch.contiguous_format)
                return guarded(t.shape, kwargs.get("dtype", t.dtype), dev, mf, fill, name)
            return real[name](t, **kwargs)
        return fn

    def full(size, fill_value, **kwargs):
        if is_cuda(kwargs.get("device")) and plain(kwargs) and isinstance(fill_value, (int, float)) and kwargs.get("dtype") is not None:
            return guarded(tuple(size), kwargs["dtype"], kwargs["device"], None, fill_value, "full")
        return real["full"](size, fill_value, **kwargs)

    torch.empty, torch.zeros, torch.ones = make_sized("empty", None), make_sized("zeros", 0), make_sized("ones", 1)
    torch.empty_like, torch.zeros_like = make_like("empty_like", None), make_like("zeros_like", 0)
    torch.full = full


def _guard_check(test_name: str):
    bad = []
    if _guard_live and torch.cuda.is_available():
        torch.cuda.synchronize()
    for raw, nbytes, what in _guard_live:
        lo, hi = raw[:_GUARD_BYTES], raw[_GUARD_BYTES + nbytes:]
        if not (bool((lo == _GUARD_PATTERN).all()) and bool((hi == _GUARD_PATTERN).all())):
            nlo, nhi = int((lo != _GUARD_PATTERN).sum()), int((hi != _GUARD_PATTERN).sum())
            bad.append(f"{what}: {nlo} bytes overwritten in front, {nhi} behind")
    _guard_live.clear()
    _guard_stats["tests"] += 1
    _guard_stats["violations"] += len(bad)
    return bad


if _GUARD_ON:
    _guard_install()

    @pytest.fixture(autouse=True)
    def _guard_bands(request):
        _guard_live.clear()
        yield
        bad = _guard_check(request.node.nodeid)
        assert not bad, "guard bands overwritten during " + request.node.nodeid + ":\n  " + "\n  ".join(bad)

    def pytest_terminal_summary(terminalreporter):
        s = _guard_stats
        terminalreporter.write_line(f"[guard bands] {s['tests']} tests, {s['buffers']} device buffers ({s['bytes'] / 1e9:.1f} GB of payload) between "
                                    f"{_GUARD_BYTES // 1024} KiB bands of 0x{_GUARD_PATTERN:02X}: {s['violations']} violations")
