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
