"""The guard-band harness of tests/conftest.py (EQA_GUARD=1) must itself be shown to detect what it is there for."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("EQA_GUARD", "0") != "1", reason="only meaningful under EQA_GUARD=1")
def test_guard_bands_detect_a_write_past_the_end_and_in_front_of_a_buffer():
    import conftest

    dev = torch.device("cuda:0")
    conftest._guard_live.clear()
    t = torch.empty(1000, dtype=torch.float32, device=dev)          # a guarded allocation: a view into a larger raw buffer
    t.fill_(1.0)
    assert conftest._guard_check("clean") == []
    t = torch.empty((3, 5, 7), dtype=torch.float32, device=dev)
    t.as_strided((3 * 5 * 7 + 2,), (1,)).fill_(2.0)                 # two floats past the end
    bad = conftest._guard_check("past the end")
    assert len(bad) == 1 and bad[0].endswith("0 bytes overwritten in front, 8 behind"), bad
    t = torch.zeros((4, 4), dtype=torch.float32, device=dev)
    raw, _, _ = conftest._guard_live[-1]
    raw[conftest._GUARD_BYTES - 4:conftest._GUARD_BYTES] = 0         # one float in front
    bad = conftest._guard_check("in front")
    assert len(bad) == 1 and "4 bytes overwritten in front" in bad[0], bad
    conftest._guard_stats["violations"] -= 2                         # the two planted ones are not findings
    # a library kernel with a wrong size argument is caught the same way: the canonicalizing transform told the batch has one
    # image more than its output buffer holds
    from equiadapt_amd import _lib
    from equiadapt_amd.images.utils import device_tables

    lib = _lib.load()
    th, fl = device_tables("canonicalize", 4, False, (64, 64), dev)
    x = torch.randn(3, 3, 32, 32, device=dev)
    y = torch.empty((2, 3, 32, 32), dtype=torch.float32, device=dev)
    gidx = torch.zeros(3, dtype=torch.int32, device=dev)
    assert lib.eqa_canon_transform_fwd(x.data_ptr(), y.data_ptr(), gidx.data_ptr(), th.data_ptr(), None, 4, 3, 3, 32, 32, 16,
                                       torch.cuda.current_stream().cuda_stream) == 0
    bad = conftest._guard_check("one image too many")
    assert any("behind" in b and "empty(2, 3, 32, 32)" in b for b in bad), bad
    conftest._guard_stats["violations"] -= len(bad)
