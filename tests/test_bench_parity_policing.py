"""bench.py polices its own parity records: `ok` against the stated tolerance per record, `parity_ok` on the line (VERDICT r04,
weak 2: a cfg4 record outside its printed tolerance went unnoticed).  CPU: the record logic only, no product code."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_mod"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_image_parity_record_ok_follows_its_tolerances():
    b = _bench()
    good = {"index_match": 1.0, "acts_max_err": 2e-9, "canonicalize_max_err": 0.0, "invert_max_err": 3e-6}
    assert b.image_parity_ok(dict(good), 1e-4, 4.6e-3)["ok"] is True
    for key, bad in (("index_match", 0.996), ("acts_max_err", 1e-6), ("canonicalize_max_err", 2e-5), ("invert_max_err", 1.1e-5)):
        rec = dict(good)
        rec[key] = bad
        out = b.image_parity_ok(rec, 1e-4, 4.6e-3)
        assert out["ok"] is False and "tolerance" in out, key
    with_targets = dict(good, masks_bit_exact=False, boxes_max_err=0.0)
    assert b.image_parity_ok(with_targets, 1e-3, 1.0)["ok"] is False
    assert b.image_parity_ok(dict(good, masks_bit_exact=True, boxes_max_err=5e-4), 1e-3, 1.0)["ok"] is True
    assert b.PIXEL_TOL == 1e-5


def test_bench_source_exits_nonzero_on_a_parity_breach():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'line["parity_ok"] = parity_ok' in src and "sys.exit(3)" in src and "if parity_ok is False" in src
