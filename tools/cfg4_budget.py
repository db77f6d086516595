"""fp64 error budget of the point-cloud chain (VNSmall -> Gram-Schmidt -> rotate) on the bench's own clouds: the HIP path and the
fp32 oracle, each against the fp64 evaluation on the same neighbour sets.  Runs on the GPU box; prints one JSON object."""
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402
from equiadapt_amd import ops  # noqa: E402
from oracle import pointcloud_ops as po  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hp = types.SimpleNamespace(n_knn=20, pooling="mean")
    torch.manual_seed(2)
    vn = ea.VNSmall(hp)
    sd = {k: v.clone() for k, v in vn.state_dict().items()}
    vn = vn.to(dev).eval()
    out = {}
    for name, B, seed in (("bench_seed12_b8", 8, 12), ("cfg4_b64_seed0", 64, 0)):
        pcs = torch.randn(B, 3, 1024, generator=torch.Generator().manual_seed(seed))
        bud = po.fp64_error_budget(pcs, sd)
        with torch.no_grad():
            vec, R, y = ops.vnsmall_canonicalize(pcs.to(dev), vn.packed_parameters(), 20, "mean")
        vec, R, y = vec.cpu().double(), R.cpu().double(), y.cpu().double()
        vs = bud["v64"].abs().amax(dim=(1, 2))
        rec = {"cond": bud["cond"].tolist(),
               "v_rel_err_hip": ((vec - bud["v64"]).abs().amax(dim=(1, 2)) / vs).tolist(),
               "v_rel_err_oracle": ((bud["v32"].double() - bud["v64"]).abs().amax(dim=(1, 2)) / vs).tolist(),
               "R_err_hip": (R - bud["R64"]).abs().amax(dim=(1, 2)).tolist(), "R_err_oracle": bud["oracle_rotation_err"].tolist(),
               "R_hip_vs_gs64_of_hipvec": (R - po.gram_schmidt(vec)).abs().amax(dim=(1, 2)).tolist(),
               "y_err_hip": (y - bud["y64"]).abs().amax(dim=(1, 2)).tolist(), "y_err_oracle": bud["oracle_coords_err"].tolist(),
               "R_hip_vs_oracle": (R - bud["R32"].double()).abs().amax(dim=(1, 2)).tolist(),
               "y_hip_vs_oracle": (y - bud["y32"].double()).abs().amax(dim=(1, 2)).tolist()}
        out[name] = rec
        worst = max(range(B), key=lambda b: rec["cond"][b])
        print(name, "worst cond", rec["cond"][worst], {k: v[worst] for k, v in rec.items()}, file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
