#!/bin/bash
# HBM traffic of the dominant kernel, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (TCC slots), kernel-trace only.  Run on the GPU box:  bash tools/collect_traffic.sh <outdir>
set -e
out=${1:-gpurun_out/traffic}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export REPS=3
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o p --output-format csv -- python tools/prof_kernels.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o p --output-format csv -- python tools/prof_kernels.py > /dev/null 2>&1
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = {}
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/{sub}/p_counter_collection.csv")):
        if r["Counter_Name"] == name:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        short = "group_action_kernel" if "group_action_kernel" in k else ("group_pool_partial" if "group_pool_partial" in k else None)
        if short:
            res.setdefault(short, {})[name + "_KB_per_launch"] = sum(v) / len(v)
ga = res["group_action_kernel"]
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a streaming read
# (128-B requests tallied at 64 B) -> doubled; calibrated in this run on group_pool_partial, whose read volume is known.
gp = res["group_pool_partial"]
known = 256 * 32 * 8 * 84 * 84 * 4
ga["calibration_fetch_factor_on_group_pool"] = known / (gp["FETCH_SIZE_KB_per_launch"] * 1024)
ga["read_bytes"] = ga["FETCH_SIZE_KB_per_launch"] * 1024 * 2
ga["write_bytes"] = ga["WRITE_SIZE_KB_per_launch"] * 1024
ga["traffic_bytes_per_launch"] = ga["read_bytes"] + ga["write_bytes"]
ga["algorithmic_bytes_per_launch"] = 256 * 2 * 3 * 224 * 224 * 4
sys.path.insert(0, ".")
from equiadapt_amd import _lib
ga["kernel_source_sha1"] = _lib.source_hash("group_action.hip")   # bench.py flags the figure as stale when the kernel's source moves on
json.dump(ga, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(ga))
PY
