#!/bin/bash
# SQ counter table of a command's kernels (one counter-only rocprofv3 pass of eight SQ counters; MI355X_MICROARCH.md: WAIT_ANY +
# WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles):   bash tools/collect_sq_counters.sh <outdir> <name filter> python <script> ...
set -e
out=$1; filt=$2; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS \
  -d "$out/sq" -o p --output-format csv -- "$@" > "$out/run.log" 2>&1
python - "$out" "$filt" <<'PY'
import csv, sys, collections, re
out, filt = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{out}/sq/p_counter_collection.csv")):
    if filt in r["Kernel_Name"]:
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"([A-Za-z_][A-Za-z_0-9]*(?:<[^()]*>)?)\(", nm)
        key = (m.group(1) if m else nm[:60], r.get("Grid_Size", ""))
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS"]
print("| kernel (grid size) | launches | " + " | ".join(names) + " | WAIT_ANY / WAVE_CYCLES | WAIT_INST_ANY / | ACTIVE_INST_ANY / | VALU instr. per wave |")
print("|---" * (len(names) + 6) + "|")
for (k, g), c in sorted(per.items()):
    m = {n: sum(c[n]) / max(1, len(c[n])) for n in names}
    wc = m["SQ_WAVE_CYCLES"] or 1.0
    print(f"| `{k}` ({g}) | {len(c['SQ_WAVES'])} | " + " | ".join(f"{m[n]:.4g}" for n in names) +
          f" | {m['SQ_WAIT_ANY'] / wc:.2f} | {m['SQ_WAIT_INST_ANY'] / wc:.2f} | {m['SQ_ACTIVE_INST_ANY'] / wc:.2f} | {m['SQ_INSTS_VALU'] / max(1.0, m['SQ_WAVES']):.0f} |")
PY
