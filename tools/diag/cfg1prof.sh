cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -- python $GRAFT_REPO_ROOT/tools/prof_cfg.py cfg1 > /dev/null 2>&1
f=$(ls /tmp/prof_c1/*/*kernel_stats.csv | head -1)
python3 - $f <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
