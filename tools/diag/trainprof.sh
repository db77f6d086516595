# kernel averages of the canonicalizer's training step at B = 256: bash tools/diag/trainprof.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --batch 256 --steps 5 2>&1 | grep "training step"
f=$(ls /tmp/prof_tr/*/*kernel_stats.csv | head -1)
python3 - $f <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]: print(r["Name"][:72], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
