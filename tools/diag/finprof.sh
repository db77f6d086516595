# A/B of libeqa variants under the kernel trace: bash tools/diag/finprof.sh variant...   (base = the in-tree library)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = base ]; then L=""; else L="EQA_LIB=$GRAFT_REPO_ROOT/build_variants/libeqa_$v.so"; fi
  rm -rf /tmp/prof_$v
  env $L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  f=$(ls /tmp/prof_$v/*/*kernel_stats.csv | head -1)
  python3 - $f $v <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("finalize", "gemv", "inv_pipe")): print(sys.argv[2], r["Name"][:64], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done
