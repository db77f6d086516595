#!/bin/bash
# Round-3 evidence, collected on the GPU box into gpurun_out/r03/ (copied to profiles/r03/ afterwards):
#   bash tools/collect_r03.sh [bench|stats|pmc|traffic ...]      (default: all)
# rocprofv3 runs from /tmp with TMPDIR=/tmp; counters are collected in counter-only passes (no trace domains beside them).
out=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p "$out"
export TMPDIR=/tmp
what=${@:-bench stats pmc traffic}
cd "$GRAFT_REPO_ROOT"
stats() {  # stats <name> <top_n> <command...>: per-kernel time table of a command
  local name=$1 top=$2; shift 2
  rm -rf /tmp/st_$name
  # (the command is `python <script in the repo> args...`: run from /tmp with the script's absolute path)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o s -- "$1" "$GRAFT_REPO_ROOT/$2" "${@:3}" > "$out/stats_$name.log" 2>&1)
  local f=$(ls /tmp/st_$name/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$out/${name}_kernel_stats.csv" && python tools/stats_md.py "$f" "$top" > "$out/rocprofv3_kernel_stats_${name}.md"
}
for w in $what; do
  case $w in
    bench)
      python bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"
      ;;
    stats)
      stats bench 16 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline
      stats train_images_canonicalizer 24 python tools/bench_train.py --batch 256 --steps 5
      stats train_images_leg 24 python bench.py --mode train --train-steps 5 --train-warmup 2 --no-cpu-baseline
      stats train_pointcloud 20 python tools/bench_train_pc.py --steps 20
      stats cfg1 14 python tools/prof_cfg.py cfg1
      stats c4_64 16 python tools/prof_cfg.py c4_64
      stats cloud_k16 10 python tools/prof_cfg.py cloud_k16
      ;;
    pmc)
      # memory path of the fused FFT transforms inside the step (lines in flight per CU by Little's law)
      i=0
      for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"; do
        i=$((i+1))
        rm -rf /tmp/pmc_$i
        (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --mode forward --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
        python tools/pmc_summary.py /tmp/pmc_$i >> "$out/pmc_memory_path_raw.txt"
      done
      ;;
    traffic)
      bash tools/collect_traffic.sh "$out/traffic_ga" > "$out/traffic_group_action.log" 2>&1 && cp "$out/traffic_ga/traffic.json" "$out/traffic_group_action.json"
      bash tools/collect_traffic_net.sh "$out/traffic_net" > "$out/traffic_canon_net.log" 2>&1
      ;;
  esac
done
ls -la "$out" | tail -40
