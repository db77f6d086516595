#!/bin/bash
# Round-4 evidence, collected on the GPU box into gpurun_out/r05/ (copied to profiles/r05/ afterwards):
#   bash tools/collect_r05.sh [bench|stats|traffic|kbench ...]      (default: all)
# rocprofv3 runs from /tmp with TMPDIR=/tmp; counters are collected in counter-only passes (no trace domains beside them).
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
export TMPDIR=/tmp
what=${@:-bench stats traffic kbench}
cd "$GRAFT_REPO_ROOT"
stats() {  # stats <name> <top_n> <command...>: per-kernel time table of a command
  local name=$1 top=$2; shift 2
  rm -rf /tmp/st_$name
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
      stats step 10 python bench.py --mode forward --steps 20 --warmup 10 --step-only
      python bench.py --mode forward --steps 20 --warmup 10 --step-only > "$out/bench_step_only.json" 2>/dev/null
      stats tutorial_prior 28 python tools/bench_tutorial.py --leg prior --steps 5 --warmup 3
      stats tutorial_optimized 20 python tools/bench_tutorial.py --leg optimized --steps 5 --warmup 3
      stats train_images_leg 20 python bench.py --mode train --train-steps 5 --train-warmup 2 --no-cpu-baseline
      ;;
    traffic)
      bash tools/collect_traffic.sh "$out/traffic_ga" > "$out/traffic_group_action.log" 2>&1 && cp "$out/traffic_ga/traffic.json" "$out/traffic_group_action.json"
      bash tools/collect_traffic_net.sh "$out/traffic_net" > "$out/traffic_canon_net.log" 2>&1 && cp "$out/traffic_net/traffic_net.json" "$out/traffic_canon_net.json"
      rm -rf "$out/traffic_ga" "$out/traffic_net"
      ;;
    kbench)
      python tools/kbench.py --reps 60 > "$out/kbench.txt" 2>&1
      python tools/kbench_aa.py > "$out/kbench_crop_resize.txt" 2>&1
      python tools/kbench_angle.py > "$out/kbench_angle_gradient.txt" 2>&1
      python tools/bench_small_batches.py > "$out/small_batches.txt" 2>&1
      python tools/kbench_vn.py > "$out/kbench_vnsmall.txt" 2>&1
      ;;
  esac
done
ls -la "$out" | tail -40
