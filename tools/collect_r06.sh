#!/bin/bash
# Round-6 evidence, collected on the GPU box into gpurun_out/r06c/ (copied to profiles/r06/ afterwards):
#   bash tools/collect_r06.sh [bench|stats|traffic|kbench|rccl1 ...]      (default: all)
# rocprofv3 runs from /tmp with TMPDIR=/tmp; counters are collected in counter-only passes (no trace domains beside them).
out=$GRAFT_REPO_ROOT/gpurun_out/r06c
mkdir -p "$out"
export TMPDIR=/tmp
what=${@:-bench stats traffic kbench rccl1}
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
      stats step 10 python bench.py --mode forward --steps 20 --warmup 10 --step-only
      python bench.py --mode forward --steps 20 --warmup 10 --step-only > "$out/bench_step_only.json" 2>/dev/null
      stats bench 16 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline
      EQA_LIFT_FFT_FUSED=0 EQA_LIFT_FFT_FORM=f32 EQA_FFT_GEMM_PIECES=f32 python bench.py --mode forward --steps 20 --warmup 10 --step-only > "$out/bench_step_only_round5_forms.json" 2>/dev/null
      EQA_LIFT_FFT_FORM=f32 EQA_FFT_GEMM_PIECES=6 python bench.py --mode forward --steps 20 --warmup 10 --step-only > "$out/bench_step_only_six_bf16_products.json" 2>/dev/null
      EQA_LIFT_FFT_FORM=f32 python bench.py --mode forward --steps 20 --warmup 10 --step-only > "$out/bench_step_only_fp32_lifting.json" 2>/dev/null
      ;;
    traffic)
      bash tools/collect_traffic.sh "$out/traffic_ga" > "$out/traffic_group_action.log" 2>&1 && cp "$out/traffic_ga/traffic.json" "$out/traffic_group_action.json"
      bash tools/collect_traffic_net.sh "$out/traffic_net" > "$out/traffic_canon_net.log" 2>&1 && cp "$out/traffic_net/traffic_net.json" "$out/traffic_canon_net.json"
      rm -rf "$out/traffic_ga" "$out/traffic_net"
      ;;
    kbench)
      python tools/kbench.py --reps 60 > "$out/kbench.txt" 2>&1
      python tools/kbench_lift_fft.py > "$out/kbench_lift_fft.txt" 2>&1
      [ -f build_variants/libeqa_lfclock.so ] && EQA_LIB=$PWD/build_variants/libeqa_lfclock.so EQA_LIFT_FFT_FORM=f32 python tools/kbench_lift_fft.py >> "$out/kbench_lift_fft.txt" 2>&1
      [ -f build_variants/libeqa_lfclock.so ] && EQA_LIB=$PWD/build_variants/libeqa_lfclock.so EQA_LIFT_FFT_FORM=h2 python tools/kbench_lift_fft.py 2>&1 | tail -n 3 >> "$out/kbench_lift_fft.txt"
      [ -x tools/micro/_bin/mfma16_chains ] && tools/micro/_bin/mfma16_chains > "$out/mfma16_chains.txt" 2>&1
      [ -x tools/micro/_bin/store_pattern ] && tools/micro/_bin/store_pattern > "$out/store_pattern.txt" 2>&1
      [ -x tools/micro/_bin/permlane_swap ] && tools/micro/_bin/permlane_swap > "$out/permlane_swap.txt" 2>&1
      python tools/kbench_gemm_error.py > "$out/kbench_gemm_error.txt" 2>&1
      python tools/kbench_gemm_pieces.py > "$out/kbench_gemm_pieces.txt" 2>&1
      python tools/kbench_invert_c1.py > "$out/kbench_invert_c1.txt" 2>&1
      python tools/kbench_invert_c1.py --batch 4 >> "$out/kbench_invert_c1.txt" 2>&1
      [ -x tools/micro/_bin/f16x2_gemm_check ] && tools/micro/_bin/f16x2_gemm_check > "$out/f16x2_gemm_check.txt" 2>&1
      for v in 1 2; do [ -f build_variants/libeqa_clock$v.so ] && EQA_LIB=$PWD/build_variants/libeqa_clock$v.so python tools/kbench_gemm_clock.py; done > "$out/kbench_gemm_clock.txt" 2>&1
      python tools/host_time_cfg5.py 2>&1 | head -4 > "$out/host_time_cfg5.txt"
      python tools/bench_small_batches.py > "$out/small_batches.txt" 2>&1
      ;;
    rccl1)
      # the same forward bench bare and as a 1-rank RCCL job under the launcher the driver uses: `value` must agree within 2 %
      # (two rounds, alternating: the chip's clocks differ by 2-3 % from run to run)
      for i in 1 2; do
        python bench.py --mode forward --no-cpu-baseline > "$out/bench_forward_bare_$i.json" 2>/dev/null
        python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2957$i bench.py --gpus 1 --mode forward --no-cpu-baseline > "$out/bench_forward_rccl_world1_$i.json" 2> "$out/bench_forward_rccl_world1_$i.err"
      done
      EQA_BENCH_BACKEND=gloo python bench.py --gpus 4 --mode forward --steps 5 --warmup 2 --batch 64 --check-images 16 > "$out/bench_gloo_world4_one_gpu.json" 2>/dev/null
      python - "$out" <<'PY'
import json, sys
o = sys.argv[1]
rows = []
for i in (1, 2):
    a = json.loads(open(o + f"/bench_forward_bare_{i}.json").read().strip().splitlines()[-1])
    b = json.loads(open(o + f"/bench_forward_rccl_world1_{i}.json").read().strip().splitlines()[-1])
    rows.append(f"round {i}: bare {a['value']:.0f} img/s (rccl_ranks {a['rccl_ranks']}); torch.distributed.run --nproc-per-node 1: {b['value']:.0f} img/s "
                f"(rccl_ranks {b['rccl_ranks']}, backend {b['backend']}); ratio {b['value'] / a['value']:.4f}")
open(o + "/rccl_world1_vs_bare.txt", "w").write("\n".join(rows) + "\n")
PY
      cat "$out/rccl_world1_vs_bare.txt"
      ;;
  esac
done
ls -la "$out" | tail -40
