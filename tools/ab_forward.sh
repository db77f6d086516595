#!/bin/bash
# A/B of forward-step variants on one box: bash tools/ab_forward.sh variant...   (variant = "base", a build_variants/libeqa_<v>.so name,
# or ENV=VALUE:variant to set an environment switch for that run).  Prints ms_per_step and the stage times, every variant twice, interleaved.
for rep in 1 2; do for spec in "$@"; do
  envs=""; v=$spec
  if [[ $spec == *:* ]]; then envs=${spec%%:*}; v=${spec##*:}; fi
  if [ $v = base ]; then L=""; else L="EQA_LIB=$PWD/build_variants/libeqa_$v.so"; fi
  env $L $envs python bench.py --mode forward --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']; print('$spec', round(d['ms_per_step'],3), {k:round(s[k]['ms'],4) for k in ('lift_conv','fft_input','fft_gemm','fft_output_sums')})"
done; done
