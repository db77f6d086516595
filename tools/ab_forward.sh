#!/bin/bash
# usage: ab.sh variant... ; prints ms_per_step and stage ms (each variant twice, interleaved)
for rep in 1 2; do for v in "$@"; do
  if [ $v = base ]; then L=""; else L="EQA_LIB=$PWD/build_variants/libeqa_$v.so"; fi
  env $L python bench.py --mode forward --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']; print('$v', round(d['ms_per_step'],3), {k:round(s[k]['ms'],4) for k in ('lift_conv','fft_input','fft_gemm','fft_output_sums')})"
done; done
