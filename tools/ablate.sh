#!/bin/bash
# Ablation of group_action_kernel (results: DESIGN.md section 3.1).  Builds four variants into build_variants/ (CPU, hipcc
# cross-compiles) -- run this part in the build container; then, on the GPU box:
#   for v in base noload nostore neither; do EQA_LIB=$PWD/build_variants/libeqa_$v.so python tools/kbench.py | grep canon; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
for v in "base:" "noload:-DEQA_ABL_NOLOAD" "nostore:-DEQA_ABL_NOSTORE" "neither:-DEQA_ABL_NOLOAD -DEQA_ABL_NOSTORE"; do
  n=${v%%:*}; f=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $f equiadapt_amd/csrc/eqa_hip.hip -o build_variants/libeqa_$n.so
done
ls build_variants
