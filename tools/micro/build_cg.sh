#!/bin/bash
# Builds tools/micro/cgemm3m_bench.hip against csrc/cgemm3m.hip once per experiment variant into build_variants/cg_<variant>
# (scratch, git-ignored; the binaries travel to the GPU box with gpurun).  usage: tools/micro/build_cg.sh [variant ...]
cd "$(dirname "$0")/../.." && mkdir -p build_variants
for v in "${@:-base}"; do
  d=""; [ "$v" != base ] && d="-DEQA_CGEMM_$v"
  # one translation unit (the harness includes the kernel source) so that the experiment's device symbols are visible to it
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -w -I include -I equiadapt_amd/csrc $d -DEQA_CG_INCLUDE_KERNEL \
      tools/micro/cgemm3m_bench.hip -o build_variants/cg_$v || echo "FAILED $v"
done
