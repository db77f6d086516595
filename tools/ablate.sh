#!/bin/bash
# Ablation builds behind the numbers in HISTORY.md.  Builds variants of libeqa_hip.so into build_variants/ (hipcc
# cross-compiles without a GPU; build_variants/ is git-ignored but travels to the GPU box).  Then, on the GPU box:
#   group-action kernel:
#     for v in base noload nostore neither; do EQA_LIB=$PWD/build_variants/libeqa_$v.so python tools/kbench.py | grep canon; done
#   Winograd input transform:
#     for v in base wnoload wnostore; do EQA_LIB=$PWD/build_variants/libeqa_$v.so python tools/kbench_wino.py --m 2 --n 64 | grep input; done
#   lifting convolution:
#     for v in base lnoload lnostore lneither; do EQA_LIB=$PWD/build_variants/libeqa_$v.so python tools/kbench_wino.py --lift | grep MFMA; done
#   VNSmall occupancy:
#     for v in vn2 vn3 vn4; do EQA_LIB=$PWD/build_variants/libeqa_$v.so python tools/kbench_pc.py | grep B=2048; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
build() { hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC -Iinclude $2 equiadapt_amd/csrc/*.hip -o build_variants/libeqa_$1.so; }
build base "" &
build noload "-DEQA_ABL_NOLOAD" &
build nostore "-DEQA_ABL_NOSTORE" &
build neither "-DEQA_ABL_NOLOAD -DEQA_ABL_NOSTORE" &
wait
build wnoload "-DEQA_WABL_NOLOAD" &
build wnostore "-DEQA_WABL_NOSTORE" &
build lnoload "-DEQA_LABL_NOLOAD" &
build lnostore "-DEQA_LABL_NOSTORE" &
wait
build lneither "-DEQA_LABL_NOLOAD -DEQA_LABL_NOSTORE" &
build vn2 "-DEQA_VN_MIN_BLOCKS=2" &
build vn3 "-DEQA_VN_MIN_BLOCKS=3" &
build vn4 "-DEQA_VN_MIN_BLOCKS=4" &
wait
ls build_variants
