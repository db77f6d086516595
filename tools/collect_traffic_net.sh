#!/bin/bash
# HBM traffic of the canonicalization network's hand-written kernels inside bench.py's step (B=256), collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, counters only.
# Run on the GPU box:  bash tools/collect_traffic_net.sh <outdir>
set -e
out=${1:-gpurun_out/traffic_net}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d "$out/$c" -o p --output-format csv -- python bench.py --mode forward --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
done
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
B = 256
alg = {  # algorithmic bytes per launch, fp32
    "winograd_k5_input_kernel": B * 92 * 92 * 256 * 4 + B * 484 * 64 * 256 * 4,
    "winograd_k5_output_sums_kernel": B * 484 * 64 * 256 * 4,
    "fft48_fwd_fused_kernel": B * 92 * 92 * 256 * 4 + 1154 * B * 4 * 512 * 4,
    "fft48_inv_fused_kernel": 1154 * B * 4 * 512 * 4 + B * 10 * 2 * 256 * 9 * 4,
    # the pipelined form writes one interior piece per consumer wave: 8 border rows + 5 x 2 tile rows, x 2 tile columns
    "fft48_inv_pipe_kernel": 1154 * B * 4 * 512 * 4 + B * 18 * 2 * 256 * 9 * 4,
    "lift_conv_mfma_kernel": B * 96 * 96 * 3 * 4 + B * 92 * 92 * 256 * 4,
    "lift_conv_dense_kernel": B * 96 * 96 * 3 * 4 + B * 92 * 92 * 256 * 4,
    "crop_resize_aa_kernel": B * 3 * 180 * 180 * 4 + B * 3 * 96 * 96 * 4,
    "crop_resize_aa_staged_kernel": B * 3 * 180 * 180 * 4 + B * 3 * 96 * 96 * 4,
    "window_sums_nhwc_finalize_kernel": B * 18 * 2 * 256 * 9 * 4 + B * 256 * 25 * 8,  # 8 border rows + 5 x 2 interior pieces, x 2 tile columns
    "group_action_kernel": B * 2 * 3 * 224 * 224 * 4,
    # V in + B3 (F x 256 x 256 x 3 floats) in + Mo out; the filter spectra are re-read from L2 by a frequency's 16 row tiles
    "fft_cgemm3m_kernel": 2 * 1154 * (B * 4 | 1) * 512 * 4 + 1154 * 256 * 256 * 3 * 4,
    # round 6: the six-product piece form (default at the headline shape): V in + the pre-split filter spectra (3 bf16 pieces per value
    # = 1.5 x the fp32 bytes) + Mo out
    "fft_cgemm3m_bf16_block_kernel<6>": 2 * 1154 * (B * 4 | 1) * 512 * 4 + 1154 * 256 * 256 * 3 * 3 * 2,
    # the fp16 form (the step's default since the second half of round 6): V in + the filter spectra as two fp16 pieces per value + Mo out
    "fft_cgemm3m_bf16_block_kernel<3>": 2 * 1154 * (B * 4 | 1) * 512 * 4 + 1154 * 256 * 256 * 3 * 2 * 2,
    # round 6: lifting convolution fused into the forward transform: the 96 x 96 x 3 input in (re-read by the 16 channel groups of a
    # tile out of L2), the spectra out -- the lifted map is never written
    "lift5_fft48_fused_kernel": B * 96 * 96 * 3 * 4 + 1154 * B * 4 * 512 * 4,
    "lift5_fft48_fused_h2_kernel": B * 96 * 96 * 3 * 4 + 1154 * B * 4 * 512 * 4,      # the same kernel with its convolution on two fp16 pieces
}
res = {k: {"algorithmic_bytes_per_launch": v} for k, v in alg.items()}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/{name}/p_counter_collection.csv")):
        if r["Counter_Name"] == name:
            for k in alg:
                if k in r["Kernel_Name"]:
                    per[k].append(float(r["Counter_Value"]))
    for k, v in per.items():
        res[k][name + "_KB_per_launch"] = sum(v) / len(v)
for k, v in res.items():
    if "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
        # gfx950: FETCH_SIZE tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled
        v["read_bytes"] = v["FETCH_SIZE_KB_per_launch"] * 1024 * 2
        v["write_bytes"] = v["WRITE_SIZE_KB_per_launch"] * 1024
        v["traffic_bytes_per_launch"] = v["read_bytes"] + v["write_bytes"]
        v["traffic_over_algorithmic"] = v["traffic_bytes_per_launch"] / v["algorithmic_bytes_per_launch"]
json.dump(res, open(f"{out}/traffic_net.json", "w"), indent=1)
for k, v in res.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
