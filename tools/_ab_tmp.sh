out=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/bench_tutorial.py --leg optimized --steps 20 --warmup 5 > $out/tut_opt_now.json 2>/dev/null; cat $out/tut_opt_now.json
EQA_CONVNET_TRAIN_MODE=plain python tools/bench_tutorial.py --leg optimized --steps 20 --warmup 5 2>/dev/null
export TMPDIR=/tmp
rm -rf /tmp/st_x; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_x -o s -- python $GRAFT_REPO_ROOT/tools/bench_tutorial.py --leg optimized --steps 5 --warmup 3 > $out/stats_tut.log 2>&1)
f=$(ls /tmp/st_x/*kernel_stats.csv | head -1); python tools/stats_md.py $f 24 > $out/rocprofv3_kernel_stats_tutorial_optimized_a.md; cat $out/rocprofv3_kernel_stats_tutorial_optimized_a.md
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_full.log 2>&1; tail -3 $out/pytest_full.log
