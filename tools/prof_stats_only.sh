cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_g/stats -o stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r04_g/bench_under_rocprof.json 2>/dev/null
ls $GRAFT_REPO_ROOT/gpurun_out/prof_r04_g/stats/*/ | head
