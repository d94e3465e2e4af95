set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=gpurun_out/r02a; mkdir -p $T
timeout -k 10 700 python -m pytest tests -m gpu -x -q --timeout 600 -s -k "not baseline_size_run" > $T/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $T/gpu_tests.log
tail -5 $T/gpu_tests.log
timeout -k 10 200 python tools/sampler_diag.py 300 > $T/sampler_diag.json 2> $T/sampler_diag.err; tail -12 $T/sampler_diag.err
timeout -k 10 120 tools/mb_grid_barrier > $T/microbench_grid_barrier.txt 2>&1; cat $T/microbench_grid_barrier.txt
timeout -k 10 400 python bench.py > $T/bench_default.json 2> $T/bench_default.err; echo "bench rc=$?"; head -c 1500 $T/bench_default.json
