set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=gpurun_out/r02b; mkdir -p $T
timeout -k 10 800 python -m pytest tests -m gpu -x -q --timeout 600 -s -k "not baseline_size_run" > $T/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $T/gpu_tests.log
grep -E "passed|failed|free-running|teacher" $T/gpu_tests.log | tail -12
timeout -k 10 300 python bench.py --batch 64 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > $T/bench_b64_gemm3.json 2> $T/bench_b64_gemm3.err; echo "rc=$?"
python -c "import json;j=json.load(open('$T/bench_b64_gemm3.json'));print('gemm3 b64', j['value'], j['decode_ms_per_frame_step'], j['stages']['code_predictor']['ms_per_step'], j['stages']['talker_decode']['ms_per_step'])"
Q3TTS_GEMM2=1 timeout -k 10 300 python bench.py --batch 64 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > $T/bench_b64_gemm2.json 2> $T/bench_b64_gemm2.err; echo "rc=$?"
python -c "import json;j=json.load(open('$T/bench_b64_gemm2.json'));print('gemm2 b64', j['value'], j['decode_ms_per_frame_step'], j['stages']['code_predictor']['ms_per_step'], j['stages']['talker_decode']['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b64 > $T/bench_b1.json 2> $T/bench_b1.err; echo "rc=$?"
python -c "import json;j=json.load(open('$T/bench_b1.json'));print('b1', j['value'], j['decode_ms_per_frame_step'], j['stages']['sampler'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $T/trace_b64 -o b -- python bench.py --batch 64 --frames 48 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $T/trace_b64.log 2>&1
python tools/rocpd_summary.py $T/trace_b64/b_results.db 30 > $T/decode_b64_f48_eager_by_grid.txt; rm -rf $T/trace_b64
head -24 $T/decode_b64_f48_eager_by_grid.txt
