set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=gpurun_out/r02e; mkdir -p $T
timeout -k 10 800 python -m pytest tests -m gpu -x -q --timeout 600 -s -k "not baseline_size_run" > $T/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $T/gpu_tests.log
grep -E "passed|failed|free-running|teacher|Error|assert" $T/gpu_tests.log | tail -12
B64="--batch 64 --frames 256 --steps 3 --warmup 1 --no-cpu-baseline"
P='import json,sys;j=json.load(open(sys.argv[1]));print(sys.argv[2], j["value"], j["decode_ms_per_frame_step"], j["stages"]["code_predictor"]["ms_per_step"], j["stages"]["talker_decode"]["ms_per_step"], j["stages"]["sampler"]["ms_per_step"])'
for i in 1 2; do
timeout -k 10 300 python bench.py $B64 > $T/b64_la4_$i.json 2> $T/b64_la4.err; python -c "$P" $T/b64_la4_$i.json la4_$i
Q3TTS_GEMM3_LA=2 timeout -k 10 300 python bench.py $B64 > $T/b64_la2_$i.json 2> $T/b64_la2.err; python -c "$P" $T/b64_la2_$i.json la2_$i
done
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b64 > $T/bench_b1.json 2> $T/bench_b1.err; python -c "$P" $T/bench_b1.json b1
