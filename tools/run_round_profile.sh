set -e
# Round-end evidence run on the GPU box: tests, bench lines, eager kernel trace summary, PMC traffic.  Heavy rocprof outputs are
# reduced to summaries and deleted (gpurun copies back at most 64 MiB).  Usage: bash tools/run_round_profile.sh [tag] [--skip-tests]
TAG=${1:-r01f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
# 1. full GPU test suite
if [ "$2" != "--skip-tests" ]; then timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -3 | tee gpurun_out/$TAG/gpu_tests.log; fi
# 2. headline bench (with cpu baseline) + b=64
timeout -k 10 600 python bench.py > gpurun_out/$TAG/bench_b1.json 2> gpurun_out/$TAG/bench_b1.err
cat gpurun_out/$TAG/bench_b1.json
timeout -k 10 600 python bench.py --batch 64 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_b64.json 2> gpurun_out/$TAG/bench_b64.err
cat gpurun_out/$TAG/bench_b64.json
timeout -k 10 600 python bench.py --batch 8 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_b8.json 2> gpurun_out/$TAG/bench_b8.err
cat gpurun_out/$TAG/bench_b8.json
timeout -k 10 600 python bench.py --model 1.7b --batch 8 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/bench_1p7b_b8.json 2> gpurun_out/$TAG/bench_1p7b_b8.err
cat gpurun_out/$TAG/bench_1p7b_b8.json
# 3. kernel trace (eager) of the b=1 workload, 1024 frames
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/trace_b1 -o b1 -- python bench.py --frames 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > gpurun_out/$TAG/trace_b1.log 2>&1
python tools/rocpd_summary.py gpurun_out/$TAG/trace_b1/b1_results.db 40 > gpurun_out/$TAG/decode_b1_f1024_eager_by_grid.txt
head -20 gpurun_out/$TAG/decode_b1_f1024_eager_by_grid.txt
rm -rf gpurun_out/$TAG/trace_b1
# 3a. the batched decode steps (b=8: k_gemv16 family, b=64: k_gemm2 + finish), 48 frames each
for BB in 8 64; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/trace_b$BB -o b -- python bench.py --batch $BB --frames 48 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > gpurun_out/$TAG/trace_b$BB.log 2>&1
  python tools/rocpd_summary.py gpurun_out/$TAG/trace_b$BB/b_results.db 30 > gpurun_out/$TAG/decode_b${BB}_f48_eager_by_grid.txt
  rm -rf gpurun_out/$TAG/trace_b$BB
done
# 3b. codec decoder alone, 2048 frames
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/trace_codec -o c -- python tools/codec_bench.py --reps 2 > gpurun_out/$TAG/codec_bench.log 2>&1
python tools/rocpd_summary.py gpurun_out/$TAG/trace_codec/c_results.db 45 > gpurun_out/$TAG/codec_f2048_by_grid.txt
rm -rf gpurun_out/$TAG/trace_codec
grep frames= gpurun_out/$TAG/codec_bench.log
# 4. PMC traffic, separate passes, 12 eager steps on the default stream
Q3TTS_NULL_STREAM=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/$TAG/pmc_fetch -o f --output-format csv -- tools/pmc_bisect 12 > gpurun_out/$TAG/pmc_fetch.log 2>&1
Q3TTS_NULL_STREAM=1 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/$TAG/pmc_write -o w --output-format csv -- tools/pmc_bisect 12 > gpurun_out/$TAG/pmc_write.log 2>&1
ls gpurun_out/$TAG/pmc_fetch gpurun_out/$TAG/pmc_write
python tools/pmc_traffic.py gpurun_out/$TAG/pmc_fetch/f_counter_collection.csv gpurun_out/$TAG/pmc_write/w_counter_collection.csv 12 1 > gpurun_out/$TAG/pmc_traffic_b1.json
cat gpurun_out/$TAG/pmc_traffic_b1.json
rm -rf gpurun_out/$TAG/pmc_fetch gpurun_out/$TAG/pmc_write
# 5. matrix-core issue rate of the codec decoder from hardware counters (SQ_INSTS_VALU_MFMA_MOPS_*: FLOP / 512), default stream
Q3TTS_NULL_STREAM=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 -d gpurun_out/$TAG/pmc_mfma -o m --output-format csv -- python tools/codec_bench.py --frames 2048 --reps 1 > gpurun_out/$TAG/pmc_mfma.log 2>&1
python tools/pmc_mfma.py gpurun_out/$TAG/pmc_mfma/m_counter_collection.csv 4096 > gpurun_out/$TAG/pmc_mfma_codec.json
cat gpurun_out/$TAG/pmc_mfma_codec.json
rm -rf gpurun_out/$TAG/pmc_mfma
