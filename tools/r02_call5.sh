set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=gpurun_out/r02h; mkdir -p $T
timeout -k 10 600 python -m pytest tests/test_gpu_codec.py tests/test_gpu_full.py tests/test_gpu_cli.py tests/test_gpu_batch.py -m gpu -x -q --timeout 600 -s -k "codec or vocoder or chunked or cli or batch12 or continuous" > $T/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $T/gpu_tests.log
grep -E "passed|failed|codec rms|Error|assert" $T/gpu_tests.log | tail -8
for FF in 64 256 2048; do python tools/codec_bench.py --frames $FF --reps 3 | grep frames= | tee -a $T/codec_fused.txt; done
for FF in 64 256 2048; do Q3TTS_NO_FUSED_RES=1 python tools/codec_bench.py --frames $FF --reps 3 | grep frames= | tee -a $T/codec_unfused.txt; done
