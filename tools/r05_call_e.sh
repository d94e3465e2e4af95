set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05_e
O=gpurun_out/r05_e
timeout -k 10 900 python -m pytest tests/test_gpu_b64.py tests/test_gpu_batch.py tests/test_gpu_dev_api.py tests/test_gpu_wide.py -m gpu -x -q > $O/tests_subset.log 2>&1 || { tail -30 $O/tests_subset.log; exit 1; }
tail -3 $O/tests_subset.log
for i in 1 2; do
bash tools/gpu_ab.sh r05_e b64 Q3TTS_ATTN_TINY2=0
bash tools/gpu_ab.sh r05_e b64 Q3TTS_DUMMY=1
done
bash tools/gpu_ab.sh r05_e b8 Q3TTS_ATTN_TINY2=0
bash tools/gpu_ab.sh r05_e b8 Q3TTS_DUMMY=1
