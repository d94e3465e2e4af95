set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05_h
O=gpurun_out/r05_h
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_dev_api.py tests/test_gpu_batch.py -m gpu -x -q > $O/tests_sampler.log 2>&1 || { tail -30 $O/tests_sampler.log; exit 1; }
tail -3 $O/tests_sampler.log
Q3TTS_LIB=$PWD/tools/exp/libprof.so timeout -k 10 200 python tools/kernel_phases.py > $O/kernel_phases.txt 2>&1
cat $O/kernel_phases.txt
timeout -k 10 900 python -m pytest tests/test_gpu_full.py -m gpu -x -q -k "free_running or greedy or sampled" > $O/tests_full_subset.log 2>&1 || { tail -30 $O/tests_full_subset.log; exit 1; }
tail -3 $O/tests_full_subset.log
