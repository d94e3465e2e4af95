set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ah
O=gpurun_out/r02ah
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python tools/ragged_bench.py --batch 64 --utterances 256 | tail -1
