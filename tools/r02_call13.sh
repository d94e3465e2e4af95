set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02y
O=gpurun_out/r02y
timeout -k 10 600 python -m pytest tests/test_gpu_codec.py -x -q --timeout 500 -k "ab_switches" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
