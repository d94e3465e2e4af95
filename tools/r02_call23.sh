set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02aq
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:randomly > gpurun_out/r02aq/gpu_tests_again.log 2>&1 || { tail -40 gpurun_out/r02aq/gpu_tests_again.log; exit 1; }
tail -2 gpurun_out/r02aq/gpu_tests_again.log
