set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02k
O=gpurun_out/r02k
timeout -k 10 300 python -m pytest tests/test_gpu_kvpool.py -x -q --timeout 280 -s > $O/kvpool.log 2>&1 || { tail -40 $O/kvpool.log; exit 1; }
tail -5 $O/kvpool.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "import json;j=json.load(open('$O/bench.json'));print('b1', j['value'], j['decode_ms_per_frame_step'], '| b64', j['b64']['value'], j['b64']['decode_ms_per_frame_step'])"
