set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ag
O=gpurun_out/r02ag
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_full.py tests/test_gpu_b64.py -x -q --timeout 600 > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-b64 > $O/bench_$i.json 2> $O/bench_$i.err; python -c "import json;j=json.load(open('$O/bench_$i.json'));print('b1', j['value'], j['decode_ms_per_frame_step'], j['stages']['talker_decode']['ms_per_step'])"; done
