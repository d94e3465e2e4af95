set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02m
O=gpurun_out/r02m
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
rocprofv3 --kernel-trace --stats -d $O/trace_b64 -o b -- python bench.py --batch 64 --frames 48 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $O/trace_b64.log 2>&1
python tools/rocpd_summary.py $O/trace_b64/b_results.db 30 > $O/decode_b64_f48_eager_by_grid.txt
rm -rf $O/trace_b64
grep -E "k_attn|k_finish" $O/decode_b64_f48_eager_by_grid.txt | cut -c1-120
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "import json;j=json.load(open('$O/bench.json'));print('b1', j['value'], j['decode_ms_per_frame_step'], '| b64', j['b64']['value'], j['b64']['decode_ms_per_frame_step'])"
