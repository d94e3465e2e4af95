set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02al
O=gpurun_out/r02al
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
for K in 0 1; do
  if [ $K = 1 ]; then export Q3TTS_NO_ATTN_TINY=1; fi
  python bench.py --batch 64 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_b64_$K.json 2> $O/bench_b64_$K.err
  python bench.py --batch 8 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_b8_$K.json 2> $O/bench_b8_$K.err
  python -c "import json;a=json.load(open('$O/bench_b64_$K.json'));b=json.load(open('$O/bench_b8_$K.json'));print('no_tiny=$K b64', a['value'], a['decode_ms_per_frame_step'], 'b8', b['value'], b['decode_ms_per_frame_step'])"
done
