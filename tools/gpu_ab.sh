set -e
# One parameterised GPU-box script for A/B runs of the decode step and the codec decoder (replaces the per-call r02_call*.sh records).
#   bash tools/gpu_ab.sh <tag> <mode> [args]
# modes:  tests              full `-m gpu` suite
#         b64 [ENV=1 ...]    bench.py --batch 64 --frames 256 with the given environment knobs set
#         b1  [ENV=1 ...]    bench.py --batch 1  --frames 512
#         codec [ENV=1 ...]  tools/codec_bench.py at F = 2048 / 256 / 64
#         ctx [ENV=1 ...]    tools/ctx_bench.py (decode step at CTX tokens of context, both KV dtypes, per-stage split)
#         default            the default bench line
# Every mode appends its result lines to gpurun_out/<tag>/ab.txt; each GPU step runs under its own `timeout -k`.
TAG=$1; MODE=$2; shift 2 || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
KNOBS="$*"
pick() { python -c "import json,sys;j=json.load(open(sys.argv[1]));print(sys.argv[2], 'RTF', j['value'], 'step_ms', j['decode_ms_per_frame_step'], 'codec_ms/frame', j['codec_decode_ms_per_frame'], 'stages', {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in (j.get('stages') or {}).items()})" "$1" "$2"; }
case $MODE in
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q --timeout 600 > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
  tail -3 $O/gpu_tests.log ;;
b64)
  env $KNOBS timeout -k 10 300 python bench.py ${KNOBS:+--hooks} --batch 64 --frames 256 --steps 3 --warmup 1 --no-cpu-baseline > $O/b64_$$.json 2> $O/b64_$$.err
  pick $O/b64_$$.json "b64 [$KNOBS]" | tee -a $O/ab.txt ;;
b8)
  env $KNOBS timeout -k 10 300 python bench.py ${KNOBS:+--hooks} --batch 8 --frames 256 --steps 3 --warmup 1 --no-cpu-baseline > $O/b8_$$.json 2> $O/b8_$$.err
  pick $O/b8_$$.json "b8 [$KNOBS]" | tee -a $O/ab.txt ;;
b1)
  env $KNOBS timeout -k 10 300 python bench.py ${KNOBS:+--hooks} --batch 1 --frames 512 --steps 3 --warmup 1 --no-cpu-baseline --no-b64 > $O/b1_$$.json 2> $O/b1_$$.err
  pick $O/b1_$$.json "b1 [$KNOBS]" | tee -a $O/ab.txt ;;
codec)
  for F in 2048 256 64; do
    echo -n "codec [$KNOBS] " | tee -a $O/ab.txt
    env $KNOBS timeout -k 10 200 python tools/codec_bench.py --frames $F --reps 5 | tee -a $O/ab.txt
  done ;;
ctx)
  # decode step at a long talker context (tools/ctx_bench.py): CTX / BATCH / KVS from the environment, knobs as arguments
  for KV in ${KVS:-fp32 bf16}; do
    echo -n "ctx [$KNOBS] " | tee -a $O/ab.txt
    env $KNOBS timeout -k 10 300 python tools/ctx_bench.py --batch ${BATCH:-64} --ctx ${CTX:-1024} --kv $KV --stages --max-ctx ${MAXCTX:-2112} | tee -a $O/ab.txt
  done ;;
default)
  timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
  python -c "import json;j=json.load(open('$O/bench_default.json'));print('b1', j['value'], j['decode_ms_per_frame_step'], j['roofline']['frac'], *[(k, j[k].get('value'), j[k].get('decode_ms_per_frame_step')) for k in ('b64','b8','b64_f2048') if k in j], '| cpu', j.get('cpu_baseline',{}).get('value'))" | tee -a $O/ab.txt ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
