set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05_b
O=gpurun_out/r05_b
timeout -k 10 560 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json;j=json.load(open('$O/bench_default.json'))
print('b1', j['value'], j['decode_ms_per_frame_step'], j['roofline']['frac'])
for k in ('b64','b8','b64_f2048','b64_f2048_kv_bf16','b1_f2048_kv_bf16','b8_1p7b_clone','capacity'):
    r=j.get(k) or {}
    print(k, r.get('value'), r.get('decode_ms_per_frame_step'), r.get('error'), r.get('clone_front_end_ms_per_utterance'), r.get('wall_ms_per_step_of_all'))
print('prefill', j['stages'].get('prefill'))
print('b64 prefill', j['b64']['stages'].get('prefill'))
print('cpu', j['cpu_baseline'].get('value'), j['cpu_baseline'].get('threads4'))
"
Q3TTS_LIB=$PWD/tools/exp/libprof.so timeout -k 10 200 python tools/seam_phases.py > $O/seam_phases.txt 2>&1
tail -40 $O/seam_phases.txt
for F in 2048 256; do timeout -k 10 200 python tools/codec_bench.py --frames $F --reps 3 | grep frames= >> $O/codec_sizes.txt; done
cat $O/codec_sizes.txt
Q3TTS_LIB=$PWD/tools/exp/libspill.so timeout -k 10 500 python -m pytest tests/test_gpu_codec_stress.py -m gpu -q -s > $O/stress_spill.log 2>&1 || true
tail -8 $O/stress_spill.log
