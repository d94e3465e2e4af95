set -e
# Round-end evidence run on the GPU box: tests, bench lines, eager kernel trace summaries, PMC traffic.  Heavy rocprof outputs are
# reduced to summaries and deleted (gpurun copies back at most 64 MiB).  Usage: bash tools/run_round_profile.sh [tag] [--skip-tests]
# Modes (second argument): main = tests, bench lines, kernel traces, codec; pmc = the rocprofv3 --pmc passes + the default bench line that quotes
# them.  The counter passes run LAST and in a call of their own: rocprofv3 --pmc has hung on this image (a WRITE_SIZE pass over 24 eager
# steps sat in hipStreamSynchronize until its timeout), and after a killed GPU step nothing else should run in the same call.
TAG=${1:-r03}
MODE=${2:-main}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
if [ "$MODE" = "pmc" ]; then
# PMC traffic, separate passes, eager steps on the default stream; two step counts so that the prefill cancels in the difference.
# Configurations: batch:context:kv, one per record of the bench line at that record's own mean context (8 + frames / 2); slots moved there with
# q3tts_measure_skip_frames (tools/pmc_bisect <steps> <batch> <ctx> <bf16>)
for CFG in ${PMC_CFGS:-1:1032:fp32:0.6b 64:136:fp32:0.6b 8:136:fp32:0.6b 64:1032:fp32:0.6b 64:1032:bf16:0.6b 1:1032:bf16:0.6b 8:136:fp32:1.7b}; do
  IFS=: read B CTX KV MODEL <<< "$CFG"; MODEL=${MODEL:-0.6b}
  BF=0; [ "$KV" = "bf16" ] && BF=1
  if [ $B = 1 ]; then S1=6; S2=12; else S1=4; S2=8; fi
  T=${B}_${CTX}_${KV}_${MODEL}
  for S in $S1 $S2; do
    Q3TTS_NULL_STREAM=1 timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f_${T}_$S -o f --output-format csv -- tools/pmc_bisect $S $B $CTX $BF $MODEL > $O/pmc_fetch.log 2>&1
    Q3TTS_NULL_STREAM=1 timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w_${T}_$S -o w --output-format csv -- tools/pmc_bisect $S $B $CTX $BF $MODEL > $O/pmc_write.log 2>&1
  done
  python tools/pmc_traffic.py --batch $B --ctx $CTX --kv $KV --model $MODEL --fetch $(ls $O/pmc_f_${T}_$S1/*/f_counter_collection.csv $O/pmc_f_${T}_$S1/f_counter_collection.csv 2>/dev/null | head -1) --write $(ls $O/pmc_w_${T}_$S1/*/w_counter_collection.csv $O/pmc_w_${T}_$S1/w_counter_collection.csv 2>/dev/null | head -1) --frames $S1 \
      --fetch2 $(ls $O/pmc_f_${T}_$S2/*/f_counter_collection.csv $O/pmc_f_${T}_$S2/f_counter_collection.csv 2>/dev/null | head -1) --write2 $(ls $O/pmc_w_${T}_$S2/*/w_counter_collection.csv $O/pmc_w_${T}_$S2/w_counter_collection.csv 2>/dev/null | head -1) --frames2 $S2 \
      --merge-into $O/decode_step_traffic.json > $O/pmc_traffic_b${T}.json
  cat $O/pmc_traffic_b${T}.json
  rm -rf $O/pmc_f_${T}_* $O/pmc_w_${T}_*
done
cp $O/decode_step_traffic.json profiles/decode_step_traffic.json   # on the box only: the committed copy is made from gpurun_out/ afterwards
timeout -k 10 600 python bench.py > $O/bench_default_with_traffic.json 2> $O/bench_default_with_traffic.err
python -c "import json;j=json.load(open('$O/bench_default_with_traffic.json'));print('b1', j['value'], j['roofline']['frac'], j['roofline']['traffic'], '| b64', j['b64']['value'], j['b64']['roofline']['traffic'])"
exit 0
fi
# 1. full GPU test suite
if [ "$3" != "--skip-tests" ]; then timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 600 -s > $O/gpu_tests.log 2>&1 || true; grep -E "passed|failed|free-running|teacher" $O/gpu_tests.log | tail -8; fi
# 2. headline bench (configs[1] + the b64 sub-record + cpu baseline), then b=8 and 1.7B b=8
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json;j=json.load(open('$O/bench_default.json'));print('b1', j['value'], j['decode_ms_per_frame_step'], j['roofline']['frac'], *[(k, j[k].get('value'), j[k].get('decode_ms_per_frame_step'), (j[k].get('roofline') or {}).get('frac')) for k in ('b64','b8','b64_f2048','b64_f2048_kv_bf16','b1_f2048_kv_bf16') if k in j], '| cpu', j['cpu_baseline']['value'])"
timeout -k 10 600 python bench.py --model 1.7b --batch 8 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_1p7b_b8.json 2> $O/bench_1p7b_b8.err
timeout -k 10 600 python bench.py --batch 128 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_b128.json 2> $O/bench_b128.err
python -c "import json;print(*[(t, json.load(open('$O/bench_%s.json' % t))['value']) for t in ('1p7b_b8','b128')])"
# 3. kernel traces (eager launches: rocprofv3 cannot trace hipGraphLaunch on this ROCm): b=1 x 1024 frames, b=8 / b=64 x 48 frames
rocprofv3 --kernel-trace --stats -d $O/trace_b1 -o b1 -- python bench.py --frames 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-b64 > $O/trace_b1.log 2>&1
python tools/rocpd_summary.py $O/trace_b1/b1_results.db 40 > $O/decode_b1_f1024_eager_by_grid.txt
head -14 $O/decode_b1_f1024_eager_by_grid.txt
rm -rf $O/trace_b1
for BB in 8 64; do
  rocprofv3 --kernel-trace --stats -d $O/trace_b$BB -o b -- python bench.py --batch $BB --frames 48 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $O/trace_b$BB.log 2>&1
  python tools/rocpd_summary.py $O/trace_b$BB/b_results.db 30 > $O/decode_b${BB}_f48_eager_by_grid.txt
  rm -rf $O/trace_b$BB
done
head -22 $O/decode_b64_f48_eager_by_grid.txt
# 3a. the batched step at a long context (where the attention streams the KV cache): 64 slots moved to context 1024, both KV dtypes
for KV in fp32 bf16; do
  rocprofv3 --kernel-trace --stats -d $O/trace_ctx_$KV -o t -- python tools/ctx_bench.py --batch 64 --ctx 1024 --kv $KV --no-graph --steps 4 --max-ctx 2112 > $O/trace_ctx_$KV.log 2>&1
  python tools/rocpd_summary.py $O/trace_ctx_$KV/t_results.db 24 > $O/decode_b64_ctx1024_${KV}_eager_by_grid.txt
  rm -rf $O/trace_ctx_$KV
  python tools/ctx_bench.py --batch 64 --ctx 1024 --kv $KV --stages --max-ctx 2112 >> $O/ctx_bench.txt
done
grep -h k_attn_stream $O/decode_b64_ctx1024_*_eager_by_grid.txt; cut -c1-300 $O/ctx_bench.txt
# 3b. codec decoder alone, 2048 frames + short utterances
rocprofv3 --kernel-trace --stats -d $O/trace_codec -o c -- python tools/codec_bench.py --reps 2 > $O/codec_bench.log 2>&1
python tools/rocpd_summary.py $O/trace_codec/c_results.db 45 > $O/codec_f2048_by_grid.txt
rm -rf $O/trace_codec
for FF in 25 64 256 2048; do python tools/codec_bench.py --frames $FF --reps 3 | grep frames= >> $O/codec_sizes.txt; done
cat $O/codec_sizes.txt
# 5. matrix-core issue rate of the codec decoder from hardware counters (SQ_INSTS_VALU_MFMA_MOPS_*: FLOP / 512), default stream
Q3TTS_NULL_STREAM=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_mfma -o m --output-format csv -- python tools/codec_bench.py --frames 2048 --reps 1 > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma.py $(ls $O/pmc_mfma/*/m_counter_collection.csv $O/pmc_mfma/m_counter_collection.csv 2>/dev/null | head -1) 4096 > $O/pmc_mfma_codec.json || true
cat $O/pmc_mfma_codec.json
rm -rf $O/pmc_mfma
