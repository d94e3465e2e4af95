set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02n
O=gpurun_out/r02n
export Q3TTS_NULL_STREAM=1
timeout -k 10 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/p1 -o p --output-format csv -- python tools/codec_bench.py --frames 1024 --reps 1 > $O/p1.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/p2 -o p --output-format csv -- python tools/codec_bench.py --frames 1024 --reps 1 > $O/p2.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $O/p3 -o p --output-format csv -- python tools/codec_bench.py --frames 1024 --reps 1 > $O/p3.log 2>&1
F1=$(ls $O/p1/*/p_counter_collection.csv $O/p1/p_counter_collection.csv 2>/dev/null | head -1)
F2=$(ls $O/p2/*/p_counter_collection.csv $O/p2/p_counter_collection.csv 2>/dev/null | head -1)
F3=$(ls $O/p3/*/p_counter_collection.csv $O/p3/p_counter_collection.csv 2>/dev/null | head -1)
python tools/pmc_sq_summary.py $F1 $F2 $F3 --top 16 > $O/codec_sq_counters.txt
rm -rf $O/p1 $O/p2 $O/p3
cat $O/codec_sq_counters.txt
