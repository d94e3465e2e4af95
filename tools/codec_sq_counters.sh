set -e
# SQ counter table of one 2048-frame codec decode, per kernel (tools/pmc_sq_summary.py): four rocprofv3 --pmc passes of the same command,
# counters only (no trace domains), default stream (Q3TTS_NULL_STREAM=1: --pmc crashes on user streams on this image).
#   bash tools/codec_sq_counters.sh <tag>        -> gpurun_out/<tag>/codec_sq_counters.txt
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
i=0
for SET in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i + 1))
  Q3TTS_NULL_STREAM=1 timeout -k 10 200 rocprofv3 --pmc $SET -d $O/sq$i -o s --output-format csv -- python tools/codec_bench.py --frames 2048 --reps 1 > $O/sq$i.log 2>&1
done
python tools/pmc_sq_summary.py $(for k in 1 2 3 4; do ls $O/sq$k/*/s_counter_collection.csv $O/sq$k/s_counter_collection.csv 2>/dev/null | head -1; done) --top 16 > $O/codec_sq_counters.txt
rm -rf $O/sq1 $O/sq2 $O/sq3 $O/sq4
head -40 $O/codec_sq_counters.txt | cut -c1-330
