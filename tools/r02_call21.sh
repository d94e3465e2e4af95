set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02am
O=gpurun_out/r02am
rocprofv3 --kernel-trace --stats -d $O/trace_b64 -o b -- python bench.py --batch 64 --frames 48 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $O/trace_b64.log 2>&1
python tools/rocpd_summary.py $O/trace_b64/b_results.db 40 --pct > $O/decode_b64_pct.txt
rm -rf $O/trace_b64
grep -E "k_attn" $O/decode_b64_pct.txt | cut -c1-190
