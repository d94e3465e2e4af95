set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02af
O=gpurun_out/r02af
export Q3TTS_NO_HEAD_SLABS=1
rocprofv3 --kernel-trace --stats -d $O/trace_b64 -o b -- python bench.py --batch 64 --frames 48 --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $O/trace_b64.log 2>&1
python tools/rocpd_summary.py $O/trace_b64/b_results.db 40 > $O/decode_b64_noslabs_by_grid.txt
rm -rf $O/trace_b64
grep -E "k_sample|k_gemm|k_gemv" $O/decode_b64_noslabs_by_grid.txt | cut -c1-120
