set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02v
O=gpurun_out/r02v
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
rocprofv3 --kernel-trace --stats -d $O/trace_c64 -o c -- python tools/codec_bench.py --frames 64 --reps 3 > $O/codec_bench64.log 2>&1
python tools/rocpd_summary.py $O/trace_c64/c_results.db 60 > $O/codec_f64_by_grid.txt
rm -rf $O/trace_c64
grep -v "k_fill_synth\|k_absmax\|k_repack\|k_split_planes\|rocclr\|k_spk\|k_snake_pre" $O/codec_f64_by_grid.txt | cut -c1-125 | head -50
