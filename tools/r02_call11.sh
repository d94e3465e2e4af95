set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ac
O=gpurun_out/r02ac
timeout -k 10 600 python -m pytest tests/test_gpu_codec.py tests/test_gpu_full.py -x -q --timeout 500 > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for FF in 64 256 2048; do python tools/codec_bench.py --frames $FF --reps 3 | grep frames= >> $O/codec_sizes.txt; done
for FF in 256 2048; do Q3TTS_CONV_FP32_ACT=1 python tools/codec_bench.py --frames $FF --reps 3 | grep frames= >> $O/codec_sizes_fp32act.txt; done
echo fast; cat $O/codec_sizes.txt; echo fp32act; cat $O/codec_sizes_fp32act.txt
for V in; do
  Q3TTS_LIB=$PWD/tools/exp/libprof_$V.so python tools/conv_phases.py --frames 2048 > $O/conv_phases_$V.txt 2>&1
  echo == $V; head -5 $O/conv_phases_$V.txt
done
rocprofv3 --kernel-trace --stats -d $O/trace_codec -o c -- python tools/codec_bench.py --reps 2 > $O/codec_bench.log 2>&1
python tools/rocpd_summary.py $O/trace_codec/c_results.db 30 > $O/codec_f2048_by_grid.txt
rm -rf $O/trace_codec
cut -c1-125 $O/codec_f2048_by_grid.txt | head -24
