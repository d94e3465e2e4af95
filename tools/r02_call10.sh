set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02p
O=gpurun_out/r02p
for V in f96 c7 c1; do
  Q3TTS_LIB=$PWD/tools/exp/libprof_$V.so python tools/conv_phases.py --frames 2048 > $O/conv_phases_$V.txt 2>&1
  echo == $V; cat $O/conv_phases_$V.txt
done
