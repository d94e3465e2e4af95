set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05_i
O=gpurun_out/r05_i
bash tools/gpu_ab.sh r05_i tests
if [ -f tools/exp/libq3_prev.so ]; then
for i in 1 2; do
Q3TTS_LIB=$PWD/tools/exp/libq3_prev.so bash tools/gpu_ab.sh r05_i b1 Q3TTS_PREV_LIB=1
bash tools/gpu_ab.sh r05_i b1 Q3TTS_DUMMY=1
done
Q3TTS_LIB=$PWD/tools/exp/libq3_prev.so bash tools/gpu_ab.sh r05_i b64 Q3TTS_PREV_LIB=1
bash tools/gpu_ab.sh r05_i b64 Q3TTS_DUMMY=1
fi
