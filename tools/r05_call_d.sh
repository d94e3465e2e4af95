set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05_d
for i in 1 2; do
bash tools/gpu_ab.sh r05_d b64 Q3TTS_DUMMY=1
bash tools/gpu_ab.sh r05_d b64 Q3TTS_SEAM_BIG=1
bash tools/gpu_ab.sh r05_d b64 Q3TTS_SEAM_BIG=2
bash tools/gpu_ab.sh r05_d b64 Q3TTS_SEAM_BIG=3
done
