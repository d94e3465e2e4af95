set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05_c
for i in 1 2; do
bash tools/gpu_ab.sh r05_c b64 Q3TTS_GEMM_PLAIN_SLABS=1
bash tools/gpu_ab.sh r05_c b64 Q3TTS_DUMMY=1
done
bash tools/gpu_ab.sh r05_c b8 Q3TTS_DUMMY=1
