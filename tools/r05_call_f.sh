set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05_f
for i in 1 2; do
bash tools/gpu_ab.sh r05_f b64 Q3TTS_DUMMY=1
bash tools/gpu_ab.sh r05_f b64 Q3TTS_SEAM_GU_KS=2
done
for i in 1 2 3; do
bash tools/gpu_ab.sh r05_f b8 Q3TTS_ATTN_TINY2=0
bash tools/gpu_ab.sh r05_f b8 Q3TTS_DUMMY=1
done
bash tools/gpu_ab.sh r05_f b1 Q3TTS_DUMMY=1
