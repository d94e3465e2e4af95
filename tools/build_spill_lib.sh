#!/bin/bash
# Builds tools/exp/libspill.so = libq3tts_hip.so whose fused 96-channel residual units (k_conv_split<..., F2 = true>) use a private
# (scratch) segment on purpose (-DQ3_FORCE_SPILL).  Round 4 saw one wrong batched-vocoder result on a build whose dilation-9 fused unit
# spilled 8 bytes; this variant puts scratch back under the concurrent vocoder lanes so that the stress test can say whether "a kernel
# with scratch on concurrent lane streams" reproduces it:
#     Q3TTS_LIB=$PWD/tools/exp/libspill.so python -m pytest tests/test_gpu_codec_stress.py -m gpu -q
# Needs a prior `python leaxer-qwen3-tts_amd/build.py` (links its other objects).  Not shipped, not part of build().
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/exp" /tmp/q3spill
cd /tmp/q3spill
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DQ3_FORCE_SPILL -x hip -c "$ROOT/leaxer-qwen3-tts_amd/csrc/q3_codec_kernels.hip" -o ck_spill.o
B="$ROOT/leaxer-qwen3-tts_amd/build"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/exp/libspill.so" ck_spill.o "$B/q3_decode_kernels.hip.o" "$B/q3_gemm_kernels.hip.o" \
    "$B/q3_speaker_kernels.hip.o" "$B/q3_engine.cpp.o" "$B/q3_codec.cpp.o" "$B/q3_speaker.cpp.o" "$B/q3_audio.cpp.o" "$B/q3_bpe.cpp.o" "$B/q3_capi.cpp.o"
python "$ROOT/tools/kernel_resources.py" "$ROOT/tools/exp/libspill.so" k_conv_split | awk '$(NF-2) != 0' | head -20
echo "$ROOT/tools/exp/libspill.so"
