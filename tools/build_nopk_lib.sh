#!/bin/bash
# Builds tools/exp/libnopk.so = libq3tts_hip.so with packed fp32 instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) switched off
# for every kernel (-target-feature -packed-fp32-ops).  Round 5 found the vocoder's last conv returning wrong sums from v_pk_fma_f32
# while the chip was in the state heavy kernels leave behind for a few hundred microseconds (profiles/r05_hunt/README.txt); that kernel
# is scalar by construction now, this variant prices the belt-and-braces option of a library without any packed fp32:
#     Q3TTS_LIB=$PWD/tools/exp/libnopk.so python tools/codec_bench.py --frames 2048
#     Q3TTS_LIB=$PWD/tools/exp/libnopk.so python bench.py --batch 64 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline
# Needs a prior `python leaxer-qwen3-tts_amd/build.py` (links its host objects).  Not shipped, not part of build().
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/exp" /tmp/q3nopk
cd /tmp/q3nopk
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops"
S="$ROOT/leaxer-qwen3-tts_amd/csrc"
hipcc $F -x hip -c "$S/q3_codec_kernels.hip" -o ck.o &
hipcc $F -mllvm -amdgpu-kernarg-preload-count=16 -x hip -c "$S/q3_decode_kernels.hip" -o dk.o &
hipcc $F -mllvm -amdgpu-kernarg-preload-count=16 -x hip -c "$S/q3_gemm_kernels.hip" -o gk.o &
hipcc $F -x hip -c "$S/q3_speaker_kernels.hip" -o sk.o &
wait
B="$ROOT/leaxer-qwen3-tts_amd/build"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/exp/libnopk.so" ck.o dk.o gk.o sk.o \
    "$B/q3_engine.cpp.o" "$B/q3_codec.cpp.o" "$B/q3_speaker.cpp.o" "$B/q3_audio.cpp.o" "$B/q3_bpe.cpp.o" "$B/q3_capi.cpp.o"
python "$ROOT/tools/kernel_resources.py" "$ROOT/tools/exp/libnopk.so" | awk '{ for (i = 1; i <= NF; ++i) if ($i == "scratch" && $(i + 1) != 0) print }' | head
echo "$ROOT/tools/exp/libnopk.so"
