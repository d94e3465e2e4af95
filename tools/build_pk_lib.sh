#!/bin/bash
# Builds tools/exp/libpk.so = libq3tts_hip.so the way rounds 1-5 shipped it: kernels compiled WITH packed fp32 instructions
# (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  The product library is built without them (leaxer-qwen3-tts_amd/build.py: NO_PK) since
# round 5 found them returning wrong results under load on this hardware (profiles/r05_hunt/README.txt).  This variant is the reproducer:
#     Q3TTS_LIB=$PWD/tools/exp/libpk.so python tools/vocoder_stress.py --caps 120,100,90,2 --phase packed:Q3TTS_COUT1_PACKED=1   # ~50 % of jobs wrong
#     Q3TTS_LIB=$PWD/tools/exp/libpk.so python tools/decode_beside_vocoder.py --batch 24                                        # sampled ids move beside a busy vocoder
# Needs a prior `python leaxer-qwen3-tts_amd/build.py` (links its host objects).  Not shipped, not part of build().
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/exp" /tmp/q3pk
cd /tmp/q3pk
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
S="$ROOT/leaxer-qwen3-tts_amd/csrc"
hipcc $F -x hip -c "$S/q3_codec_kernels.hip" -o ck.o &
hipcc $F -mllvm -amdgpu-kernarg-preload-count=16 -x hip -c "$S/q3_decode_kernels.hip" -o dk.o &
hipcc $F -mllvm -amdgpu-kernarg-preload-count=16 -x hip -c "$S/q3_gemm_kernels.hip" -o gk.o &
hipcc $F -x hip -c "$S/q3_speaker_kernels.hip" -o sk.o &
wait
B="$ROOT/leaxer-qwen3-tts_amd/build"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/exp/libpk.so" ck.o dk.o gk.o sk.o \
    "$B/q3_engine.cpp.o" "$B/q3_codec.cpp.o" "$B/q3_speaker.cpp.o" "$B/q3_audio.cpp.o" "$B/q3_bpe.cpp.o" "$B/q3_capi.cpp.o"
echo "$ROOT/tools/exp/libpk.so"
