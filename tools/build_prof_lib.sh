#!/bin/bash
# Builds tools/exp/libprof.so = libq3tts_hip.so with -DQ3_SAMPLE_PROF (phase timestamps inside k_sample / k_gemv1 / k_cp_attn_oproj) plus
# an accessor; use with Q3TTS_LIB=$PWD/tools/exp/libprof.so python tools/kernel_phases.py.  Needs a prior `python leaxer-qwen3-tts_amd/build.py`.
set -e
PROF_DEFS=("$@")
# PROF_DEFS (bash array, e.g. PROF_DEFS=('-DQ3_CONV_PROF_SEL=(a.taps==7&&a.C_in==192)' -DQ3_CONV_PROF_WG=2000); source this script or export is not enough for arrays: pass them as arguments instead): extra -D flags for the codec kernels (Q3_CONV_PROF_SEL / Q3_CONV_PROF_WG: which k_conv_split launch and workgroup is stamped)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/exp" /tmp/q3prof
cd /tmp/q3prof
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DQ3_SAMPLE_PROF -mllvm -amdgpu-kernarg-preload-count=16 -x hip -c "$ROOT/leaxer-qwen3-tts_amd/csrc/q3_decode_kernels.hip" -o dk_prof.o
# the codec kernels take minutes to compile: SKIP_CODEC=1 links the regular object (no stamps inside k_conv_split) and stubs the accessor
CK=ck_prof.o
if [ "$SKIP_CODEC" = "1" ]; then
  CK="$ROOT/leaxer-qwen3-tts_amd/build/q3_codec_kernels.hip.o"
  echo 'namespace q3 { void conv_prof_read(long long* out) { for (int i = 0; i < 64; ++i) out[i] = 0; } }' > conv_stub.cpp
  g++ -O2 -fPIC -c conv_stub.cpp -o conv_stub.o
else
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DQ3_SAMPLE_PROF "${PROF_DEFS[@]}" -x hip -c "$ROOT/leaxer-qwen3-tts_amd/csrc/q3_codec_kernels.hip" -o ck_prof.o
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -DQ3_SAMPLE_PROF -mllvm -amdgpu-kernarg-preload-count=16 -x hip -c "$ROOT/leaxer-qwen3-tts_amd/csrc/q3_gemm_kernels.hip" -o gk_prof.o
cat > prof_api.cpp <<'EOC'
namespace q3 { void sample_prof_read(long long* out); void conv_prof_read(long long* out); void gemm_prof_read(long long* out); void seam_prof_read(long long* out); }
extern "C" void q3_seam_prof(long long* out) { q3::seam_prof_read(out); }
extern "C" void q3_kernel_prof(long long* out) { q3::sample_prof_read(out); }
extern "C" void q3_conv_prof(long long* out) { q3::conv_prof_read(out); }
extern "C" void q3_gemm_prof(long long* out) { q3::gemm_prof_read(out); }
EOC
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -x hip -c prof_api.cpp -o prof_api.o
B="$ROOT/leaxer-qwen3-tts_amd/build"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/exp/libprof.so" dk_prof.o $CK $([ "$SKIP_CODEC" = "1" ] && echo conv_stub.o) gk_prof.o prof_api.o \
    "$B/q3_speaker_kernels.hip.o" "$B/q3_engine.cpp.o" "$B/q3_codec.cpp.o" "$B/q3_speaker.cpp.o" "$B/q3_audio.cpp.o" "$B/q3_bpe.cpp.o" "$B/q3_capi.cpp.o"
echo "$ROOT/tools/exp/libprof.so"
