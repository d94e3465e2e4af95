#!/bin/bash
# AddressSanitizer + UBSan run of the host-only parsers (tokenizer files, WAV reader, resampler, mel) on mutated inputs; CPU only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I "$ROOT/leaxer-qwen3-tts_amd/csrc" \
    "$ROOT/tools/host_sanitize.cpp" "$ROOT/leaxer-qwen3-tts_amd/csrc/q3_bpe.cpp" "$ROOT/leaxer-qwen3-tts_amd/csrc/q3_audio.cpp" -o "$D/host_sanitize"
"$D/host_sanitize" "${1:-400}" "$D"
rm -rf "$D"
