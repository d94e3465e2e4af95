"""Build libq3tts_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so ships with the repo
snapshot to the GPU box.  `python leaxer-qwen3-tts_amd/build.py [--force]`."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libq3tts_hip.so")
SOURCES = ["q3_decode_kernels.hip", "q3_gemm_kernels.hip", "q3_codec_kernels.hip", "q3_speaker_kernels.hip", "q3_engine.cpp", "q3_codec.cpp",
           "q3_speaker.cpp", "q3_audio.cpp", "q3_bpe.cpp", "q3_capi.cpp"]
HEADERS = ["q3_common.h", "q3_engine.h", "q3_kvpool.h", "q3_bpe.h", "q3_audio.h", os.path.join("..", "..", "include", "q3tts.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
# Kernels are built WITHOUT packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): on this hardware they return wrong
# results now and then while other kernels keep the chip loaded — the vocoder's last conv (two PCM samples off by up to 1.7e-2 in up to
# 91 % of stressed jobs) and the decode step's logits (~1e-6, enough to move marginal sampled ids) beside a busy vocoder — and never with
# scalar fp32 (profiles/r05_hunt/README.txt; tests/test_kernel_resources.py checks the built code objects).  Cost: nothing on the codec
# and the batched step, 2 % on the b=1 step.  Q3TTS_BUILD_PACKED_FP32=1 builds the old way (tools/build_pk_lib.sh: the reproducer).
NO_PK = [] if os.environ.get("Q3TTS_BUILD_PACKED_FP32") == "1" else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _digest(paths):
    h = hashlib.sha1(" ".join(FLAGS + NO_PK).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _compile(src, hdr_digest, force):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    tag = obj + ".tag"
    dg = _digest([path]) + hdr_digest
    if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == dg:
        return obj, False
    extra = ["-mllvm", "-amdgpu-kernarg-preload-count=16"] if src in ("q3_decode_kernels.hip", "q3_gemm_kernels.hip") else []   # leading scalar kernel args arrive in SGPRs
    if src.endswith(".hip"):
        extra = extra + NO_PK
    cmd = ["hipcc"] + FLAGS + extra + ["-x", "hip", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
    with open(tag, "w") as f:
        f.write(dg)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdr_digest = _digest([os.path.join(CSRC, h) for h in HEADERS])
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr_digest, force), srcs))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        os.replace(LIB + ".tmp", LIB)
        if verbose:
            print("linked", LIB)
    _build_cli(force)
    _build_pmc_driver(force)
    return LIB


CLI = os.path.join(HERE, "leaxer-tts")


def _build_cli(force=False):
    """leaxer-tts: the reference's CLI surface over TTSEngine (plain g++-style host code, links the .so)."""
    srcs = [os.path.join(CSRC, "tts_engine.cpp"), os.path.join(CSRC, "main.cpp")]
    deps = srcs + [os.path.join(CSRC, "tts_engine.h"), os.path.join(HERE, "..", "include", "q3tts.h"), LIB]
    if not force and os.path.exists(CLI) and all(os.path.getmtime(CLI) >= os.path.getmtime(d) for d in deps):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", CLI] + srcs + ["-L" + HERE, "-lq3tts_hip", "-lstdc++fs", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building leaxer-tts failed:\n" + r.stderr[-4000:])
    return CLI


def _build_pmc_driver(force=False):
    """tools/pmc_bisect: a C driver of the decode step for the rocprofv3 --pmc passes (the program after `--` must be the program
    itself, not python).  It holds a q3tts_config on its stack, so it is rebuilt whenever the C-ABI header changes."""
    root = os.path.join(HERE, "..")
    src, out = os.path.join(root, "tools", "pmc_bisect.cpp"), os.path.join(root, "tools", "pmc_bisect")
    deps = [src, os.path.join(root, "include", "q3tts.h"), LIB]
    if not os.path.exists(src):
        return None
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-o", out, src, "-L" + HERE, "-lq3tts_hip", "-Wl,-rpath,$ORIGIN/../leaxer-qwen3-tts_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building tools/pmc_bisect failed:\n" + r.stderr[-4000:])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
