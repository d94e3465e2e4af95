"""Build oracle/_ref/libleaxer_ref.so from the REFERENCE's own source files, where they lie under
/root/reference (never copied into this repo): src/io/{tokenizer,wav_reader,mel}.cpp have no ONNX Runtime
dependency, so plain g++ on those files works.  (The hot-path file src/tts_onnx.cpp needs the ONNX Runtime
headers + library and is NOT buildable here — DESIGN.md section 2.)  Outputs only into oracle/_ref/ (git-ignored,
but shipped to the GPU box with the snapshot).  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")
SO = os.path.join(OUT, "libleaxer_ref.so")


def build(force=False):
    if not os.path.isdir(os.path.join(REF, "src", "io")):
        return SO if os.path.exists(SO) else None   # GPU box: use the prebuilt file if it travelled
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(REF, "src", "io", f) for f in ("tokenizer.cpp", "wav_reader.cpp", "mel.cpp")]
    shim = os.path.join(HERE, "ref_shim", "ref_io_shim.cpp")
    deps = srcs + [shim]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(REF, "src"), "-o", SO + ".tmp"] + srcs + [shim]
    subprocess.run(cmd, check=True)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force=True))
