"""The drop-in boundary seen from the reference's side: its own caller, src/main_onnx.cpp, compiled UNCHANGED against this build's
`leaxer_qwen::TTSEngine` (csrc/tts_engine.h placed on the include path under the name the reference includes, tts_onnx.h) and linked
to libq3tts_hip.so.  Needs /root/reference (build container only; the file is read where it lies, nothing is copied into the repo),
no GPU: -h and the two error paths of main() that end before any synthesis (reference src/main_onnx.cpp:126-156)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = "/root/reference/src/main_onnx.cpp"
PKG = os.path.join(ROOT, "leaxer-qwen3-tts_amd")


@pytest.fixture(scope="module")
def ref_cli(tmp_path_factory):
    if not os.path.exists(REF_MAIN):
        pytest.skip("reference sources not present (GPU box)")
    if not os.path.exists(os.path.join(PKG, "libq3tts_hip.so")):
        pytest.skip("libq3tts_hip.so not built")
    d = tmp_path_factory.mktemp("refcli")
    inc = d / "inc"
    inc.mkdir()
    # the ONLY adaptation: the header's file name.  main_onnx.cpp says #include "tts_onnx.h"; INTEGRATION.md section 1.
    shutil.copy(os.path.join(PKG, "csrc", "tts_engine.h"), inc / "tts_onnx.h")
    exe = d / "leaxer-tts-ref-main"
    # The reference source goes in through stdin: an #include "..." looks in the including FILE's directory first, and next to
    # main_onnx.cpp sits the reference's own tts_onnx.h (ONNX Runtime members: a different class layout).  Read from stdin, the
    # "current file" has no directory and the lookup starts at -I, i.e. at this build's header.
    obj = d / "main_onnx.o"
    with open(REF_MAIN, "rb") as src:
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-x", "c++", "-c", "-", "-I", str(inc), "-I", os.path.join(ROOT, "include"), "-o", str(obj)],
                           stdin=src, capture_output=True, text=True, cwd=str(d))
    assert r.returncode == 0, "the reference's main_onnx.cpp no longer compiles against tts_engine.h:\n" + r.stderr[-3000:]
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(obj), os.path.join(PKG, "csrc", "tts_engine.cpp"),
           "-L", PKG, "-lq3tts_hip", "-lstdc++fs", "-Wl,-rpath," + PKG, "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "linking the reference's main against tts_engine.cpp + libq3tts_hip.so failed:\n" + r.stderr[-3000:]
    return str(exe), d


def test_reference_main_builds_and_prints_its_usage(ref_cli):
    exe, _ = ref_cli
    r = subprocess.run([exe, "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    for flag in ("-m, --model", "-p, --prompt", "-o, --output", "--lang", "--ref", "--temp", "--top-k", "--top-p", "--max-tokens"):
        assert flag in r.stdout, flag


def test_reference_main_error_paths(ref_cli):
    exe, d = ref_cli
    r = subprocess.run([exe, "-p", "hi"], capture_output=True, text=True, timeout=60)                 # :126-130
    assert r.returncode == 1 and "--model and --prompt are required" in r.stderr
    r = subprocess.run([exe, "-m", str(d / "nope"), "-p", "hi"], capture_output=True, text=True, timeout=60)   # :132-135
    assert r.returncode == 1 and "model directory not found" in r.stderr
    empty = d / "empty_model_dir"
    empty.mkdir()
    # an existing directory without model files: the engine constructor fails, is_ready() is false, main prints get_error() (:151-156)
    r = subprocess.run([exe, "-m", str(empty), "-p", "hi", "-o", str(d / "o.wav")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and r.stderr.startswith("Error: ") and "model.q3w" in r.stderr, r.stderr
    assert not (d / "o.wav").exists()
