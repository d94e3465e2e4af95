"""The reference-shaped C++ surface: weight file round trip, TTSEngine via the leaxer-tts CLI
(flags of reference src/main_onnx.cpp:99-124), 16-bit WAV writer semantics (:47-54)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from util import frame_tokens, tiny_pair, to_osampling

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "leaxer-qwen3-tts_amd", "leaxer-tts")


def read_wav16(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[36:40] == b"data"
    fmt, ch, rate, _, align, bits = struct.unpack("<HHIIHH", b[20:36])
    assert (fmt, ch, rate, align, bits) == (1, 1, 24000, 2, 16)
    n = struct.unpack("<I", b[40:44])[0]
    assert struct.unpack("<I", b[4:8])[0] == 36 + n and len(b) == 44 + n
    return np.frombuffer(b[44:], np.int16)


def test_weight_file_roundtrip_and_cli(tmp_path):
    import q3tts
    eng, orc, w = tiny_pair(seed=4, max_batch=1, max_ctx=96)
    mdir = tmp_path / "model"
    mdir.mkdir()
    eng.save_weights(str(mdir / "model.q3w"))
    eng2 = q3tts.Engine(eng.cfg, device=0, max_batch=1, max_ctx=96)
    eng2.load_weights(str(mdir / "model.q3w"))
    for name in ("talker.layers.1.q_proj", "cd.dec.blocks.2.res.1.conv1.w", "cp.embed.7", "text.fc2.b"):
        assert np.array_equal(eng2.get_tensor(name, w[name].shape), w[name]), name
    eng2.close()
    # the Python writer produces a file the C loader accepts and that holds the same values
    import sys
    sys.path.insert(0, ROOT)
    from tools.pack_weights import write_q3w
    write_q3w(str(mdir / "py.q3w"), eng.cfg, w)
    eng3 = q3tts.Engine(eng.cfg, device=0, max_batch=1, max_ctx=96)
    eng3.load_weights(str(mdir / "py.q3w"))
    assert np.array_equal(eng3.get_tensor("cp.layers.0.down_proj", w["cp.layers.0.down_proj"].shape), w["cp.layers.0.down_proj"])
    eng3.close()

    out = tmp_path / "o" / "x.wav"
    out.parent.mkdir()
    r = subprocess.run([CLI, "-m", str(mdir), "--tokens", "11,22,33,44", "-o", str(out), "--lang", "ja", "--temp", "0.8",
                        "--top-k", "50", "--top-p", "0.95", "--max-tokens", "12", "--seed", "7", "--bogus-flag"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Generated" in r.stdout and "Saved to" in r.stdout
    got = read_wav16(str(out))
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=12)
    ids = frame_tokens([11, 22, 33, 44])
    codes = orc.generate(orc.build_prompt(ids, 3), to_osampling(sp), seed=7, stream=0, cp_cached=True, ignore_eos=False)
    ref = orc.vocoder(codes)
    ref16 = (np.clip(ref, -1, 1) * 32767.0).astype(np.int16)   # truncation toward zero, like the reference writer
    assert got.shape == ref16.shape
    d = got.astype(np.float64) - ref16.astype(np.float64)
    assert np.sqrt(np.mean(d ** 2)) < 3.3 and np.abs(d).max() <= 64   # north_star: 1e-4 RMS = 3.3 LSB of int16
    eng.close()
    orc.close()


def test_cli_text_prompt_uses_the_tokenizer(tmp_path):
    """-p TEXT == --tokens <ids of the tokenizer> (the ids themselves are pinned against the
    reference's tokenizer in tests/test_tokenizer.py); files in the reference's location (tts_onnx.cpp:110-112)."""
    import json
    import q3tts
    eng, orc, _ = tiny_pair(seed=5, max_batch=1, max_ctx=96)
    mdir = tmp_path / "models" / "onnx"
    mdir.mkdir(parents=True)
    eng.save_weights(str(mdir / "model.q3w"))
    eng.close()
    orc.close()
    tdir = tmp_path / "models" / "models" / "Qwen3-TTS-12Hz-0.6B-Base"   # <parent of model_dir>/models/<name>
    tdir.mkdir(parents=True)
    vocab = {ch: 10 + k for k, ch in enumerate("abcdefghijklmnopqrstuvwxyz")}
    vocab.update({"\u0120": 40, "he": 41, "ll": 42, "hell": 43, "hello": 44, "\u0120w": 45, "or": 46, "\u0120wor": 47, ",": 48})
    json.dump(vocab, open(tdir / "vocab.json", "w"))
    open(tdir / "merges.txt", "w").write("#version: 0.2\nh e\nl l\nhe ll\nhell o\n\u0120 w\no r\n\u0120w or\n")
    tok = q3tts.Tokenizer(str(tdir / "vocab.json"), str(tdir / "merges.txt"))
    ids = [int(t) for t in tok.encode("hello, world")]
    tok.close()
    assert ids == [44, 48, 47, 21, 13]
    common = ["-m", str(mdir), "--max-tokens", "8", "--seed", "3"]
    a, b = tmp_path / "a.wav", tmp_path / "b.wav"
    r = subprocess.run([CLI, "-p", "hello, world", "-o", str(a)] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([CLI, "--tokens", ",".join(map(str, ids)), "-o", str(b)] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(a, "rb").read() == open(b, "rb").read() and len(read_wav16(str(a))) > 0
    # present but unreadable tokenizer files are a constructor error (tts_onnx.cpp:114-117)
    open(tdir / "vocab.json", "w").write("[1,2]")
    r = subprocess.run([CLI, "-p", "hello", "-o", str(a)] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Failed to load tokenizer" in r.stderr


def test_cli_streaming_equals_one_shot(tmp_path):
    """--stream-chunk N: audio delivered in chunks while the frames are generated is the one-shot decode, sample for sample."""
    eng, orc, _ = tiny_pair(seed=6, max_batch=1, max_ctx=96)
    mdir = tmp_path / "m"
    mdir.mkdir()
    eng.save_weights(str(mdir / "model.q3w"))
    eng.close()
    orc.close()
    common = ["-m", str(mdir), "--tokens", "3,1,4,1,5,9", "--max-tokens", "23", "--seed", "8"]
    a, b = str(tmp_path / "a.wav"), str(tmp_path / "b.wav")
    r = subprocess.run([CLI, "-o", a] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([CLI, "-o", b, "--stream-chunk", "5"] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "First" in r.stdout and "chunks" in r.stdout, r.stdout + r.stderr
    wa, wb = read_wav16(a), read_wav16(b)
    assert wa.shape == wb.shape and wa.size > 0 and np.abs(wa.astype(int) - wb.astype(int)).max() <= 1


def test_cli_errors_like_the_reference(tmp_path):
    r = subprocess.run([CLI, "-p", "hello"], capture_output=True, text=True)
    assert r.returncode == 1 and "required" in r.stderr
    r = subprocess.run([CLI, "-m", str(tmp_path / "nope"), "-p", "hello"], capture_output=True, text=True)
    assert r.returncode == 1 and "model directory not found" in r.stderr
    r = subprocess.run([CLI, "-m", "synthetic:0", "-p", "hello", "--max-tokens", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Tokenizer not ready" in r.stderr and "synthesis failed" in r.stderr


def test_cli_synthetic_1p7b_spec(tmp_path):
    """-m synthetic-1.7b:<seed>: the 1.7B dims (talker 2048 wide, predictor behind cp.proj) through the reference-shaped CLI."""
    out = str(tmp_path / "o.wav")
    r = subprocess.run([CLI, "-m", "synthetic-1.7b:3", "--tokens", "1001,2002,3003", "-o", out, "--max-tokens", "6", "--seed", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    pcm = read_wav16(out)
    assert pcm.size > 6 * 1920 - 1920 and np.abs(pcm).max() > 0


def test_onnx_converter_output_drives_the_engine(tmp_path):
    """f4, the ONNX half: the eight self-made .onnx-format graphs (tools/make_onnx_fixture.py: the tiny config's tensors, Linear weights
    anonymous and transposed) -> tools/import_onnx.py --by-shape-order (as a command line, the way a user would run it) -> model.q3w ->
    q3tts_load_weights_file: the engine synthesizes exactly the codes and PCM of an engine filled with the same tensors directly, and the
    codes the oracle makes from them.  A format round trip (the files a reference user holds are the graphs of
    /root/reference/src/tts_onnx.cpp:91-107); no claim about a real export's initialiser names."""
    import json
    import sys
    import q3tts
    import q3_oracle as qo
    sys.path.insert(0, ROOT)
    from tools.make_onnx_fixture import write_fixture
    eng, orc, w = tiny_pair(seed=9, max_batch=2, max_ctx=96)
    try:
        keys = json.load(open(os.path.join(ROOT, "tests", "golden", "hf_state_dict_keys.json")))
        paths = write_fixture(str(tmp_path / "onnx"), qo.tensor_specs(orc.cfg), w, keys)
        cfg_json = tmp_path / "cfg.json"
        cfg_json.write_text(json.dumps(orc.cfg.to_dict()))
        mdir = tmp_path / "model"
        mdir.mkdir()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "import_onnx.py"), "--out", str(mdir / "model.q3w"), "--config", str(cfg_json),
                            "--by-shape-order", "--assume-square-transposed"] + paths, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "[shape-order]" in r.stdout and "0 registry tensors missing" in r.stdout
        conv = q3tts.Engine(eng.cfg, device=0, max_batch=2, max_ctx=96)
        conv.load_weights(str(mdir / "model.q3w"))
        sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=14)
        toks = [frame_tokens([3, 1, 4, 1, 5]), frame_tokens([9, 2, 6])]
        pa, ca, na = eng.synthesize_batch(toks, sp, lang=0, seed=2, ignore_eos=True)
        pb, cb, nb_ = conv.synthesize_batch(toks, sp, lang=0, seed=2, ignore_eos=True)
        conv.close()
        for u in range(2):
            assert np.array_equal(ca[u], cb[u]) and np.array_equal(pa[u], pb[u]), u
            ref = orc.generate(orc.build_prompt(toks[u], 0), to_osampling(sp), seed=2, stream=u, cp_cached=True, ignore_eos=True)
            assert np.array_equal(cb[u], ref), u
    finally:
        eng.close()
        orc.close()
