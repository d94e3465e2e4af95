"""SURVEY.md 8f-2, host part — the product's WAV reader / resampler / log-mel extractor (q3tts_read_wav_host,
q3tts_resample_host, q3tts_mel_host; host-only code in libq3tts_hip.so) against the REFERENCE's own
src/io/wav_reader.cpp and src/io/mel.cpp compiled into oracle/_ref/libleaxer_ref.so (oracle/build_ref.py).
Reader and resampler: bit-exact (same arithmetic on the same bytes).  Mel: the FFT differs in summation
order, so energies are compared with a relative + noise-floor tolerance stated below."""
import ctypes as C
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "leaxer-qwen3-tts_amd"))

import build_ref  # noqa: E402


@pytest.fixture(scope="module")
def ref():
    so = build_ref.build()
    if not so or not os.path.exists(so):
        pytest.skip("oracle/_ref/libleaxer_ref.so not built (needs /root/reference)")
    L = C.CDLL(so)
    L.ref_read_wav.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.ref_resample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.ref_mel.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return L


def ref_read(L, path, cap=1 << 22):
    buf = np.zeros(cap, np.float32)
    sr = C.c_int(-1)
    n = L.ref_read_wav(os.fsencode(path), buf.ctypes.data, cap, C.byref(sr))
    assert n <= cap
    return (buf[:n].copy(), sr.value) if n > 0 else None


def chunk(cid, body):
    return cid + struct.pack("<I", len(body)) + body


def wav_bytes(fmt_tag, channels, rate, bits, data, extra_before=b"", extra_after=b"", fmt_ext=b"", riff=b"RIFF", wave=b"WAVE",
              data_size=None):
    fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits) + fmt_ext
    d = b"data" + struct.pack("<I", len(data) if data_size is None else data_size) + data
    body = wave + chunk(b"fmt ", fmt) + extra_before + d + extra_after
    return riff + struct.pack("<I", len(body)) + body


def pcm_cases():
    rng = np.random.default_rng(0)
    n = 777
    i16 = rng.integers(-32768, 32768, (n, 2), dtype=np.int64).astype("<i2")
    i32 = rng.integers(-2 ** 31, 2 ** 31, (n, 1), dtype=np.int64).astype("<i4")
    u8 = rng.integers(0, 256, (n, 3), dtype=np.int64).astype(np.uint8)
    f32 = rng.standard_normal((n, 2)).astype("<f4")
    i24 = rng.integers(-2 ** 23, 2 ** 23, (n, 2), dtype=np.int64)
    b24 = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in i24.reshape(-1))
    yield "pcm16_stereo", wav_bytes(1, 2, 44100, 16, i16.tobytes())
    yield "pcm16_mono", wav_bytes(1, 1, 16000, 16, i16[:, 0].copy().tobytes())
    yield "pcm32_mono", wav_bytes(1, 1, 48000, 32, i32.tobytes())
    yield "pcm8_3ch", wav_bytes(1, 3, 8000, 8, u8.tobytes())
    yield "pcm24_stereo", wav_bytes(1, 2, 22050, 24, b24)
    yield "float32_stereo", wav_bytes(3, 2, 24000, 32, f32.tobytes())
    yield "float64_unsupported_width", wav_bytes(3, 1, 24000, 64, rng.standard_normal(64).astype("<f8").tobytes())
    yield "fmt_ext_18", wav_bytes(1, 1, 24000, 16, i16[:, 0].copy().tobytes(), fmt_ext=b"\x00\x00")
    yield "list_before_data", wav_bytes(1, 1, 24000, 16, i16[:, 0].copy().tobytes(), extra_before=chunk(b"LIST", b"INFOabcd1234"))
    yield "odd_chunk_no_pad", wav_bytes(1, 1, 24000, 16, i16[:, 0].copy().tobytes(), extra_before=chunk(b"junk", b"abc"))
    yield "chunk_after_data", wav_bytes(1, 1, 24000, 16, i16[:, 0].copy().tobytes(), extra_after=chunk(b"id3 ", b"x" * 10))
    yield "two_data_chunks", wav_bytes(1, 1, 24000, 16, i16[:100, 0].copy().tobytes(), extra_after=chunk(b"data", i16[100:300, 1].copy().tobytes()))
    yield "truncated_data", wav_bytes(1, 1, 24000, 16, i16[:50, 0].copy().tobytes(), data_size=400)
    yield "partial_frame", wav_bytes(1, 2, 24000, 16, i16.tobytes()[:-3])
    yield "extensible_rejected", wav_bytes(0xFFFE, 1, 24000, 16, i16[:, 0].copy().tobytes(), fmt_ext=b"\x16\x00" + b"\x00" * 22)
    yield "zero_channels", wav_bytes(1, 0, 24000, 16, i16[:, 0].copy().tobytes())
    yield "zero_rate", wav_bytes(1, 1, 0, 16, i16[:, 0].copy().tobytes())
    yield "empty_data", wav_bytes(1, 1, 24000, 16, b"")
    yield "no_data_chunk", b"RIFF" + struct.pack("<I", 28) + b"WAVE" + chunk(b"fmt ", struct.pack("<HHIIHH", 1, 1, 24000, 48000, 2, 16))
    yield "bad_riff", wav_bytes(1, 1, 24000, 16, i16[:, 0].copy().tobytes(), riff=b"RIFX")
    yield "bad_wave", wav_bytes(1, 1, 24000, 16, i16[:, 0].copy().tobytes(), wave=b"WAVX")
    yield "header_only", b"RIFF\x04\x00\x00\x00WAVE"
    yield "tiny", b"RI"


def test_read_wav_matches_reference(ref, tmp_path):
    import q3tts
    seen_ok = seen_fail = 0
    for name, blob in pcm_cases():
        p = str(tmp_path / (name + ".wav"))
        open(p, "wb").write(blob)
        want = ref_read(ref, p)
        got = q3tts.read_wav(p)
        if want is None:
            assert got is None, name
            seen_fail += 1
        else:
            assert got is not None, name
            assert got[1] == want[1], name
            assert got[0].shape == want[0].shape and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), name
            seen_ok += 1
    assert seen_ok >= 12 and seen_fail >= 8
    assert q3tts.read_wav(str(tmp_path / "missing.wav")) is None and ref_read(ref, str(tmp_path / "missing.wav")) is None


@pytest.mark.parametrize("src,dst", [(16000, 24000), (44100, 24000), (48000, 24000), (8000, 24000), (22050, 24000), (24000, 24000), (24000, 16000), (11025, 48000)])
def test_resample_matches_reference(ref, src, dst):
    import q3tts
    rng = np.random.default_rng(src + dst)
    for n in (1, 2, 3, 1000, 12345):
        a = rng.standard_normal(n).astype(np.float32)
        out = np.zeros(int(n * dst / src) + 8, np.float32)
        m = ref.ref_resample(a.ctypes.data, n, src, dst, out.ctypes.data, out.size)
        got = q3tts.resample(a, src, dst)
        assert got.size == m, (n, src, dst)
        assert np.array_equal(got.view(np.uint32), out[:m].view(np.uint32)), (n, src, dst)


def signals():
    rng = np.random.default_rng(3)
    t = np.arange(60000) / 24000.0
    yield "speechlike", (0.3 * np.sin(2 * np.pi * 180 * t) + 0.2 * np.sin(2 * np.pi * 1250 * t + 1.0) + 0.05 * np.sin(2 * np.pi * 7100 * t)
                          + 0.01 * rng.standard_normal(t.size)).astype(np.float32)
    yield "noise", (0.1 * rng.standard_normal(30000)).astype(np.float32)
    yield "silence", np.zeros(5000, np.float32)
    yield "short", (0.5 * rng.standard_normal(100)).astype(np.float32)
    yield "exact_window", (0.5 * rng.standard_normal(1024)).astype(np.float32)
    yield "window_plus_one", (0.5 * rng.standard_normal(1025)).astype(np.float32)
    yield "one_hop_more", (0.5 * rng.standard_normal(1280)).astype(np.float32)
    yield "impulse", np.concatenate([np.zeros(3000), [1.0], np.zeros(3000)]).astype(np.float32)
    yield "loud_clipped", np.clip(3.0 * rng.standard_normal(20000), -1, 1).astype(np.float32)


def test_log_mel_matches_reference(ref):
    """Tolerance: |E - E_ref| <= 1e-4 E_ref + 1e-6 sqrt(E_ref E_max) + 1e-10 on the mel energies E = exp(logmel) - 1e-10
    (fp32 FFTs with different butterfly order: errors scale with the frame's peak, not with the bin)."""
    import q3tts
    for name, a in signals():
        cap = 128 * (a.size // 256 + 2)
        out = np.zeros(cap, np.float32)
        k = ref.ref_mel(a.ctypes.data, a.size, out.ctypes.data, cap)
        frames = k // 128
        want = out[:k].reshape(128, frames)
        got = q3tts.log_mel(a)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert frames == (1 if a.size < 1024 else (a.size - 1024) // 256 + 1)
        e_w = np.exp(want.astype(np.float64)) - 1e-10
        e_g = np.exp(got.astype(np.float64)) - 1e-10
        emax = e_w.max(axis=0, keepdims=True)
        tol = 1e-4 * np.abs(e_w) + 1e-6 * np.sqrt(np.abs(e_w) * emax) + 1e-10
        bad = np.abs(e_g - e_w) > tol
        assert not bad.any(), (name, int(bad.sum()), float(np.abs(e_g - e_w).max()))
        # where a band carries real energy the log values themselves agree to 1e-3
        strong = e_w > 1e-6 * emax
        assert np.abs(got - want)[strong].max() < 1e-3, name
    assert q3tts.log_mel(np.zeros(0, np.float32)).shape == (128, 0)
    assert ref.ref_mel(None, 0, None, 0) == 0


def test_host_parsers_survive_garbage_under_sanitizers():
    """AddressSanitizer + UBSan build of the tokenizer-file readers, the WAV reader, the resampler and the mel extractor, driven with
    seeded mutations of well-formed files (tools/host_sanitize.cpp): they parse what users supply, so no input may make them touch
    memory they do not own.  CPU only (GPU sanitizers are not available on the pool)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "host_sanitize.sh"), "150"], capture_output=True, timeout=600)   # stderr echoes garbage bytes
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert b"no sanitizer report" in r.stdout
