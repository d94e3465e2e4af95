"""SURVEY.md 8f-2 on the GPU: the ECAPA-TDNN speaker encoder (q3tts_speaker_encoder_host), the wav -> embedding
chain (q3tts_extract_speaker_embedding_host) and clone synthesis (q3tts_synthesize_clone_batch_host) against the
CPU oracle and the transformers golden.  Tolerances: embedding 1e-4 of its max magnitude (fp32 both sides,
different summation order); generated codes bit-exact; PCM 1e-4 RMS."""
import os
import struct
import subprocess

import numpy as np
import pytest

import q3_oracle as qo
from util import frame_tokens, load_gold, tiny_pair, to_osampling, to_q3cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "leaxer-qwen3-tts_amd", "leaxer-tts")


def close(a, b, rel=1e-4):
    return float(np.abs(a - b).max()) <= rel * max(1.0, float(np.abs(b).max()))


def write_wav16(path, samples, rate, channels=1):
    pcm = (np.clip(samples, -1, 1) * 32767).astype("<i2")
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * 2, channels * 2, 16) + \
        b"data" + struct.pack("<I", pcm.size * 2) + pcm.tobytes()
    open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)


def voice(seconds, rate, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * rate)) / rate
    f0 = 120 + 40 * np.sin(2 * np.pi * 0.7 * t)
    x = sum(a * np.sin(2 * np.pi * k * np.cumsum(f0) / rate) for k, a in ((1, 0.3), (2, 0.2), (3, 0.12), (5, 0.05)))
    return (x * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t)) + 0.01 * rng.standard_normal(t.size)).astype(np.float32)


def test_speaker_encoder_vs_transformers_golden():
    w, d = load_gold("hf_speaker.npz")
    eng, orc, _ = tiny_pair(seed=2, max_batch=1, max_ctx=64, extra=w)
    assert eng.has_speaker_encoder
    for T in (5, 9, 40):
        got = eng.speaker_encoder(d[f"mel_{T}"])
        assert close(got, d[f"embed_{T}"]), (T, float(np.abs(got - d[f"embed_{T}"]).max()))
    with pytest.raises(RuntimeError, match="at least 5 mel frames"):
        eng.speaker_encoder(d["mel_5"][:, :4])
    eng.close()
    orc.close()


def test_speaker_encoder_vs_oracle_random_weights_and_lengths():
    eng, orc, _ = tiny_pair(seed=6, max_batch=1, max_ctx=64)
    rng = np.random.default_rng(1)
    for T in (5, 16, 17, 63, 200, 1001):
        mel = (2.0 * rng.standard_normal((128, T)) - 4.0).astype(np.float32)
        got, want = eng.speaker_encoder(mel), orc.speaker_encoder(mel)
        assert close(got, want), (T, float(np.abs(got - want).max()), float(np.abs(want).max()))
    eng.close()
    orc.close()


def test_extract_and_clone_synthesis_vs_oracle(tmp_path):
    import q3tts
    eng, orc, _ = tiny_pair(seed=8, max_batch=2, max_ctx=96)
    wav = str(tmp_path / "ref.wav")
    write_wav16(wav, np.stack([voice(1.3, 16000, 0), voice(1.3, 16000, 1)], 1).reshape(-1), 16000, channels=2)
    # the chain of extract_speaker_embedding (tts_onnx.cpp:331-365), piece by piece through the host entry points ...
    audio, sr = q3tts.read_wav(wav)
    assert sr == 16000 and audio.size == int(1.3 * 16000)
    mel = q3tts.log_mel(q3tts.resample(audio, sr, 24000))
    want = orc.speaker_encoder(mel)
    # ... equals the one-call form
    spk = eng.extract_speaker_embedding(wav)
    assert close(spk, want), float(np.abs(spk - want).max())
    with pytest.raises(RuntimeError, match="Failed to read audio"):
        eng.extract_speaker_embedding(str(tmp_path / "missing.wav"))

    # synthesize_clone: speaker row spliced before CODEC_BOS (tts_onnx.cpp:481-498) -> one more prompt row
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=10)
    toks = [frame_tokens([11, 22, 33, 44]), frame_tokens([5, 6, 7])]
    spk_b = qo.bf16_round(spk)   # any embedding works; a bf16-representable one keeps both sides on identical inputs
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=2, seed=4, ignore_eos=True, speakers=[spk_b, None])
    for u, t in enumerate(toks):
        prompt = orc.build_prompt(t, 2, speaker=spk_b if u == 0 else None)
        assert prompt.shape[0] == (10 if u == 0 else 9)
        ref = orc.generate(prompt, to_osampling(sp), seed=4, stream=u, cp_cached=True, ignore_eos=True)
        assert np.array_equal(codes[u], ref), u
        ref_pcm = orc.vocoder(ref)
        assert pcm[u].shape == ref_pcm.shape and float(np.sqrt(np.mean((pcm[u] - ref_pcm) ** 2))) < 1e-4
    # the speaker row changes what is generated
    pcm0, codes0, _ = eng.synthesize_batch(toks[:1], sp, lang=2, seed=4, ignore_eos=True)
    assert not np.array_equal(codes0[0], codes[0])
    eng.close()
    orc.close()


def test_full_size_speaker_encoder_vs_oracle():
    """0.6B dims (512/1536 channels -> 1024) on seeded weights; 3 s of reference audio."""
    import q3tts
    cfg = q3tts.default_config("0.6b")
    ocfg = qo.Config.from_dict(cfg.to_dict())
    eng = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=64)
    eng.fill_synthetic(seed=0)
    rng = np.random.default_rng(5)
    orc = qo.Oracle(ocfg, max_ctx=8)
    for name, shape, kind in qo.tensor_specs(ocfg):
        if not name.startswith("spk."):
            continue
        fan = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        a = qo.bf16_round(rng.standard_normal(shape).astype(np.float32) * (1.0 / np.sqrt(fan) if kind == "w" else 0.1))
        eng.set_tensor(name, a)
        orc.set_tensor(name, a)
    eng.finalize()
    mel = q3tts.log_mel(voice(3.0, 24000, 2))
    got, want = eng.speaker_encoder(mel), orc.speaker_encoder(mel)
    assert got.shape == (1024,) and close(got, want), (float(np.abs(got - want).max()), float(np.abs(want).max()))
    eng.close()
    orc.close()


def test_cli_ref_flag(tmp_path):
    """--ref WAV through the reference-shaped CLI == the same clone synthesis through the Python binding."""
    import q3tts
    from test_gpu_cli import read_wav16
    eng, orc, _ = tiny_pair(seed=9, max_batch=1, max_ctx=96)
    mdir = tmp_path / "m"
    mdir.mkdir()
    eng.save_weights(str(mdir / "model.q3w"))
    wav = str(tmp_path / "ref.wav")
    write_wav16(wav, voice(1.0, 24000, 3), 24000)
    out = str(tmp_path / "o.wav")
    r = subprocess.run([CLI, "-m", str(mdir), "--tokens", "11,22,33", "--ref", wav, "-o", out, "--max-tokens", "8", "--seed", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    sp = q3tts.Sampling(max_new_tokens=8)
    pcm, _, _ = eng.synthesize_batch([frame_tokens([11, 22, 33])], sp, lang=0, seed=2, speakers=[eng.extract_speaker_embedding(wav)])
    want16 = (np.clip(pcm[0], -1, 1) * 32767.0).astype(np.int16)
    got16 = read_wav16(out)
    assert got16.shape == want16.shape and np.abs(got16.astype(int) - want16.astype(int)).max() <= 1
    r = subprocess.run([CLI, "-m", str(mdir), "--tokens", "11", "--ref", str(tmp_path / "none.wav"), "-o", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Failed to read audio" in r.stderr
    eng.close()
    orc.close()
