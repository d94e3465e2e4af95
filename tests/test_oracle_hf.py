"""Oracle vs. golden vectors generated from the `transformers` implementation of the public
architecture (tests/golden/make_hf_goldens.py).  The reference itself holds no golden vectors for
the network arithmetic: these fixtures pin the [HINT] layer math instead.
Tolerance: fp32 both sides, different summation order -> 2e-5 absolute on O(1) values."""
import os

import numpy as np
import pytest

import q3_oracle as qo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5


def load(name):
    z = np.load(os.path.join(GOLD, name))
    w = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    d = {k: z[k] for k in z.files if not k.startswith("w:")}
    return w, d


@pytest.fixture(scope="module")
def oracle():
    o = qo.Oracle(qo.config_tiny(), max_ctx=64)
    yield o
    o.close()


def test_talker_prefill_and_decode(oracle):
    w, d = load("hf_talker.npz")
    oracle.load(w)
    logits, lh = oracle.prefill(d["prefill_in"])
    assert np.abs(logits - d["prefill_logits"]).max() < TOL
    assert np.abs(lh - d["prefill_last_hidden"]).max() < TOL
    for i in range(d["decode_in"].shape[0]):
        lg, h = oracle.decode(d["decode_in"][i])
        assert np.abs(lg - d["decode_logits"][i]).max() < TOL, i
        assert np.abs(h - d["decode_last_hidden"][i]).max() < TOL, i


def test_code_predictor_reference_call_pattern(oracle):
    w, d = load("hf_predictor.npz")
    oracle.load(w)
    for j in range(d["logits"].shape[0]):
        lg = oracle.code_predictor(d["seq"][: j + 2], j)
        assert np.abs(lg - d["logits"][j]).max() < TOL, j


@pytest.mark.parametrize("F", [1, 3, 7])
def test_code2wav(oracle, F):
    w, d = load("hf_code2wav.npz")
    oracle.load(w)
    pcm = oracle.vocoder(d[f"codes_{F}"])
    ref = d[f"pcm_{F}"]
    assert pcm.shape == ref.shape
    assert np.abs(pcm - ref).max() < 2e-4
    rms = float(np.sqrt(np.mean((pcm - ref) ** 2)))
    assert rms < 2e-5, rms  # north_star PCM tolerance is 1e-4 RMS
    assert float(np.sqrt(np.mean(ref ** 2))) > 0.05  # fixture is not silent / not saturated


@pytest.mark.parametrize("T", [5, 9, 40])
def test_speaker_encoder(oracle, T):
    """ECAPA-TDNN speaker encoder vs transformers' ECAPA_TimeDelayNet."""
    w, d = load("hf_speaker.npz")
    oracle.load(w)
    out = oracle.speaker_encoder(d[f"mel_{T}"])
    ref = d[f"embed_{T}"]
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-4 * max(1.0, float(np.abs(ref).max())), float(np.abs(out - ref).max())
    assert float(np.abs(ref).max()) > 0.05


def test_speaker_encoder_needs_five_frames(oracle):
    w, d = load("hf_speaker.npz")
    oracle.load(w)
    with pytest.raises(RuntimeError):
        oracle.speaker_encoder(d["mel_5"][:, :4])
