"""GPU parity of the 12 Hz codec decoder (reference run_vocoder, src/tts_onnx.cpp:759-776) and of the
whole synthesize_tokens pipeline, through the C-ABI.  north_star tolerance: PCM within 1e-4 RMS;
the fp32-MFMA path is expected (and asserted) to be ~10x tighter."""
import numpy as np
import pytest

import q3_oracle as qo
from util import frame_tokens, load_gold, tiny_pair, to_osampling

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    eng, orc, w = tiny_pair(seed=2, max_batch=3, max_ctx=192)
    yield eng, orc, w
    eng.close()
    orc.close()


@pytest.mark.parametrize("F", [1, 3, 7])
def test_goldens_code2wav(pair, F):
    eng, _, w0 = pair
    w, d = load_gold("hf_code2wav.npz")
    eng.load({**w0, **w})
    pcm = eng.codec_decode(d[f"codes_{F}"])
    ref = d[f"pcm_{F}"]
    assert pcm.shape == ref.shape
    rms = float(np.sqrt(np.mean((pcm - ref) ** 2)))
    assert rms < 2e-5 and np.abs(pcm - ref).max() < 2e-4, (rms, np.abs(pcm - ref).max())
    eng.load(w0)


@pytest.mark.parametrize("F", [1, 2, 5, 17, 40, 130])
def test_codec_vs_oracle(pair, F):
    """Lengths chosen to straddle the 64-row GEMM tiles, the 4-token attention window and the
    transposed-conv phase boundaries."""
    eng, orc, _ = pair
    rng = np.random.default_rng(F)
    codes = rng.integers(0, eng.cfg.cd_codebook, (F, eng.cfg.n_groups)).astype(np.int64)
    pcm = eng.codec_decode(codes)
    ref = orc.vocoder(codes)
    assert pcm.shape == ref.shape == (eng.codec_decode_len(F),)
    err = pcm - ref
    rms = float(np.sqrt(np.mean(err ** 2)))
    assert rms < 2e-5, (F, rms, float(np.abs(err).max()), float(np.sqrt(np.mean(ref ** 2))))
    assert float(np.sqrt(np.mean(ref ** 2))) > 1e-3   # not silent


def test_synthesize_batch_vs_oracle(pair):
    """TTSEngine::synthesize_tokens (tts_onnx.cpp:405-436) end to end for a ragged batch > max_batch."""
    import q3tts
    eng, orc, _ = pair
    sp = q3tts.Sampling(temperature=0.9, top_p=0.95, top_k=40, max_new_tokens=20)
    texts = [[5, 6, 7, 8], [9], [100, 200, 300, 400, 500, 600, 700], [1, 2], [3, 3, 3, 3, 3]]
    toks = [frame_tokens(t) for t in texts]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=1, seed=77, ignore_eos=False)
    for u, t in enumerate(toks):
        po = orc.build_prompt(t, 1)
        ref_codes = orc.generate(po, to_osampling(sp), seed=77, stream=u, cp_cached=True, ignore_eos=False)
        assert nfr[u] == len(ref_codes), u
        assert np.array_equal(codes[u], ref_codes), u
        if len(ref_codes) == 0:
            assert len(pcm[u]) == 0     # reference returns an empty vector (tts_onnx.cpp:418)
            continue
        ref_pcm = orc.vocoder(ref_codes)
        assert pcm[u].shape == ref_pcm.shape
        assert float(np.sqrt(np.mean((pcm[u] - ref_pcm) ** 2))) < 1e-4
