"""Batched decode (M > 8 rows -> bf16-MFMA skinny GEMM with hi/lo-split activations) against the oracle:
configs[2]-style batches at test size, and a 16-utterance batch at 0.6B dims."""
import numpy as np
import pytest

import q3_oracle as qo
from util import frame_tokens, tiny_pair, to_ocfg, to_osampling

pytestmark = pytest.mark.gpu


def test_batch12_generation_medium_config():
    import q3tts
    eng, orc, _ = tiny_pair(seed=6, max_batch=12, max_ctx=160, ocfg=qo.config_medium())
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=40)
    rng = np.random.default_rng(0)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(1, 24, 12)]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=21, ignore_eos=False)
    n_eos = 0
    for u, t in enumerate(toks):
        ref = orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=21, stream=u, cp_cached=True, ignore_eos=False)
        assert nfr[u] == len(ref) and np.array_equal(codes[u], ref), (u, nfr[u], len(ref))
        n_eos += len(ref) < sp.max_new_tokens
        if len(ref):
            ref_pcm = orc.vocoder(ref)
            assert float(np.sqrt(np.mean((pcm[u] - ref_pcm) ** 2))) < 1e-4
    eng.close()
    orc.close()


def test_batch12_logits_tolerance_medium_config():
    """Teacher-forced single steps: the MFMA path's logits stay within fp32-summation noise of the oracle."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=7, max_batch=12, max_ctx=96, ocfg=qo.config_medium())
    rng = np.random.default_rng(3)
    x = rng.standard_normal((12, eng.cfg.hidden)).astype(np.float32)
    lg, lh = eng.prefill(x)            # S = 12 rows > 8 -> MFMA GEMM inside one slot's prefill
    lo, ho = orc.prefill(x)
    assert np.abs(lg - lo).max() < 2e-4 and np.abs(lh - ho).max() < 2e-4
    seq = rng.standard_normal((14, eng.cfg.hidden)).astype(np.float32)
    assert np.abs(eng.code_predictor(seq, 12) - orc.code_predictor(seq, 12)).max() < 2e-4
    eng.close()
    orc.close()


def test_batch16_full_size_greedy():
    """0.6B dims, 16 utterances in one batch, greedy, 4 frames: codec ids bit-exact vs the oracle."""
    import q3tts
    cfg = q3tts.default_config("0.6b")
    eng = q3tts.Engine(cfg, device=0, max_batch=16, max_ctx=128)
    eng.fill_synthetic(seed=0)
    orc = qo.Oracle(to_ocfg(cfg), max_ctx=48)
    for name, shape in eng.tensor_infos():
        if not name.startswith("cd."):
            orc.set_tensor(name, eng.get_tensor(name, shape))
    sp = q3tts.Sampling(temperature=1.0, top_p=1.0, top_k=1, max_new_tokens=4)
    rng = np.random.default_rng(5)
    toks = [frame_tokens(rng.integers(0, 151643, 16)) for _ in range(16)]
    for b, t in enumerate(toks):
        p, tr = eng.build_prompt(t, 0)
        eng.slot_begin(b, p, tr, sp, seed=2, stream_id=b, ignore_eos=True)
    assert eng.decode_steps(4) == 0
    bad = []
    for b, t in enumerate(toks):
        got = eng.slot_codes(b)
        ref = orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=2, stream=b, cp_cached=True, ignore_eos=True)
        if not np.array_equal(got, ref):
            bad.append(b)
    assert not bad, bad
    eng.close()
    orc.close()


def test_continuous_batching_is_schedule_independent():
    """11 ragged utterances (EOS live) through 3 slots: a finished slot is re-armed with the next utterance while its codes go to an
    asynchronous vocoder lane.  Every utterance equals the oracle's one-at-a-time result, and the 11-slot run, bit for bit."""
    import q3tts
    eng, orc, w = tiny_pair(seed=18, max_batch=3, max_ctx=96)
    sp = q3tts.Sampling(temperature=0.9, top_p=0.95, top_k=40, max_new_tokens=40)
    rng = np.random.default_rng(11)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(1, 20, 11)]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=5, ignore_eos=False)
    lens = set()
    for u, t in enumerate(toks):
        ref = orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=5, stream=u, cp_cached=True, ignore_eos=False)
        assert nfr[u] == len(ref) and np.array_equal(codes[u], ref), u
        lens.add(len(ref))
        if len(ref):
            ref_pcm = orc.vocoder(ref)
            assert pcm[u].shape == ref_pcm.shape and float(np.sqrt(np.mean((pcm[u] - ref_pcm) ** 2))) < 1e-4, u
        else:
            assert len(pcm[u]) == 0
    assert len(lens) > 3                      # the lengths really are ragged
    eng.close()
    wide = q3tts.Engine(eng.cfg, device=0, max_batch=11, max_ctx=96)
    wide.load(w)
    pcm2, codes2, nfr2 = wide.synthesize_batch(toks, sp, lang=0, seed=5, ignore_eos=False)
    for u in range(11):
        assert np.array_equal(codes2[u], codes[u]) and np.array_equal(pcm2[u], pcm[u]), u
    wide.close()
    orc.close()


def test_per_utterance_caps_fix_ragged_lengths():
    """q3tts_synthesize_schedule_host: max_new_per_utt with EOS suppressed gives exactly those lengths, each utterance a prefix of its
    uncapped generation (7 utterances through 2 slots)."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=19, max_batch=2, max_ctx=96)
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=30)
    rng = np.random.default_rng(13)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(1, 12, 7)]
    caps = np.array([5, 30, 1, 17, 9, 30, 12], np.int32)
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=2, ignore_eos=True, max_new_per_utt=caps)
    assert np.array_equal(nfr, caps)
    for u, t in enumerate(toks):
        ref = orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=2, stream=u, cp_cached=True, ignore_eos=True)
        assert np.array_equal(codes[u], ref[: caps[u]]), u
        assert len(pcm[u]) == eng.codec_decode_len(int(caps[u]))
    with pytest.raises(ValueError):
        eng.synthesize_batch(toks, sp, max_new_per_utt=caps[:3])
    eng.close()
    orc.close()


def test_scattered_slots_share_one_prefill_pass():
    """Slots 0, 2, 4, 6 finish at the same look (caps 6 / 20 alternate), so the scheduler re-arms four non-consecutive slots at once:
    their prompts go through the talker stack as one row block addressed through a slot map.  Every utterance still equals the oracle."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=23, max_batch=8, max_ctx=96, ocfg=qo.config_medium())
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=20)
    rng = np.random.default_rng(17)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(2, 14, 24)]
    caps = np.array([6, 20] * 12, np.int32)
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=1, seed=12, ignore_eos=True, max_new_per_utt=caps)
    assert np.array_equal(nfr, caps)
    for u, t in enumerate(toks):
        ref = orc.generate(orc.build_prompt(t, 1), to_osampling(sp), seed=12, stream=u, cp_cached=True, ignore_eos=True)
        assert np.array_equal(codes[u], ref[: caps[u]]), u
    eng.close()
    orc.close()


def test_more_than_128_rows_per_projection():
    """70 utterances in one batch: predictor pass 0 has 140 rows, which the matrix-core GEMM walks as 128-row blocks (the slabs keep
    their [slice][all rows] shape).  Codes of utterances on both sides of the block boundary bit-exact vs the oracle."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=31, max_batch=70, max_ctx=64, ocfg=qo.config_medium())
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=10)
    rng = np.random.default_rng(19)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(1, 10, 70)]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=44, ignore_eos=True)
    for u in (0, 31, 62, 63, 64, 65, 69):
        ref = orc.generate(orc.build_prompt(toks[u], 0), to_osampling(sp), seed=44, stream=u, cp_cached=True, ignore_eos=True)
        assert nfr[u] == 10 and np.array_equal(codes[u], ref), u
    eng.close()
    orc.close()
