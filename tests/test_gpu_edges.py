"""Edge cases of the path through the C-ABI: empty text, one-frame generation, more utterances than slots, the error
contract (message + engine still usable), input validation of the session-shaped calls."""
import numpy as np
import pytest

import q3_oracle as qo
from util import IM_END, IM_START, ASSISTANT, TTS_BOS, TTS_EOS, frame_tokens, tiny_pair, to_osampling

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    eng, orc, w = tiny_pair(seed=12, max_batch=3, max_ctx=64)
    yield eng, orc
    eng.close()
    orc.close()


def test_empty_text_follows_the_reference_indexing(pair):
    """synthesize(""): the 5-token frame.  The reference takes input_ids[3] (= TTS_EOS) as the first text token and
    leaves only [tts_eos] in the trailing block (tts_onnx.cpp:515-536); 4 ids are the least it can index."""
    import q3tts
    eng, orc = pair
    ids = frame_tokens([])
    assert list(ids) == [IM_START, ASSISTANT, TTS_BOS, TTS_EOS, IM_END]
    p, t = eng.build_prompt(ids, 0)
    po = orc.build_prompt(ids, 0)
    to, _ = orc.trailing()
    assert p.shape == po.shape == (8, eng.cfg.hidden) and t.shape == to.shape == (1, eng.cfg.hidden)
    assert np.abs(p - po).max() < 1e-5 and np.abs(t - to).max() < 1e-5
    sp = q3tts.Sampling(max_new_tokens=6)
    pcm, codes, _ = eng.synthesize_batch([ids], sp, lang=0, seed=3, ignore_eos=True)
    ref = orc.generate(po, to_osampling(sp), seed=3, stream=0, cp_cached=True, ignore_eos=True)
    assert np.array_equal(codes[0], ref)
    with pytest.raises(RuntimeError, match="too short"):
        eng.build_prompt(ids[:3], 0)
    with pytest.raises(RuntimeError):
        orc.build_prompt(ids[:3], 0)


def test_single_frame_and_more_utterances_than_slots(pair):
    import q3tts
    eng, orc = pair
    rng = np.random.default_rng(5)
    toks = [frame_tokens(rng.integers(0, 1000, n)) for n in (1, 7, 2, 30, 3, 4, 11)]      # 7 utterances, 3 slots
    for max_new in (1, 5):
        sp = q3tts.Sampling(temperature=0.9, top_p=0.9, top_k=20, max_new_tokens=max_new)
        pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=4, seed=21, ignore_eos=True)
        assert list(nfr) == [max_new] * len(toks)
        for u, t in enumerate(toks):                 # utterance u draws from stream u wherever it was scheduled
            ref = orc.generate(orc.build_prompt(t, 4), to_osampling(sp), seed=21, stream=u, cp_cached=True, ignore_eos=True)
            assert np.array_equal(codes[u], ref), (max_new, u)
            ref_pcm = orc.vocoder(ref)
            assert pcm[u].shape == ref_pcm.shape and float(np.sqrt(np.mean((pcm[u] - ref_pcm) ** 2))) < 1e-4


def test_errors_carry_a_message_and_leave_the_engine_usable(pair):
    import q3tts
    eng, orc = pair
    H = eng.cfg.hidden
    ids = frame_tokens([1, 2, 3])
    good, _, _ = eng.synthesize_batch([ids], q3tts.Sampling(max_new_tokens=3), seed=1, ignore_eos=True)
    with pytest.raises(RuntimeError, match="exceeds max_ctx"):
        eng.synthesize_batch([ids], q3tts.Sampling(max_new_tokens=64), seed=1)
    with pytest.raises(RuntimeError, match="out of range"):
        eng.text_project([eng.cfg.text_vocab])
    with pytest.raises(RuntimeError, match="out of range"):
        eng.codec_embed([-1])
    with pytest.raises(RuntimeError, match="out of range"):
        eng.codec_decode(np.full((2, eng.cfg.n_groups), eng.cfg.cd_codebook, np.int64))
    with pytest.raises(RuntimeError, match="out of range"):
        eng.codec_decode(np.zeros((0, eng.cfg.n_groups), np.int64))
    with pytest.raises(RuntimeError, match="slot out of range"):
        eng.slot_begin(3, np.zeros((8, H), np.float32), np.zeros((1, H), np.float32), q3tts.Sampling(max_new_tokens=2))
    with pytest.raises(RuntimeError, match="too long"):
        eng.build_prompt(frame_tokens(np.ones(1100, np.int64)), 0)
    with pytest.raises(RuntimeError, match="unknown tensor"):
        eng.set_tensor("talker.layers.99.q_proj", np.zeros(4, np.float32))
    with pytest.raises(RuntimeError, match="expected"):
        eng.set_tensor("talker.norm", np.zeros(H + 1, np.float32))
    again, _, _ = eng.synthesize_batch([ids], q3tts.Sampling(max_new_tokens=3), seed=1, ignore_eos=True)
    assert np.array_equal(again[0], good[0])


def test_sampling_parameter_extremes(pair):
    """top_k larger than the vocabulary / 0, top_p = 1 and tiny, temperature 0 (= T 1, not greedy) and huge."""
    import q3tts
    eng, orc = pair
    rng = np.random.default_rng(9)
    logits = (rng.standard_normal(eng.cfg.vocab) * 3).astype(np.float32)
    for kw in (dict(top_k=0, top_p=1.0, temperature=1.0), dict(top_k=100000, top_p=1.0, temperature=0.7), dict(top_k=3, top_p=1e-6, temperature=1.0),
               dict(top_k=50, top_p=0.95, temperature=0.0), dict(top_k=50, top_p=0.5, temperature=1e6), dict(top_k=1, top_p=0.1, temperature=5.0)):
        sp = q3tts.Sampling(max_new_tokens=1, **kw)
        for u in (0.0, 0.3, 0.999999):
            assert eng.sample(logits, sp, u) == orc.sample(logits, to_osampling(sp), u), (kw, u)


def test_scheduler_error_leaves_the_engine_usable():
    """A job with an invalid utterance (fewer than the 4 ids the reference indexes, or an id outside the text vocabulary) fails as a whole
    with the reason; the next job on the same handle runs normally and matches the oracle."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=27, max_batch=2, max_ctx=64)
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=6)
    good = frame_tokens([4, 5, 6])
    with pytest.raises(RuntimeError, match="too short"):
        eng.synthesize_batch([good, np.array([1, 2, 3], np.int64), good], sp)
    with pytest.raises(RuntimeError, match="text id out of range"):
        eng.synthesize_batch([good, frame_tokens([10 ** 7])], sp)
    pcm, codes, nfr = eng.synthesize_batch([good, good, good], sp, seed=3, ignore_eos=True)
    for u in range(3):
        ref = orc.generate(orc.build_prompt(good, 0), to_osampling(sp), seed=3, stream=u, cp_cached=True, ignore_eos=True)
        assert np.array_equal(codes[u], ref), u
    eng.close()
    orc.close()


def test_corrupt_weight_files_are_rejected(tmp_path):
    """The weight file is user input: a wrong magic, a truncated file, an absurd element count or an unknown tensor name end in an error
    message, never in a huge allocation or a partially loaded engine being used."""
    import struct
    import q3tts
    eng, orc, _ = tiny_pair(seed=29, max_batch=1, max_ctx=64)
    good = tmp_path / "good.q3w"
    eng.save_weights(str(good))
    raw = good.read_bytes()
    cfg_bytes = struct.unpack("<I", raw[8:12])[0]
    first = 8 + 4 + cfg_bytes + 4                       # offset of the first tensor record: u16 name length, name, u8 dtype, u64 numel
    nl = struct.unpack("<H", raw[first:first + 2])[0]
    numel_off = first + 2 + nl + 1
    cases = {
        "magic": b"XXXXXXXX" + raw[8:],
        "truncated": raw[: len(raw) // 3],
        "numel": raw[:numel_off] + struct.pack("<Q", 1 << 60) + raw[numel_off + 8:],
        "name": raw[:first + 2] + b"Z" * nl + raw[first + 2 + nl:],
        "cfg": raw[:8] + struct.pack("<I", 4096) + raw[12:],
    }
    for tag, blob in cases.items():
        p = tmp_path / f"{tag}.q3w"
        p.write_bytes(blob)
        e2 = q3tts.Engine(eng.cfg, device=0, max_batch=1, max_ctx=64)
        with pytest.raises(RuntimeError):
            e2.load_weights(str(p))
        e2.close()
    e3 = q3tts.Engine(eng.cfg, device=0, max_batch=1, max_ctx=64)
    e3.load_weights(str(good))
    ids = frame_tokens([3, 4, 5])
    assert np.array_equal(e3.build_prompt(ids, 0)[0], eng.build_prompt(ids, 0)[0])
    e3.close()
    eng.close()
    orc.close()
