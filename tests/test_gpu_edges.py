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


@pytest.mark.parametrize("fail_at,caps", [(2, None), (1, (9, 9, 8, 8, 3)), (4, (9, 2, 8, 1, 3))])
def test_vocoder_phase_error_leaves_the_engine_usable(fail_at, caps):
    """A failure INSIDE the vocoder phase (here injected at the fail_at-th lane submit; in the field an arena or pinned-buffer allocation
    failing) used to leave lanes busy with the failed job's pcm_out / pcm_len pointers: the next job's drain then wrote through freed
    host memory.  Now the job fails as a whole with the reason, nothing is delivered, and the next job on the same handle matches the
    oracle.  (This config's decoder is too narrow for the batched kernels: one utterance per lane.  The batched-group path is covered at
    0.6B dims by tests/test_gpu_full.py::test_vocoder_group_failure_leaves_the_engine_usable.)"""
    import gc
    import os
    import q3tts
    eng, orc, _ = tiny_pair(seed=33, max_batch=3, max_ctx=64, flags=q3tts.FLAG_TEST_HOOKS)   # only such an engine reads the injection variable
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=9)
    rng = np.random.default_rng(41)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in (3, 5, 2, 7, 4)]
    mx = None if caps is None else np.array(caps, np.int32)
    os.environ["Q3TTS_TEST_FAIL_VOCODER_SUBMIT"] = str(fail_at)
    try:
        with pytest.raises(RuntimeError, match="injected"):
            eng.synthesize_batch(toks, sp, seed=8, ignore_eos=True, max_new_per_utt=mx)
    finally:
        del os.environ["Q3TTS_TEST_FAIL_VOCODER_SUBMIT"]
    gc.collect()                      # the failed job's PCM buffers are gone: a stale pending item would now be a use-after-free
    junk = [np.full(200000, 7.0, np.float32) for _ in range(8)]
    pcm, codes, nfr = eng.synthesize_batch(toks, sp, seed=8, ignore_eos=True, max_new_per_utt=mx)
    assert all(np.all(j == 7.0) for j in junk)
    for u, t in enumerate(toks):
        ref = orc.generate(orc.build_prompt(t, 0), to_osampling(sp), seed=8, stream=u, cp_cached=True, ignore_eos=True)
        n = 9 if caps is None else caps[u]
        assert nfr[u] == n and np.array_equal(codes[u], ref[:n]), u
        ref_pcm = orc.vocoder(ref[:n])
        assert pcm[u].shape == ref_pcm.shape and float(np.sqrt(np.mean((pcm[u] - ref_pcm) ** 2))) < 1e-4, u
    eng.close()
    orc.close()


def test_speaker_embedding_length_is_checked():
    """build_prompts copies `hidden` floats from a speaker row: a shorter caller-supplied vector must be refused, not read out of bounds."""
    import q3tts
    eng, orc, _ = tiny_pair(seed=35, max_batch=1, max_ctx=64)
    ids = frame_tokens([5, 6, 7])
    sp = q3tts.Sampling(max_new_tokens=2)
    short = np.zeros(eng.cfg.hidden - 8, np.float32)
    with pytest.raises(ValueError, match="speaker embedding"):
        eng.synthesize_batch([ids], sp, speakers=[short])
    with pytest.raises(ValueError, match="speaker embedding"):
        eng.build_prompt(ids, 0, speaker=short)
    with pytest.raises(ValueError, match="per utterance"):
        eng.synthesize_batch([ids, ids], sp, speakers=[None])
    bad = q3tts.Config.from_dict(dict(eng.cfg.to_dict(), spk_enc_dim=eng.cfg.hidden // 2))
    with pytest.raises(RuntimeError, match="spk_enc_dim"):
        q3tts.Engine(bad, device=0, max_batch=1, max_ctx=32)
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
    # record boundaries: u16 name length, name, u8 dtype, u64 numel, payload (2 or 4 bytes per element)
    recs, off = [], first
    while off < len(raw):
        n_l = struct.unpack("<H", raw[off:off + 2])[0]
        dt = raw[off + 2 + n_l]
        ne = struct.unpack("<Q", raw[off + 3 + n_l:off + 11 + n_l])[0]
        end = off + 11 + n_l + ne * (2 if dt == 1 else 4)
        recs.append((off, end))
        off = end
    n_off = first - 4
    n_rec = struct.unpack("<I", raw[n_off:n_off + 4])[0]
    assert n_rec == len(recs) and recs[-1][1] == len(raw)
    # a shorter tensor list (count patched, records dropped) and a duplicated record in place of its successor both used to load
    # "successfully" and leave tensors as uninitialised HBM
    cases["short_list"] = raw[:n_off] + struct.pack("<I", n_rec - 3) + raw[first:recs[-4][1]]
    cases["duplicate"] = raw[:recs[1][0]] + raw[recs[0][0]:recs[0][1]] + raw[recs[1][1]:]
    cases["one_missing"] = raw[:n_off] + struct.pack("<I", n_rec - 1) + raw[first:recs[4][0]] + raw[recs[4][1]:]
    for tag, blob in cases.items():
        p = tmp_path / f"{tag}.q3w"
        p.write_bytes(blob)
        e2 = q3tts.Engine(eng.cfg, device=0, max_batch=1, max_ctx=64)
        with pytest.raises(RuntimeError) as ei:
            e2.load_weights(str(p))
        if tag in ("short_list", "one_missing"):
            assert "missing" in str(ei.value), (tag, str(ei.value))
        if tag == "duplicate":
            assert "twice" in str(ei.value), str(ei.value)
        e2.close()
    e3 = q3tts.Engine(eng.cfg, device=0, max_batch=1, max_ctx=64)
    e3.load_weights(str(good))
    ids = frame_tokens([3, 4, 5])
    assert np.array_equal(e3.build_prompt(ids, 0)[0], eng.build_prompt(ids, 0)[0])
    e3.close()
    eng.close()
    orc.close()
