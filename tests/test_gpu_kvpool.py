"""The talker's KV page pool through the C-ABI (q3tts_create_pooled / q3tts_kv_pool_info): the reference grows one KVCache per utterance
(tts_onnx.h:108-115, one token per run_decode); here slots take 64-token pages from a bounded pool.  Results must not depend on which
pages a slot happens to own or on how many utterances the pool admits at once."""
import numpy as np
import pytest

import q3_oracle as qo
from util import calibrate_codec, frame_tokens, to_osampling, to_q3cfg

pytestmark = pytest.mark.gpu


def _engine(w, ocfg, max_batch, max_ctx, pool_tokens):
    import q3tts
    eng = q3tts.Engine(to_q3cfg(ocfg), device=0, max_batch=max_batch, max_ctx=max_ctx, kv_pool_tokens=pool_tokens)
    eng.load(w)
    return eng


@pytest.fixture(scope="module")
def setup():
    ocfg = qo.config_tiny()
    w = calibrate_codec(qo.random_weights(ocfg, 31), ocfg)
    orc = qo.Oracle(ocfg, max_ctx=320, weights=w)
    yield ocfg, w, orc
    orc.close()


def test_pool_accounting_and_exhaustion(setup):
    import q3tts
    ocfg, w, orc = setup
    eng = _engine(w, ocfg, 3, 320, 4 * 64)                 # 4 pages for 3 slots of up to 5 pages each
    try:
        assert eng.kv_pool_info() == (64, 4, 4)
        ids = frame_tokens(np.arange(5))
        p, t = eng.build_prompt(ids, 0)
        sp2 = q3tts.Sampling(max_new_tokens=100)            # prompt 9 + 100 tokens: 2 pages
        eng.slot_begin(0, p, t, sp2, seed=1, stream_id=0, ignore_eos=True)
        assert eng.kv_pool_info() == (64, 4, 2)
        eng.slot_begin(1, p, t, sp2, seed=1, stream_id=1, ignore_eos=True)
        assert eng.kv_pool_info()[2] == 0
        with pytest.raises(RuntimeError, match="KV page pool exhausted"):
            eng.slot_begin(2, p, t, q3tts.Sampling(max_new_tokens=10), seed=1, stream_id=2, ignore_eos=True)
        assert eng.decode_steps(3) == 2                     # the refused slot was not armed; the other two are unharmed
        eng.slot_release(0)
        assert eng.kv_pool_info()[2] == 2
        eng.slot_begin(2, p, t, sp2, seed=1, stream_id=2, ignore_eos=True)      # takes the pages slot 0 gave back
        while eng.decode_steps(16) > 0:
            pass
        ref = {u: orc.generate(orc.build_prompt(ids, 0), to_osampling(sp2), seed=1, stream=u, cp_cached=True, ignore_eos=True) for u in (1, 2)}
        assert np.array_equal(eng.slot_codes(1), ref[1]) and np.array_equal(eng.slot_codes(2), ref[2])
        # re-arming a slot with a smaller cap gives pages back; session-shaped calls grow the slot's share on demand
        eng.slot_begin(1, p, t, q3tts.Sampling(max_new_tokens=10), seed=1, stream_id=1, ignore_eos=True)
        assert eng.kv_pool_info()[2] == 1
        for s in range(3):
            eng.slot_release(s)
        assert eng.kv_pool_info() == (64, 4, 4)
        x = np.random.default_rng(0).standard_normal((70, ocfg.hidden)).astype(np.float32) * 0.1
        eng.prefill(x[:8], slot=0)
        assert eng.kv_pool_info()[2] == 3
        lg = None
        for i in range(8, 70):                             # crosses into a second page at position 64
            lg, _ = eng.decode(x[i], slot=0)
        assert eng.kv_pool_info()[2] == 2
        orc.prefill(x[:8])
        for i in range(8, 70):
            lo, _ = orc.decode(x[i])
        assert np.abs(lg - lo).max() < 2e-4
        eng.slot_release(0)
    finally:
        eng.close()


def test_scattered_pages_and_bounded_admission_give_the_same_codes(setup):
    import q3tts
    ocfg, w, orc = setup
    rng = np.random.default_rng(9)
    toks = [frame_tokens(rng.integers(0, 1000, n)) for n in (3, 9, 1, 14, 6, 2, 8)]
    caps = np.array([130, 20, 70, 150, 64, 5, 90], np.int32)         # 1..3 pages each
    sp = q3tts.Sampling(temperature=0.9, top_p=0.95, top_k=30, max_new_tokens=150)
    full = _engine(w, ocfg, 4, 320, 0)
    try:
        assert full.kv_pool_info() == (64, 4 * 5, 4 * 5)
        _, codes_full, nfr_full = full.synthesize_batch(toks, sp, lang=1, seed=77, ignore_eos=True, max_new_per_utt=caps)
    finally:
        full.close()
    small = _engine(w, ocfg, 4, 320, 4 * 64)                          # 4 pages: one to three utterances at a time on 4 slots
    try:
        pcm, codes, nfr = small.synthesize_batch(toks, sp, lang=1, seed=77, ignore_eos=True, max_new_per_utt=caps)
        assert list(nfr) == list(caps) == list(nfr_full)
        for u in range(len(toks)):
            assert np.array_equal(codes[u][:caps[u]], codes_full[u][:caps[u]]), u
        assert small.kv_pool_info() == (64, 4, 4)                     # every page came back
        for u in (0, 3):                                              # and both agree with the oracle's contiguous cache
            sp_u = q3tts.Sampling(temperature=0.9, top_p=0.95, top_k=30, max_new_tokens=int(caps[u]))
            ref = orc.generate(orc.build_prompt(toks[u], 1), to_osampling(sp_u), seed=77, stream=u, cp_cached=True, ignore_eos=True)
            assert np.array_equal(codes[u][:caps[u]], ref), u
        with pytest.raises(RuntimeError, match="more KV pages than the pool holds"):
            small.synthesize_batch(toks[:1], q3tts.Sampling(max_new_tokens=300), lang=1, seed=1, ignore_eos=True)   # 5 pages > 4
        pcm2, codes2, _ = small.synthesize_batch(toks[:2], sp, lang=1, seed=77, ignore_eos=True, max_new_per_utt=caps[:2])   # still usable
        assert np.array_equal(codes2[0][:caps[0]], codes[0][:caps[0]])
    finally:
        small.close()


def test_on_demand_growth_and_preemption(setup):
    """EOS-terminated generation (lengths unknown): slots grow page by page, and a pool too small for every running utterance's cap
    preempts the youngest, which is generated again later — codes and frame counts equal the unbounded engine's."""
    import q3tts
    ocfg, w, orc = setup
    rng = np.random.default_rng(4)
    toks = [frame_tokens(rng.integers(0, 1000, n)) for n in (5, 2, 12, 7, 3, 9)]
    sp = q3tts.Sampling(temperature=0.9, top_p=0.95, top_k=30, max_new_tokens=150)      # up to 3 pages each, 12 for four slots
    full = _engine(w, ocfg, 4, 320, 0)
    try:
        _, codes_full, nfr_full = full.synthesize_batch(toks, sp, lang=2, seed=13)
        assert full.sched_stats()[1] == 0
    finally:
        full.close()
    small = _engine(w, ocfg, 4, 320, 5 * 64)
    try:
        pcm, codes, nfr = small.synthesize_batch(toks, sp, lang=2, seed=13)
        admitted, preempted, peak = small.sched_stats()
        print(f"5-page pool, 4 slots, 6 utterances of up to 3 pages: admitted {admitted}, preempted {preempted}, peak live {peak}, frames {list(nfr)}")
        assert list(nfr) == list(nfr_full)
        for u in range(len(toks)):
            assert np.array_equal(codes[u][:nfr[u]], codes_full[u][:nfr[u]]), u
        assert preempted > 0 and admitted == len(toks) + preempted and peak >= 2
        assert small.kv_pool_info() == (64, 5, 5)
        ref = orc.generate(orc.build_prompt(toks[0], 2), to_osampling(sp), seed=13, stream=0, cp_cached=True)
        assert np.array_equal(codes[0][:nfr[0]], ref)
    finally:
        small.close()
