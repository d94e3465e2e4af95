"""The batch-first, device-pointer half of the C-ABI (include/q3tts.h "_dev" entry points): q3tts_talker_prefill_dev,
q3tts_talker_decode_dev, q3tts_code_predictor_dev, q3tts_sample_dev.  Every entry is compared with its "_host" twin — bit for bit where
the two take the same launches (one row; the sampler at any batch) — and with the CPU oracle where the batch changes the kernels.

Reference seam: run_prefill / run_decode / run_code_predictor / sample_token, /root/reference/src/tts_onnx.cpp:615-757, 878-950;
predict_subcodes :851-872.  The long-context case is also the parity test of k_attn_stream (the batched step's attention past 512
tokens of context) against the oracle."""
import numpy as np
import pytest

import q3_oracle as qo
from util import Hip, to_ocfg, to_osampling

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hip():
    h = Hip()
    yield h
    h.free()


def _engine_and_oracle(max_batch, max_ctx, flags=0, kv_bf16=False, oracle_ctx=None):
    import q3tts
    cfg = q3tts.default_config("0.6b")
    eng = q3tts.Engine(cfg, device=0, max_batch=max_batch, max_ctx=max_ctx, flags=flags)
    eng.fill_synthetic(seed=0)
    orc = qo.Oracle(to_ocfg(cfg), max_ctx=oracle_ctx or max_ctx, kv_bf16=kv_bf16)
    for name, shape in eng.tensor_infos():
        if not name.startswith(("cd.", "spk.")):
            orc.set_tensor(name, eng.get_tensor(name, shape))
    return eng, orc


def test_sample_dev_equals_the_host_twin_row_for_row(hip):
    """q3tts_sample_dev over 40 rows (code0 vocabulary with suppression, sub-code vocabulary without; greedy, top-k / top-p, the general
    path) == q3tts_sample_host on each row with the same uniform: the ids are integers, so bit for bit."""
    import q3tts
    from util import tiny_pair
    eng, orc, _ = tiny_pair(seed=3, max_batch=2, max_ctx=32)
    rng = np.random.default_rng(5)
    try:
        for V, suppress in ((eng.cfg.vocab, True), (eng.cfg.sub_vocab, False)):
            for kw in (dict(temperature=1.0, top_p=1.0, top_k=1), dict(temperature=0.8, top_p=0.95, top_k=50), dict(temperature=1.3, top_p=0.7, top_k=0)):
                sp = q3tts.Sampling(max_new_tokens=1, **kw)
                B = 40
                lg = (rng.standard_normal((B, V)) * 3.0).astype(np.float32)
                u = rng.random(B).astype(np.float32)
                lg_d, u_d, ids_d = hip.put(lg), hip.put(u), hip.alloc(B * 8)
                eng.sample_dev(lg_d, B, V, sp, u_d, suppress, ids_d)      # returns at once: ordered on the engine's stream
                hip.rt.hipStreamSynchronize(eng.stream)
                got = hip.get(ids_d, (B,), np.int64)
                want = np.array([eng.sample(lg[b], sp, float(u[b]), suppress) for b in range(B)], np.int64)
                assert np.array_equal(got, want), (V, kw)
                ref = np.array([orc.sample(lg[b], to_osampling(sp), float(u[b])) for b in range(B)], np.int64) if not suppress else None
                if ref is not None:
                    assert np.array_equal(got, ref), (V, kw)
    finally:
        eng.close()
        orc.close()


def test_talker_dev_one_row_takes_the_host_twins_launches_bit_for_bit(hip):
    """batch 1: q3tts_talker_prefill_dev + q3tts_talker_decode_dev on device buffers == q3tts_talker_prefill_host + q3tts_talker_decode_host
    bit for bit (same kernels, same order), 0.6B dims, a caller stream of its own."""
    import q3tts
    cfg = q3tts.default_config("0.6b")
    a = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=128)
    b = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=128)
    try:
        a.fill_synthetic(seed=0)
        b.fill_synthetic(seed=0)
        H, V = cfg.hidden, cfg.vocab
        rng = np.random.default_rng(11)
        x = (rng.standard_normal((8, H)) * 0.05).astype(np.float32)
        lg_h, lh_h = a.prefill(x)
        st = hip.stream()
        x_d, lg_d, lh_d = hip.put(x), hip.alloc(V * 4), hip.alloc(H * 4)
        b.talker_prefill_dev(x_d, 1, 8, None, lg_d, lh_d, stream=st.value)
        hip.rt.hipStreamSynchronize(st)
        assert np.array_equal(hip.get(lg_d, (V,), np.float32), lg_h[-1]) and np.array_equal(hip.get(lh_d, (H,), np.float32), lh_h)
        e_d = hip.alloc(H * 4)
        for i in range(20):
            e = (rng.standard_normal(H) * 0.05).astype(np.float32)
            lg_h, lh_h = a.decode(e)
            hip.write(e_d, e)
            b.talker_decode_dev(e_d, 1, None, lg_d, lh_d, stream=st.value)
            hip.rt.hipStreamSynchronize(st)
            assert np.array_equal(hip.get(lg_d, (V,), np.float32), lg_h) and np.array_equal(hip.get(lh_d, (H,), np.float32), lh_h), i
        hip.rt.hipStreamDestroy(st)
    finally:
        a.close()
        b.close()


def test_talker_dev_ragged_batch_with_masked_rows_vs_oracle_and_host_twin(hip):
    """16 slots, ragged prompts (lens 8 / 5 / 3 in runs: equal-length runs share a pass, a lone slot takes the single-slot launches), then 6
    batched decode steps with some rows masked out at some steps.  Checked slots against the oracle (run_prefill / run_decode semantics)
    within the fp32 logit bound; masked rows' outputs untouched; every slot against the host twin driven one slot at a time."""
    import q3tts
    B, S = 16, 8
    eng, orc = _engine_and_oracle(B, 64)
    twin = q3tts.Engine(eng.cfg, device=0, max_batch=B, max_ctx=64)
    try:
        twin.fill_synthetic(seed=0)
        H, V = eng.cfg.hidden, eng.cfg.vocab
        rng = np.random.default_rng(21)
        lens = np.array([8] * 6 + [5] * 5 + [3] + [8] * 4, np.int32)
        x = (rng.standard_normal((B, S, H)) * 0.05).astype(np.float32)
        x_d, lg_d, lh_d = hip.put(x), hip.alloc(B * V * 4), hip.alloc(B * H * 4)
        eng.talker_prefill_dev(x_d, B, S, lens, lg_d, lh_d)
        lg, lh = hip.get(lg_d, (B, V), np.float32), hip.get(lh_d, (B, H), np.float32)
        worst_o = worst_t = 0.0
        check = (0, 5, 6, 10, 11, 15)
        for b in range(B):
            tl, th = twin.prefill(x[b, : lens[b]], slot=b)
            worst_t = max(worst_t, float(np.abs(lg[b] - tl[-1]).max()), float(np.abs(lh[b] - th).max()))
        lg0, lh0 = lg, lh
        steps = 6
        masks = (rng.random((steps, B)) > 0.25).astype(np.uint8)
        masks[:, 0] = 1
        e_all = (rng.standard_normal((steps, B, H)) * 0.05).astype(np.float32)
        e_d = hip.alloc(B * H * 4)
        got = []
        for i in range(steps):
            hip.write(e_d, e_all[i])
            hip.write(lg_d, np.full((B, V), 7.0, np.float32))
            hip.write(lh_d, np.full((B, H), 7.0, np.float32))
            eng.talker_decode_dev(e_d, B, masks[i], lg_d, lh_d)
            lg, lh = hip.get(lg_d, (B, V), np.float32), hip.get(lh_d, (B, H), np.float32)
            for b in range(B):
                if masks[i, b]:
                    tl, th = twin.decode(e_all[i, b], slot=b)
                    worst_t = max(worst_t, float(np.abs(lg[b] - tl).max()), float(np.abs(lh[b] - th).max()))
                else:
                    assert (lg[b] == 7.0).all() and (lh[b] == 7.0).all(), (i, b)      # a masked row leaves no trace in the outputs
            got.append((lg, lh))
        for b in check:      # the oracle, one slot at a time: prefill of the slot's prompt, then the steps the slot took part in
            lo_all, ho = orc.prefill(x[b, : lens[b]])
            lo = lo_all[-1] if lo_all.ndim == 2 else lo_all
            worst_o = max(worst_o, float(np.abs(lg0[b] - lo).max()), float(np.abs(lh0[b] - ho).max()))
            for i in range(steps):
                if masks[i, b]:
                    lo, ho = orc.decode(e_all[i, b])
                    worst_o = max(worst_o, float(np.abs(got[i][0][b] - lo).max()), float(np.abs(got[i][1][b] - ho).max()))
        print("talker _dev, 16 ragged slots, masked steps: worst |difference| vs oracle %.3g, vs the one-slot host twin %.3g" % (worst_o, worst_t))
        assert worst_o < 2e-4 and worst_t < 2e-4, (worst_o, worst_t)
    finally:
        eng.close()
        twin.close()
        orc.close()


@pytest.mark.parametrize("kv", ["fp32", "bf16"])
def test_batched_decode_at_long_context_streaming_attention_vs_oracle(hip, kv):
    """k_attn_stream against the oracle at depth: 16 slots decode together (q3tts_talker_decode_dev: the batched step's talker stage — MFMA
    slab GEMMs and, with max_ctx past 512 tokens, the streaming attention kernel over 256-token (fp32 cache) / 512-token (bf16 cache)
    splits) for 700 teacher-forced steps from an 8-row prompt, crossing two / one split boundaries; then 6 more steps whose logits and
    hidden rows are compared for slots 0 and 9 against the oracle, which builds the same 708-token caches with one prefill pass each
    (the trick of test_talker_decode_at_context_2048_full_size).  fp32 cache: the fp32 logit bound; bf16 cache: that mode's bound
    (tests/test_gpu_full.py, bf16 KV note).  Reference: run_decode over the grown KVCache, /root/reference/src/tts_onnx.cpp:667-732."""
    import q3tts
    B, T = 16, 700
    bf = kv == "bf16"
    eng, orc = _engine_and_oracle(B, 1024, flags=q3tts.FLAG_KV_BF16 if bf else 0, kv_bf16=bf, oracle_ctx=T + 32)
    try:
        H, V = eng.cfg.hidden, eng.cfg.vocab
        rng = np.random.default_rng(77)
        check = (0, 9)
        xs = {b: (rng.standard_normal((8 + T, H)) * 0.05).astype(np.float32) for b in check}      # the checked slots' whole input history
        x0 = (rng.standard_normal((B, 8, H)) * 0.05).astype(np.float32)
        for b in check:
            x0[b] = xs[b][:8]
        x_d, lg_d, lh_d = hip.put(x0), hip.alloc(B * V * 4), hip.alloc(B * H * 4)
        eng.talker_prefill_dev(x_d, B, 8, None, 0, 0)
        e_d = hip.alloc(B * H * 4)
        filler = (rng.standard_normal((B, H)) * 0.05).astype(np.float32)
        for i in range(T):
            e = filler.copy()
            e[1:] *= np.float32(1.0 + 0.001 * (i % 7))
            for b in check:
                e[b] = xs[b][8 + i]
            hip.write(e_d, e)
            eng.talker_decode_dev(e_d, B, None, lg_d if i == T - 1 else 0, lh_d if i == T - 1 else 0)
        lg, lh = hip.get(lg_d, (B, V), np.float32), hip.get(lh_d, (B, H), np.float32)
        worst = 0.0
        tail = {b: (rng.standard_normal((6, H)) * 0.05).astype(np.float32) for b in check}
        outs = []
        for i in range(6):
            e = filler.copy()
            for b in check:
                e[b] = tail[b][i]
            hip.write(e_d, e)
            eng.talker_decode_dev(e_d, B, None, lg_d, lh_d)
            outs.append((hip.get(lg_d, (B, V), np.float32), hip.get(lh_d, (B, H), np.float32)))
        for b in check:
            lo_all, ho = orc.prefill(xs[b])
            lo = lo_all[-1] if lo_all.ndim == 2 else lo_all
            worst = max(worst, float(np.abs(lg[b] - lo).max()), float(np.abs(lh[b] - ho).max()))
            for i in range(6):
                lo, ho = orc.decode(tail[b][i])
                worst = max(worst, float(np.abs(outs[i][0][b] - lo).max()), float(np.abs(outs[i][1][b] - ho).max()))
        bound = 2e-2 if bf else 3e-4
        print("batched decode at context %d..%d, %s KV cache: worst |logit / hidden difference| vs the oracle %.3g (bound %.0e)" % (8 + T, 8 + T + 6, kv, worst, bound))
        assert worst < bound, worst
    finally:
        eng.close()
        orc.close()


def test_code_predictor_dev_fused_subcodes_vs_the_reference_call_pattern(hip):
    """q3tts_code_predictor_dev (15 KV-cached passes, on-device sampling) for 12 utterances against predict_subcodes restated with the
    oracle's session calls in the REFERENCE's pattern (/root/reference/src/tts_onnx.cpp:851-872: the whole growing sequence re-run per
    sub-code, no cache; sample_token with the same uniforms).  Greedy and sampled; a mismatch must sit on a decision whose oracle margin
    is below the fp32 logit noise (then the row stops being compared, as in the free-running tests)."""
    import q3tts
    B = 12
    eng, orc = _engine_and_oracle(B, 64)
    try:
        H, G = eng.cfg.hidden, eng.cfg.n_groups
        rng = np.random.default_rng(31)
        x = (rng.standard_normal((B, 8, H)) * 0.05).astype(np.float32)
        x_d, lh_d = hip.put(x), hip.alloc(B * H * 4)
        eng.talker_prefill_dev(x_d, B, 8, None, 0, lh_d)
        lh = hip.get(lh_d, (B, H), np.float32)
        code0 = rng.integers(0, 2048, B).astype(np.int64)
        c0_d, sub_d = hip.put(code0), hip.alloc(B * (G - 1) * 4)
        for kw, frame in ((dict(temperature=1.0, top_p=1.0, top_k=1), 0), (dict(temperature=0.8, top_p=0.95, top_k=50), 5)):
            sp = q3tts.Sampling(max_new_tokens=1, **kw)
            eng.code_predictor_dev(lh_d, c0_d, B, sp, 99, 40, frame, sub_d)
            sub = hip.get(sub_d, (B, G - 1), np.int32)
            exact = 0
            for b in (0, 5, 11):
                seq = [lh[b], orc.codec_embed([int(code0[b])])[0]]
                for j in range(G - 1):
                    lo = orc.code_predictor(np.stack(seq), j)
                    u = q3tts.rng_uniform(99, 40 + b, frame, j + 1)
                    tok, margin = orc.sample_margin(lo, to_osampling(sp), u)
                    if int(sub[b, j]) != tok:
                        print("code_predictor_dev: row %d parts from the reference pattern at sub-code %d, oracle margin %.3g" % (b, j, margin))
                        assert margin < 2e-4, (b, j, margin)
                        break
                    exact += 1
                    seq.append(orc.cp_embed(tok, j))
            print("code_predictor_dev %s: %d of 45 checked sub-code decisions identical to the uncached reference pattern" % ("greedy" if kw["top_k"] == 1 else "sampled", exact))
            assert exact >= 40
    finally:
        eng.close()
        orc.close()


def test_dev_entries_on_the_projected_tiny_config_with_a_pooled_cache(hip):
    """The "_dev" sessions where the other tests do not go: the 1.7B structure (predictor narrower than the talker behind cp.proj) on the
    tiny config (GEMV-family kernels at 5 rows: dims that are no multiple of 128), and a bounded KV page pool (pages handed out as the
    batched decode grows the contexts; table-mapped attention).  Every row against the oracle: prefill + 70 decode steps across a
    64-token page boundary, then the fused sub-code predictor against the uncached reference pattern."""
    import q3tts
    from util import to_q3cfg
    ocfg = qo.config_tiny_proj()
    w = qo.random_weights(ocfg, 12)
    B = 5
    eng = q3tts.Engine(to_q3cfg(ocfg), device=0, max_batch=B, max_ctx=192, kv_pool_tokens=10 * 64)   # 10 pages for 5 slots of up to 76 tokens (a full pool would be 15)
    eng.load(w)
    orcs = [qo.Oracle(ocfg, max_ctx=192, weights=w) for _ in range(2)]
    try:
        H, V, G = eng.cfg.hidden, eng.cfg.vocab, eng.cfg.n_groups
        rng = np.random.default_rng(3)
        lens = np.array([6, 6, 3, 6, 2], np.int32)
        x = rng.standard_normal((B, 6, H)).astype(np.float32)
        x_d, lg_d, lh_d, e_d = hip.put(x), hip.alloc(B * V * 4), hip.alloc(B * H * 4), hip.alloc(B * H * 4)
        eng.talker_prefill_dev(x_d, B, 6, lens, lg_d, lh_d)
        check = (1, 4)
        for o, b in zip(orcs, check):
            lo_all, ho = o.prefill(x[b, : lens[b]])
            lo = lo_all[-1] if lo_all.ndim == 2 else lo_all
            assert np.abs(hip.get(lg_d, (B, V), np.float32)[b] - lo).max() < 2e-4 and np.abs(hip.get(lh_d, (B, H), np.float32)[b] - ho).max() < 2e-4, b
        worst = 0.0
        for i in range(70):
            e = rng.standard_normal((B, H)).astype(np.float32)
            hip.write(e_d, e)
            eng.talker_decode_dev(e_d, B, None, lg_d, lh_d)
            lg, lh = hip.get(lg_d, (B, V), np.float32), hip.get(lh_d, (B, H), np.float32)
            for o, b in zip(orcs, check):
                lo, ho = o.decode(e[b])
                worst = max(worst, float(np.abs(lg[b] - lo).max()), float(np.abs(lh[b] - ho).max()))
        print("talker _dev on the projected tiny config, pooled cache, 5 rows x 70 steps: worst |difference| vs oracle %.3g" % worst)
        assert worst < 2e-4, worst
        sp = q3tts.Sampling(max_new_tokens=1, temperature=1.0, top_p=1.0, top_k=1)
        code0 = rng.integers(0, ocfg.sub_vocab, B).astype(np.int64)
        c0_d, sub_d = hip.put(code0), hip.alloc(B * (G - 1) * 4)
        eng.code_predictor_dev(lh_d, c0_d, B, sp, 7, 0, 3, sub_d)
        sub = hip.get(sub_d, (B, G - 1), np.int32)
        lh = hip.get(lh_d, (B, H), np.float32)
        o = orcs[0]
        for b in range(B):
            seq = [lh[b], o.codec_embed([int(code0[b])])[0]]
            for j in range(G - 1):
                tok, margin = o.sample_margin(o.code_predictor(np.stack(seq), j), to_osampling(sp), 0.5)
                if int(sub[b, j]) != tok:
                    assert margin < 2e-4, (b, j, margin)
                    break
                seq.append(o.cp_embed(tok, j))
    finally:
        eng.close()
        for o in orcs:
            o.close()


def test_bf16_cache_equals_rounded_fp32_cache_bit_for_bit_in_the_batched_step(hip):
    """The pin of the bf16 KV data path (tests/test_gpu_full.py::test_bf16_kv_storage_equals_rounded_fp32_storage does it for the b = 1
    kernels) for the batched step's streaming attention: 16 slots, 8-row prompts, 560 teacher-forced batched decode steps — across the
    512-token split boundary — with the cache in bf16 (Q3TTS_FLAG_KV_BF16) and, on a second engine, in fp32 holding the same rounded rows
    (Q3TTS_FLAG_KV_ROUND_BF16: the bf16 kernel's lane mapping, batch shape and splits on 4-byte storage).  Same rows before rounding,
    same rounding, same arithmetic in the same order: logits and hidden rows identical bit for bit at every compared step."""
    import q3tts
    cfg = q3tts.default_config("0.6b")
    B, T = 16, 560
    a = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=640, flags=q3tts.FLAG_KV_BF16)
    b = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=640, flags=q3tts.FLAG_KV_ROUND_BF16)
    try:
        a.fill_synthetic(seed=0)
        b.fill_synthetic(seed=0)
        H, V = cfg.hidden, cfg.vocab
        rng = np.random.default_rng(5)
        x0 = (rng.standard_normal((B, 8, H)) * 0.05).astype(np.float32)
        x_d, e_d = hip.put(x0), hip.alloc(B * H * 4)
        la, ha, lb, hb = hip.alloc(B * V * 4), hip.alloc(B * H * 4), hip.alloc(B * V * 4), hip.alloc(B * H * 4)
        a.talker_prefill_dev(x_d, B, 8, None, la, ha)
        b.talker_prefill_dev(x_d, B, 8, None, lb, hb)
        assert np.array_equal(hip.get(la, (B, V), np.float32), hip.get(lb, (B, V), np.float32))
        compared = 0
        for i in range(T):
            e = (rng.standard_normal((B, H)) * 0.05).astype(np.float32)
            hip.write(e_d, e)
            look = i < 4 or i % 37 == 0 or i >= T - 6 or 500 <= i <= 510      # incl. the steps around context 512
            a.talker_decode_dev(e_d, B, None, la if look else 0, ha if look else 0)
            b.talker_decode_dev(e_d, B, None, lb if look else 0, hb if look else 0)
            if look:
                assert np.array_equal(hip.get(la, (B, V), np.float32), hip.get(lb, (B, V), np.float32)), i
                assert np.array_equal(hip.get(ha, (B, H), np.float32), hip.get(hb, (B, H), np.float32)), i
                compared += 1
        assert compared > 30
    finally:
        a.close()
        b.close()
