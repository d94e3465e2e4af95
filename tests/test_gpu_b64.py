"""BASELINE.json configs[2] at its own dims: Qwen3-TTS-0.6B, 64 utterances decoding in one batch (the 17..128-row bf16-MFMA GEMM
path: k_gemm2 + its finish kernels / prologues, batched attention, 64-row sampler grids), against the CPU oracle.

Reference semantics per utterance: generate_codes / predict_subcodes, /root/reference/src/tts_onnx.cpp:782-872 (the reference runs them
one utterance at a time; batching is this build's, so every utterance of the batch must equal its own single-utterance oracle run)."""
import numpy as np
import pytest

import q3_oracle as qo
from util import frame_tokens, to_ocfg, to_osampling

pytestmark = pytest.mark.gpu

CHECK = (0, 16, 17, 31, 63)     # oracle cost: a spread of utterances, both sides of the 16/17-row kernel switch


@pytest.fixture(scope="module", params=["fp32kv", "bf16kv", "finish-launches", "seam-rescue", "attn-splits", "attn-stream", "attn-stream-bf16kv"])
def wide(request):
    """64 slots at 0.6B dims.  Rounds: the default engine (split-K seam inside k_gemm3: the slab GEMMs reduce their own slabs, deferred
    RMSNorm); the talker KV cache in bf16 (Q3TTS_FLAG_KV_BF16, oracle in the same mode); Q3TTS_SEAM=0, the k_finish* launches the seam
    replaces; Q3TTS_SEAM_SPIN=1, every chunk owner gives up after one look, so the abandon / compare-and-swap rescue path of the seam
    produces the planes (a path a chip that runs the whole grid at once never takes);
    Q3TTS_ATTN_KEEP_SPLITS + 64-token splits on a 192-token cache: the talker's attention as split-T partials + the combine launch, on
    k_attn (Q3TTS_ATTN_STREAM=0: the kernel engines without the streaming kernel's shape keep) and, rounds attn-stream / -bf16kv, on
    k_attn_stream with one-page splits (what a 64-utterance batch runs beyond 512 tokens of context — the b64_f2048 bench — and never at
    the 64-token contexts of the other rounds): 56 frames, contexts crossing the first split boundary, two-deep K/V ring, new token in
    the last active split."""
    import os
    import q3tts
    cfg = q3tts.default_config("0.6b")
    bf = request.param in ("bf16kv", "attn-stream-bf16kv")
    long_run = request.param.startswith("attn-")
    env = {"finish-launches": {"Q3TTS_SEAM": "0"}, "seam-rescue": {"Q3TTS_SEAM_SPIN": "1"},
           "attn-splits": {"Q3TTS_ATTN_KEEP_SPLITS": "1", "Q3TTS_ATTN_CHUNK": "64", "Q3TTS_ATTN_STREAM": "0"},
           "attn-stream": {"Q3TTS_ATTN_KEEP_SPLITS": "1", "Q3TTS_ATTN_STREAM_CHUNK": "64"},
           "attn-stream-bf16kv": {"Q3TTS_ATTN_KEEP_SPLITS": "1", "Q3TTS_ATTN_STREAM_CHUNK": "64"}}.get(request.param, {})
    os.environ.update(env)             # read at engine creation
    try:
        eng = q3tts.Engine(cfg, device=0, max_batch=64, max_ctx=192 if long_run else 64, flags=(q3tts.FLAG_KV_BF16 if bf else 0) | q3tts.FLAG_TEST_HOOKS)   # the rounds' A/B knobs need the hooks flag
    finally:
        for k in env:
            del os.environ[k]
    eng.creation_env = env
    eng.fill_synthetic(seed=0)
    eng.margin_noise = 2e-2 if bf else 2e-4     # tests/test_gpu_full.py, bf16 KV note: what logit agreement this cache mode can honour
    eng.logit_bound = 2e-2 if bf else 2e-4
    # bf16 KV: ids are NOT bit-exact against the oracle in this mode (rounding decorrelates two implementations); the margin gate above opens
    # at the first decision whose top-2 gap is under 2e-2.  So that the gate cannot open at frame 0 unnoticed, the greedy test asserts a
    # floor on the bit-exact prefix of every checked utterance and prints the count (fp32 KV rounds: the whole run).
    eng.min_exact_frames = 5 if bf else None     # measured (round 4): 11 and 7 bit-exact frames for utterances 0 and 17, utterance 63 the whole run
    eng.long_run = long_run     # the 32-frame greedy test runs 56 frames there: contexts cross the 64-token split
    orc = qo.Oracle(to_ocfg(cfg), max_ctx=96 if eng.long_run else 48, kv_bf16=bf)
    for name, shape in eng.tensor_infos():
        if not name.startswith(("cd.", "spk.")):
            orc.set_tensor(name, eng.get_tensor(name, shape))
    rng = np.random.default_rng(64)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(3, 20, 64)]
    yield eng, orc, toks
    eng.close()
    orc.close()


@pytest.mark.parametrize("nb", [24, 64])
@pytest.mark.parametrize("sampled", [False, True])
def test_batched_generation_full_size(wide, nb, sampled):
    """nb utterances x 8 frames at 0.6B dims through q3tts_synthesize_schedule_host (batched prefill of nb x 8 prompt rows, then the
    hipGraph step at nb rows; predictor pass 0 runs 2 nb rows): codec ids bit-exact vs the oracle for a spread of utterances."""
    import q3tts
    eng, orc, toks = wide
    kw = dict(temperature=0.8, top_p=0.95, top_k=50) if sampled else dict(temperature=1.0, top_p=1.0, top_k=1)
    sp = q3tts.Sampling(max_new_tokens=8, **kw)
    pcm, codes, nfr = eng.synthesize_batch(toks[:nb], sp, lang=0, seed=77, ignore_eos=True)
    assert all(int(n) == 8 for n in nfr)
    bad = []
    for u in [u for u in CHECK if u < nb] + [nb - 1]:
        ref, mg = orc.generate_margins(orc.build_prompt(toks[u], 0), to_osampling(sp), seed=77, stream=u, cp_cached=True, ignore_eos=True)
        if not np.array_equal(codes[u], ref):   # margin-aware verdict (tests/test_gpu_full.py check_free_running): one decision to explain
            f, g = [int(v) for v in np.argwhere(codes[u] != ref)[0]]
            print("utterance %d parts from the oracle at frame %d group %d, oracle decision margin %.3g" % (u, f, g, float(mg[f, 2 + g])))
            if not (float(mg[f, 2 + g]) < eng.margin_noise and np.array_equal(codes[u][:f], ref[:f]) and np.array_equal(codes[u][f, :g], ref[f, :g])):
                bad.append((u, f, g, float(mg[f, 2 + g])))
        assert np.isfinite(pcm[u]).all() and len(pcm[u]) == eng.codec_decode_len(8)
    assert not bad, bad
    # the batch is deterministic and every utterance independent of its neighbours: the first 24 of a 64-batch == the 24-batch
    if nb == 24:
        _, codes64, _ = eng.synthesize_batch(toks, sp, lang=0, seed=77, ignore_eos=True)
        for u in range(24):
            assert np.array_equal(codes64[u], codes[u]), u


def test_every_row_of_the_batch_against_its_own_oracle_run(wide):
    """All 64 utterances, not a spread: 2 free-running greedy frames of the 64-row batch (32 code decisions per utterance, prompts of 8
    to 24 tokens) against 64 single-utterance oracle runs — the other batched tests check 3 to 6 rows against the oracle and the rest only
    against another batch.  fp32-cache default engine only (the oracle makes ~7 frames per second); margin-aware like the others."""
    import q3tts
    eng, orc, toks = wide
    if eng.creation_env or eng.min_exact_frames is not None:
        pytest.skip("one round is enough: 64 oracle runs")
    sp = q3tts.Sampling(max_new_tokens=2, temperature=1.0, top_p=1.0, top_k=1)
    _, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=13, ignore_eos=True)
    assert all(int(n) == 2 for n in nfr)
    exact = 0
    for u in range(64):
        ref, mg = orc.generate_margins(orc.build_prompt(toks[u], 0), to_osampling(sp), seed=13, stream=u, cp_cached=True, ignore_eos=True)
        bad = np.argwhere(codes[u] != ref)
        if bad.size == 0:
            exact += 1
            continue
        f, g = int(bad[0][0]), int(bad[0][1])
        print("all-rows check, utterance %d: first divergence at frame %d group %d, oracle margin %.3g" % (u, f, g, float(mg[f, 2 + g])))
        assert float(mg[f, 2 + g]) < eng.margin_noise, (u, f, g, float(mg[f, 2 + g]))
        assert np.array_equal(codes[u][:f], ref[:f]) and np.array_equal(codes[u][f, :g], ref[f, :g])
    print("all-rows check: %d of 64 utterances bit-exact over 2 frames (32 decisions each)" % exact)
    assert exact >= 60


def test_batched_greedy_32_frames_margin_aware(wide):
    """64 utterances x 32 free-running greedy frames in one batch (the split-K slab GEMM path for every projection, 32 hipGraph replays):
    a spread of utterances against their own single-utterance oracle runs.  Margin-aware: a divergence is a failure unless the oracle's
    top-2 logit gap at that decision is below the logit-noise bound (then it is printed and everything before it must match)."""
    import q3tts
    eng, orc, toks = wide
    F = 56 if eng.long_run else 32
    sp = q3tts.Sampling(max_new_tokens=F, temperature=1.0, top_p=1.0, top_k=1)
    _, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=9, ignore_eos=True)
    assert all(int(n) == F for n in nfr)
    for u in (0, 17, 63):
        ref, mg = orc.generate_margins(orc.build_prompt(toks[u], 0), to_osampling(sp), seed=9, stream=u, cp_cached=True, ignore_eos=True)
        bad = np.argwhere(codes[u] != ref)
        if bad.size == 0:
            print("b=64 greedy, utterance %d: %d frames bit-exact, smallest top-2 margin %.3g" % (u, F, float(mg[:, 2:].min())))
            continue
        f, g = int(bad[0][0]), int(bad[0][1])
        print("b=64 greedy, utterance %d: %d frames + %d decisions bit-exact, first divergence at frame %d group %d, oracle margin %.3g"
              % (u, f, g, f, g, float(mg[f, 2 + g])))
        assert float(mg[f, 2 + g]) < eng.margin_noise, (u, f, g, float(mg[f, 2 + g]))
        if eng.min_exact_frames is not None:
            assert f >= eng.min_exact_frames, "bf16 KV: only %d bit-exact frames for utterance %d (floor %d)" % (f, u, eng.min_exact_frames)
        assert np.array_equal(codes[u][:f], ref[:f]) and np.array_equal(codes[u][f, :g], ref[f, :g])


def test_teacher_forced_logits_64_rows_full_size(wide):
    """One 64-row decode step, teacher-forced: after the scheduler's batched prefill (64 x 8 prompt rows in 128-row GEMM blocks) and one
    hipGraph step (64-row talker pass, 128- and 64-row predictor passes: every k_gemm2 / finish variant of the layer), the talker logits
    and last_hidden the fused path holds for frame 1 stay within 2e-4 of the oracle fed with the same frame-0 codes."""
    import q3tts
    eng, orc, toks = wide
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=1)
    G = eng.cfg.n_groups
    _, codes, nfr = eng.synthesize_batch(toks, sp, seed=5, ignore_eos=True)      # utterance u runs in slot u
    assert all(int(n) == 1 for n in nfr)
    worst = 0.0
    for u in CHECK:
        po = orc.build_prompt(toks[u], 0)
        ref = orc.generate(po, to_osampling(sp), seed=5, stream=u, cp_cached=True, ignore_eos=True)
        if not np.array_equal(codes[u], ref):   # bf16 KV mode: the frame may already hold a sub-noise decision; teacher-force what the ENGINE emitted
            _, mg = orc.generate_margins(po, to_osampling(sp), seed=5, stream=u, cp_cached=True, ignore_eos=True)
            g = int(np.argwhere(codes[u][0] != ref[0])[0][0])
            print("teacher-forced: utterance %d differs from the oracle at group %d, oracle decision margin %.3g" % (u, g, float(mg[0, 2 + g])))
            assert eng.margin_noise > 1e-3 and float(mg[0, 2 + g]) < eng.margin_noise and np.array_equal(codes[u][0, :g], ref[0, :g]), (u, g, float(mg[0, 2 + g]))
            ref = codes[u]
        # teacher forcing: frame 0's embedding sum (tts_onnx.cpp:824-842) into the oracle's run_decode
        tro, _ = orc.trailing()
        orc.prefill(po)
        x = orc.codec_embed([int(ref[0, 0])])[0].copy()
        for j in range(G - 1):
            x = x + orc.cp_embed(int(ref[0, j + 1]), j)
        x = x + tro[0]
        lo, ho = orc.decode(x)
        lg, lh = eng.slot_logits(u)
        worst = max(worst, float(np.abs(lg - lo).max()), float(np.abs(lh - ho).max()))
    print("teacher-forced 64-row step: max |logit / hidden error| %.3g (bound %.0e)" % (worst, eng.logit_bound))
    assert worst < eng.logit_bound, worst


def test_gemm3_slabs_are_bit_identical_to_gemm2(wide):
    """k_gemm3 (straight-line K chunks, pinned fragment reads, LDS-transposed stores) performs the same operations on the same operands
    in the same order as the second-generation kernel it replaces: with Q3TTS_GEMM2=1 (the A/B knob) the same engine must produce
    bit-identical logits after a batched prefill + one 64-row step, and identical codes over 6 frames."""
    import os
    import q3tts
    eng, _, toks = wide
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=6)
    _, codes3, _ = eng.synthesize_batch(toks, sp, seed=21, ignore_eos=True)
    lg3 = [eng.slot_logits(u) for u in (0, 17, 63)]
    os.environ["Q3TTS_GEMM2"] = "1"
    os.environ.update(eng.creation_env)     # the twin engine runs the same round (seam / finish launches) as the fixture's
    try:
        e2 = q3tts.Engine(eng.cfg, device=0, max_batch=64, max_ctx=192 if eng.long_run else 64, flags=eng.flags)     # a fresh engine: its step graph is captured with the old kernel
        e2.fill_synthetic(seed=0)
        _, codes2, _ = e2.synthesize_batch(toks, sp, seed=21, ignore_eos=True)
        lg2 = [e2.slot_logits(u) for u in (0, 17, 63)]
        e2.close()
    finally:
        del os.environ["Q3TTS_GEMM2"]
        for k in eng.creation_env:
            del os.environ[k]
    for u in range(64):
        assert np.array_equal(codes3[u], codes2[u]), u
    for (a, ah), (b, bh) in zip(lg3, lg2):
        assert np.array_equal(a, b) and np.array_equal(ah, bh)


@pytest.mark.parametrize("nb", [64, 8])
def test_fragment_packed_weights_are_bit_identical_to_row_major(nb):
    """Round 5: k_gemm3 (12..128 rows) and k_gemv16 (3..16 rows) stream a fragment-packed copy of every projection matrix — per (16-row
    tile, 32-wide k-step) the 64 lanes' 16 bytes back to back, one contiguous KB per load instruction instead of sixteen half-used lines —
    registered by the engine at finalize.  The same values reach the same registers: with Q3TTS_PACKED_W=0 (the A/B knob: row-major
    matrices) a twin engine must produce bit-identical codes over 6 sampled frames and bit-identical logits, at 64 rows (slab GEMM + seam)
    and at 8 rows (the GEMV-contract kernel; its prefill too)."""
    import os
    import q3tts
    cfg = q3tts.default_config("0.6b")
    rng = np.random.default_rng(65)
    toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(3, 20, nb)]
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=6)
    res = []
    for knob in (None, "0"):
        if knob is not None:
            os.environ["Q3TTS_PACKED_W"] = knob
        try:
            e = q3tts.Engine(cfg, device=0, max_batch=nb, max_ctx=64, flags=q3tts.FLAG_TEST_HOOKS)
            e.fill_synthetic(seed=0)
            _, codes, _ = e.synthesize_batch(toks, sp, seed=22, ignore_eos=True)
            lg = [e.slot_logits(u) for u in (0, nb // 2, nb - 1)]
            e.close()
        finally:
            os.environ.pop("Q3TTS_PACKED_W", None)
        res.append((codes, lg))
    (c1, l1), (c0, l0) = res
    for u in range(nb):
        assert np.array_equal(c1[u], c0[u]), u
    for (a, ah), (b, bh) in zip(l1, l0):
        assert np.array_equal(a, b) and np.array_equal(ah, bh)


def test_batch_of_80_crosses_the_128_row_block():
    """80 utterances in one batch at 0.6B dims: the talker's projections run 80 rows (64-row blocks of the split-K seam GEMM), predictor
    pass 0 runs 160 rows — past the 128 rows one GEMM block and the seam cover, so those launches walk 128-row blocks and keep the
    finish launches — and the later passes 80.  Sampled, 8 frames; a spread of utterances against their single-utterance oracle runs."""
    import q3tts
    cfg = q3tts.default_config("0.6b")
    eng = q3tts.Engine(cfg, device=0, max_batch=80, max_ctx=64)
    orc = qo.Oracle(to_ocfg(cfg), max_ctx=48)
    try:
        eng.fill_synthetic(seed=0)
        for name, shape in eng.tensor_infos():
            if not name.startswith(("cd.", "spk.")):
                orc.set_tensor(name, eng.get_tensor(name, shape))
        rng = np.random.default_rng(80)
        toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(3, 20, 80)]
        sp = q3tts.Sampling(max_new_tokens=8, temperature=0.8, top_p=0.95, top_k=50)
        pcm, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=5, ignore_eos=True)
        assert all(int(n) == 8 for n in nfr)
        bad = []
        for u in (0, 63, 64, 79):
            ref, mg = orc.generate_margins(orc.build_prompt(toks[u], 0), to_osampling(sp), seed=5, stream=u, cp_cached=True, ignore_eos=True)
            if not np.array_equal(codes[u], ref):
                f, g = [int(v) for v in np.argwhere(codes[u] != ref)[0]]
                print("b=80, utterance %d parts from the oracle at frame %d group %d, oracle decision margin %.3g" % (u, f, g, float(mg[f, 2 + g])))
                if not (float(mg[f, 2 + g]) < 2e-4 and np.array_equal(codes[u][:f], ref[:f]) and np.array_equal(codes[u][f, :g], ref[f, :g])):
                    bad.append((u, f, g, float(mg[f, 2 + g])))
            assert np.isfinite(pcm[u]).all()
        assert not bad, bad
    finally:
        eng.close()
        orc.close()


def test_batch_of_32_with_long_prompts_one_split_attention_walks_several_batches():
    """32 utterances with 100..150-token prompts, 12 greedy frames, at 0.6B dims.  With >= 256 (utterance, kv head) pairs the batched step
    runs the talker's attention as ONE split per pair that walks the whole context in batches of 128 tokens with an online softmax across
    batches — the b=64 x 256-frame bench spends most of its steps there (contexts 130-280), while the other batched tests stop at 60 tokens
    and the b=1 tests take 64-token splits of one batch each.  Also: the scheduler's batched prefill over ~4000 prompt rows (128-row GEMM
    blocks, prompts longer than one 64-token KV page).  A spread of utterances against their single-utterance oracle runs, margin-aware."""
    import q3tts
    cfg = q3tts.default_config("0.6b")
    eng = q3tts.Engine(cfg, device=0, max_batch=32, max_ctx=192)
    orc = qo.Oracle(to_ocfg(cfg), max_ctx=192)
    try:
        eng.fill_synthetic(seed=0)
        for name, shape in eng.tensor_infos():
            if not name.startswith(("cd.", "spk.")):
                orc.set_tensor(name, eng.get_tensor(name, shape))
        rng = np.random.default_rng(32)
        toks = [frame_tokens(rng.integers(0, 151643, int(n))) for n in rng.integers(100, 150, 32)]
        F = 12
        sp = q3tts.Sampling(max_new_tokens=F, temperature=1.0, top_p=1.0, top_k=1)
        _, codes, nfr = eng.synthesize_batch(toks, sp, lang=0, seed=3, ignore_eos=True)
        assert all(int(n) == F for n in nfr)
        for u in (0, 15, 31):
            ref, mg = orc.generate_margins(orc.build_prompt(toks[u], 0), to_osampling(sp), seed=3, stream=u, cp_cached=True, ignore_eos=True)
            bad = np.argwhere(codes[u] != ref)
            if bad.size == 0:
                print("b=32, %d-token prompt, utterance %d: %d frames bit-exact, smallest top-2 margin %.3g" % (len(toks[u]), u, F, float(mg[:, 2:].min())))
                continue
            f, g = int(bad[0][0]), int(bad[0][1])
            print("b=32, utterance %d: first divergence at frame %d group %d, oracle margin %.3g" % (u, f, g, float(mg[f, 2 + g])))
            assert float(mg[f, 2 + g]) < 2e-4, (u, f, g, float(mg[f, 2 + g]))
            assert np.array_equal(codes[u][:f], ref[:f]) and np.array_equal(codes[u][f, :g], ref[f, :g])
    finally:
        eng.close()
        orc.close()
