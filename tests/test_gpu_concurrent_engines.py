"""Several engines stepping on one GPU at the same time (bench.py's `capacity` record; a server with more than one model instance per GPU)
must produce what each produces alone: every kernel's result is a function of its inputs, whatever else shares the chip.

Round 5 found a kernel for which that did not hold — the vocoder's last conv returned wrong partial sums only while other lanes' kernels
were running (profiles/r05_hunt/) — so the talker / code-predictor / sampler chain gets the same treatment here:
three engines x 24 slots, 40 sampled steps each (hipGraph replays) from three host threads, ids bit-identical to each engine's own
solo run; repeated, because the vocoder's fault showed in one job out of five."""
import threading

import numpy as np
import pytest

from util import frame_tokens

pytestmark = pytest.mark.gpu


def _arm(eng, g, B, sp, prompt, trailing):
    for b in range(B):
        eng.slot_release(b)
    for b in range(B):
        eng.slot_begin(b, prompt, trailing, sp, seed=5, stream_id=g * B + b, ignore_eos=True)


def test_three_engines_stepping_concurrently_equal_their_solo_runs():
    import q3tts
    cfg = q3tts.default_config("0.6b")
    n_eng, B, steps = 3, 24, 40
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=steps + 4)
    engs = []
    try:
        for g in range(n_eng):
            e = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=96)
            e.fill_synthetic(seed=0)
            engs.append(e)
        prompt, trailing = engs[0].build_prompt(np.asarray(frame_tokens([11, 22, 33, 44, 55]), np.int64), 0)
        solo = []
        for g, e in enumerate(engs):
            _arm(e, g, B, sp, prompt, trailing)
            e.decode_steps(steps)
            solo.append([e.slot_codes(b).copy() for b in range(B)])
            assert all(c.shape[0] == steps for c in solo[-1])
        assert not np.array_equal(solo[0][0], solo[1][0])      # different RNG streams: the engines do different work
        for rep in range(6):
            for g, e in enumerate(engs):
                _arm(e, g, B, sp, prompt, trailing)
            bar = threading.Barrier(n_eng)
            errs = []

            def run(e):
                try:
                    bar.wait()
                    e.decode_steps(steps)
                except Exception as ex:   # noqa: BLE001 - reported below
                    errs.append(ex)
            th = [threading.Thread(target=run, args=(e,)) for e in engs]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not errs, errs
            for g, e in enumerate(engs):
                for b in range(B):
                    got = e.slot_codes(b)
                    if not np.array_equal(got, solo[g][b]):
                        first = int(np.argwhere((got != solo[g][b]).any(axis=1))[0][0])
                        raise AssertionError("repetition %d engine %d slot %d: ids differ from the solo run from frame %d on" % (rep, g, b, first))
    finally:
        for e in engs:
            e.close()


def test_decode_steps_beside_a_busy_vocoder_equal_the_solo_run():
    """The serving mix: one engine decodes (24 slots, sampled, hipGraph replays) while another engine's vocoder keeps the chip loaded with
    batched jobs (the state in which round 5's packed last conv went wrong).  The decoding engine's ids must equal its solo run, and the
    vocoding engine's PCM its own single decodes, every repetition."""
    import q3tts
    cfg = q3tts.default_config("0.6b")
    B, steps = 24, 48
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=steps + 4)
    dec = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=96)
    voc = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=192)
    try:
        dec.fill_synthetic(seed=0)
        voc.fill_synthetic(seed=0)
        prompt, trailing = dec.build_prompt(np.asarray(frame_tokens([9, 8, 7, 6]), np.int64), 0)
        _arm(dec, 0, B, sp, prompt, trailing)
        dec.decode_steps(steps)
        solo = [dec.slot_codes(b).copy() for b in range(B)]
        rng = np.random.default_rng(3)
        caps = [120, 100, 90, 2]
        codes = [rng.integers(0, cfg.cd_codebook, (f, cfg.n_groups)).astype(np.int64) for f in caps]
        alone = [voc.codec_decode(c) for c in codes]
        for rep in range(5):
            _arm(dec, 0, B, sp, prompt, trailing)
            stop, errs, worst = threading.Event(), [], [0.0]

            def vocode():
                try:
                    while not stop.is_set():
                        pcm = voc.codec_decode_batch(codes)
                        for u in range(len(caps)):
                            worst[0] = max(worst[0], float(np.abs(pcm[u] - alone[u]).max()))
                except Exception as ex:   # noqa: BLE001 - reported below
                    errs.append(ex)
            t = threading.Thread(target=vocode)
            t.start()
            try:
                dec.decode_steps(steps)
            finally:
                stop.set()
                t.join()
            assert not errs, errs
            assert worst[0] < 2e-5, ("vocoder beside a decoding engine", rep, worst[0])
            for b in range(B):
                got = dec.slot_codes(b)
                if not np.array_equal(got, solo[b]):
                    first = int(np.argwhere((got != solo[b]).any(axis=1))[0][0])
                    raise AssertionError("repetition %d slot %d: ids differ from the solo run from frame %d on" % (rep, b, first))
    finally:
        dec.close()
        voc.close()
