"""Do an engine's decode steps produce the same ids while another engine's vocoder loads the chip?

    python tools/decode_beside_vocoder.py --batch 24 --steps 48 --reps 6 [--knob Q3TTS_SEAM=0] [--no-vocoder] [--no-graph]

Arms B slots (sampled, fixed seeds), decodes `steps` frames solo, then repeats the same decode with a second engine running batched
vocoder jobs from another host thread, and reports which slots' ids moved and from which frame.  --no-vocoder repeats the solo run
instead (the control: the decode chain against itself).  Knobs are set before the decoding engine is created (it is a test-hook engine)."""
import argparse
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "leaxer-qwen3-tts_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--knob", action="append", default=[])
    ap.add_argument("--no-vocoder", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--greedy", action="store_true")
    a = ap.parse_args()
    import q3tts
    for kv in a.knob:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    cfg = q3tts.default_config("0.6b")
    B = a.batch
    flags = q3tts.FLAG_TEST_HOOKS | (q3tts.FLAG_NO_GRAPH if a.no_graph else 0)
    dec = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=a.steps + 48, flags=flags)
    dec.fill_synthetic(seed=0)
    voc = None
    if not a.no_vocoder:
        voc = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=192)
        voc.fill_synthetic(seed=0)
    sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=1 if a.greedy else 50, max_new_tokens=a.steps + 4)
    prompt, trailing = dec.build_prompt(np.array([151644, 77091, 151672, 9, 8, 7, 6, 151673, 151645], np.int64), 0)

    def arm():
        for b in range(B):
            dec.slot_release(b)
        for b in range(B):
            dec.slot_begin(b, prompt, trailing, sp, seed=5, stream_id=b, ignore_eos=True)
    arm()
    dec.decode_steps(a.steps)
    solo = [dec.slot_codes(b).copy() for b in range(B)]
    rng = np.random.default_rng(3)
    caps = [120, 100, 90, 2]
    codes = [rng.integers(0, cfg.cd_codebook, (f, cfg.n_groups)).astype(np.int64) for f in caps]
    total_moved = 0
    for rep in range(a.reps):
        arm()
        stop, errs, jobs = threading.Event(), [], [0]

        def vocode():
            try:
                while not stop.is_set():
                    voc.codec_decode_batch(codes)
                    jobs[0] += 1
            except Exception as ex:   # noqa: BLE001
                errs.append(ex)
        t = None
        if voc is not None:
            t = threading.Thread(target=vocode)
            t.start()
        try:
            dec.decode_steps(a.steps)
        finally:
            stop.set()
            if t is not None:
                t.join()
        if errs:
            raise errs[0]
        moved = []
        for b in range(B):
            got = dec.slot_codes(b)
            if not np.array_equal(got, solo[b]):
                fr = int(np.argwhere((got != solo[b]).any(axis=1))[0][0])
                g = int(np.argwhere(got[fr] != solo[b][fr])[0][0])
                moved.append((b, fr, g))
        total_moved += len(moved)
        print("rep %d: %d of %d slots moved%s; (slot, first frame, first group) %s" % (rep, len(moved), B, "" if voc is None else " (%d vocoder jobs beside)" % jobs[0], moved[:10]), flush=True)
    print("batch %d, %d steps, knobs %s, %s: %d moved slots in %d repetitions" % (B, a.steps, a.knob or "-", "solo control" if voc is None else "vocoder beside", total_moved, a.reps), flush=True)
    dec.close()
    if voc is not None:
        voc.close()
    return 1 if total_moved else 0


if __name__ == "__main__":
    sys.exit(main())
