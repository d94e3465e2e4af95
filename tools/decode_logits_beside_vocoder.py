"""Are the decode step's logits bit-identical to a solo run while another engine's vocoder loads the chip?

    [REPS=8] [Q3TTS_LIB=tools/exp/libpk.so] python tools/decode_logits_beside_vocoder.py BATCH SLOT STEPS

One eager decode step at a time (q3tts_step_logits_host: the 16 logits rows behind a frame's decisions of SLOT), first solo twice (the
control), then with a second engine running batched vocoder jobs from another host thread; reports the first frame / code group whose row
differs and by how much.  Round 5: with packed fp32 instructions in the kernels the rows differ by 0.5e-6 .. 9e-6 beside the vocoder, with
the product build (no packed fp32) they are bit-identical (profiles/r05_hunt/README.txt)."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "leaxer-qwen3-tts_amd"))
import q3tts
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
slot = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = q3tts.default_config("0.6b")
dec = q3tts.Engine(cfg, device=0, max_batch=B, max_ctx=steps + 48, flags=q3tts.FLAG_TEST_HOOKS)
dec.fill_synthetic(seed=0)
voc = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=192)
voc.fill_synthetic(seed=0)
sp = q3tts.Sampling(temperature=0.8, top_p=0.95, top_k=50, max_new_tokens=steps + 4)
prompt, trailing = dec.build_prompt(np.array([151644, 77091, 151672, 9, 8, 7, 6, 151673, 151645], np.int64), 0)
def arm():
    for b in range(B): dec.slot_release(b)
    for b in range(B): dec.slot_begin(b, prompt, trailing, sp, seed=5, stream_id=b, ignore_eos=True)
arm()
solo = [dec.step_logits(slot).copy() for _ in range(steps)]
arm()
solo2 = [dec.step_logits(slot).copy() for _ in range(steps)]
print("solo vs solo: identical", all(np.array_equal(a, b) for a, b in zip(solo, solo2)))
rng = np.random.default_rng(3)
codes = [rng.integers(0, cfg.cd_codebook, (f, cfg.n_groups)).astype(np.int64) for f in (120, 100, 90, 2)]
for rep in range(int(os.environ.get("REPS", "4"))):
    arm()
    stop = threading.Event()
    def vocode():
        while not stop.is_set(): voc.codec_decode_batch(codes)
    t = threading.Thread(target=vocode); t.start()
    try:
        got = [dec.step_logits(slot).copy() for _ in range(steps)]
    finally:
        stop.set(); t.join()
    for f in range(steps):
        d = np.abs(got[f] - solo[f])
        if d.max() > 0:
            g = int(np.argwhere(d.max(axis=1) > 0)[0][0])
            nz = np.argwhere(d > 0)
            print("rep %d: first differing logits at frame %d group %d: %d of %d values of that row differ, max |diff| %.3g (row max |logit| %.3g); rows differing in this frame: %s" % (
                rep, f, g, int((d[g] > 0).sum()), d.shape[1], d[g].max(), np.abs(solo[f][g]).max(), sorted(set(int(x) for x in nz[:, 0]))))
            break
    else:
        print("rep %d: all %d frames bit-identical" % (rep, steps))
