"""A 70-line packed-fp32 kernel outside the product, beside the product's vocoder (tools/pk_probe.hip):

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/pk_probe.hip -o tools/exp/libpkprobe.so
    python tools/pk_probe_beside_vocoder.py [--reps 200]

Phases: packed / scalar build of the probe kernel with the chip otherwise idle, then the same while a q3tts engine runs batched vocoder
jobs from another host thread.  Every launch's outputs are compared bit for bit with a reference taken on the idle chip."""
import argparse
import ctypes as C
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "leaxer-qwen3-tts_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import q3tts
    P = C.CDLL(os.path.join(HERE, "exp", "libpkprobe.so"))
    P.pk_probe_run.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_float)]
    tiles = P.pk_probe_init()
    assert tiles > 0
    cfg = q3tts.default_config("0.6b")
    voc = q3tts.Engine(cfg, device=0, max_batch=1, max_ctx=192)
    voc.fill_synthetic(seed=0)
    rng = np.random.default_rng(3)
    codes = [rng.integers(0, cfg.cd_codebook, (f, cfg.n_groups)).astype(np.int64) for f in (120, 100, 90, 2)]
    voc.codec_decode_batch(codes)

    def run(packed):
        bs, w = C.c_long(0), C.c_float(0)
        bl = P.pk_probe_run(packed, a.reps, C.byref(bs), C.byref(w))
        return bl, bs.value, w.value
    for beside in (False, True, False):
        stop, jobs = threading.Event(), [0]

        def vocode():
            while not stop.is_set():
                voc.codec_decode_batch(codes)
                jobs[0] += 1
        t = None
        if beside:
            t = threading.Thread(target=vocode)
            t.start()
        try:
            for packed in (1, 0, 1):
                bl, bs, w = run(packed)
                print("%-28s %-7s %4d launches of %d workgroups: %4d with a wrong output, %7d wrong samples, worst |error| %.3g" % (
                    "beside the product's vocoder" if beside else "chip otherwise idle", "packed" if packed else "scalar", a.reps, tiles, bl, bs, w), flush=True)
        finally:
            stop.set()
            if t is not None:
                t.join()
        if beside:
            print("   (%d vocoder jobs ran beside)" % jobs[0], flush=True)
    voc.close()


if __name__ == "__main__":
    main()
