"""Repetition stress of the job vocoder (q3tts_codec_decode_batch_host) against single-utterance decodes, without the talker in the loop:

    python tools/vocoder_stress.py --reps 2000 [--caps 40,9,120,2,90,19,8,100,6,44] [--lanes N] [--phase NAME:KNOB=V,KNOB=V ...]

Every repetition poisons the vocoder's reusable workspace (NaN), runs the job on random codes (fixed per run) and compares each
utterance with its own single decode taken once up front (and re-taken every 50th repetition: the single path under the same stress).
A mismatch above --tol prints the sample range, both sides' values around it and which side moved.  Phases run back to back on one
engine with the given A/B knobs set in the environment (the engine is created with Q3TTS_FLAG_TEST_HOOKS), so a knob that makes the
mismatches disappear names the kernel.  Used for round 5's hunt of the intermittent two-sample mismatch."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "leaxer-qwen3-tts_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=500)
    ap.add_argument("--caps", default="40,9,120,2,90,19,8,100,6,44")
    ap.add_argument("--lanes", type=int, default=None)
    ap.add_argument("--tol", type=float, default=2e-5)
    ap.add_argument("--model", default="0.6b")
    ap.add_argument("--no-poison", action="store_true")
    ap.add_argument("--check-input", action="store_true", help="every repetition: the last batched group's final-conv input must equal the first clean job's, bit for bit (all upstream kernels of that group under the same stress)")
    ap.add_argument("--clock", type=int, default=0, help="with Q3TTS_COUT1_PACKED=2: print the conv's shader clock for the first N jobs of each phase, wrong or not")
    ap.add_argument("--diag", type=int, default=0, help="for the first N mismatching jobs read back the last batched group's final-conv input and output and say which is wrong")
    ap.add_argument("--phase", action="append", default=[], help="NAME:KNOB=V,KNOB=V (repeatable); default: one phase without knobs")
    a = ap.parse_args()
    import q3tts
    caps = [int(x) for x in a.caps.split(",")]
    if a.lanes is not None:
        os.environ["Q3TTS_CODEC_LANES"] = str(a.lanes)
    eng = q3tts.Engine(q3tts.default_config(a.model), device=0, max_batch=1, max_ctx=192, flags=q3tts.FLAG_TEST_HOOKS)
    eng.fill_synthetic(seed=0)
    rng = np.random.default_rng(5)
    codes = [rng.integers(0, eng.cfg.cd_codebook, (f, eng.cfg.n_groups)).astype(np.int64) for f in caps]
    alone0 = [eng.codec_decode(c) for c in codes]
    alone1 = [eng.codec_decode(c) for c in codes]
    for u in range(len(caps)):
        assert np.array_equal(alone0[u], alone1[u]), "single decode is not repeatable"
    # the final conv on the host (float64): out[t] = b + sum_tap sum_c w[0][c][tap] x[t - 6 + tap][c]
    shapes = dict(eng.tensor_infos())
    w_out = eng.get_tensor("cd.dec.conv_out.w", shapes["cd.dec.conv_out.w"]).astype(np.float64).reshape(1, -1, 7)[0]   # [c][tap]
    b_out = float(eng.get_tensor("cd.dec.conv_out.b", shapes["cd.dec.conv_out.b"]).reshape(-1)[0])

    def host_conv(x):   # x [T][C] -> [T]
        T = x.shape[0]
        xp = np.concatenate([np.zeros((6, x.shape[1])), x.astype(np.float64)])
        out = np.full(T, b_out)
        for tap in range(7):
            out += xp[tap:tap + T] @ w_out[:, tap]
        return np.clip(out, -1.0, 1.0)

    sx_good, diag_left = None, a.diag
    phases = a.phase or ["default:"]
    total_bad = 0
    for ph in phases:
        name, _, kv = ph.partition(":")
        knobs = dict(x.split("=", 1) for x in kv.split(",") if x)
        for k, v in knobs.items():
            os.environ[k] = v
        bad, worst, t0, alone_bad, n_in, in_bad = 0, 0.0, time.time(), 0, 0, 0
        for rep in range(a.reps):
            if not a.no_poison:
                eng.poison_workspace()
            pcm = eng.codec_decode_batch(codes)
            bad_before = bad
            for u in range(len(caps)):
                if pcm[u].shape != alone0[u].shape:
                    print("%s rep %d utterance %d: length %d != %d" % (name, rep, u, pcm[u].size, alone0[u].size), flush=True)
                    bad += 1
                    continue
                dd = np.abs(pcm[u] - alone0[u])
                d = float(dd.max()) if dd.size else 0.0
                if not np.isfinite(d):
                    d = float("inf")
                worst = max(worst, d)
                if d > a.tol:
                    bad += 1
                    off = np.nonzero(~(dd <= a.tol))[0]
                    lo, hi = int(off[0]), int(off[-1])
                    w0, w1 = max(lo - 3, 0), min(min(hi, lo + 8) + 4, pcm[u].size)
                    print("%s rep %d utterance %d (%d frames, %d samples): %d samples off in [%d, %d], max %.3g" % (name, rep, u, caps[u], pcm[u].size, off.size, lo, hi, d))
                    print("   job  ", np.array2string(pcm[u][w0:w1], precision=6, max_line_width=250))
                    print("   alone", np.array2string(alone0[u][w0:w1], precision=6, max_line_width=250), flush=True)
            if a.clock and rep < a.clock:
                part_c = eng.final_conv_partials()
                if part_c is not None:
                    ticks, rt = part_c[:, 0, 7].astype(np.float64), part_c[:, 1, 7].astype(np.float64)
                    mhz = ticks / np.maximum(rt, 1.0) * 100.0
                    print("%s rep %d: %s; conv shader clock median %.0f MHz (min %.0f, max %.0f), FMA passes median %.2f us" % (
                        name, rep, "WRONG" if bad > bad_before else "clean", np.median(mhz), mhz.min(), mhz.max(), np.median(rt) / 100.0), flush=True)
            if a.check_input and sx_good is not None:
                sx_now, _ = eng.group_final_conv()
                n_in += 1
                if not np.array_equal(sx_now, sx_good):
                    in_bad += 1
                    dx = np.argwhere(sx_now != sx_good)
                    print("%s rep %d: final-conv INPUT differs from the clean job's at %d elements: %s" % (name, rep, len(dx), dx[:8].tolist()), flush=True)
            if (a.diag or a.check_input) and (sx_good is None or (diag_left > 0 and bad > bad_before)):
                sx, gp = eng.group_final_conv()
                if sx.size:
                    ref = np.stack([host_conv(sx[i]) for i in range(sx.shape[0])])
                    conv_err = np.abs(ref - gp)
                    if bad == bad_before:
                        if sx_good is None:
                            sx_good = sx
                            print("%s rep %d: clean job; group %s, |host conv(sx) - device pcm| max %.3g" % (name, rep, sx.shape, conv_err.max()), flush=True)
                    else:
                        diag_left -= 1
                        wrong = np.argwhere(conv_err > a.tol)
                        print("%s rep %d DIAG: device final conv differs from the host conv of ITS OWN input at %d samples: %s" % (name, rep, len(wrong), wrong[:12].tolist()))
                        # which per-row, per-tap partial sum D[row][tap] = sum_c w[tap][c] x[row][c] went wrong, and what took its place?
                        # out[t] = sum_tap D[t - 6 + tap][tap]; a wrong pair (t, t + 1) shares row t - 5 + k: slot (row, k + 1) feeds t, slot (row, k) feeds t + 1
                        for (q, t) in wrong[:8:2].tolist():
                            xq = np.concatenate([np.zeros((6, sx.shape[2])), sx[q].astype(np.float64)])
                            lo_r, hi_r = max(t - 80, 0), min(t + 80, sx.shape[1])
                            D = xq[lo_r:hi_r + 6] @ w_out            # D[i][tap] of padded row lo_r + i  (padded row p = source row p - 6)
                            d0, d1 = float(gp[q, t]) - ref[q, t], float(gp[q, t + 1]) - ref[q, t + 1]
                            for k in range(6):
                                pr = t + 1 + k                         # padded row index of the shared source row
                                x1, x0 = D[pr - lo_r, k + 1] + d0, D[pr - lo_r, k] + d1
                                m1 = np.argwhere(np.abs(D - x1) < 2e-6)
                                m0 = np.argwhere(np.abs(D - x0) < 2e-6)
                                if len(m1) and len(m0):
                                    print("      pair at (%d, %d): if row %d taps (%d, %d) were stored wrong: stored values equal D[row][tap] at %s and %s%s" % (
                                        q, t, pr - 6, k + 1, k, [(int(r_) + lo_r - 6, int(k_)) for r_, k_ in m1[:4]], [(int(r_) + lo_r - 6, int(k_)) for r_, k_ in m0[:4]],
                                        "; zero too" if abs(x1) < 2e-6 and abs(x0) < 2e-6 else ""))
                        part = eng.final_conv_partials()
                        if part is not None:
                            ticks, rt = part[:, 0, 7].astype(np.float64), part[:, 1, 7].astype(np.float64)
                            mhz = ticks / np.maximum(rt, 1.0) * 100.0
                            print("      shader clock over the FMA passes, per tile: median %.0f MHz (min %.0f, max %.0f); passes take median %.2f us" % (
                                np.median(mhz), mhz.min(), mhz.max(), np.median(rt) / 100.0))
                            part = part.copy(); part[:, :3, 7] = 0.0
                        if part is not None:   # [tile][row][tap]: tile b of sequence q covers padded rows 250 b .. 250 b + 255
                            tiles = part.shape[0] // sx.shape[0]
                            for q in sorted(set(int(x) for x in wrong[:, 0])):
                                xq = np.concatenate([np.zeros((6, sx.shape[2])), sx[q].astype(np.float64), np.zeros((512, sx.shape[2]))])
                                for b in sorted(set(int(t) // 250 for (qq, t) in wrong.tolist() if qq == q)):
                                    D = xq[250 * b: 250 * b + 256] @ w_out                         # [256][7]
                                    valid = (np.arange(250 * b, 250 * b + 256) - 6 < sx.shape[1])[:, None]
                                    e = np.abs(part[q * tiles + b][:, :7] - D * valid)
                                    badp = np.argwhere(e > 1e-6)
                                    print("      sequence %d tile %d: %d partial sums in LDS differ from the host's: (row, tap) %s" % (q, b, len(badp), badp[:12].tolist()))
                                    for (r_, k_) in badp[:6]:
                                        got = float(part[q * tiles + b][r_, k_])
                                        same = np.argwhere(np.abs(D - got) < 3e-7)
                                        print("         row %d tap %d: LDS %r, host %r; LDS value equals host D at (row, tap) %s" % (r_, k_, got, float(D[r_, k_]), same[:4].tolist()))
                                        xr = xq[250 * b + r_]
                                        P = np.array([[float(xr[12 * s_:12 * s_ + 12] @ w_out[12 * s_:12 * s_ + 12, kk]) for s_ in range(8)] for kk in range(7)])   # [tap][lane slice]
                                        print("            LDS - host = %.9f; the row's eight lane partials of this tap: %s" % (got - float(D[r_, k_]), np.array2string(P[k_], precision=9, max_line_width=250)))
                                        # one channel of every lane slice multiplied with ANOTHER channel's input (a stale broadcast register)?
                                        diff = got - float(D[r_, k_])
                                        for e_ in range(12):
                                            for e2 in range(12):
                                                if e2 != e_:
                                                    idx, idx2 = np.arange(8) * 12 + e_, np.arange(8) * 12 + e2
                                                    dd_ = float((w_out[idx, k_] * (xr[idx2] - xr[idx])).sum())
                                                    if abs(dd_ - diff) < 3e-7:
                                                        print("            = every lane's channel %d weight times channel %d's input instead of its own" % (e_, e2))
                                        # does the stored value equal the sum with lane slices taken from ANOTHER tap of the same row?
                                        for kk in range(7):
                                            for mask in range(1, 256):
                                                sel = np.array([(mask >> s_) & 1 for s_ in range(8)], bool)
                                                v_ = P[k_][~sel].sum() + P[kk][sel].sum() if kk != k_ else P[k_][~sel].sum()
                                                if abs(v_ - got) < 3e-7:
                                                    print("            = this tap's lanes %s + %s of lanes %s" % (np.nonzero(~sel)[0].tolist(), "tap %d" % kk if kk != k_ else "nothing", np.nonzero(sel)[0].tolist()))
                        if sx_good is not None:
                            dx = np.argwhere(sx != sx_good)
                            print("%s rep %d DIAG: final-conv INPUT differs from a clean job's at %d elements (seq, row, channel): %s" % (name, rep, len(dx), dx[:16].tolist()))
                            for (q, r, ch) in dx[:6]:
                                print("      sx[%d][%d][%d] = %r, clean %r" % (q, r, ch, float(sx[q, r, ch]), float(sx_good[q, r, ch])))
                        sys.stdout.flush()
            if rep % 50 == 49:
                for u in range(len(caps)):
                    al = eng.codec_decode(codes[u])
                    if not np.array_equal(al, alone0[u]):
                        alone_bad += 1
                        dd = np.abs(al - alone0[u])
                        off = np.nonzero(dd > 0)[0]
                        print("%s rep %d utterance %d: SINGLE decode moved: %d samples in [%d, %d], max %.3g" % (name, rep, u, off.size, off[0], off[-1], float(dd.max())), flush=True)
        for k in knobs:
            os.environ.pop(k, None)
        print("phase %-24s knobs %-40s reps %d: %d mismatching utterances, %d moved single decodes, worst |job - alone| %.3g, %.1f s%s"
              % (name, kv or "-", a.reps, bad, alone_bad, worst, time.time() - t0,
                 "; final-conv input checked on %d jobs: %d differ" % (n_in, in_bad) if a.check_input else ""), flush=True)
        total_bad += bad + alone_bad + in_bad
    eng.close()
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
