"""MFMA issue rate of the codec decoder from hardware counters:
    Q3TTS_NULL_STREAM=1 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 -d OUT -o m --output-format csv \
        -- python tools/codec_bench.py --frames 2048 --reps 1
    python tools/pmc_mfma.py OUT/m_counter_collection.csv FRAMES_DECODED > profiles/..._pmc_mfma_codec.json
SQ_INSTS_VALU_MFMA_MOPS_* count the matrix cores' math operations in units of 512 (one v_mfma_f32_32x32x16_f16 = 32 768 FLOP = 64 units
per wave), so counted FLOP / kernel time is the matrix-core rate actually issued; against the 2.5 PFLOP/s dense 16-bit peak that is
the MFMA utilisation.  The split-precision path issues 3 products per fp32 product: algorithmic rate = issued rate / 3."""
import csv
import json
import sys
from collections import defaultdict

PEAK_16BIT_TFLOPS = 2500.0
path, frames = sys.argv[1], int(sys.argv[2])
flop = defaultdict(float)
dur = {}
name_of = {}
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0]
    for pre in ("void q3::", "q3::"):
        if k.startswith(pre):
            k = k[len(pre):]
    d = r["Dispatch_Id"]
    name_of[d] = k
    dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    if r["Counter_Name"].startswith("SQ_INSTS_VALU_MFMA_MOPS"):
        flop[d] += float(r["Counter_Value"]) * 512.0
by = defaultdict(lambda: [0.0, 0.0, 0])
for d, k in name_of.items():
    fam = k.split("<")[0]
    by[fam][0] += flop.get(d, 0.0)
    by[fam][1] += dur[d]
    by[fam][2] += 1
tot_f = sum(v[0] for v in by.values())
mf = {k: v for k, v in by.items() if v[0] > 0}
mf_t = sum(v[1] for v in mf.values())
out = {
    "frames_decoded": frames,
    "mfma_flop_counted": tot_f,
    "mfma_gflop_per_frame_counted": round(tot_f / frames / 1e9, 3),
    "note": "counted = issued by the matrix cores (hi*hi + hi*lo + lo*hi per fp32 product on the split path, plus tile padding)",
    "kernels_with_mfma": {
        k: {"dispatches": v[2], "seconds": round(v[1], 6), "TFLOP/s_issued": round(v[0] / v[1] / 1e12, 1),
            "mfma_util_vs_2.5PF": round(v[0] / v[1] / 1e12 / PEAK_16BIT_TFLOPS, 4)}
        for k, v in sorted(mf.items(), key=lambda kv: -kv[1][1])},
    "all_mfma_kernels": {"seconds": round(mf_t, 6), "TFLOP/s_issued": round(tot_f / mf_t / 1e12, 1) if mf_t else None,
                         "mfma_util_vs_2.5PF": round(tot_f / mf_t / 1e12 / PEAK_16BIT_TFLOPS, 4) if mf_t else None},
}
print(json.dumps(out, indent=1))
