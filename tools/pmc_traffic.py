"""HBM traffic per decode step from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section):
bytes = FETCH_SIZE*1024*2 (gfx950 reports exactly half of a wide coalesced read stream) + WRITE_SIZE*1024.
Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <frames> <batch>"""
import csv
import json
import sys
from collections import defaultdict

DECODE = ("k_gemv", "k_gemm", "k_attn", "k_cp_attn_oproj", "k_sample", "k_finish", "k_rmsnorm_split")


def total(path, counter):
    tot = defaultdict(float)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            tot[name.split("(")[0].replace("void q3::", "").replace("q3::", "")] += float(r["Counter_Value"])
    return tot


fetch, write = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE")
frames, batch = int(sys.argv[3]), int(sys.argv[4])
rd = sum(v for k, v in fetch.items() if k.startswith(DECODE)) * 1024 * 2
wr = sum(v for k, v in write.items() if k.startswith(DECODE)) * 1024
out = {"frames": frames, "batch": batch, "read_bytes_per_step": rd / frames, "write_bytes_per_step": wr / frames,
       "hbm_bytes_per_step": (rd + wr) / frames,
       "note": "decode kernels only (k_gemv*/k_gemm*/k_attn*/k_cp_attn_oproj/k_sample/k_finish*), includes the one prefill; FETCH_SIZE doubled per the gfx950 correction",
       "by_kernel_read_MB_per_step": {k: round(v * 2048 / frames / 1e6, 2) for k, v in sorted(fetch.items(), key=lambda kv: -kv[1])[:8]}}
print(json.dumps(out))
