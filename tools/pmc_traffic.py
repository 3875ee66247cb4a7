#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced read, so it is doubled (checked
here on kernels whose traffic is known: attention reads the 310 MB qkv tensor -> FETCH 151 MB; LayerNorm reads
206 MB -> FETCH 99 MB; WRITE_SIZE matched the known output sizes 1:1)."""
import csv, json, sys
from collections import defaultdict

def mean_per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            a = acc[row["Kernel_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: (s / n, n) for k, (s, n) in acc.items()}

fetch = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
# template arguments: <BN, WM, WN, NSTAGE, EPI, COMP (0 | 1 | 2), PERS>
TAGS = {"vit.fc1": "gemm_f16_v2_kernel<256, 2, 4, 4, 1, 0, true>", "vit.qkv": "gemm_f16_v2_kernel<256, 2, 4, 4, 0, 0, true>",
        "vit.fc2": "gemm_f16_v2_kernel<256, 2, 4, 4, 2, 0, true>", "vit.proj": "gemm_f16_v2_kernel<128, 2, 2, 3, 2, 0, false>",
        "vit.fc1+mxfp4": "gemm_f16_v2_kernel<256, 2, 4, 4, 1, 2, false>", "vit.fc2+mxfp4": "gemm_f16_v2_kernel<256, 2, 4, 4, 2, 2, false>",
        "vit.attn": "attention_pers_kernel<13>", "vit.ln": "layernorm_blk_kernel<4, 8>"}
# algorithmic bytes per launch of TILES tiles (argv[4], default 128 = one lane of the two-lane bench; 256 when the passes ran with --opt streams=1):
# operands read once + outputs written once (+ fp32 residual read-modify-write)
TILES = int(sys.argv[4]) if len(sys.argv) > 4 else 128
M = TILES * 197
ALGO = {"vit.fc1": 2 * (M * 1024 + 4096 * 1024) + 2 * M * 4096, "vit.qkv": 2 * (M * 1024 + 3072 * 1024) + 2 * M * 3072,
        "vit.attn": 2 * M * 3072 + 2 * M * 1024,
        "vit.proj": 2 * (M * 1024 + 1024 * 1024) + 8 * M * 1024, "vit.fc2": 2 * (M * 4096 + 4096 * 1024) + 8 * M * 1024}
out = {}
for tag, pat in TAGS.items():
    f = [(v, n) for k, (v, n) in fetch.items() if pat in k]
    w = [(v, n) for k, (v, n) in write.items() if pat in k]
    if f and w:
        out[tag] = {"kernel": pat, "fetch_kib_raw": round(f[0][0], 1), "write_kib": round(w[0][0], 1), "dispatches": f[0][1],
                    "bytes_per_launch": round((2 * f[0][0] + w[0][0]) * 1024), "tiles_per_launch": TILES}
        if tag in ALGO:
            out[tag]["algorithmic_bytes_per_launch"] = ALGO[tag]
            out[tag]["traffic_over_algorithmic"] = round(out[tag]["bytes_per_launch"] / ALGO[tag], 2)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
