#!/usr/bin/env python3
"""How much do the two lanes of keep_encode_image actually overlap?  From a rocprofv3 --kernel-trace CSV of encode steps: for every kernel family the time it
spends alone on the GPU and the time it shares with a kernel of the OTHER queue (and which), plus the wall time covered by 0 / 1 / 2+ kernels.

    rocprofv3 --kernel-trace -d /tmp/kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-sustained --no-breakdown
    python tools/lane_overlap.py /tmp/kt/**/*kernel_trace.csv
"""
import csv, glob, sys
from collections import defaultdict

def family(name):
    n = name
    if "gemm_f16_v2_kernel<256, 2, 4, 4, 0, 0, true>" in n: return "qkv"
    if "gemm_f16_v2_kernel<256, 2, 4, 4, 1, 0, true>" in n: return "fc1"
    if "gemm_f16_v2_kernel<256, 2, 4, 4, 2, 0, true>" in n: return "fc2"
    if "gemm_f16_v2_kernel<128, 2, 2, 3, 2, 0, false>" in n: return "proj"
    if "gemm_f16_v2_kernel" in n: return "gemm(other)"
    if "attention" in n: return "attn"
    if "layernorm" in n: return "ln"
    if "skinny" in n: return "cls-gemm"
    return "other"

rows = []
for path in sys.argv[1:]:
    for f in glob.glob(path, recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), family(r["Kernel_Name"])))
rows.sort()
# keep the steady part: the last 60 % of the trace (warm-up, calibration and probes come first)
t_lo = rows[0][0] + 0.4 * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
events = []
for i, (a, b, q, fam) in enumerate(rows):
    events.append((a, 1, i)); events.append((b, -1, i))
events.sort()
active, last = set(), events[0][0]
cover = defaultdict(int)                       # number of active kernels -> ns
alone, shared = defaultdict(int), defaultdict(lambda: defaultdict(int))
for t, d, i in events:
    dt = t - last
    if dt > 0:
        cover[min(len(active), 3)] += dt
        fams = [rows[j][3] for j in active]
        queues = {rows[j][2] for j in active}
        for j in active:
            if len(queues) > 1:
                for k in active:
                    if rows[k][2] != rows[j][2]:
                        shared[rows[j][3]][rows[k][3]] += dt
            else:
                alone[rows[j][3]] += dt
    last = t
    (active.add if d > 0 else active.discard)(i)
tot = sum(cover.values())
print(f"wall time analysed {tot / 1e6:.1f} ms: idle {100 * cover[0] / tot:.1f} %, one kernel {100 * cover[1] / tot:.1f} %, two {100 * cover[2] / tot:.1f} %, three or more {100 * cover[3] / tot:.1f} %")
dur = defaultdict(int)
for a, b, q, fam in rows:
    dur[fam] += b - a
print(f"{'kernel':12s} {'sum of durations':>18s} {'alone':>8s} {'beside the other lane':>22s}   with what")
for fam in sorted(dur, key=lambda f: -dur[f]):
    sh = sum(shared[fam].values())
    top = ", ".join(f"{k} {100 * v / max(sh, 1):.0f}%" for k, v in sorted(shared[fam].items(), key=lambda kv: -kv[1])[:4])
    print(f"{fam:12s} {dur[fam] / 1e6:15.2f} ms {100 * alone[fam] / max(dur[fam], 1):7.1f}% {100 * sh / max(dur[fam], 1):21.1f}%   {top}")
