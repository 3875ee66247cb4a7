#!/usr/bin/env python3
"""Per-kernel evidence table: for every kernel of one 256-tile encode_image step on ONE internal stream,
   launches per step, average duration, share of the step, algorithmic work rate against its roofline, matrix-pipe busy share,
   effective shader clock while it ran, and HBM-side bytes per launch against the algorithmic bytes.

   python tools/kernel_table.py <kernel_stats_single_stream.csv> <pmc_mfma_counter_collection.csv> <hbm_traffic.json> <steps profiled> > table.md

Inputs come from tools/refresh_profiles.sh (rocprofv3 --kernel-trace --stats; --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
GRBM_GUI_ACTIVE in its own pass, also single-stream; FETCH_SIZE / WRITE_SIZE passes reduced by tools/pmc_traffic.py)."""
import csv, json, re, sys
from collections import defaultdict

stats_csv, pmc_csv, traffic_json, steps = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
bench_json = sys.argv[5] if len(sys.argv) > 5 else None        # optional: bench.py's line of the same build (HIP-event split of the proj / fc2 launches)
M = 256 * 197                       # rows of a 256-tile step (one lane when streams = 1)
PEAK_TF, PEAK_HBM = 2516.6, 8000.0
# kernel-name pattern -> (label, algorithmic FLOP per launch or None, algorithmic bytes per launch or None, roofline)
G = lambda n, k: 2.0 * M * n * k
# template arguments of gemm_f16_v2_kernel: <BN, WM, WN, NSTAGE, EPI, COMP (0 none | 1 W_lo term | 2 both terms), PERS>
KERNELS = [
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 0, 0, true>", "qkv GEMM (persistent, bias -> fp16)", G(3072, 1024), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 1, 0, true>", "fc1 GEMM (persistent, bias + GELU -> fp16)", G(4096, 1024), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 2, 0, true>", "fc2 GEMM (persistent, LayerScale + fp32 residual RMW)", G(1024, 4096), None, "mfma"),
    (r"gemm_f16_v2_kernel<128, 2, 2, 3, 2, 0, false>", "proj GEMM (256x128 tiles, two workgroups per CU, LayerScale + fp32 residual RMW)", G(1024, 1024), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 1, 2, false>", "fc1 GEMM + MX-fp4 correction phase (both terms)", G(4096, 1024), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 2, 2, false>", "fc2 GEMM + MX-fp4 correction phase (both terms)", G(1024, 4096), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 0, 2, false>", "qkv GEMM + MX-fp4 correction phase (the split-attention blocks)", G(3072, 1024), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 0, 0, false>", "qkv GEMM, split product (3 fp16 passes)", G(3072, 1024), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 2, 0, false>", "proj GEMM, split product (the split-attention blocks)", G(1024, 1024), None, "mfma"),
    (r"gemm_f16_v2_kernel<256, 2, 4, 4, 3, 0, false>", "patch-embed GEMM, split product", 2.0 * 256 * 196 * 768 * 1024, None, "mfma"),
    (r"attention_pers_kernel<13>", "attention (197 tokens, 16 heads; persistent, next pair's K / V staged under the compute)", 4.0 * 256 * 16 * 197 * 197 * 64, 2.0 * M * 4096, "mfma"),
    (r"attention_kernel<13, false, 8>", "attention, last block (CLS query only)", None, None, "mfma"),
    (r"attention_kernel<13, true, 4>", "attention, split product (the split-attention blocks)", 4.0 * 256 * 16 * 197 * 197 * 64, 4.0 * M * 4096, "mfma"),
    (r"layernorm_blk_kernel<4, 8>", "LayerNorm (fp32 in, fp16 K-blocked out; the average includes the 256-row launches of the CLS-row path)", None, None, "hbm"),
    (r"gemm_skinny_partial_wide_kernel", "CLS rows: small-M GEMM, 128x128 tiles, K-sliced partial products (proj / fc1 / fc2 of the CLS-row chain, last block's tail)", None, None, "mfma"),
    (r"gemm_skinny_partial_kernel", "small-M GEMM, 32x128 tiles (calls of < 64 rows)", None, None, "mfma"),
    (r"gemm_skinny_reduce", "CLS rows: partial-sum reduce + epilogue", None, None, "hbm"),
    (r"im2col_kernel", "im2col (+ cls / pos rows)", None, 256 * 3 * 224 * 224 * 2 + 2 * 2.0 * 256 * 196 * 768, "hbm"),
]

stats = {}
for row in csv.DictReader(open(stats_csv)):
    stats[row["Name"]] = (int(row["Calls"]), float(row["TotalDurationNs"]), float(row["AverageNs"]))
pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
pmc_dur = defaultdict(dict)          # kernel -> {dispatch id: ns}: durations of the SAME (counter-collecting) run, when the CSV carries timestamps
for row in csv.DictReader(open(pmc_csv)):
    a = pmc[row["Kernel_Name"]][row["Counter_Name"]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
    if row.get("Start_Timestamp") and row.get("End_Timestamp"):
        pmc_dur[row["Kernel_Name"]][row.get("Dispatch_Id", len(pmc_dur[row["Kernel_Name"]]))] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
traffic = json.load(open(traffic_json))
tr_by_pat = {v["kernel"]: v for v in traffic.values()}

total_ns = sum(t for _, t, _ in stats.values())
print("| kernel | launches / step | avg µs | % of step | algorithmic rate | frac of roofline | matrix pipe busy | eff. clock (MHz) | HBM-side bytes / launch (÷ algorithmic) |")
print("|---|---|---|---|---|---|---|---|---|")
seen = 0.0
for pat, label, flop, abytes, roof in KERNELS:
    hit = [(n, v) for n, v in stats.items() if pat in n]
    if not hit:
        continue
    name, (calls, tot, avg) = hit[0]
    seen += tot
    rate = frac = "—"
    if roof == "mfma" and flop:
        tf = flop / avg / 1e3
        rate, frac = f"{tf:.0f} TFLOP/s", f"{tf / PEAK_TF:.3f}"
    elif roof == "hbm" and abytes:
        gb = abytes / avg
        rate, frac = f"{gb:.0f} GB/s", f"{gb / PEAK_HBM:.3f}"
    kn, c = next(((n, v) for n, v in pmc.items() if pat in n), (None, None))
    busy = clk = "—"
    if c and c.get("GRBM_GUI_ACTIVE") and c["GRBM_GUI_ACTIVE"][1]:
        gui = c["GRBM_GUI_ACTIVE"][0] / c["GRBM_GUI_ACTIVE"][1] / 8.0     # the counter is summed over the 8 XCDs (an idle-chip probe kernel reads 8 x 2.4 GHz x its duration)
        mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / max(c["SQ_VALU_MFMA_BUSY_CYCLES"][1], 1) if "SQ_VALU_MFMA_BUSY_CYCLES" in c else 0.0
        busy = f"{mf / (gui * 1024):.2f}"                         # busy SIMD-cycles / (kernel cycles x 1024 SIMDs)
        d = pmc_dur.get(kn)
        dur = sum(d.values()) / len(d) if d else avg             # same-run duration when the CSV has timestamps, else the stats run's
        clk = f"{gui / dur * 1e3:.0f}" + ("" if d else "*")
    t = tr_by_pat.get(pat)
    tb = "—"
    if t:
        tb = f"{t['bytes_per_launch'] / 1e6:.0f} MB" + (f" ({t['traffic_over_algorithmic']}x)" if "traffic_over_algorithmic" in t else "")
    print(f"| `{label}` | {calls / steps:.1f} | {avg / 1e3:.1f} | {100 * tot / total_ns:.1f} | {rate} | {frac} | {busy} | {clk} | {tb} |")
print(f"| everything else | | | {100 * (total_ns - seen) / total_ns:.1f} | | | | | |")
print(f"\nstep = {total_ns / steps / 1e6:.2f} ms of kernel time on one stream ({steps:.0f} steps profiled); peaks: {PEAK_TF} TFLOP/s dense fp16 at 2.4 GHz, {PEAK_HBM:.0f} GB/s HBM.")
print("`matrix pipe busy` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); `eff. clock` = GRBM_GUI_ACTIVE / 8 / duration of the same dispatches "
      "(* = duration taken from the kernel-trace run: the counter CSV had no timestamps); counter-collecting runs serialise kernels and clock a little lower than plain ones. "
      "traffic rows: FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch of the single-stream run (256-tile launches).")
