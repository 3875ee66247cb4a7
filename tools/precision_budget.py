#!/usr/bin/env python3
"""Where the 1e-4 cosine tolerance is cheapest to buy: the MEASURED variance share and the MEASURED cost of every precision knob.

The 'comp' mode spends a per-block budget (keep_set_block_precision): block i's attention side plain / split / split with a compensated qkv /
compensated qkv only, its MLP plain / compensated (both MX-fp4 terms) / compensated (W_lo term only).  For every (block, knob) this tool measures

  d_var   the cosine-error variance that site contributes: the probe (256 seeded tiles x 64 seeded prompts) encoded with EVERYTHING as split
          products except that one site, minus the same with nothing downgraded (the engine's fp32-class floor).  Measured this way -- not as the
          difference between two fast settings -- because a change upstream re-draws the activation rounding of every block after it, which
          buries a 1 % share under the sampling noise of the other 99 %;
  d_ms    what the knob costs: 256-tile encode steps with the knob switched on in an otherwise plain-fp16 plan, minus the plain plan,
          interleaved A/B on this box (the part runs at its power cap: only same-box, same-minute comparisons mean anything).

and then builds plans greedily (largest d_var per d_ms first) for the calibration targets, next to the prefix ladder rungs, each verified by
encoding the probe.  Output: a markdown table + JSON (profiles/r05_precision_budget.{md,json}).

    python tools/precision_budget.py [--out gpurun_out/budget] [--steps 12] [--rounds 2] [--skip-cost]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keep_amd import KEEPModel, _lib                                                     # noqa: E402
from keep_amd.config import KEEPShape                                                    # noqa: E402
from keep_amd.model import (CALIBRATION_POPULATION, COMP_LADDER, max_sigmas_quantile, plan_string, prefix_plan)   # noqa: E402
from keep_amd.synth import synth_prompts, synth_state_dict                               # noqa: E402

A_PLAIN, A_SPLIT, A_SPLITQ, A_COMPQ = _lib.ATTN_PLAIN, _lib.ATTN_SPLIT, _lib.ATTN_SPLIT_COMPQKV, _lib.ATTN_COMPQKV
M_PLAIN, M_SPLIT, M_COMP, M_COMPW, M_CLS = _lib.MLP_PLAIN, _lib.MLP_SPLIT, _lib.MLP_COMP, _lib.MLP_COMP_W, _lib.MLP_CLS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/budget")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--skip-cost", action="store_true")
    ap.add_argument("--family", default="default")
    ap.add_argument("--no-bias", action="store_true", help="without the mean-input bias compensation (keep_calibrate_bias) the product applies at load")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    shape = KEEPShape()
    m = KEEPModel(shape)
    m.auto_calibrate = False
    m.load_state_dict(synth_state_dict(shape, seed=0, family=args.family))
    m.to(dev).eval()
    m.reserve(tiles=256)
    if not args.no_bias:
        m.calibrate_bias()
    depth = shape.vision.depth
    g = torch.Generator(device=dev).manual_seed(20250929)
    tiles = torch.randn(args.tiles, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
    toks = synth_prompts(64, 64, seed=20250929 % 100003)
    bank = m.encode_text({k: v.to(dev) for k, v in toks.items()})

    def cos(plan=None, precision="comp"):
        m.set_precision(precision)
        if plan is not None:
            m.set_plan(plan)
        out = torch.cat([m.similarity(m.encode_image(tiles[i:i + 256]), bank) for i in range(0, tiles.shape[0], 256)])
        return out

    ref = cos(precision="strict")
    split_all = [(A_SPLIT, M_SPLIT)] * depth
    plain_all = [(A_PLAIN, M_PLAIN)] * depth

    def var_of(plan):
        d = cos(plan) - ref
        return float(d.double().pow(2).mean()), float(d.abs().max())

    floor, _ = var_of(split_all)
    total, total_max = var_of(plain_all)
    print(f"probe: {tiles.shape[0]} tiles x {bank.shape[0]} prompts; all-plain rms {total ** 0.5:.3e} max {total_max:.3e}; all-split floor rms {floor ** 0.5:.3e}", flush=True)

    # ---- variance per (block, site treatment), everything else split -------------------------------------------------------------
    knobs = {"attn_plain": (A_PLAIN, None), "attn_compqkv_plain_rest": (A_COMPQ, None), "attn_split_compqkv": (A_SPLITQ, None),
             "mlp_plain": (None, M_PLAIN), "mlp_comp": (None, M_COMP), "mlp_comp_w": (None, M_COMPW), "mlp_cls": (None, M_CLS)}
    var = {k: [] for k in knobs}
    for i in range(depth):
        for k, (am, mm) in knobs.items():
            p = list(split_all)
            p[i] = (am if am is not None else A_SPLIT, mm if mm is not None else M_SPLIT)
            v, _ = var_of(p)
            var[k].append(max(v - floor, 0.0))
        print(f"block {i:2d}: " + "  ".join(f"{k} {var[k][i] / total * 100:6.2f}%" for k in knobs), flush=True)

    # ---- cost per knob: interleaved timing --------------------------------------------------------------------------------------
    xt = tiles[:256]

    def ms_of(plan):
        m.set_precision("comp")
        m.set_plan(plan)
        for _ in range(2):
            m.encode_image(xt)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m.encode_image(xt)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / args.steps * 1e3

    cost_knobs = {"attn_split": (A_SPLIT, None), "attn_split_compqkv": (A_SPLITQ, None), "attn_compqkv": (A_COMPQ, None),
                  "mlp_comp": (None, M_COMP), "mlp_comp_w": (None, M_COMPW), "mlp_cls": (None, M_CLS)}
    cost = {k: {} for k in cost_knobs}
    base_ms = []
    if not args.skip_cost:
        # the cost of a knob is (nearly) the same in every block but the last (CLS-rows-only tail): sample blocks 1, 12 and 23 and 4-block groups
        sample_blocks = [0, 1, depth // 2, depth - 1]
        for r in range(args.rounds):
            base_ms.append(ms_of(plain_all))
            for k, (am, mm) in cost_knobs.items():
                for i in sample_blocks:
                    p = list(plain_all)
                    p[i] = (am if am is not None else A_PLAIN, mm if mm is not None else M_PLAIN)
                    cost[k].setdefault(i, []).append(ms_of(p))
                # four blocks at once (blocks 4..7): a cleaner per-block figure than one block's difference of two ~38 ms steps
                p = list(plain_all)
                for i in range(4, 8):
                    p[i] = (am if am is not None else A_PLAIN, mm if mm is not None else M_PLAIN)
                cost[k].setdefault("4-7", []).append(ms_of(p))
            base_ms.append(ms_of(plain_all))
            print(f"cost round {r}: plain {base_ms[-2]:.3f} / {base_ms[-1]:.3f} ms", flush=True)
    base = sum(base_ms) / len(base_ms) if base_ms else float("nan")
    per_block_ms = {}
    for k in cost_knobs:
        if cost[k]:
            per_block_ms[k] = (sum(cost[k]["4-7"]) / len(cost[k]["4-7"]) - base) / 4.0
    print("per-block cost (ms per 256-tile step, two lanes, from the 4-block groups):", {k: round(v, 3) for k, v in per_block_ms.items()}, flush=True)

    # ---- plans: prefix ladder rungs and greedy plans, each verified -------------------------------------------------------------
    targets = {"location (round 4's rule)": 1e-4 / 5.51, "q=0.90": 1e-4 / max_sigmas_quantile(CALIBRATION_POPULATION, 0.90),
               "q=0.99": 1e-4 / max_sigmas_quantile(CALIBRATION_POPULATION, 0.99)}
    c = {"attn_split": per_block_ms.get("attn_split", 1.0), "attn_split_compqkv": per_block_ms.get("attn_split_compqkv", 0.8),
         "attn_compqkv": per_block_ms.get("attn_compqkv", 0.2), "mlp_comp": per_block_ms.get("mlp_comp", 0.52), "mlp_comp_w": per_block_ms.get("mlp_comp_w", 0.3),
         "mlp_cls": max(per_block_ms.get("mlp_cls", 0.05), 0.01)}
    # what is left of a site's variance under each treatment (measured above, per block)
    left_a = {A_PLAIN: lambda i: var["attn_plain"][i], A_COMPQ: lambda i: var["attn_compqkv_plain_rest"][i], A_SPLITQ: lambda i: var["attn_split_compqkv"][i], A_SPLIT: lambda i: 0.0}
    left_m = {M_PLAIN: lambda i: var["mlp_plain"][i], M_COMPW: lambda i: var["mlp_comp_w"][i], M_COMP: lambda i: var["mlp_comp"][i], M_SPLIT: lambda i: 0.0,
              M_CLS: lambda i: var["mlp_cls"][i]}
    cost_a = {A_PLAIN: 0.0, A_COMPQ: c["attn_compqkv"], A_SPLITQ: c["attn_split_compqkv"], A_SPLIT: c["attn_split"]}
    cost_m = {M_PLAIN: 0.0, M_COMPW: c["mlp_comp_w"], M_COMP: c["mlp_comp"], M_CLS: c["mlp_cls"]}

    def predict(plan):
        return floor + sum(left_a[a](i) + left_m[mm](i) for i, (a, mm) in enumerate(plan))

    def plan_cost(plan):
        return sum(cost_a[a] + cost_m.get(mm, 3 * c["mlp_comp"]) for a, mm in plan)

    def greedy(target_var, allow):
        am, mm = [A_PLAIN] * depth, [M_PLAIN] * depth
        order = []
        while predict(list(zip(am, mm))) > target_var:
            best, gain = None, 0.0
            for i in range(depth):
                for a in allow["attn"]:
                    dc, dv = cost_a[a] - cost_a[am[i]], left_a[am[i]](i) - left_a[a](i)
                    if dc > 0 and dv / dc > gain:
                        best, gain = ("a", i, a), dv / dc
                for mo in allow["mlp"]:
                    dc, dv = cost_m[mo] - cost_m[mm[i]], left_m[mm[i]](i) - left_m[mo](i)
                    if dc > 0 and dv / dc > gain:
                        best, gain = ("m", i, mo), dv / dc
            if best is None:
                break
            (am if best[0] == "a" else mm)[best[1]] = best[2]
            order.append((best, gain))
        return list(zip(am, mm)), order

    results = []

    def verify(name, plan, timed=True):
        v, mx = var_of(plan)
        ms = ms_of(plan) if (timed and not args.skip_cost) else float("nan")
        row = {"name": name, "plan": plan_string(plan), "probe_rms": v ** 0.5, "probe_max": mx, "predicted_rms": predict(plan) ** 0.5,
               "predicted_extra_ms": plan_cost(plan), "ms_per_step": ms, "tiles_per_s": 256e3 / ms if ms == ms else None}
        results.append(row)
        print(f"{name:44s} rms {row['probe_rms']:.3e} (predicted {row['predicted_rms']:.3e}) max {mx:.3e}  {ms:7.3f} ms/step  {plan_string(plan)}", flush=True)
        return row

    for full, mlp in COMP_LADDER[2:9]:
        verify(f"prefix {full}/{mlp}", prefix_plan(depth, full, mlp))
    verify("all plain", plain_all)
    verify("every MLP plain + CLS rows split", [(A_PLAIN, M_CLS)] * depth)
    verify("block 0 attn split+comp qkv; every MLP plain + CLS rows split", [(A_SPLITQ, M_CLS)] + [(A_PLAIN, M_CLS)] * (depth - 1))
    verify("block 0 attn split+comp qkv, MLP comp; other MLPs plain + CLS rows split", [(A_SPLITQ, M_COMP)] + [(A_PLAIN, M_CLS)] * (depth - 1))
    allows = {"all knobs": {"attn": [A_COMPQ, A_SPLITQ, A_SPLIT], "mlp": [M_CLS, M_COMPW, M_COMP]},
              "no CLS-row knob (round-5 mid-way)": {"attn": [A_COMPQ, A_SPLITQ, A_SPLIT], "mlp": [M_COMPW, M_COMP]},
              "attn split + mlp comp (the prefix family's knobs, any block)": {"attn": [A_SPLIT], "mlp": [M_COMP]},
              "the product's default knobs": {"attn": [A_SPLITQ, A_SPLIT], "mlp": [M_CLS, M_COMP]}}
    for tname, trms in targets.items():
        for aname, allow in allows.items():
            plan, order = greedy(trms ** 2, allow)
            row = verify(f"greedy[{aname}] -> {tname} ({trms:.3e})", plan)
            # the prediction adds variances; if the verified rms misses, tighten the target by the miss and try once more
            if row["probe_rms"] > trms:
                plan2, _ = greedy((trms ** 2) * (trms / row["probe_rms"]) ** 2, allow)
                verify(f"  ... tightened", plan2)
    if not args.skip_cost:
        base_ms.append(ms_of(plain_all))
    out = {"probe": f"{tiles.shape[0]} tiles x {bank.shape[0]} prompts, bench weights (family {args.family}), mean-input bias compensation {'off' if args.no_bias else 'on'}", "all_plain_rms": total ** 0.5, "floor_rms": floor ** 0.5,
           "variance_share_of_all_plain": {k: [v / total for v in vs] for k, vs in var.items()}, "variance_abs": var,
           "plain_ms_per_step": base_ms, "cost_samples_ms": {k: {str(b): v for b, v in d.items()} for k, d in cost.items()}, "per_block_cost_ms": per_block_ms,
           "targets_rms": targets, "plans": results}
    json.dump(out, open(os.path.join(args.out, "precision_budget.json"), "w"), indent=1)
    with open(os.path.join(args.out, "precision_budget.md"), "w") as f:
        f.write(f"Probe: {out['probe']}.  All-plain rms {total ** 0.5:.3e}; all-split floor {floor ** 0.5:.3e}.  Shares in % of the all-plain variance; "
                f"'left' = what remains of the site's share under that treatment.  Costs in ms per 256-tile step (plain plan: {base:.2f} ms).\n\n")
        f.write("| block | attn side plain | left: comp. qkv only | left: split + comp. qkv | MLP plain | left: MLP comp (2 terms) | left: MLP comp (W_lo only) | left: MLP plain + CLS rows split |\n|---|---|---|---|---|---|---|---|\n")
        for i in range(depth):
            f.write(f"| {i} | " + " | ".join(f"{var[k][i] / total * 100:.2f}" for k in ("attn_plain", "attn_compqkv_plain_rest", "attn_split_compqkv", "mlp_plain", "mlp_comp", "mlp_comp_w", "mlp_cls")) + " |\n")
        f.write("\n| knob | ms per block |\n|---|---|\n")
        for k, v in per_block_ms.items():
            f.write(f"| {k} | {v:.3f} |\n")
        f.write("\n| plan | probe rms | predicted rms | probe max | ms / step | tiles/s | per-block modes |\n|---|---|---|---|---|---|---|\n")
        for r in results:
            f.write(f"| {r['name']} | {r['probe_rms']:.3e} | {r['predicted_rms']:.3e} | {r['probe_max']:.3e} | {r['ms_per_step']:.3f} | "
                    f"{(r['tiles_per_s'] or 0):.0f} | {r['plan']} |\n")
    print(open(os.path.join(args.out, "precision_budget.md")).read())


if __name__ == "__main__":
    main()
