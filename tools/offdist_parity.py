#!/usr/bin/env python3
"""Does the calibrated 'comp' plan hold the 1e-4 cosine tolerance OFF the distribution its calibration probes?

Up to round 5 `KEEPModel.calibrate()` / `calibrate_bias()` probed seeded N(0,1) tiles only, and every parity figure of this repository was measured on N(0,1)
tiles too (`--probe gaussian --bias-probe gaussian` reproduces that build's choices; profiles/r06_offdist_parity_before_repair_gaussian_probe.json).  This tool
evaluates the model -- calibrated exactly as `load_state_dict` leaves it -- on the structured tile families of `keep_amd.synth.synth_tile_family` (real-image
crops, Beer-Lambert stain fields, glass background, half / half) and on the N(0,1) control:

  oracle leg   `--oracle-tiles` tiles per family in 'comp' and 'strict' against the fp32 CPU oracle (cosines against 64 prompts)
  slide leg    `--tiles` tiles per family in 'comp' against 'strict': every cosine against 64 and 264 prompts, with the bias compensation
               on and off, plus the all-plain fp16 plan as the yardstick of how hard the family is (`--only-calibrated`: the calibrated plan alone)

    python tools/offdist_parity.py [--tiles 12500] [--oracle-tiles 64] [--probe default|mixture] [--out gpurun_out/offdist.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                              # noqa: E402
from keep_amd import KEEPModel                                            # noqa: E402
from keep_amd.config import KEEPShape
from keep_amd.model import CALIBRATION_POPULATION, mixture_exceedance                                     # noqa: E402
from keep_amd.synth import (normalise_u8, synth_prompts, synth_state_dict, synth_tile_family, synth_tiles_device)   # noqa: E402

FAMILIES = ("he_crops", "stain_field", "background", "half")


def family_tiles(family, a, b, dev, seed):
    """float tiles [b-a,3,224,224] bf16-free: uint8 families go through the engine's own uint8 path"""
    if family == "gaussian":
        return synth_tiles_device(a, b, dev, torch.bfloat16, seed=seed)
    return synth_tile_family(family, a, b, dev, seed=seed)


def encode(model, x):
    return model.encode_image_uint8(x) if x.dtype == torch.uint8 else model.encode_image(x)


def encode_all(model, family, n, dev, seed):
    out = torch.empty(n, 768, device=dev)
    for a in range(0, n, 256):
        b = min(a + 256, n)
        out[a:b] = encode(model, family_tiles(family, a, b, dev, seed))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=12_500)
    ap.add_argument("--oracle-tiles", type=int, default=64)
    ap.add_argument("--families", nargs="*", default=list(FAMILIES) + ["gaussian"])
    ap.add_argument("--seed", type=int, default=7000)
    ap.add_argument("--probe", default="mixture", choices=["mixture", "gaussian"], help="what the load-time calibration probes (KEEPModel.calibration_probe)")
    ap.add_argument("--bias-probe", default=None, choices=["mixture", "gaussian", "off"], help="what calibrate_bias probes (default: the same as --probe)")
    ap.add_argument("--weight-family", default="default", help="weight family of keep_amd.synth (default | heavy_tail | small_ls)")
    ap.add_argument("--weight-seed", type=int, default=0)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--only-calibrated", action="store_true", help="skip the no-compensation / all-plain yardsticks (full-size runs)")
    ap.add_argument("--out", default="gpurun_out/offdist_parity.json")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    shape = KEEPShape()
    sd = synth_state_dict(shape, seed=args.weight_seed, family=args.weight_family)
    model = KEEPModel(shape)
    model.calibration_probe = args.probe
    if args.bias_probe is not None and args.bias_probe != args.probe:
        model.auto_calibrate = False
    model.load_state_dict(sd)
    model.to(dev).eval()
    if not model.auto_calibrate:
        if args.bias_probe == "off":
            model.set_option("bias_correction", 0)
        else:
            model.calibrate_bias(probe=args.bias_probe)
        model.calibrate(probe=args.probe)
    model.reserve(tiles=256)
    own = model.get_plan()
    depth = len(own)
    bias_was = int(model.get_option("bias_correction"))
    res = {"calibration": model.calibration, "families": {}, "weight_family": args.weight_family, "weight_seed": args.weight_seed}
    xb = synth_tiles_device(0, 256, dev, torch.bfloat16, seed=1234)
    for _ in range(5):
        model.encode_image(xb)
    res["ms_per_256_tile_step"] = round(bench.time_gpu(lambda: model.encode_image(xb), dev, 30) * 1e3, 3)
    cal = model.calibration or {}
    print(f"[calibration] probe {cal.get('probe_distribution')} plan {cal.get('plan')} governing {cal.get('governing_group')} predicted {cal.get('predicted_max_abs_dcos')} "
          f"tried {len(cal.get('tried', []))}; {res['ms_per_256_tile_step']} ms per 256-tile step", flush=True)
    for t in cal.get("tried", []):
        print(f"    tried {t['plan']} iso {t.get('isotropic_rms_by_group')} aniso {t.get('anisotropy_factor')} governing {t.get('governing_group')} predicted {t['predicted_max_abs_dcos']}", flush=True)
    for name, v in (cal.get("per_group") or {}).items():
        print(f"    {name}: {v}", flush=True)
    for name, v in ((cal.get("variance_shares") or {}).get("by_group") or {}).items():
        tot = v["floor"] + sum(v["attn"]) + sum(v["mlp"])
        print(f"    shares[{name}] all-plain rms {tot ** 0.5:.3e} floor {v['floor']:.2e} attn% {[round(100 * a / tot, 1) for a in v['attn']]} mlp% {[round(100 * a / tot, 1) for a in v['mlp']]} "
              f"cls_left {v['residual_mlp'][4]} res_m {[(k, r) for k, r in v['residual_mlp'].items() if k != 4]} res_a {v['residual_attn']}", flush=True)
    toks64, toksD = synth_prompts(64, 256, seed=1), synth_prompts(264, 256, seed=5)

    def banks():
        t64 = model.encode_text({k: v.to(dev) for k, v in toks64.items()})
        tD = torch.cat([model.encode_text({k: v[i:i + 64].to(dev) for k, v in toksD.items()}) for i in range(0, 264, 64)])
        return t64, tD

    model.set_precision("comp")
    model.set_plan(own)
    t64_c, tD_c = banks()
    model.set_precision("strict")
    t64_s, tD_s = banks()
    if not args.no_oracle:
        from oracle import keep_oracle as O
        torch.set_num_threads(min(bench.usable_cpus(), 64))
        with torch.no_grad():
            t64_o = O.encode_text(sd, toks64)
    for fam in args.families:
        t0 = time.perf_counter()
        r = {}
        if not args.no_oracle and args.oracle_tiles > 0:
            x = family_tiles(fam, 0, args.oracle_tiles, dev, args.seed + 1)
            xf = (normalise_u8(x) if x.dtype == torch.uint8 else x.float()).cpu()
            with torch.no_grad():
                ref = O.similarity(O.encode_image(sd, xf), t64_o)
            for mode in ("comp", "strict"):
                model.set_precision(mode)
                model.set_plan(own)
                f = encode(model, x)
                txt = t64_c if mode == "comp" else t64_s
                sim = model.similarity(f, txt).cpu()
                d = (sim - ref).abs()
                lab = model.classify(x, txt)[1].cpu() if mode == "comp" else sim.argmax(1)
                r[f"{mode}_vs_oracle"] = {"tiles": args.oracle_tiles, "max_abs_dcos": float(f"{d.max():.3e}"), "rms_dcos": float(f"{d.pow(2).mean().sqrt():.3e}"),
                                          "labels_equal": int((lab.long() == ref.argmax(1)).sum()), "distinct_oracle_labels": int(ref.argmax(1).unique().numel())}
        n = args.tiles
        model.set_precision("strict")
        f_s = encode_all(model, fam, n, dev, args.seed)
        s64_s, sD_s = model.similarity(f_s, t64_s), model.similarity(f_s, tD_s)
        model.set_precision("comp")
        variants = (("calibrated_plan", own, bias_was),) if args.only_calibrated else (
            ("calibrated_plan", own, 1), ("calibrated_plan_no_bias_compensation", own, 0),
            ("all_plain_fp16", [(0, 0)] * depth, 1), ("all_plain_fp16_no_bias_compensation", [(0, 0)] * depth, 0))
        for label, plan, bias in variants:
            model.set_plan(plan)
            model.set_option("bias_correction", bias)
            f = encode_all(model, fam, n, dev, args.seed)
            e = (f - f_s)
            sig = e.double().pow(2).sum(1).div(768).sqrt().tolist()
            r[label] = {"population_exceedance_from_slide_tile_errors": float(f"{mixture_exceedance(sig, CALIBRATION_POPULATION, 1e-4):.3e}"),
                        "hardest_tile_over_rms": round(max(sig) / (sum(v * v for v in sig) / len(sig)) ** 0.5, 2),
                        "cos_vs_64_prompts": bench.diff_stats(model.similarity(f, t64_c) - s64_s),
                        "cos_vs_264_distinct_prompts": bench.diff_stats(model.similarity(f, tD_c) - sD_s),
                        "isotropic_rms": float(f"{float(e.pow(2).sum(1).mean().div(768).sqrt()):.3e}"),
                        "mean_error_vector_norm_over_rms_norm": round(float(e.mean(0).norm() / e.pow(2).sum(1).mean().sqrt()), 3)}
        model.set_option("bias_correction", bias_was)
        model.set_plan(own)
        # how different the family's features are from each other (a near-constant family collapses to few points)
        r["feature_spread"] = {"mean_pairwise_cos_first_512": round(float((f_s[:512] @ f_s[:512].t()).mean()), 4)}
        r["seconds"] = round(time.perf_counter() - t0, 1)
        res["families"][fam] = r
        c = r["calibrated_plan"]
        if args.only_calibrated:
            print(f"[{fam}, {n} tiles] 64 prompts: max {c['cos_vs_64_prompts']['max_abs']:.3e} rms {c['cos_vs_64_prompts']['rms']:.3e} over {c['cos_vs_64_prompts']['over_1e-4']}; "
                  f"264 prompts: max {c['cos_vs_264_distinct_prompts']['max_abs']:.3e} rms {c['cos_vs_264_distinct_prompts']['rms']:.3e} over {c['cos_vs_264_distinct_prompts']['over_1e-4']} "
                  f"max/rms {c['cos_vs_264_distinct_prompts']['max_over_rms']}; isotropic rms {c['isotropic_rms']:.3e} hardest tile / rms {c['hardest_tile_over_rms']}; "
                  f"population exceedance from this slide {c['population_exceedance_from_slide_tile_errors']:.2e}", flush=True)
            continue
        print(f"[{fam}] calibrated: 64p max {c['cos_vs_64_prompts']['max_abs']:.3e} rms {c['cos_vs_64_prompts']['rms']:.3e} over {c['cos_vs_64_prompts']['over_1e-4']}; "
              f"exceed {c['population_exceedance_from_slide_tile_errors']:.2e} hardest/rms {c['hardest_tile_over_rms']} iso {c['isotropic_rms']:.3e}; "
              f"264p max {c['cos_vs_264_distinct_prompts']['max_abs']:.3e} rms {c['cos_vs_264_distinct_prompts']['rms']:.3e} over {c['cos_vs_264_distinct_prompts']['over_1e-4']}; "
              f"no-bias rms {r['calibrated_plan_no_bias_compensation']['cos_vs_264_distinct_prompts']['rms']:.3e}; plain rms {r['all_plain_fp16']['cos_vs_264_distinct_prompts']['rms']:.3e} "
              f"(no bias {r['all_plain_fp16_no_bias_compensation']['cos_vs_264_distinct_prompts']['rms']:.3e}); oracle {r.get('comp_vs_oracle')} {r.get('strict_vs_oracle')}", flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
