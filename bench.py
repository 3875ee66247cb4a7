#!/usr/bin/env python3
"""Headline benchmark: 224x224 tiles encoded per second through the ViT-L/16 image tower
(BASELINE.json configs[1]: batch 256 synthetic bf16 tiles per GPU, random-init weights).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                      (starts the N ranks itself: re-executes under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...        (the same, launched from outside)

One step = KEEPModel.encode_image on one batch of 256 device-resident tiles per rank (+ the RCCL
all-gather of the [256,768] embeddings when N > 1: the slide-level pooling exchange of config 4).
Rank 0 prints ONE JSON line.  `value` is measured over exactly --steps steps, in the DEFAULT precision mode
('comp': the fastest mode whose cosines stay within the 1e-4 reference tolerance); on top of that contract the line carries
  sustained   the same step repeated for >= 10 s (the part runs at its power cap; a sub-second burst is not a sustained rate)
  roofline    the time-dominant kernel (the persistent fp32-residual GEMM: vit.proj + vit.fc2), HIP-event timed on one stream with nothing
              else on the GPU (a KERNEL figure); the same launches inside the two-lane timed region under a separate key; every other kernel
  parity      BASELINE config 3 (4096 tiles x 64 prompts through both towers, similarity + argmax) against the committed
              fp32-oracle fixture tests/golden/c3_dual_tower.npz: 262 144 cosines, match-rate, argmax agreement
  configs     c3 (dual tower), c4 (100 000-tile slides: every cosine of the benched setting against the split-product mode, screening scores,
              slide label, tumour ratio), c4_structured (the same on a slide of STRUCTURED uint8 tiles: real-image crops, stain fields, glass
              background -- not the N(0,1) pixels the headline is timed on) and c5 (100 000-tile x 2-class fp16 probability map, all of it
              against the oracle)
  cpu_baseline  the oracle's encode_image and encode_text on the host cores (bounded samples)
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from keep_amd.distributed import StepExchange, rccl_env, timed_steps       # noqa: E402

rccl_env()        # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: what RCCL's multi-process transport needs on these nodes) before HIP starts

from keep_amd import KEEPModel, PROFILE_TAGS, vit_flops_per_tile          # noqa: E402
from keep_amd.config import KEEPShape                                      # noqa: E402
from keep_amd.model import plan_prefix, plan_string                       # noqa: E402
from keep_amd.synth import synth_prompts, synth_state_dict, synth_tiles    # noqa: E402

PEAK_F16_TFLOPS = 2516.6     # 256 CU x 4096 FLOP/clk x 2.4 GHz, dense
PEAK_HBM_GBS = 8000.0
# The kernel that takes the largest share of a step: gemm_f16_v2_kernel<256,2,4,4,EPI_RESID_LS,0,true> -- the persistent 256x256 GEMM with the
# LayerScale + fp32 residual read-modify-write epilogue, launched for fc2 ([B*197,4096] x [4096,1024]) of every block that runs it as a single fp16
# pass (26-27 % of the step: profiles/r06_per_kernel_table.md; proj has its own kernel since round 5).  The engine times those launches under this tag.
DOMINANT_TAGS = ("vit.fc2",)
DOMINANT_KERNEL = "keepk::gemm_f16_v2_kernel<256,2,4,4,EPI_RESID_LS,0,true> (persistent; the vit.fc2 launches)"
BERT_FLOPS_PER_PROMPT_256 = 45_903_642_624
DTYPE_NAME = {"fp16": "fp16", "comp": "fp16+mxfp4", "strict": "fp16x3"}

T_START = time.perf_counter()


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(sd, tiles: int = 8, iters: int = 40, min_seconds: float = 10.0):
    """The oracle (fp32 torch-CPU restatement of the reference's encode_image / encode_text) on the host cores: tiles/s and, as SURVEY.md 8(d)
    asks, prompts/s on 8 prompts x 256 tokens.  Thread policy: torch intra-op threads = the cores this process may use (affinity mask capped by
    the cgroup quota), at most 64; one process."""
    from oracle import keep_oracle as O
    torch.set_num_threads(min(usable_cpus(), 64))
    x = synth_tiles(tiles, seed=0)
    with torch.no_grad():
        O.encode_image(sd, x[:1])                                   # warm-up
        t0 = time.perf_counter()
        done = 0
        for _ in range(iters):
            O.encode_image(sd, x)
            done += 1
            if time.perf_counter() - t0 > min_seconds:              # bounded sample: ~10 s of CPU work
                break
        dt = time.perf_counter() - t0
        toks = synth_prompts(8, 256, seed=1)
        O.encode_text(sd, {k: v[:1] for k, v in toks.items()})      # warm-up
        t1 = time.perf_counter()
        pdone = 0
        for _ in range(12):
            O.encode_text(sd, toks)
            pdone += 1
            if time.perf_counter() - t1 > 4.0:                      # ~4 s more
                break
        pdt = time.perf_counter() - t1
    iters = done
    return {"value": round(tiles * iters / dt, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "prompts_per_s": round(8 * pdone / pdt, 2),
            "thread_policy": "one process, torch intra-op threads = min(cores granted by affinity mask and cgroup quota, 64)",
            "sample": f"{iters} x {tiles} synthetic 224x224 tiles through encode_image and {pdone} x 8 prompts x 256 tokens (padded length, as the "
                      "reference computes it) through encode_text, fp32 torch-CPU restatement (oracle/keep_oracle.py) of KEEPModel, same "
                      f"synthetic weights; CPU: {cpu_model_name()}"}


def tiles_per_s_for_ceiling(world, B, steps, elapsed) -> float:
    return B * steps / elapsed           # per GPU


def lib_sha16() -> str:
    import hashlib
    from keep_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return "unknown"


def time_gpu(fn, dev, reps: int):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps


def config3(model, dev):
    """BASELINE config 3: full dual tower, 4096 tiles x 64 prompts, similarity matrix + argmax, one GPU -- timed, and
    compared with the fp32-oracle fixture (tools/make_golden.py c3; same seeded weights, tiles and prompts)."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "c3_dual_tower.npz")
    if not os.path.exists(path):
        return None, None
    g = np.load(path)
    n, chunk, P = int(g["n_tiles"]), int(g["chunk"]), int(g["input_ids"].shape[0])
    tiles = torch.empty(n, 3, 224, 224, device=dev, dtype=torch.float32)        # fp32 pixels: exactly what the oracle saw
    for c in range(n // chunk):
        tiles[c * chunk:(c + 1) * chunk] = synth_tiles(chunk, seed=int(g["tile_seed0"]) + c).to(dev)
    toks = {"input_ids": torch.from_numpy(g["input_ids"].astype("int64")).to(dev),
            "attention_mask": torch.from_numpy(g["attention_mask"].astype("int64")).to(dev)}
    toks["token_type_ids"] = torch.zeros_like(toks["input_ids"])
    model.reserve(tiles=chunk, prompts=P, seq=256)

    txt = model.encode_text(toks)

    def run():
        # the default path of config 3: KEEPModel.classify = default-precision encode of every tile, similarity + argmax, and a second,
        # split-product encode of the tiles whose top-2 margin is below the engine's label_margin (keep_classify in the C ABI)
        return model.classify(tiles, txt, return_features=True)

    run()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    txt = model.encode_text(toks)
    sim, lab, feats = run()
    torch.cuda.synchronize(dev)
    t_all = time.perf_counter() - t0
    rechecked = model.last_rechecked
    t_cls = time_gpu(run, dev, 1)
    t_img = time_gpu(lambda: [model.encode_image(tiles[i:i + chunk]) for i in range(0, n, chunk)], dev, 1)
    t_txt = time_gpu(lambda: model.encode_text(toks), dev, 5)
    plain = torch.cat([model.encode_image(tiles[i:i + chunk]) for i in range(0, n, chunk)])
    psim, plab = model.similarity(plain, txt, mode="argmax")             # what the labels would be without the second look
    t_sim = time_gpu(lambda: model.similarity(plain, txt, mode="argmax"), dev, 20)
    ref = torch.from_numpy(g["sims"]).to(dev)
    ref_lab = torch.from_numpy(g["argmax"].astype("int64")).to(dev)
    margin = torch.from_numpy(g["margin"]).to(dev)
    d = (sim - ref).abs()
    dp = (psim - ref).abs()
    d_img = (plain @ torch.from_numpy(g["txt"]).to(dev).t() - ref).abs()            # image-tower share: GPU tiles x oracle text features
    same = lab.long() == ref_lab
    psame = plab.long() == ref_lab
    near = margin < 2e-4                                                            # tiles whose two best prompts are closer than 2 x tolerance
    parity = {"max_abs_dcos": float(f"{max(d.max().item(), dp.max().item()):.3e}"), "rms_dcos": float(f"{dp.pow(2).mean().sqrt().item():.3e}"),
              "n_cosines": int(d.numel()), "sim_match_rate_1e-4": round(float(((d <= 1e-4) & (dp <= 1e-4)).float().mean().item()), 6),
              "argmax_match_rate": round(float(same.float().mean().item()), 6),
              "argmax_mismatches": int((~same).sum().item()), "labels_bit_exact": bool(same.all().item()),
              "labels_from": "KEEPModel.classify (keep_classify): default-precision encode + split-product second look at tiles with a top-2 margin "
                             f"< label_margin = {model.get_option('label_margin'):.2e}",
              "tiles_encoded_twice": int(rechecked), "tiles": n,
              "without_second_look": {"argmax_mismatches": int((~psame).sum().item()),
                                      "max_margin_of_a_mismatch": float(f"{(margin[~psame].max().item() if (~psame).any() else 0.0):.3e}")},
              "near_tie_tiles": int(near.sum().item()), "near_tie_argmax_match": int((same & near).sum().item()),
              "smallest_oracle_margin": float(f"{margin.min().item():.3e}"),
              "image_tower_only_max_abs_dcos": float(f"{d_img.max().item():.3e}"),
              "north_star_tolerance": 1e-4, "precision_mode": model.precision_name,
              "reference": "tests/golden/c3_dual_tower.npz: fp32 CPU oracle on BASELINE config 3 (4096 tiles x 64 prompts, both towers, "
                           "seed-0 weights); max / rms / match-rate cover BOTH the first-pass cosines and the returned ones"}
    exec_T = model.last_text_length
    bytes_sim = 4 * (768 * n + 768 * P + n * P)
    cfg = {"workload": "config 3: 4096 tiles (16 x 256, fp32 pixels) + 64 prompts x 256 tokens -> sim [4096,64] fp32 + argmax, 1 GPU",
           "seconds": round(t_all, 4), "tiles_per_s": round(n / t_cls, 1), "tiles_per_s_without_second_look": round(n / t_img, 1),
           "second_look_cost": round(t_cls / t_img - 1.0, 4),
           "prompts_per_s_padded_equivalent": round(P / t_txt, 1),
           "text_tflops_padded_equivalent": round(P * BERT_FLOPS_PER_PROMPT_256 / t_txt / 1e12, 1),
           "text_executed_length": int(exec_T), "text_ms": round(t_txt * 1e3, 3),
           "sim_argmax_us": round(t_sim * 1e6, 1), "sim_GBps": round(bytes_sim / t_sim / 1e9, 1), "sim_frac_of_hbm_peak": round(bytes_sim / t_sim / 1e9 / PEAK_HBM_GBS, 4)}
    return cfg, parity


def rcc_shaped_bank(txt, K=1782, C=4, seed=11):
    """K prompt sets x C classes drawn from few distinct strings, as the RCC prompt file: class c draws from its own
    share of the distinct text embeddings; a classifier column is a unit text embedding."""
    g = torch.Generator().manual_seed(seed)
    per_class = txt.shape[0] // C
    picks = torch.stack([torch.randint(0, per_class, (K,), generator=g) + c * per_class for c in range(C)], 1)      # [K, C]
    return [txt[p.to(txt.device)].t().contiguous() for p in picks]


def diff_stats(d):
    from keep_amd.model import max_sigmas_gumbel, max_sigmas_quantile
    mx, rms = float(d.abs().max()), float(d.double().pow(2).mean().sqrt())
    a, b = max_sigmas_gumbel(d.numel())
    return {"max_abs": float(f"{mx:.3e}"), "rms": float(f"{rms:.3e}"), "max_over_rms": round(mx / max(rms, 1e-30), 2), "n": int(d.numel()),
            "gaussian_max_over_rms": {"location": round(b, 2), "mean": round(b + 0.5772 / a, 2), "quantile_0.99": round(max_sigmas_quantile(d.numel(), 0.99), 2)},
            "over_1e-4": int((d.abs() > 1e-4).sum())}


C4_SEEDS = (1000, 2000, 3000)        # tile seeds of the config-4 leg: each is a different 100 000-tile synthetic slide of N(0,1) tiles (+ one of structured tiles)


def config4(model, dev, n: int = 100_000, distinct: int = 264, K: int = 1782, topn: int = 50, settings=None, seeds=C4_SEEDS, deadline=None):
    """BASELINE config 4 at its stated size on one GPU, on SEVERAL synthetic slides (`seeds`): n tiles (generated on the device, 256 per batch)
    encoded in the model's current 'comp' plan and in 'strict' (split products: pinned to 6.4e-7 of the fp32 oracle on config 3).  Per slide: every
    cosine difference against a 64-prompt bank (n x 64) and against the `distinct` prompt strings a K x 4 classifier bank of the RCC shape is built
    from (n x 264 distinct cosines = everything the n x 7128 screening logits can contain) -- text features of the comp side encoded in the
    benched mode, of the reference side in 'strict'.  On the first slide also the reference's subtyping / detection flow on both feature sets:
    screening scores, the selected ensemble, the slide label, the tumour ratio.  `settings`: extra plans to measure on the first slide the same way
    ((comp_full_blocks, comp_mlp_blocks) pairs or per-block plans; tools/c4_parity.py); the model's own plan is always measured and restored."""
    from keep_amd import wsi
    from keep_amd.model import plan_prefix, plan_string, prefix_plan
    from keep_amd.synth import synth_tiles_device

    def encode_all(seed):
        out = torch.empty(n, 768, device=dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for a in range(0, n, 512):             # 512 tiles per call: two lanes of 256 (+1.7 % over 256-tile calls, profiles/r05_step_boundary_bubble.txt)
            b = min(a + 512, n)
            out[a:b] = model.encode_image(synth_tiles_device(a, b, dev, torch.bfloat16, seed=seed))
        torch.cuda.synchronize(dev)
        return out, time.perf_counter() - t0

    def text_banks():
        t64 = model.encode_text({k: v.to(dev) for k, v in toks64.items()})
        tD = torch.cat([model.encode_text({k: v[i:i + 64].to(dev) for k, v in toksD.items()}) for i in range(0, distinct, 64)])
        return t64, tD

    own = model.get_plan()
    depth = len(own)
    prec_was, sb_was = model._options["precision"], int(model._options.get("strict_blocks", 0))
    toks64, toksD = synth_prompts(64, 256, seed=1), synth_prompts(distinct, 256, seed=5)
    out = {"workload": f"config 4 on one GPU: {len(seeds)} synthetic slides of {n} tiles (device-generated, bf16; tile seeds {list(seeds)}), 64-prompt bank, "
                       f"{distinct} distinct prompts -> K = {K} x C = 4 classifier bank ({K * 4} columns), prompt screening (topn {topn}), subtyping on a "
                       "256-px grid, detection (C = 2)",
           "reference": "the engine's 'strict' mode on the same tiles and prompts (split products; 6.4e-7 of the fp32 oracle on config 3: tests/golden/c3_dual_tower.npz)"}
    plans = [own]
    for st in settings or []:
        pl = prefix_plan(depth, int(st[0]), int(st[1])) if len(st) == 2 and not isinstance(st[0], (tuple, list)) else [tuple(x) for x in st]
        if pl != own:
            plans.append(pl)
    try:
        model.set_precision("comp", sb_was)
        model.set_plan(own)
        txt64_c, txtD_c = text_banks()                  # the comp side's text features: the benched mode (the text tower runs split products there too)
        model.set_precision("strict", sb_was)
        txt64, txtD = text_banks()
        per_seed, strict_rates, comp_rates = [], [], []
        for si, seed in enumerate(seeds):
            # the default run must end within minutes on any box: a slide takes ~50 s; the ones that would start past the budget are left out and named
            if si > 0 and deadline is not None and time.perf_counter() - T_START > deadline:
                out["slides_skipped"] = {"tile_seeds": list(seeds[si:]), "why": f"the run was {time.perf_counter() - T_START:.0f} s old (budget {deadline:.0f} s: --c4-budget-seconds)"}
                break
            model.set_precision("strict", sb_was)
            f_s, t_s = encode_all(seed)
            strict_rates.append(n / t_s)
            sim64_s, simD_s = model.similarity(f_s, txt64), model.similarity(f_s, txtD)
            if si == 0:
                bank4 = rcc_shaped_bank(txtD, K)
                bank2 = [c[:, :2].contiguous() for c in bank4]
                side = int(n ** 0.5) + 1
                idx = torch.arange(n)
                coords = torch.stack([(idx % side) * 256, (idx // side) * 256], 1).numpy()
                sc_s = wsi.prompt_scores(f_s, bank4, model=model)
                ens4_s = wsi.zero_shot_prompt_select(bank4, f_s, topn, dev, model=model)
                ens2_s = wsi.zero_shot_prompt_select(bank2, f_s, topn, dev, model=model)
                label_s = int(wsi.zero_shot_subtyping(ens4_s, f_s, coords, 256, True, model=model))
                p2_s = model.similarity(f_s, ens2_s.t().contiguous(), scale=10.0, mode="softmax")
                ratio_s = wsi.zero_shot_detection(ens2_s, f_s, coords, 256, False, model=model)
            for pi, plan in enumerate(plans if si == 0 else plans[:1]):
                model.set_precision("comp", sb_was)
                model.set_plan(plan)
                f, t = encode_all(seed)
                e = (f - f_s).norm(dim=1)
                pre = plan_prefix(plan)
                r = {"tile_seed": seed, "comp_full_blocks": pre[0] if pre else None, "comp_mlp_blocks": pre[1] if pre else None, "plan": plan_string(plan),
                     "tiles_per_s_incl_tile_generation": round(n / t, 1),
                     "feature_error_norm": {"max": float(f"{float(e.max()):.3e}"), "rms": float(f"{float(e.pow(2).mean().sqrt()):.3e}")},
                     "cos_vs_64_prompts": diff_stats(model.similarity(f, txt64_c) - sim64_s),
                     f"cos_vs_{distinct}_distinct_prompts": diff_stats(model.similarity(f, txtD_c) - simD_s)}
                if pi == 0:
                    comp_rates.append(n / t)
                    per_seed.append(r)
                if si > 0:
                    continue
                sc = wsi.prompt_scores(f, bank4, model=model)
                ens4 = wsi.zero_shot_prompt_select(bank4, f, topn, dev, model=model)
                ens2 = wsi.zero_shot_prompt_select(bank2, f, topn, dev, model=model)
                r["screening_scores"] = {"max_abs_diff": float(f"{float((sc - sc_s).abs().max()):.3e}"),
                                         "same_top_n": bool(set(torch.topk(sc, topn).indices.tolist()) == set(torch.topk(sc_s, topn).indices.tolist()))}
                r["ensemble_classifier_max_abs_diff"] = float(f"{float((ens4 - ens4_s).abs().max()):.3e}")
                r["slide_label"] = [int(wsi.zero_shot_subtyping(ens4, f, coords, 256, True, model=model)), label_s]
                r["slide_label_equal"] = r["slide_label"][0] == label_s
                ratio = wsi.zero_shot_detection(ens2, f, coords, 256, False, model=model)
                p2 = model.similarity(f, ens2_s.t().contiguous(), scale=10.0, mode="softmax")           # same classifier on both sides: the tile-level comparison
                flipped = (p2[:, 1] > 0.5) != (p2_s[:, 1] > 0.5)
                cs = model.similarity(f_s, ens2_s.t().contiguous())
                cosm = (cs[:, 1] - cs[:, 0]).abs()
                r["tumour_ratio"] = [ratio, ratio_s]
                r["tumour_tile_calls"] = {"tiles_whose_call_differs": int(flipped.sum()),
                                          "largest_strict_cos_margin_of_such_a_tile": float(f"{(float(cosm[flipped].max()) if bool(flipped.any()) else 0.0):.3e}"),
                                          "note": "a tile's tumour call is argmax over two cosines: it can only differ where the two are closer than twice the cosine tolerance "
                                                  "(KEEPModel.classify re-encodes such tiles when the labels themselves are the product)"}
                r["prob_map_max_abs_diff"] = float(f"{float((p2 - p2_s).abs().max()):.3e}")
                out["headline_setting" if pi == 0 else f"setting_{pi}_{plan_string(plan)}"] = r
        keyD = f"cos_vs_{distinct}_distinct_prompts"
        out["slides_run"] = len(per_seed)
        out["strict_tiles_per_s"] = round(sum(strict_rates) / len(strict_rates), 1)
        out["tiles_per_s_incl_tile_generation"] = round(sum(comp_rates) / len(comp_rates), 1)
        out["per_seed"] = [{"tile_seed": r["tile_seed"], "cos_vs_64_prompts": r["cos_vs_64_prompts"], keyD: r[keyD],
                            "feature_error_norm": r["feature_error_norm"]} for r in per_seed]
        out["worst_over_seeds"] = {"max_abs_dcos_64_prompts": max(r["cos_vs_64_prompts"]["max_abs"] for r in per_seed),
                                   f"max_abs_dcos_{distinct}_distinct_prompts": max(r[keyD]["max_abs"] for r in per_seed),
                                   "over_1e-4": sum(r["cos_vs_64_prompts"]["over_1e-4"] + r[keyD]["over_1e-4"] for r in per_seed),
                                   "cosines_compared": sum(r["cos_vs_64_prompts"]["n"] + r[keyD]["n"] for r in per_seed),
                                   "largest_max_over_rms": max(max(r["cos_vs_64_prompts"]["max_over_rms"], r[keyD]["max_over_rms"]) for r in per_seed)}
        out["max_abs_dcos_vs_strict"] = out["worst_over_seeds"]["max_abs_dcos_64_prompts"]
        out[f"max_abs_dcos_vs_strict_{distinct}_distinct_prompts"] = out["worst_over_seeds"][f"max_abs_dcos_{distinct}_distinct_prompts"]
        out["within_1e-4"] = bool(out["worst_over_seeds"]["over_1e-4"] == 0)
    finally:
        model.set_precision({0: "fp16", 1: "strict", 2: "comp"}[int(prec_was)], sb_was)
        model.set_plan(own)
    return out


def structured_slide_parity(model, dev, family: str = "mixed", n: int = 100_000, seed: int = 7000, distinct: int = 264, banks=None):
    """The 1e-4 tolerance OFF the i.i.d. N(0,1) pixels of config 2: a synthetic slide of `n` uint8 tiles of one of keep_amd.synth.TILE_FAMILIES (real-image
    crops, Beer-Lambert stain fields, glass background, half / half; "mixed" = a quarter each, interleaved) through `encode_image_uint8` -- ToTensor +
    Normalize fused into the first kernel, as the reference transform feeds the model (keep_inference.py:88-93) -- in the model's current 'comp' plan
    and in 'strict'; every cosine against a 64-prompt bank and `distinct` prompts, and what the slide's own per-tile errors predict for the
    calibration population (keep_amd.model.mixture_exceedance).  `banks`: (txt64_comp, txtD_comp, txt64_strict, txtD_strict) to reuse."""
    from keep_amd.model import CALIBRATION_POPULATION, mixture_exceedance
    from keep_amd.synth import synth_tile_family
    prec_was, sb_was = model._options["precision"], int(model._options.get("strict_blocks", 0))
    own = model.get_plan()

    def encode_all():
        out = torch.empty(n, 768, device=dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for a in range(0, n, 512):
            b = min(a + 512, n)
            out[a:b] = model.encode_image_uint8(synth_tile_family(family, a, b, dev, seed=seed))
        torch.cuda.synchronize(dev)
        return out, time.perf_counter() - t0

    def text_banks():
        toks64, toksD = synth_prompts(64, 256, seed=1), synth_prompts(distinct, 256, seed=5)
        t64 = model.encode_text({k: v.to(dev) for k, v in toks64.items()})
        tD = torch.cat([model.encode_text({k: v[i:i + 64].to(dev) for k, v in toksD.items()}) for i in range(0, distinct, 64)])
        return t64, tD

    try:
        model.set_precision("comp", sb_was)
        model.set_plan(own)
        t64_c, tD_c = banks[:2] if banks else text_banks()
        f, t_c = encode_all()
        model.set_precision("strict", sb_was)
        t64_s, tD_s = banks[2:] if banks else text_banks()
        f_s, t_s = encode_all()
        e = f - f_s
        sig = e.double().pow(2).sum(1).div(e.shape[1]).sqrt()
        st64, stD = diff_stats(model.similarity(f, t64_c) - model.similarity(f_s, t64_s)), diff_stats(model.similarity(f, tD_c) - model.similarity(f_s, tD_s))
        return {"workload": f"{n} uint8 tiles of the '{family}' family (keep_amd.synth.synth_tile_family, seed {seed}) through encode_image_uint8, the calibrated "
                            f"'comp' plan against 'strict'; cosines against 64 and {distinct} prompts",
                "family": family, "tiles": n, "plan": plan_string(own), "tiles_per_s_incl_tile_generation": round(n / t_c, 1), "strict_tiles_per_s": round(n / t_s, 1),
                "cos_vs_64_prompts": st64, f"cos_vs_{distinct}_distinct_prompts": stD, "over_1e-4": st64["over_1e-4"] + stD["over_1e-4"],
                "max_abs_dcos": max(st64["max_abs"], stD["max_abs"]),
                "isotropic_rms": float(f"{float(sig.pow(2).mean().sqrt()):.3e}"), "hardest_tile_over_rms": round(float(sig.max() / sig.pow(2).mean().sqrt()), 2),
                "population_exceedance_from_this_slide": float(f"{mixture_exceedance(sig.tolist(), CALIBRATION_POPULATION, 1e-4):.3e}"),
                "mean_pairwise_feature_cos_first_512": round(float((f_s[:512] @ f_s[:512].t()).mean()), 4)}
    finally:
        model.set_precision({0: "fp16", 1: "strict", 2: "comp"}[int(prec_was)], sb_was)
        model.set_plan(own)


def config5(model, dev, n: int = 100_000):
    """BASELINE config 5: per-tile dense similarity map, fp16: softmax(10 * cos) over 2 classes for a 100 000-tile slide."""
    from oracle import keep_oracle as O
    g = torch.Generator().manual_seed(55)
    feats = torch.nn.functional.normalize(torch.randn(n, 768, generator=g), dim=-1)
    cls = torch.nn.functional.normalize(torch.randn(2, 768, generator=g), dim=-1)
    fd, cd = feats.to(dev), cls.to(dev)
    out = model.similarity(fd, cd, scale=10.0, mode="softmax_f16")
    t = time_gpu(lambda: model.similarity(fd, cd, scale=10.0, mode="softmax_f16"), dev, 50)
    ref = O.sim_softmax(O.similarity(feats, cls), 10.0)                 # the whole map: 100 000 x 2 x 768 is a 0.3 GFLOP matmul on the host
    err = (out.float().cpu() - ref).abs().max().item()
    half_ulp = 2.0 ** -12                                                # probabilities in [0.5, 1) are stored to 2^-11: half an ulp there, + the fp32 path's 1e-6
    nbytes = n * 768 * 4 + n * 2 * 2                      # 768*s read + C*s written per tile (features are fp32 here)
    return {"workload": f"config 5: {n} tiles x 2 classes, fp16 probability map softmax(10 cos), 1 GPU", "us": round(t * 1e6, 1),
            "tiles_per_s": round(n / t, 0), "GBps": round(nbytes / t / 1e9, 1), "frac_of_hbm_peak": round(nbytes / t / 1e9 / PEAK_HBM_GBS, 4),
            "max_abs_err_vs_oracle": float(f"{err:.2e}"), "tiles_checked_against_oracle": n, "fp16_half_ulp_at_0.5_plus_1e-6": half_ulp + 1e-6,
            "within_fp16_rounding_of_oracle": bool(err <= half_ulp + 1e-6)}


def slide_leg(model, dev, n: int, rank: int, world: int, distinct: int = 264, K: int = 1782, topn: int = 50):
    """BASELINE configs 4 and 5 in their stated MULTI-GPU layout: the synthetic slide's `n` tiles in contiguous shards of ceil(n / world)
    per rank (12 500 at n = 100 000, world = 8), encoded in batches of 512 with the RCCL all-gather of batch j's [512, 768] embeddings issued under
    the encode of batch j + 1 (keep_amd.distributed.encode_tiles_sharded); then ON EVERY RANK, from the gathered [n, 768] matrix: prompt screening
    over a K x 4 classifier bank, the ensemble, the slide label (subtyping on a 256-px grid) and the tumour ratio; and config 5's fp16 probability
    map of the rank's own shard.  Every rank must end with the same embedding matrix (checksum), the same label and the same ratio."""
    import torch.distributed as dist
    from keep_amd import wsi
    from keep_amd.distributed import encode_tiles_sharded, shard_bounds
    from keep_amd.synth import synth_tiles_device
    toksD = synth_prompts(distinct, 256, seed=5)
    txtD = torch.cat([model.encode_text({k: v[i:i + 64].to(dev) for k, v in toksD.items()}) for i in range(0, distinct, 64)])
    bank4 = rcc_shaped_bank(txtD, K)
    bank2 = [c[:, :2].contiguous() for c in bank4]
    load = lambda a, b: synth_tiles_device(a, b, dev, torch.bfloat16, seed=C4_SEEDS[0])
    fence = lambda: (dist.barrier() if dist.is_initialized() else None, torch.cuda.synchronize(dev))
    encode_tiles_sharded(model.encode_image, min(n, 1024 * world), load, batch=512)         # warm-up (workspace, communicator)
    fence()
    t0 = time.perf_counter()
    feats = encode_tiles_sharded(model.encode_image, n, load, batch=512)                    # 512 tiles per call = two lanes of 256; the all-gather of call j under call j + 1
    fence()
    t_enc = time.perf_counter() - t0
    side = int(n ** 0.5) + 1
    idx = torch.arange(n)
    coords = torch.stack([(idx % side) * 256, (idx // side) * 256], 1).numpy()
    t1 = time.perf_counter()
    ens4 = wsi.zero_shot_prompt_select(bank4, feats, topn, dev, model=model)
    label = int(wsi.zero_shot_subtyping(ens4, feats, coords, 256, True, model=model))
    ens2 = wsi.zero_shot_prompt_select(bank2, feats, topn, dev, model=model)
    ratio = float(wsi.zero_shot_detection(ens2, feats, coords, 256, False, model=model))
    lo, hi = shard_bounds(n, rank, world)
    pmap = model.similarity(feats[lo:hi], ens2.t().contiguous(), scale=10.0, mode="softmax_f16")      # config 5: this rank's share of the dense map
    torch.cuda.synchronize(dev)
    t_slide = time.perf_counter() - t1
    row = torch.tensor([float(feats.double().sum()), float(feats.double().pow(2).sum()), float(label), ratio, float(pmap.float().sum()), t_enc, t_slide],
                       dtype=torch.float64, device=dev)
    rows = [torch.empty_like(row) for _ in range(world)]
    if dist.is_initialized():
        dist.all_gather(rows, row)
    else:
        rows = [row]
    got = [r.tolist() for r in rows]
    same = all(g[:4] == got[0][:4] for g in got)
    return {"workload": f"configs 4 + 5 in the multi-GPU layout: {n} synthetic tiles in contiguous shards of <= {-(-n // world)} per rank x {world} rank(s), RCCL "
                        f"all-gather of every 512-tile batch's embeddings under the next batch's encode, then on every rank: screening over {K} x 4 prompts, "
                        "ensemble, slide label, tumour ratio; fp16 probability map of the rank's shard",
            "tiles": n, "ranks": world, "tiles_per_rank": hi - lo, "encode_and_gather_seconds_max_over_ranks": round(max(g[5] for g in got), 3),
            "tiles_per_s_whole_slide": round(n / max(g[5] for g in got), 1), "slide_level_seconds_max_over_ranks": round(max(g[6] for g in got), 3),
            "slide_label": label, "tumour_ratio": round(ratio, 5), "prob_map_rows_this_rank": int(pmap.shape[0]),
            "every_rank_same_embeddings_label_ratio": bool(same), "embedding_checksums": [g[0] for g in got]}


def spawn_ranks(n: int) -> int:
    """Re-run this command under `torch.distributed.run` with one rank per GPU of this node (loopback rendezvous on a free port) and return its
    exit code; rank 0 of the children prints the JSON line on this process's stdout.  Fails loudly when the node has fewer GPUs than asked for."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible on this node: not launching (a line with n_gpus != the ranks that ran is never printed)",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.pop("KEEP_BENCH_FORCE_SPAWN", None)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    if n == 1:
        env["KEEP_BENCH_FORCE_DIST"] = "1"           # the forced one-rank spawn (tests) goes through the RCCL path too
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    log(f"launching {n} rank(s): {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="tiles per GPU per step")
    ap.add_argument("--precision", default="comp", choices=["comp", "fp16", "strict"])
    ap.add_argument("--pixel-dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (e.g. comp_mlp_blocks=8)")
    ap.add_argument("--sustain-seconds", type=float, default=10.0)
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the config-3 parity / config-3 / config-4 / config-5 legs")
    ap.add_argument("--no-c4", action="store_true", help="skip the 100 000-tile config-4 leg (about a minute)")
    ap.add_argument("--no-c4-structured", action="store_true", help="skip the 100 000-tile structured-tile slide of the config-4 leg (about a minute)")
    ap.add_argument("--c4-tiles", type=int, default=100_000)
    ap.add_argument("--c4-budget-seconds", type=float, default=150.0, help="no further config-4 slide is started once the run is this old (the first one always runs)")
    ap.add_argument("--c4-seeds", type=int, default=len(C4_SEEDS), help="number of 100 000-tile synthetic slides of the config-4 leg (about 50 s each)")
    ap.add_argument("--slide", action="store_true", help="also run configs 4 + 5 in their multi-GPU layout (tiles sharded over the ranks, RCCL all-gather of the embeddings, "
                                                         "slide label / tumour ratio on every rank): the `slide` key of the line")
    ap.add_argument("--slide-tiles", type=int, default=100_000)
    ap.add_argument("--budget", default=None, choices=["ladder", "measured"], help="re-run KEEPModel.calibrate with this budget after loading")
    ap.add_argument("--plan", default=None, help="run this per-block plan instead of the calibrated one: the 'attn:<digits> mlp:<digits>' string a bench line "
                                                 "reports (profiling runs: KEEP_CALIBRATE=0 keeps the calibration's kernels out of the trace)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: keep_amd has no CPU execution path")
    launched = "WORLD_SIZE" in os.environ            # under torch.distributed.run (the documented N > 1 launch) the ranks already exist
    if not launched and (args.gpus > 1 or os.environ.get("KEEP_BENCH_FORCE_SPAWN") == "1"):
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU over RCCL), never report a line whose n_gpus
        # differs from the ranks that ran
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line's n_gpus must be the number of ranks that ran")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("KEEP_BENCH_FORCE_DIST") == "1"     # the override exercises the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    log(f"start: world={world} cpus={usable_cpus()} (os.cpu_count={os.cpu_count()})")
    torch.set_num_threads(min(usable_cpus(), 16))
    shape = KEEPShape()
    want_configs = rank == 0 and world == 1 and not args.no_configs      # the config-3 parity / c3 / c5 legs belong to the 1-GPU line; scaling runs stay lean
    # identical weights on every rank, BOTH towers everywhere: the load-time calibration measures its cosines against prompts through the loaded text
    # tower (an image-only engine falls back to random unit vectors, a harsher yardstick that picks a slower setting) -- every rank of every world
    # size must run the same setting for the scaling figures to compare like with like
    sd = synth_state_dict(shape, seed=0, text=True)
    model = KEEPModel(shape, precision=args.precision, towers=("image", "text"))
    model.precision_name = args.precision
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    if args.budget is not None and args.precision == "comp":
        model.calibrate(budget=args.budget)
    if args.plan is not None:
        a_digits, m_digits = (part.split(":")[1] for part in args.plan.split())
        model.set_plan([(int(a), int(m)) for a, m in zip(a_digits, m_digits)])
    for kv in args.opt:
        k, v = kv.split("=")
        model.set_option(k, float(v))
    model.reserve(tiles=args.batch)
    log("weights uploaded, workspace reserved")

    B = args.batch
    pix = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.pixel_dtype]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)        # per-rank tiles, generated on device
    tiles = torch.randn(B, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(pix)
    # keep_amd.distributed.StepExchange: double-buffered, so the RCCL all-gather of step i overlaps the encode of step i+1 (the same
    # class runs under gloo at world sizes 2 and 3 in tests/test_distributed_cpu.py)
    exchange = StepExchange(B, shape.projection_dim, dev) if use_dist else None

    def step():
        f = model.encode_image(tiles)
        if use_dist:
            exchange.submit(f)
        return f

    def timed(n_steps, own=None):
        if use_dist:
            return timed_steps(step, n_steps, exchange, own)     # fence (collectives + barrier + device sync), n steps, fence, MAX over ranks
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if own is not None:
            own.append(el)
        return el

    rccl_ranks_seen = None
    if use_dist:
        # proof for the driver's SCALE record that RCCL really spans --gpus ranks: an all-reduce of ones over the nccl group
        ones = torch.ones(1, device=dev, dtype=torch.float32)
        dist.all_reduce(ones)
        rccl_ranks_seen = int(round(float(ones.item())))
        if rccl_ranks_seen != world:
            raise SystemExit(f"RCCL all-reduce saw {rccl_ranks_seen} ranks, expected {world}")
        # every rank calibrates on its own at load, and the rule's verdict can sit within a rounding of its threshold: rank 0's plan (verified there, on the same
        # weights and probe) runs on every rank -- a scaling figure must not mix settings
        from keep_amd.distributed import adopt_rank0_plan, assert_same_setting
        adopt_rank0_plan(model, device=dev)
        assert_same_setting([model.get_option("precision")] + [float(v) for am in model.get_plan() for v in am],
                            "precision setting (precision, per-block plan)", device=dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    log("warm-up done")
    model.profile_enable(",".join(DOMINANT_TAGS))
    model.profile_reset()
    own_elapsed = []
    elapsed = timed(args.steps, own_elapsed)                          # THE timed region of the contract: exactly --steps steps
    in_region = {t: model.profile_read(t) for t in DOMINANT_TAGS}
    model.profile_disable()
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    per_rank = exchange_cost = None
    if use_dist:
        t = torch.tensor([own_elapsed[0]], device=dev, dtype=torch.float64)
        rows = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(rows, t)
        per_rank = [round(B * args.steps / float(r.item()), 1) for r in rows]
        # what the exchange costs on the critical path: the same steps without it, same fences
        def bare():
            model.encode_image(tiles)
        el0 = timed_steps(bare, args.steps, exchange)
        exchange_cost = {"ms_per_step_with_exchange": round(elapsed / args.steps * 1e3, 3), "ms_per_step_encode_only": round(el0 / args.steps * 1e3, 3),
                         "exposed_ms_per_step": round((elapsed - el0) / args.steps * 1e3, 3),
                         "bytes_gathered_per_step": world * B * shape.projection_dim * 4}

    slide = None
    if args.slide:
        log(f"slide leg: {args.slide_tiles} tiles over {world} rank(s) ...")
        slide = slide_leg(model, dev, args.slide_tiles, rank, world)
        if not slide["every_rank_same_embeddings_label_ratio"]:
            raise SystemExit(f"slide leg: the ranks disagree on the gathered embeddings / label / ratio: {slide}")
        model.reserve(tiles=args.batch)

    # effective shader clock under this load: a one-wave probe (shader cycles against the 100 MHz reference counter) on a side stream, right after
    # the timed region (the queue is still full of encodes: the probe lands between them) and again during the sustained region
    side = torch.cuda.Stream(device=dev)
    clock_buf = torch.zeros(16, 2, dtype=torch.int64, device=dev)     # allocated and settled before the probes' neighbours are queued
    torch.cuda.synchronize(dev)
    clock_samples = []

    def sample_clock():
        if len(clock_samples) < clock_buf.shape[0]:
            clock_samples.append(clock_buf[len(clock_samples)])
            model.clock_probe(clock_samples[-1], spin_us=300, stream=side)

    for _ in range(3):
        step()
        sample_clock()
    sustained = None
    if not args.no_sustained:
        n_sus = max(args.steps, int(args.sustain_seconds / (elapsed / args.steps)) + 1)
        el = timed(n_sus)
        sustained = {"value": round(world * B * n_sus / el, 2), "unit": "tiles/s", "steps": n_sus, "seconds": round(el, 2),
                     "ms_per_step": round(el / n_sus * 1e3, 3),
                     "note": "same step, same barriers, timed for >= 10 s: the part sits at its power cap under this load, so this is the rate to plan with"}
        for _ in range(8):                                          # (the timed regions themselves stay untouched: samples are taken around them)
            step()
            sample_clock()
        log(f"sustained region done: {n_sus} steps in {el:.1f} s")
    torch.cuda.synchronize(dev)
    mhz = sorted(100.0 * float(t[0]) / float(t[1]) for t in clock_buf[:len(clock_samples)].cpu() if float(t[1]) > 0)
    clock = {"effective_shader_MHz_median": round(mhz[len(mhz) // 2], 0), "min": round(mhz[0], 0), "max": round(mhz[-1], 0), "samples": len(mhz),
             "how": "keep_clock_probe: one wavefront counting shader cycles against the 100 MHz reference for 300 us on a side stream while encode steps run",
             "peak_quoted_at_MHz": 2400} if mhz else None

    # untimed pass on ONE internal stream: per-kernel times with nothing else on the GPU (lanes do not overlap): the breakdown, and the KERNEL figure
    # of the roofline block (HIP events around every launch, on the stream it is launched on)
    breakdown, single = None, {}
    if not args.no_breakdown and rank == 0:
        streams_was = model._options.get("streams", 2)
        model.set_option("streams", 1)
        model.profile_enable(None)
        model.profile_reset()
        n_pass = 3
        for _ in range(n_pass):
            model.encode_image(tiles)
        torch.cuda.synchronize(dev)
        breakdown = {}
        for tag in PROFILE_TAGS:
            ms, n, fl = model.profile_read(tag)
            if n:
                breakdown[tag] = round(ms / n_pass, 3)
                single[tag] = (ms, n, fl)
        model.profile_disable()
        model.set_option("streams", streams_was)

    # context for the roofline fractions: what the matrix pipes ALONE sustain on this box at the socket's power cap, no memory traffic (keep_mfma_probe)
    pipe_ceiling = None
    if rank == 0 and not args.no_breakdown:
        rnd = model.mfma_ceiling()
        zer = model.mfma_ceiling(torch.zeros(1024, dtype=torch.float16, device=dev))
        pipe_ceiling = {"random_N01_fp16_operands_TFLOPs": round(rnd, 1), "frac_of_peak": round(rnd / PEAK_F16_TFLOPS, 4),
                        "all_zero_operands_TFLOPs": round(zer, 1),
                        "note": "v_mfma_f32_32x32x16_f16 back to back on every SIMD, operands in registers, no memory access: with high-entropy operands the power cap "
                                "holds the pipes at ~1.6 GHz, so ~0.63 of the nominal peak is what the silicon offers fp16 GEMM work on real data; the fractions above stay "
                                "quoted against the nominal peak",
                        "end_to_end_frac_of_this_ceiling": round(tiles_per_s_for_ceiling(world, B, args.steps, elapsed) * vit_flops_per_tile() / (rnd * 1e12), 4) if rnd > 0 else None}
    c3 = c4 = c4s = c5 = parity = None
    if want_configs:
        log("config 3 (4096 tiles x 64 prompts, parity vs the oracle fixture) ...")
        c3, parity = config3(model, dev)
        c5 = config5(model, dev)
        if parity is not None and not parity["labels_bit_exact"]:
            log(f"WARNING: {parity['argmax_mismatches']} config-3 labels differ from the fp32 oracle's -- the north star asks for bit-exact labels")
        if not args.no_c4 and args.precision == "comp":
            log(f"config 4 ({args.c4_tiles} tiles: the benched setting against the split-product mode) ...")
            c4 = config4(model, dev, n=args.c4_tiles, seeds=C4_SEEDS[:max(1, args.c4_seeds)], deadline=args.c4_budget_seconds)
            if not c4["within_1e-4"]:
                log("WARNING: config 4 holds cosines that differ from the split-product mode by more than 1e-4 in the benched setting")
            if not args.no_c4_structured:
                log(f"config 4, structured tiles ({args.c4_tiles} uint8 tiles: real-image crops, stain fields, glass, half / half) ...")
                c4s = structured_slide_parity(model, dev, "mixed", n=args.c4_tiles)
                if c4s["over_1e-4"]:
                    log("WARNING: the structured slide holds cosines that differ from the split-product mode by more than 1e-4")
        log("configs done")

    line = None
    if rank == 0:
        tiles_per_s = world * B * args.steps / elapsed
        frac_e2e = tiles_per_s / world * vit_flops_per_tile() / (PEAK_F16_TFLOPS * 1e12)

        def rate(ms, n, fl):
            return {"achieved": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0.0, "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4) if ms > 0 else 0.0,
                    "avg_launch_ms": round(ms / max(n, 1), 4), "launches": int(n), "flops_per_launch": round(fl / max(n, 1), 1)}

        # HBM-side traffic of the dominant kernel: PMC counters need rocprofv3 around the process (separate FETCH_SIZE / WRITE_SIZE passes, the guide's
        # gfx950 correction), so it is read from the committed summary of tools/refresh_profiles.sh and labelled with the library build it was taken on
        traffic, traffic_src, traffic_tiles = None, None, None
        for rnd in ("r06", "r05", "r04", "r03"):
            tpath = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic.json")
            if os.path.exists(tpath):
                try:
                    tjs = json.load(open(tpath))
                    tj = tjs.get("vit.fc2", tjs.get("vit.proj+fc2", {}))
                    traffic = tj.get("bytes_per_launch")
                    traffic_tiles = tj.get("tiles_per_launch", 128)      # the PMC passes run the two-lane bench: 128-tile launches
                    sha_path = os.path.join(ROOT, "profiles", f"{rnd}_lib_sha16.txt")
                    sha = open(sha_path).read().strip() if os.path.exists(sha_path) else "unrecorded"
                    traffic_src = (f"profiles/{rnd}_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch of this kernel "
                                   f"({traffic_tiles}-tile launches; algorithmic: {tj.get('algorithmic_bytes_per_launch')}), measured on library build {sha}; "
                                   f"this run's library is {lib_sha16()}")
                    break
                except (OSError, ValueError):
                    traffic = None
        dom = [single.get(t, (0.0, 0, 0.0)) for t in DOMINANT_TAGS]
        dom_ms, dom_n, dom_fl = (sum(d[i] for d in dom) for i in range(3))
        roofline = {"bound": "mfma", "kernel": DOMINANT_KERNEL, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", **rate(dom_ms, dom_n, dom_fl),
                    "traffic": traffic, "traffic_tiles_per_launch": traffic_tiles, "flops_tiles_per_launch": B, "traffic_source": traffic_src,
                    "timing": "single_stream_pass: HIP events around every launch of this kernel, on the stream it is launched on, 3 encode calls of 256 tiles with one "
                              "internal stream (nothing else on the GPU): a kernel figure, comparable with profiles/r06_rocprofv3_kernel_stats_single_stream.csv",
                    "share_of_single_stream_step": round(dom_ms / max(sum(v[0] for v in single.values()), 1e-9), 4) if single else None,
                    "by_operator": {t: rate(*single[t]) for t in DOMINANT_TAGS if t in single},
                    "in_timed_region_two_lane": {**rate(*(sum(in_region[t][i] for t in DOMINANT_TAGS) for i in range(3))),
                                                 "note": "the same kernel's launches INSIDE the timed region, where two 128-tile lanes share the GPU: a launch's "
                                                         "duration includes the time it spends beside the other lane's kernels -- not a kernel figure"},
                    "other_kernels_single_stream": {t: rate(*single[t]) for t in ("vit.qkv", "vit.fc1", "vit.proj", "vit.fc1.x", "vit.fc2.x", "vit.qkv.x", "vit.proj.x", "vit.patch")
                                                    if t in single},
                    "frac_end_to_end": round(frac_e2e, 4),
                    "note": "achieved / frac = ALGORITHMIC FLOPs (2*M*N*K per launch; correction passes are never counted) / summed launch durations; "
                            "frac_end_to_end = tiles/s x 123.11 GFLOP / peak over the whole encoder (the figure the driver can check against its own clock); "
                            "peak = 256 CU x 4096 FLOP/clk x 2.4 GHz dense fp16"}
        if pipe_ceiling is not None:
            roofline["matrix_pipe_ceiling_measured"] = pipe_ceiling
        if clock is not None:
            roofline["clock"] = clock
            roofline["frac_of_peak_at_effective_clock"] = round(frac_e2e * 2400.0 / clock["effective_shader_MHz_median"], 4)
        line = {
            "metric": "224x224 tiles encoded/sec (whole node)", "value": round(tiles_per_s, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NAME[args.precision],
            "data": "synthetic (randn tiles generated on device, seeded random-init weights)",
            "config": {"workload": "ViT-L/16 image encoder only (KEEP encode_image), batch 256 synthetic 224x224 tiles per GPU "
                                   f"({args.pixel_dtype} pixels; GEMM operands fp16 with fp32 accumulation, + MX-fp4 correction passes in 'comp'), 1xMI355X per rank",
                       "tiles_per_gpu_per_step": B, "precision": args.precision,
                       "exchange": "RCCL all_gather of [256,768] fp32 embeddings per step" if use_dist else "none",
                       "comp_settings": {"comp_full_blocks": int(model.get_option("comp_full_blocks")), "comp_mlp_blocks": int(model.get_option("comp_mlp_blocks")),
                                         "plan": plan_string(model.get_plan()), "plan_is_prefix": plan_prefix(model.get_plan()) is not None,
                                         "label_margin": model.get_option("label_margin"),
                                         "bias_correction": bool(model.get_option("bias_ready") and model.get_option("bias_correction")),
                                         "chosen_by": "KEEPModel.calibrate() at load_state_dict" if model.calibration else "the plan a handle starts with"},
                       "mfma_frac_end_to_end": round(frac_e2e, 4)},
            "roofline": roofline,
        }
        if model.calibration is not None:
            line["calibration"] = model.calibration
        if sustained is not None:
            line["sustained"] = sustained
        if parity is not None:
            line["parity"] = parity
        if c3 is not None or c4 is not None or c5 is not None:
            line["configs"] = {"c3": c3, "c4": c4, "c4_structured": c4s, "c5": c5}
        if rccl_ranks_seen is not None:
            line["rccl_ranks_seen"] = rccl_ranks_seen      # sum of an all-reduced ones tensor over the nccl (= RCCL) group: must equal n_gpus
        if per_rank is not None:
            line["per_rank_tiles_per_s"] = per_rank
            line["exchange"] = exchange_cost
        if slide is not None:
            line["slide"] = slide
        if breakdown is not None:
            line["breakdown_ms_per_step_single_stream"] = breakdown
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle on host cores) ...")
            line["cpu_baseline"] = cpu_baseline(sd)
            log("cpu baseline done")
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which a pipe only sees at exit: push it out first so that the JSON
        # line is the LAST line on stdout
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
