#!/usr/bin/env python3
"""The 1e-4 cosine tolerance at BASELINE config-4 / config-5 size for SEVERAL 'comp' settings (bench.py's `configs.c4` leg measures the one the
headline ran): a 100 000-tile synthetic slide encoded in each setting and in 'strict', every cosine against a 64-prompt bank and against the 264
distinct prompts of an RCC-shaped classifier bank, the screening scores, the ensemble, the slide label and the tumour ratio (bench.config4).

    python tools/c4_parity.py [--tiles 100000] [--settings 1,6 1,8 1,10] [--out gpurun_out/c4_parity.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                              # noqa: E402
from keep_amd import KEEPModel                                            # noqa: E402
from keep_amd.config import KEEPShape                                     # noqa: E402
from keep_amd.synth import synth_state_dict                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=100_000)
    ap.add_argument("--settings", nargs="*", default=["1,8", "1,10"], help="extra prefix plans (comp_full_blocks,comp_mlp_blocks) measured on the first slide")
    ap.add_argument("--seeds", type=int, default=5, help="number of synthetic slides (tile seeds 1000, 2000, ...)")
    ap.add_argument("--family", default="default", help="weight family of keep_amd.synth (default | heavy_tail | small_ls)")
    ap.add_argument("--weight-seed", type=int, default=0, help="seed of the synthetic weights (bench.py runs seed 0)")
    ap.add_argument("--budget", default="ladder", choices=["ladder", "measured"])
    ap.add_argument("--out", default="gpurun_out/c4_parity.json")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = KEEPModel(KEEPShape())
    model.load_state_dict(synth_state_dict(KEEPShape(), seed=args.weight_seed, family=args.family))
    model.to(dev).eval()                                                  # calibrates: `headline_setting` below is what load_state_dict picked
    if args.budget != "ladder":
        model.calibrate(budget=args.budget)
    model.reserve(tiles=256)
    res = bench.config4(model, dev, n=args.tiles, settings=[tuple(int(v) for v in st.split(",")) for st in args.settings],
                        seeds=tuple(1000 * (i + 1) for i in range(args.seeds)))
    res["weight_family"] = args.family
    res["weight_seed"] = args.weight_seed
    res["calibration"] = model.calibration
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
