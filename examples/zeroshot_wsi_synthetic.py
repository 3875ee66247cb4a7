#!/usr/bin/env python3
"""Synthetic-slide run of the zero-shot WSI flow (BASELINE.json configs 4 and 5) on 1..8 GPUs.

    python examples/zeroshot_wsi_synthetic.py --tiles 4096 --prompt-sets 64
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/zeroshot_wsi_synthetic.py --tiles 100000

Mirrors WSI_evaluation/zeroshot_subtyping_WSI.py / zeroshot_detection_WSI.py / zeroshot_segmentation_WSI.py
with the offline CLAM feature extraction folded in: tiles -> encode_image (sharded over the ranks) ->
RCCL all-gather of the embeddings -> prompt screening -> refine -> slide label / tumour ratio / probability map.
No dataset, weights or tokenizer exist offline, so tiles, weights and token ids are seeded synthetic data.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keep_amd import KEEPModel, wsi                                   # noqa: E402
from keep_amd.config import KEEPShape, small_shape                    # noqa: E402
from keep_amd.distributed import adopt_rank0_plan, encode_tiles_sharded    # noqa: E402
from keep_amd.synth import synth_prompts, synth_state_dict, synth_tiles_device   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=4096)
    ap.add_argument("--prompt-sets", type=int, default=64)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--distinct-texts", type=int, default=24)
    ap.add_argument("--topn", type=int, default=50)
    ap.add_argument("--depth", type=int, default=24, help="ViT/BERT depth (24/12 = the real model; smaller for a quick look)")
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("KEEP_EXAMPLE_FORCE_DIST") == "1"       # the override pushes a 1-GPU run through RCCL as well
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    shape = KEEPShape() if args.depth >= 24 else small_shape(args.depth, max(1, args.depth // 2))
    model = KEEPModel(shape)
    model.load_state_dict(synth_state_dict(shape, seed=0))
    model.to(dev).eval()
    adopt_rank0_plan(model, device=dev)       # every rank calibrated on its own at load: the job runs rank 0's plan (a no-op without a process group)

    # tiles are generated on the device in units of 256 (a tile's pixels depend only on its global index, so every
    # world size sees the same slide); one or two randn launches per batch, no per-tile Python work
    def load_tiles(a, b):
        return synth_tiles_device(a, b, dev, torch.bfloat16, seed=1000)

    model.reserve(tiles=256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feats = encode_tiles_sharded(model.encode_image, args.tiles, load_tiles, batch=512)      # 512 tiles per call: two lanes of 256
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t0

    # prompt bank shaped like the RCC one: K prompt sets x C classes drawn from few distinct strings
    toks = synth_prompts(args.distinct_texts, 256, seed=5)
    txt = model.encode_text({k: v.to(dev) for k, v in toks.items()})
    gsel = torch.Generator().manual_seed(11)
    classifiers = []
    for _ in range(args.prompt_sets):
        pick = torch.randperm(args.distinct_texts, generator=gsel)[: args.classes]
        classifiers.append(wsi_build(txt[pick.to(dev)]))
    grid = int(args.tiles ** 0.5) + 1
    idx = torch.arange(args.tiles)
    coords256 = torch.stack([(idx % grid) * 256, (idx // grid) * 256], 1).numpy()
    coords224 = torch.stack([(idx % grid) * 224, (idx // grid) * 224], 1).numpy()

    t0 = time.perf_counter()
    ens = wsi.zero_shot_prompt_select(classifiers, feats, min(args.topn, args.prompt_sets), dev)
    label = int(wsi.zero_shot_subtyping(ens, feats, coords256, 256, True))
    ens2 = wsi.zero_shot_prompt_select([c[:, :2].contiguous() for c in classifiers], feats, min(args.topn, args.prompt_sets), dev)
    ratio = wsi.zero_shot_detection(ens2, feats, coords256, 256, False)
    prob16 = model.similarity(torch.nn.functional.normalize(feats), ens2.t().contiguous(), scale=10.0, mode="softmax_f16")
    seg = wsi.zero_shot_segment(ens2, feats, coords224, None, 224, True)
    torch.cuda.synchronize()
    t_slide = time.perf_counter() - t0
    if rank == 0:
        print(f"{args.tiles} tiles on {world} GPU(s){' (RCCL process group)' if use_dist else ''}: encode {t_enc:.3f}s ({args.tiles / t_enc:.0f} tiles/s incl. tile generation), "
              f"slide-level steps {t_slide * 1e3:.1f} ms; label={label} tumour_ratio={ratio:.4f} "
              f"seg_map={len(seg)} tiles, fp16 prob map {tuple(prob16.shape)}")
    if use_dist:
        dist.destroy_process_group()


def wsi_build(class_embeddings):
    e = torch.nn.functional.normalize(class_embeddings, dim=-1)
    return (e / e.norm(dim=-1, keepdim=True)).t().contiguous()


if __name__ == "__main__":
    main()
