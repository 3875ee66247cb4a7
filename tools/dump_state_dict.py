#!/usr/bin/env python3
"""Write a state_dict and a tile batch in the flat binary form examples/c_abi_demo.cpp reads.

    python tools/dump_state_dict.py weights.bin tiles.bin [--release-dir DIR] [--depth 24] [--tiles 4]

Without --release-dir the weights are the seeded synthetic release-layout state_dict (keep_amd.synth)."""
import argparse, os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def write_state_dict(sd, path):
    with open(path, "wb") as f:
        for key, t in sd.items():
            t = t.detach().to(torch.float32).contiguous().cpu()
            k = key.encode()
            f.write(struct.pack("<I", len(k))); f.write(k)
            f.write(struct.pack("<I", t.dim()))
            f.write(struct.pack(f"<{t.dim()}q", *t.shape))
            f.write(t.numpy().tobytes())


def write_tiles(tiles, path):
    code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.uint8: 3}[tiles.dtype]
    with open(path, "wb") as f:
        f.write(struct.pack("<qi", tiles.shape[0], code))
        f.write(tiles.contiguous().view(torch.uint8).numpy().tobytes())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("weights"); ap.add_argument("tiles")
    ap.add_argument("--release-dir"); ap.add_argument("--depth", type=int, default=24); ap.add_argument("--tiles-n", type=int, default=4)
    a = ap.parse_args()
    from keep_amd.config import KEEPShape, small_shape
    from keep_amd.synth import synth_state_dict, synth_tiles
    if a.release_dir:
        p = os.path.join(a.release_dir, "pytorch_model.bin")
        sd = torch.load(p, map_location="cpu", weights_only=True)
    else:
        sd = synth_state_dict(KEEPShape() if a.depth >= 24 else small_shape(a.depth, max(1, a.depth // 2)), seed=0)
    write_state_dict(sd, a.weights)
    write_tiles(synth_tiles(a.tiles_n, seed=1).to(torch.bfloat16), a.tiles)
