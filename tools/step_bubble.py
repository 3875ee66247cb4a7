#!/usr/bin/env python3
"""How much of a 256-tile step is the fork / join bubble between encode calls?  K calls of 256 tiles against one call of K x 256 tiles cut into the same lanes."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from keep_amd import KEEPModel
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_state_dict
dev = torch.device("cuda", 0)
shape = KEEPShape()
m = KEEPModel(shape); m.load_state_dict(synth_state_dict(shape, seed=0)); m.to(dev).eval()
K = 8
g = torch.Generator(device=dev).manual_seed(1)
tiles = torch.randn(256 * K, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
m.reserve(tiles=256)
def run_calls():
    return torch.cat([m.encode_image(tiles[i * 256:(i + 1) * 256]) for i in range(K)])
def run_one():
    return m.encode_image(tiles)
def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    best = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); o = f(); torch.cuda.synchronize(); best.append(time.perf_counter() - t)
    return min(best), o
for rnd in range(3):
    m.set_option("max_tiles", 256)
    ta, oa = timeit(run_calls)
    m.set_option("max_tiles", 128)
    tb, ob = timeit(run_one)
    m.set_option("max_tiles", 256)
    tc, oc = timeit(run_one)
    print(f"round {rnd}: {K} calls x 256: {256*K/ta:8.1f} tiles/s   one call, lanes of 128: {256*K/tb:8.1f}   one call, lanes of 256: {256*K/tc:8.1f}   equal: {torch.equal(oa, ob)} {torch.equal(oa, oc)}", flush=True)
