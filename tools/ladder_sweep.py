#!/usr/bin/env python3
"""Cost / error of 'comp' settings beyond calibrate()'s ladder: the load-time probe's statistics (256 seeded tiles x 64 prompts against the
split-product mode) and the 256-tile step time for each (comp_full_blocks, comp_mlp_blocks[, comp_qkv_from]) given on the command line.
    python tools/ladder_sweep.py 1,6 1,8 2,4 2,6 2,6,1 2,8,1"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keep_amd import KEEPModel                                             # noqa: E402
from keep_amd.config import KEEPShape                                      # noqa: E402
from keep_amd.model import CALIBRATION_POPULATION, expected_max_sigmas     # noqa: E402
from keep_amd.synth import synth_prompts, synth_state_dict                 # noqa: E402

dev = torch.device("cuda", 0)
m = KEEPModel(KEEPShape())
m.auto_calibrate = False
m.load_state_dict(synth_state_dict(KEEPShape(), seed=0))
m.to(dev).eval()
seed = 20250929
g = torch.Generator(device=dev).manual_seed(seed)
probe = torch.randn(256, 3, 224, 224, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
toks = synth_prompts(64, 64, seed=seed % 100003)
m.set_precision("strict")
bank = m.encode_text({k: v.to(dev) for k, v in toks.items()})
ref = m.similarity(m.encode_image(probe), bank)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(256, 3, 224, 224, device=dev, generator=g).to(torch.bfloat16)
z = expected_max_sigmas(CALIBRATION_POPULATION)
print(f"targets for a population of {CALIBRATION_POPULATION}: probe rms <= {1e-4 / z:.3e}, probe max <= {1e-4 * expected_max_sigmas(256 * 64) / z:.3e}")
for rep in range(2):
    for st in sys.argv[1:]:
        v = [int(t) for t in st.split(",")]
        m.set_precision("comp")
        m.set_option("comp_full_blocks", v[0])
        m.set_option("comp_mlp_blocks", v[1])
        m.set_option("comp_qkv_from", v[2] if len(v) > 2 else 1 << 20)
        d = (m.similarity(m.encode_image(probe), bank) - ref).abs()
        for _ in range(3):
            m.encode_image(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m.encode_image(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"{st:10s} probe max {float(d.max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}   {dt * 1e3:7.3f} ms/step {256 / dt:7.1f} tiles/s", flush=True)
