#!/usr/bin/env python3
"""vit.attn time per 256-tile step (HIP events, one internal stream) and the two-lane step time for a list of attn_stagger values."""
import sys, time, torch
sys.path.insert(0, '.')
from keep_amd import KEEPModel
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_state_dict
m = KEEPModel(towers=("image",)); m.auto_calibrate = False
m.load_state_dict(synth_state_dict(KEEPShape(), seed=0, text=False)); m.to("cuda:0")
m.set_option("comp_mlp_blocks", 6)
x = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16)
for st in [int(a) for a in sys.argv[1:]] or [0]:
    m.set_option("attn_stagger", st)
    m.set_option("streams", 1)
    for _ in range(3): m.encode_image(x)
    m.profile_enable("vit.attn"); m.profile_reset()
    for _ in range(5): m.encode_image(x)
    torch.cuda.synchronize(); ms, n, _ = m.profile_read("vit.attn"); m.profile_disable()
    m.set_option("streams", 2)
    for _ in range(3): m.encode_image(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): m.encode_image(x)
    torch.cuda.synchronize(); step = (time.perf_counter() - t0) / 15 * 1e3
    print(f"attn_stagger={st:6d}: vit.attn {ms / 5:.3f} ms per step on one stream ({n // 5} launches), two-lane step {step:.2f} ms")
