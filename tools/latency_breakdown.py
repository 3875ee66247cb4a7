import os, sys
sys.path.insert(0, "/root/repo")
import torch
from keep_amd import KEEPModel, PROFILE_TAGS
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_prompts, synth_state_dict
sd = synth_state_dict(KEEPShape(), seed=0)
from keep_amd.synth import towers_of
m = KEEPModel(towers=towers_of(sd)); m.load_state_dict(sd); m.to("cuda:0")
toks = {k: v.cuda() for k, v in synth_prompts(1, 256, seed=1).items()}
x = torch.randn(1, 3, 224, 224, device="cuda").to(torch.bfloat16)
for name, f in (("text P=1", lambda: m.encode_text(toks)), ("image B=1", lambda: m.encode_image(x))):
    for _ in range(3): f()
    m.profile_enable(None); m.profile_reset()
    for _ in range(10): f()
    torch.cuda.synchronize()
    print(name, {t: (round(m.profile_read(t)[0] / 10, 3), m.profile_read(t)[1] // 10) for t in PROFILE_TAGS if m.profile_read(t)[1]})
    m.profile_disable()
