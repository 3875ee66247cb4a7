#!/usr/bin/env python3
"""encode_text of BASELINE config 3's prompt bank (64 prompts x 256 tokens, valid lengths 8-32) for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats -d DIR --output-format csv -- python tools/text_profile.py --trim 1     (run at the longest valid length: T = 32)
    rocprofv3 --kernel-trace --stats -d DIR --output-format csv -- python tools/text_profile.py --trim 0     (run at the padded length, as the reference does)
Text tower only, 'comp' precision (= split products throughout the text tower), 20 timed calls after 3 warm-ups."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keep_amd import KEEPModel, bert_flops_per_prompt                      # noqa: E402
from keep_amd.config import KEEPShape                                      # noqa: E402
from keep_amd.synth import synth_prompts, synth_state_dict                 # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trim", type=int, default=1)
ap.add_argument("--prompts", type=int, default=64)
ap.add_argument("--calls", type=int, default=20)
args = ap.parse_args()
m = KEEPModel(KEEPShape(), towers=("text",))
m.load_state_dict(synth_state_dict(KEEPShape(), seed=0, vision=False))
m.to("cuda:0").eval()
m.trim_padding = bool(args.trim)
toks = {k: v.cuda() for k, v in synth_prompts(args.prompts, 256, seed=1).items()}
for _ in range(3):
    m.encode_text(toks)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.calls):
    m.encode_text(toks)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.calls
print(f"encode_text {args.prompts} prompts x 256 tokens, trim_padding={bool(args.trim)} (run at T = {m.last_text_length}): {dt * 1e3:.3f} ms per call, "
      f"{args.prompts / dt:.0f} prompts/s, {args.prompts * bert_flops_per_prompt() / dt / 1e12:.1f} TFLOP/s padded-equivalent")
