#!/usr/bin/env python3
"""Which compensation scheme brings the image tower inside |dcos| <= 1e-4 at the lowest MFMA cost?

CPU experiment on the oracle's arithmetic (full-depth ViT-L, bench weights).  Every GEMM operand site of every block
can be run as
    'x'    exact fp32 (what a hi/lo split with fp16 lo parts gives to 2^-22)
    'h'    fp16 operand, fp32 accumulate                       -- one fp16 MFMA pass
    'h8'   fp16 pass + correction term on the MX-fp8 pipe      -- lo part and its partner rounded to e4m3, unit block scale
    'h4'   fp16 pass + correction term on the MX-fp4 pipe      -- e2m1 elements, one E8M0 scale per 32 k
The correction for the A site is Q(A_lo) . Q(W_hi), for the W site Q(A_hi) . Q(W_lo); the lo x lo term is dropped.
Usage:  precision_study.py blocks | sites | scheme <name> ...
"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keep_amd.config import KEEPShape
from keep_amd.synth import synth_state_dict, synth_tiles
from oracle import keep_oracle as O

torch.set_num_threads(os.cpu_count())
NT = int(os.environ.get("TILES", "8"))
DEV = "cuda" if torch.cuda.is_available() else "cpu"       # pure torch fp32 arithmetic either way (the GPU only makes it fast)
torch.backends.cuda.matmul.allow_tf32 = False
sd = {k: v.to(DEV) for k, v in synth_state_dict(KEEPShape(), seed=0, text=False).items()}
x = synth_tiles(NT, seed=100).to(DEV)
bank = torch.nn.functional.normalize(torch.randn(64, 768, generator=torch.Generator().manual_seed(3)), dim=-1).to(DEV)
r16 = lambda t: t.to(torch.float16).to(torch.float32)
E2M1 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device="cuda" if torch.cuda.is_available() else "cpu")


def q_fp8(t, prescale=1.0):
    """e4m3 round-to-nearest with a fixed power-of-two prescale (the MX instruction runs with unit block scales)."""
    return (t * prescale).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) / prescale


def q_fp4(t):
    """MX-fp4: blocks of 32 along the last dim, E8M0 scale chosen so the block maximum does not clip."""
    shp = t.shape
    b = t.reshape(-1, 32)
    amax = b.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 6.0)))
    v = (b / scale).abs()
    idx = torch.bucketize(v, (E2M1[1:] + E2M1[:-1]) / 2)       # nearest grid point
    return (torch.sign(b) * E2M1[idx] * scale).reshape(shp)


def lin(mode_a, mode_w, a, w, b):
    ah = r16(a) if mode_a != "x" else a
    wh = r16(w) if mode_w != "x" else w
    y = ah @ wh.t()
    if mode_a in ("h8", "h4"):
        al = a - ah
        y = y + (q_fp8(al, 2048.0) @ q_fp8(wh if mode_w != "x" else r16(w)).t() if mode_a == "h8" else q_fp4(al) @ q_fp4(wh).t())
    if mode_w in ("h8", "h4"):
        wl = w - wh
        y = y + (q_fp8(ah if mode_a != "x" else r16(a)) @ q_fp8(wl, 2048.0).t() if mode_w == "h8" else q_fp4(ah) @ q_fp4(wl).t())
    return y + b


def forward(spec):
    """spec(block, site) -> mode; sites: qkv.A qkv.W qkv.out attn.P proj.A proj.W fc1.A fc1.W fc2.A fc2.W"""
    p = "visual."
    wpe = sd[p + "patch_embed.proj.weight"]
    t = O.patchify(x, 16) @ wpe.reshape(1024, -1).t() + sd[p + "patch_embed.proj.bias"]
    t = torch.cat([sd[p + "cls_token"].expand(x.shape[0], -1, -1), t], 1) + sd[p + "pos_embed"]
    for i in range(24):
        bp = f"{p}blocks.{i}."
        m = lambda s: spec(i, s)
        h = O.layer_norm(t, sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], 1e-6)
        qkv = lin(m("qkv.A"), m("qkv.W"), h, sd[bp + "attn.qkv.weight"], sd[bp + "attn.qkv.bias"])
        if m("qkv.out") != "x": qkv = r16(qkv)
        qkv = qkv.reshape(x.shape[0], 197, 3, 16, 64).permute(2, 0, 3, 1, 4)
        pr = torch.softmax((qkv[0] @ qkv[1].transpose(-1, -2)) * 0.125, -1)
        if m("attn.P") != "x": pr = r16(pr)
        a = (pr @ qkv[2]).transpose(1, 2).reshape(x.shape[0], 197, 1024)
        t = t + sd[bp + "ls1.gamma"] * lin(m("proj.A"), m("proj.W"), a, sd[bp + "attn.proj.weight"], sd[bp + "attn.proj.bias"])
        h = O.layer_norm(t, sd[bp + "norm2.weight"], sd[bp + "norm2.bias"], 1e-6)
        u = O.gelu_erf(lin(m("fc1.A"), m("fc1.W"), h, sd[bp + "mlp.fc1.weight"], sd[bp + "mlp.fc1.bias"]))
        t = t + sd[bp + "ls2.gamma"] * lin(m("fc2.A"), m("fc2.W"), u, sd[bp + "mlp.fc2.weight"], sd[bp + "mlp.fc2.bias"])
    f = O.layer_norm(t, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)[:, 0]
    return O.l2_normalize(O.visual_head(sd, f))


SITES = ["qkv.A", "qkv.W", "qkv.out", "attn.P", "proj.A", "proj.W", "fc1.A", "fc1.W", "fc2.A", "fc2.W"]
GEMM_SITES = [s for s in SITES if s.endswith(".A") or s.endswith(".W")]
FLOPS = {"qkv": 1239416832, "proj": 413138944, "fc1": 1652555776, "fc2": 1652555776}     # per tile per block


def cost(spec):
    """MFMA time relative to the all-fp16 tower (GEMMs only): fp8 correction = +1/2 pass, fp4 = +1/4, exact = +1 (fp16 lo pass)."""
    tot = base = 0.0
    for i in range(24):
        for g, fl in FLOPS.items():
            c = 1.0
            for s in (g + ".A", g + ".W"):
                c += {"h": 0.0, "h8": 0.5, "h4": 0.25, "x": 1.0}[spec(i, s)]
            tot += c * fl; base += fl
    return tot / base


def report(name, spec, ref):
    t0 = time.time()
    d = forward(spec) @ bank.t() - ref
    print(f"{name:44s} max|dcos| {d.abs().max():.2e}  rms {d.pow(2).mean().sqrt():.2e}  gemm cost x{cost(spec):.3f}  ({time.time() - t0:.0f}s)", flush=True)
    return d.pow(2).mean().item()


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "blocks"
    with torch.no_grad():
        ref = forward(lambda i, s: "x") @ bank.t()
        if what == "blocks":
            tot = report("all fp16", lambda i, s: "h", ref)
            vs = [report(f"block {b} fp16, rest exact", lambda i, s, b=b: "h" if i == b else "x", ref) for b in range(24)]
            print("variance share per block:", [round(v / sum(vs), 3) for v in vs], f"sum/all = {sum(vs) / tot:.2f}")
        elif what == "sites":
            lo, hi = int(sys.argv[2]), int(sys.argv[3])
            vs = {s: report(f"blocks {lo}..{hi - 1} {s} fp16", lambda i, q, s=s: "h" if (lo <= i < hi and q == s) else "x", ref) for s in SITES}
            tot = sum(vs.values())
            print({s: round(v / tot, 3) for s, v in vs.items()})
        elif what == "scheme":
            for name in (sys.argv[2:] or sorted(SCHEMES)):
                report(name, SCHEMES[name], ref)


def graded(n_full, n_w, full="h4", wonly="h4"):
    """first n_full blocks: both correction terms; next n_w blocks: the W-side term only; rest plain fp16"""
    def spec(i, s):
        if s in ("qkv.out", "attn.P"): return "h"
        if i < n_full: return full
        if i < n_full + n_w: return wonly if s.endswith(".W") else "h"
        return "h"
    return spec


SCHEMES = {
    "fp16": lambda i, s: "h",
    "all_h4": lambda i, s: "h4" if s in GEMM_SITES else "h",
    "all_h8": lambda i, s: "h8" if s in GEMM_SITES else "h",
    "all_x": lambda i, s: "x" if s in GEMM_SITES else "h",
    "w_h4": lambda i, s: "h4" if s.endswith(".W") else "h",
    "w_h8": lambda i, s: "h8" if s.endswith(".W") else "h",
    "a_h4": lambda i, s: "h4" if s.endswith(".A") else "h",
    "mlp_h4": lambda i, s: "h4" if s[:3] in ("fc1", "fc2") and s in GEMM_SITES else "h",
}
for k in (2, 4, 6, 8, 12):
    SCHEMES[f"first{k}_h4"] = graded(k, 0)
    SCHEMES[f"first{k}_h8"] = graded(k, 0, "h8")
    SCHEMES[f"first{k}_x"] = graded(k, 0, "x")
    SCHEMES[f"first{k}_h4_restW"] = graded(k, 24, "h4", "h4")

if __name__ == "__main__":
    main()
