"""Single-operator wrappers over the C ABI (keep_op_* in include/keep_hip.h).

Used by the parity tests to check each HIP kernel against a torch fp32/fp64 reference of the same
operator; not needed by users of ``KEEPModel``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .model import _ptr, _stream

EPI_F16, EPI_GELU_F16, EPI_RESID_LS, EPI_RESID_F32 = 0, 1, 2, 4


class Ops:
    def __init__(self, device="cuda"):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.KeepHipError("no GPU visible: keep_amd has no CPU execution path")
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self._h = C.c_void_p(0)
        rc = lib.keep_create(idx, C.byref(self._h))
        if rc:
            raise _lib.KeepHipError(f"keep_create failed ({rc})")

    def __del__(self):
        try:
            if self._h.value:
                _lib.load().keep_destroy(self._h)
        except Exception:
            pass

    def set_option(self, name: str, value: float):
        _lib.check(self._h, _lib.load().keep_set_option(self._h, name.encode(), float(value)), name)

    def debug_timeline(self, nblocks: int):
        """[nblocks,4] int64 shader-clock stamps of the last GEMM launch (needs option gemm_dbg=1)."""
        import numpy as np
        buf = np.zeros((nblocks, 4), dtype=np.int64)
        rc = _lib.load().keep_debug_read(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        _lib.check(self._h, rc, "debug_read")
        return buf

    def _f(self, t: Optional[torch.Tensor]):
        return None if t is None else t.to(self.device, torch.float32).contiguous()

    def linear(self, a, w, bias, epi=EPI_F16, split=False, ls=None, resid=None):
        a, w, bias, ls, resid = map(self._f, (a, w, bias, ls, resid))
        M, K = a.shape
        N = w.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=self.device)
        rc = _lib.load().keep_op_linear(self._h, _ptr(a), _ptr(w), _ptr(bias), _ptr(ls), _ptr(resid), M, N, K, epi,
                                        int(split), _ptr(out), _stream(self.device))
        _lib.check(self._h, rc, "op_linear")
        return out

    def mlp(self, x, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, ls, mode=0):
        """x + ls * fc2(gelu(fc1(layernorm(x)))) through the tower's kernels; mode 0 fp16, 1 split, 2 compensated."""
        x, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, ls = map(self._f, (x, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, ls))
        M, D = x.shape
        F = fc1_w.shape[0]
        out = torch.empty_like(x)
        rc = _lib.load().keep_op_mlp(self._h, _ptr(x), _ptr(ln_w), _ptr(ln_b), _ptr(fc1_w), _ptr(fc1_b), _ptr(fc2_w), _ptr(fc2_b),
                                     _ptr(ls), M, D, F, int(mode), _ptr(out), _stream(self.device))
        _lib.check(self._h, rc, "op_mlp")
        return out

    def attention(self, qkv, B, T, heads, mask=None, split=False):
        qkv = self._f(qkv)
        m = None if mask is None else mask.to(self.device, torch.int64).contiguous()
        out = torch.empty((B * T, heads * 64), dtype=torch.float32, device=self.device)
        rc = _lib.load().keep_op_attention(self._h, _ptr(qkv), _ptr(m), B, T, heads, int(split), _ptr(out),
                                           _stream(self.device))
        _lib.check(self._h, rc, "op_attention")
        return out

    def layernorm(self, x, gamma, beta, eps, add=None):
        x, gamma, beta, add = map(self._f, (x, gamma, beta, add))
        out = torch.empty_like(x)
        rc = _lib.load().keep_op_layernorm(self._h, _ptr(x), _ptr(add), _ptr(gamma), _ptr(beta), x.shape[0], x.shape[1],
                                           float(eps), _ptr(out), _stream(self.device))
        _lib.check(self._h, rc, "op_layernorm")
        return out

    def sgemm(self, a, b, bias=None, scale=1.0, act=0):
        a, b, bias = map(self._f, (a, b, bias))
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=self.device)
        rc = _lib.load().keep_op_sgemm(self._h, _ptr(a), _ptr(b), _ptr(bias), a.shape[0], b.shape[0], a.shape[1],
                                       float(scale), int(act), _ptr(out), _stream(self.device))
        _lib.check(self._h, rc, "op_sgemm")
        return out

    def l2norm_(self, x):
        rc = _lib.load().keep_op_l2norm(self._h, _ptr(x), x.shape[0], x.shape[1], _stream(self.device))
        _lib.check(self._h, rc, "op_l2norm")
        return x
