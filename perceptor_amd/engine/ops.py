"""Tensor-level wrappers over the C ABI (perceptor_amd/_hip.py).

Activations are NHWC 16-bit torch tensors ([N, H, W, C] or [M, C]); every
function launches hand-written HIP kernels on the current stream.  torch is
used only for device memory.
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Optional

import torch

from .. import _hip
from .._hip import ACT_GEGLU, ACT_NONE, DT_F16X2, IgemmArgs, call, ptr


HALO_ENABLED = True
SPLITK_ENABLED = True
WD_ENABLED = True       # weights-direct conv3x3 kernel (csrc/conv_wd.hip) where the shape is eligible
GEMM_WD_ENABLED = True       # weights-direct GEMM (csrc/gemm_wd.hip) for plain GEMMs
GEMM_WD_CONV = os.environ.get("PMI_GEMM_WD_CONV", "1") != "0"   # A-B switch: small-map 3x3 convolutions on the weights-direct GEMM (csrc/gemm_wd.hip, CONV)
FLASH_ENABLED = True    # general flash attention (csrc/attn_flash.hip) instead of batched GEMMs + softmax where the head dim is not 64


def set_halo(enabled: bool) -> None:
    """A/B switch: route conv3x3 through the LDS-halo kernel (default) or the generic implicit-GEMM kernel."""
    global HALO_ENABLED
    HALO_ENABLED = bool(enabled)
    _hip.lib().pmi_set_option(0, int(enabled))


# bench.py sets this to a list to time every 3x3-convolution launch (conv3x3_wd_kernel, conv3x3_halo_kernel) with HIP events on the launch stream
KERNEL_EVENTS = None
GEMM_TRACE = None    # tools/gemm_trace.py: list collecting (desc, flops, ev0, ev1) of every pmi_igemm launch
DEBUG_WS = None      # tools/conv_probe.py --stamps: int64 buffer the PMI_STAMPS build of conv3x3.hip writes phase timestamps to


def _empty(shape, dtype, device):
    return torch.empty(shape, dtype=dtype, device=device)


def split_group(c: int) -> int:
    """Group size of a precise (hi + lo) tensor with c logical channels (csrc/common.h: F16X2)."""
    if c % 32 == 0:
        return 32
    if c > 32 or c % 8:
        raise ValueError(f"precise tensors need a channel count that is a multiple of 32 (or of 8 below 32), got {c}")
    return c


def logical_c(x: torch.Tensor, dt: int) -> int:
    return x.shape[-1] // 2 if dt == DT_F16X2 else x.shape[-1]


class PackedLinear:
    """Weights packed for pmi_igemm: B[Npad][K] 16-bit with k = tap*Cin + c, fp32 bias.

    dt = precise (DT_F16X2): the inputs are hi + lo tensors with 2*Cin channels per pixel ([hi G | lo G] per group of G channels), so
    the f16 weights are duplicated along K in the same pattern: W*hi + W*lo accumulates in fp32 on the MFMA.  `sources` gives the
    logical channel counts of a two-pointer concat input (each source is its own precise tensor with its own grouping)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], dt: int, device, cin_pad: Optional[int] = None, sources=None):
        w = weight.detach().float()
        if w.ndim == 3:   # Conv1d k=1
            w = w[..., 0]
        if w.ndim == 2:
            w = w[:, :, None, None]
        cout, cin, kh, kw = w.shape
        self.cout, self.cin, self.taps = cout, cin, kh * kw
        self.cin_p = cin_pad if cin_pad is not None else (cin + 7) // 8 * 8
        self.n_p = (cout + 3) // 4 * 4
        packed = torch.zeros(self.n_p, kh * kw, self.cin_p, dtype=torch.float32)
        packed[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
        self.split = dt == DT_F16X2
        self.cin_l = self.cin_p                      # logical input channels
        self.self_concat = False
        if self.split:
            def dup(wt, lo_zero=False):              # [N, taps, C] -> [N, taps, 2C] in the [hi G | lo G] pattern of the input
                parts, o = [], 0
                for cs in (sources or [self.cin_l]):
                    g = split_group(cs)
                    blk = wt[:, :, o:o + cs].reshape(self.n_p, kh * kw, cs // g, 1, g).expand(-1, -1, -1, 2, -1).clone()
                    if lo_zero:
                        blk[:, :, :, 1, :] = 0
                    parts.append(blk.reshape(self.n_p, kh * kw, 2 * cs))
                    o += cs
                assert o == self.cin_l
                return torch.cat(parts, dim=2)
            w_hi = packed.to(torch.float16).float()
            w_lo = packed - w_hi                     # what f16 cannot hold of fp32 weights (zero for fp16 checkpoints / the synthetic weights)
            self._w1, self._sources = w_hi, list(sources or [self.cin_l])     # un-duplicated weights: other duplication patterns are built on demand
            packed = dup(w_hi)
            self.cin_p = 2 * self.cin_l              # channels of the (physical) input tensors
            # f16 holds the weights to within the mode's own precision: nothing to add.  Otherwise the low part becomes a second K block.
            # The measure is the low part's share of the weight NORM (what reaches a sum over K), not its largest element: bf16-exact or
            # fp16-checkpoint weights of a long-K layer have a few values under the f16 subnormal grid (< 2^-14: they lose < 3e-8 absolute,
            # ~4e-8 of the norm) -- the max-element test of round 2 doubled K for every such layer (4x the MFMA work, for nothing).
            if float(w_lo.norm()) > 2.0 ** -22 * float(w_hi.norm()):
                if sources is not None:
                    import warnings
                    warnings.warn("precise mode: fp32 weights of a two-source (concat) convolution are rounded to f16")
                else:
                    # second K block over the SAME input tensor (passed again as the second source): W_lo * x_hi
                    packed = torch.cat([packed, dup(w_lo, lo_zero=True)], dim=2)
                    self.self_concat = True
                    self.cin_p = 2 * self.cin_p
        self.w = packed.reshape(self.n_p, -1).to(device=device, dtype=_hip.TORCH_DTYPE[dt]).contiguous()
        self.b = None
        if bias is not None:
            b = torch.zeros(self.n_p, dtype=torch.float32)
            b[:cout] = bias.detach().float()
            self.b = b.to(device)
        self.dt = dt
        self._frag = {}

    @property
    def K(self):
        return self.taps * self.cin_p

    def frag_gemm(self) -> torch.Tensor:
        """Fragment order for the weights-direct GEMM (csrc/gemm_wd.hip), 16x16x32 MFMA:
        [N/32][K/128][32-deep k-step (4)][16-column block (2)][lane = 16*(k quarter) + column][8 k]."""
        if "gemm" not in self._frag:
            assert self.taps in (1, 9) and self.n_p % 32 == 0 and self.K % 32 == 0     # (taps 9: the GEMM kernel's conv mode, k = tap * Cin + c)
            kp = (self.K + 127) // 128 * 128                                    # K tail: zero weights up to a whole 128-deep chunk
            w = self.w
            if kp != self.K:
                w = torch.zeros((self.n_p, kp), dtype=self.w.dtype, device=self.w.device)
                w[:, :self.K] = self.w
            w = w.view(self.n_p // 32, 2, 16, kp // 128, 4, 4, 8)              # nb, cb, r16, chunk, k32, q4, j
            self._frag["gemm"] = w.permute(0, 3, 4, 1, 5, 2, 6).contiguous()
        return self._frag["gemm"]

    def frag16(self, ck: int, dup_g: int = 0) -> torch.Tensor:
        """Fragment order for the 16x16x32 MFMA form of the weights-direct kernel:
        [N/32][Cin/ck][dx][ck/32][dy][16-channel block][lane = 16*(k quarter) + channel][8 k].
        dup_g (split weights only): the K order [W of dup_g logical channels | the same again] per group -- the kernel's fused prologue
        over a split input stages a chunk as [yh of ck/2 channels | their yl] (csrc/conv_wd.hip, SIN = 1), so dup_g = ck / 2; the default
        is the tensors' own memory order, groups of 32."""
        key = ("mf16", ck, dup_g)
        if key not in self._frag:
            assert self.taps == 9 and self.n_p % 32 == 0 and self.cin_p % ck == 0
            src = self.w
            if dup_g and dup_g != 32:
                assert self.split and not self.self_concat and all(cs % dup_g == 0 for cs in self._sources)
                parts, o = [], 0
                for cs in self._sources:
                    blk = self._w1[:, :, o:o + cs].reshape(self.n_p, 9, cs // dup_g, 1, dup_g).expand(-1, -1, -1, 2, -1)
                    parts.append(blk.reshape(self.n_p, 9, 2 * cs))
                    o += cs
                src = torch.cat(parts, dim=2).reshape(self.n_p, -1).to(device=self.w.device, dtype=self.w.dtype).contiguous()
            w = src.view(self.n_p // 32, 2, 16, 3, 3, self.cin_p // ck, ck // 32, 4, 8)   # nb, cb, r16, dy, dx, chunk, k32, q4, j
            self._frag[key] = w.permute(0, 5, 4, 6, 3, 1, 7, 2, 8).contiguous()
        return self._frag[key]

    def frag_c8(self) -> torch.Tensor:
        """Fragment order for config 8 of the weights-direct kernel (at most 32 input channels): k = tap * Cin + c zero-padded to whole
        32-deep MFMA steps, [N/32][steps][16-channel block (2)][lane = 16*(k quarter) + channel][8 k]."""
        if "c8" not in self._frag:
            assert self.taps == 9 and self.n_p % 32 == 0 and self.cin_p % 8 == 0 and self.cin_p <= 32
            ks = (9 * self.cin_p + 31) // 32
            w = torch.zeros((self.n_p, ks * 32), dtype=self.w.dtype, device=self.w.device)
            w[:, :self.K] = self.w
            w = w.view(self.n_p // 32, 2, 16, ks, 4, 8)                       # nb, cb, r16, step, q4, j
            self._frag["c8"] = w.permute(0, 3, 1, 4, 2, 5).contiguous()
        return self._frag["c8"]

    def frag(self, ck: int) -> torch.Tensor:
        """The 3x3 weights in MFMA fragment order for the weights-direct kernel (csrc/conv_wd.hip):
        [N/32][Cin/ck][dx][ck/16][dy][lane = 32*(k half) + channel][8 k] -- every (n-block, chunk, dx, k-step, dy)
        fragment is one contiguous 1 KB piece and a wave's whole weight stream is contiguous in its loop order."""
        if ck not in self._frag:
            assert self.taps == 9 and self.n_p % 32 == 0 and self.cin_p % ck == 0
            w = self.w.view(self.n_p // 32, 32, 3, 3, self.cin_p // ck, ck // 16, 2, 8)   # nb, l31, dy, dx, chunk, ks, h, j
            self._frag[ck] = w.permute(0, 4, 3, 5, 2, 6, 1, 7).contiguous()
        return self._frag[ck]


class MixedLinear:
    """Both packings of one 3x3 convolution for the mixed mode (engine/adm_mixed.py), built on first use: `single` = plain f16 weights over the
    logical channels (the operand act(GroupNorm(hi + lo)) is rounded once to f16), `dbl` = weights duplicated along K for the hi + lo operand
    (also what the generic fallback path takes)."""

    def __init__(self, weight, bias, device, sources=None, cin_pad=None):
        self._args = (weight, bias, device, sources, cin_pad)
        self._single = self._dbl = None

    @property
    def single(self) -> PackedLinear:
        if self._single is None:
            w, b, dev, _, cp = self._args
            self._single = PackedLinear(w, b, _hip.DT_F16, dev, cin_pad=cp)
        return self._single

    @property
    def dbl(self) -> PackedLinear:
        if self._dbl is None:
            w, b, dev, src, cp = self._args
            self._dbl = PackedLinear(w, b, DT_F16X2, dev, cin_pad=cp, sources=src)
        return self._dbl


def split_convert(x: torch.Tensor, to_split: bool) -> torch.Tensor:
    """plain f16 [..., C] <-> split (hi + lo) [..., 2C] (pmi_split_convert): the level boundaries of the mixed mode."""
    c = x.shape[-1] if to_split else x.shape[-1] // 2
    y = _empty(x.shape[:-1] + ((2 * c) if to_split else c,), torch.float16, x.device)
    call("pmi_split_convert", ptr(x), ptr(y), x.numel() // x.shape[-1], c, int(to_split))
    return y


MIXED_TRACE = None      # tools / tests: list collecting (route, operand, shape) of every conv3x3_mixed call


def conv3x3_mixed(x: torch.Tensor, mlin: MixedLinear, *, operand: str, prologue, x1: Optional[torch.Tensor] = None, up: bool = False,
                  residual: Optional[torch.Tensor] = None, res_up: bool = False, nbias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Stride-1 3x3 convolution of the mixed mode: x (and x1: skip-concat) are split tensors [N, H, W, 2C]; prologue = (coef_a, coef_b, act) is
    the fused GroupNorm-apply + SiLU of the input; operand "single": the activated value is rounded once to f16 (K = 9 Cin), "dbl": it is
    kept as hi + lo (K = 18 Cin, fp32-grade product).  Split output [N, H, W, 2 Cout] with fused statistics; split residual.
    Shapes the weights-direct kernel does not take fall back to apply pass + generic split convolution (always the doubled operand)."""
    assert operand in ("single", "dbl")
    n, hin, win, _ = x.shape
    h, w = (hin * 2, win * 2) if up else (hin, win)
    ca, cb, pact = prologue
    lin = mlin.single if operand == "single" else mlin.dbl
    a = IgemmArgs()
    div = 2 if operand == "single" else 1                  # single: C0 / C1 / K count logical channels
    c0, c1 = x.shape[-1] // div, (x1.shape[-1] // div if x1 is not None else 0)
    a.H, a.W, a.Hin, a.Win, a.hw = h, w, hin, win, h * w
    m = n * h * w
    a.M, a.N, a.K, a.C0, a.C1 = m, lin.n_p, lin.K, c0, c1
    a.lda0, a.lda1 = x.stride(-2), (x1.stride(-2) if x1 is not None else 0)
    a.taps, a.stride, a.up, a.res_up, a.act, a.alpha = 9, 1, int(up), int(res_up), ACT_NONE, 1.0
    a.batch, a.batch_inner, a.dtype = 1, 1, DT_F16X2
    a.split_in, a.split_out = (2 if operand == "single" else 1), 32
    a.A0, a.A1, a.B, a.bias, a.nbias, a.R = ptr(x), ptr(x1), ptr(lin.w), ptr(lin.b), ptr(nbias), ptr(residual)
    a.ldb, a.ldr, a.ldnb = lin.K, (residual.stride(-2) if residual is not None else 0), (nbias.stride(0) if nbias is not None else 0)
    a.res_f32 = int(residual is not None and residual.dtype == torch.float32)     # an fp32 residual (the 1x1 skip_connection's output) costs the same bytes as a split one
    a.pro_a, a.pro_b, a.pro_act, a.Bf = 1, 1, pact, 1      # markers for the config query
    cfg = _hip.lib().pmi_conv3x3_halo_config(C.byref(a)) if (lin.n_p % 128 == 0 and lin.K == 9 * (c0 + c1) and WD_ENABLED and HALO_ENABLED) else -1
    if MIXED_TRACE is not None:
        MIXED_TRACE.append(("wd" if cfg >= 6 else "fallback", operand, (n, h, w, c0 + c1, lin.cout)))
    if cfg < 6:
        if residual is not None and residual.dtype == torch.float32:      # the generic split epilogue reads a split residual
            rs = _empty(residual.shape[:-1] + (2 * residual.shape[-1],), torch.float16, x.device)
            call("pmi_split_from_f32", ptr(residual), residual.stride(-2), ptr(rs), residual.numel() // residual.shape[-1], residual.shape[-1])
            residual = rs
        return igemm(x, mlin.dbl, a1=x1, up=up, residual=residual, res_up=res_up, nbias=nbias, prologue=prologue, want_stats=True)
    ck = 64 if cfg == 6 else 32
    a.Bf = ptr(lin.frag16(ck) if operand == "single" else lin.frag16(ck, dup_g=ck // 2))
    a.pro_a, a.pro_b, a.pro_act = ptr(ca), ptr(cb), pact
    out = _empty((n, h, w, 2 * lin.n_p), torch.float16, x.device)
    a.D, a.ldd = ptr(out), out.stride(-2)
    rows = _hip.lib().pmi_igemm_stats_rows(C.byref(a))
    if rows > 0:
        st = _empty((n, rows, lin.n_p, 2), torch.float32, x.device)
        a.stats, a.stats_p = ptr(st), rows
        out._pmi_stats = (st, rows)
    if DEBUG_WS is not None:
        a.ws = ptr(DEBUG_WS)
        a.reserved = 77
    if KERNEL_EVENTS is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("pmi_igemm", C.byref(a))
        e1.record()
        nbytes = (x.numel() + (x1.numel() if x1 is not None else 0)) * 2 + lin.w.numel() * 2 + out.numel() * 2 + (residual.numel() * 2 if residual is not None else 0)
        KERNEL_EVENTS.append((e0, e1, 2.0 * m * lin.cout * lin.cin * 9, float(nbytes), f"{h}x{w} {c0}+{c1}->{lin.cout} cfg{cfg} mixed-{operand}"))
        return out
    call("pmi_igemm", C.byref(a))
    return out


def igemm(a0: torch.Tensor, lin: PackedLinear, *, a1: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
          act: int = ACT_NONE, up: bool = False, stride: int = 1, res_up: bool = False, nbias: Optional[torch.Tensor] = None,
          out_f32: bool = False, out: Optional[torch.Tensor] = None, alpha: float = 1.0, prologue=None,
          want_stats: bool = False, hw: Optional[int] = None, pre_out: Optional[torch.Tensor] = None,
          act_grad_of: Optional[torch.Tensor] = None, act_grad: int = ACT_NONE, defer_reduce: bool = False,
          split_out: bool = False) -> torch.Tensor:
    """Convolution (a0 is [N,H,W,C]) or linear (a0 is [M,C]) through pmi_igemm.

    prologue = (coef_a [N,Cin], coef_b [N,Cin], act): fused GroupNorm-apply(+FiLM)+activation on the conv input
    (LDS-halo conv3x3 kernel only); when the shape is not eligible the apply kernel runs first.
    pre_out: 16-bit tensor like the output that receives the PRE-activation value (fused act epilogue keeps the backward's input);
    act_grad_of / act_grad: the output is multiplied by act'(act_grad_of) -- both only where fused_mlp_epilogues(lin) is True."""
    dt = lin.dt
    conv = a0.ndim == 4
    c0 = a0.shape[-1]
    c1 = a1.shape[-1] if a1 is not None else 0
    assert (c0 + c1) * (2 if lin.self_concat else 1) == lin.cin_p, (c0, c1, lin.cin_p)
    a = IgemmArgs()
    if conv:
        n, hin, win, _ = a0.shape
        hv, wv = (hin * 2, win * 2) if up else (hin, win)
        h, w = hv // stride, wv // stride
        m = n * h * w
        a.H, a.W, a.Hin, a.Win = h, w, hin, win
        a.hw = h * w
        oshape = (n, h, w, lin.n_p)
    else:
        assert lin.taps == 1 and not up and stride == 1
        m = a0.shape[0]
        a.hw = hw or 1
        oshape = (m, lin.n_p // 2 if act == ACT_GEGLU else lin.n_p)
    if lin.split:
        a.split_in = 1
    if (lin.split and not out_f32) or split_out:          # precise output: hi + lo pairs, 2*N 16-bit values per row (split_out: from plain f16
        a.split_out = split_group(lin.n_p)                # operands too -- the mixed mode's attention projection back onto a split stream)
        oshape = oshape[:-1] + (2 * lin.n_p,)
    if out is None:
        out = _empty(oshape, torch.float32 if out_f32 else _hip.TORCH_DTYPE[dt], a0.device)
    a.A0, a.A1, a.B = ptr(a0), ptr(a1), ptr(lin.w)
    a.bias, a.nbias, a.R, a.D = ptr(lin.b), ptr(nbias), ptr(residual), ptr(out)
    a.M, a.N, a.K = m, lin.n_p, lin.K
    a.C0, a.C1 = c0, c1
    a.lda0, a.lda1 = a0.stride(-2), (a1.stride(-2) if a1 is not None else 0)
    a.ldb, a.ldd = lin.K, out.stride(-2)
    a.ldr = residual.stride(-2) if residual is not None else 0
    a.taps, a.stride, a.up, a.res_up, a.act = lin.taps, stride, int(up), int(res_up), act
    a.out_f32 = int(out.dtype == torch.float32)
    a.res_f32 = int(residual is not None and residual.dtype == torch.float32)
    a.alpha = alpha
    a.ldnb = nbias.stride(0) if nbias is not None else 0
    a.batch, a.batch_inner = 1, 1
    a.dtype = dt
    if conv and lin.taps == 9 and stride == 1 and HALO_ENABLED and WD_ENABLED and lin.n_p % 32 == 0 and ((lin.n_p >= 128 and lin.cin_p % 64 == 0) or (lin.cin_p <= 32 and a1 is None)):
        a.Bf = 1                       # ask which tile config the weights-direct kernel would run, then hand it that packing
        a.pro_a = 1 if (prologue is not None and not lin.split) else None     # (the table size limit depends on a fused prologue)
        cfg = _hip.lib().pmi_conv3x3_halo_config(C.byref(a))
        a.pro_a = None
        a.Bf = ptr(lin.frag16(64)) if cfg == 6 else ptr(lin.frag16(32)) if cfg == 7 else ptr(lin.frag(64)) if cfg == 4 else ptr(lin.frag_c8()) if cfg == 8 else None
    if pre_out is not None or act_grad_of is not None:
        a.D2, a.aux, a.aux_act = ptr(pre_out), ptr(act_grad_of), act_grad
    # (split weights: their duplicated K is an ordinary K for the weights-direct GEMM; its epilogues write plain 16-bit or fp32 rows only)
    wd_ok = GEMM_WD_ENABLED and lin.taps == 1 and (not lin.split or out_f32) and not up and stride == 1 and lin.n_p % 32 == 0 and lin.K % 32 == 0 \
        and nbias is None and prologue is None
    if wd_ok and want_stats:
        # the weights-direct GEMM has no statistics epilogue; the generic kernel has none either once it splits K (attention proj_out on
        # 16x16 / 8x8 maps): then the faster GEMM runs and the consumer's GroupNorm takes its statistics pass as before
        wd_ok = SPLITK_ENABLED and _hip.lib().pmi_igemm_splitk(C.byref(a)) > 1
    if wd_ok:
        a.Bf = 1                               # plain GEMM: ask whether the weights-direct kernel (csrc/gemm_wd.hip) takes this shape ...
        a.Bf = ptr(lin.frag_gemm()) if _hip.lib().pmi_gemm_wd_eligible(C.byref(a)) else None      # ... and only then pack its weight order
        if a.Bf:
            want_stats = False
    if prologue is not None:
        ca, cb, pact = prologue
        if HALO_ENABLED and not lin.split and _hip.lib().pmi_conv3x3_halo_config(C.byref(a)) >= 0:
            a.pro_a, a.pro_b, a.pro_act = ptr(ca), ptr(cb), pact
        else:   # not eligible: materialise act(x*a+b) with the streaming kernel, then convolve
            n_, h_, w_, _ = a0.shape
            y = _empty((n_, h_, w_, c0 + c1), a0.dtype, a0.device)
            lc0, lc = (c0 // 2, (c0 + c1) // 2) if lin.split else (c0, c0 + c1)       # logical channel counts
            call("pmi_gn_apply", ptr(a0), ptr(a1), lc0, ptr(ca), ptr(cb), None, ptr(y), n_, h_, w_, lc, pact, 0, dt)
            a.A0, a.A1, a.C0, a.C1, a.lda0, a.lda1 = ptr(y), None, c0 + c1, 0, c0 + c1, 0
            a0 = y
    if lin.self_concat:        # precise mode, fp32 weights: their low part multiplies the SAME input again as a second source
        assert a.A1 is None
        a.A1, a.C1, a.lda1 = a.A0, a.C0, a.lda0
    conv_gemm = False
    if conv and lin.taps == 9 and stride == 1 and not up and GEMM_WD_ENABLED and GEMM_WD_CONV and not lin.split and not a.Bf \
            and not a.pro_a and lin.n_p % 32 == 0 and c0 % 128 == 0 and c1 % 128 == 0:
        a.Bf = 1           # a 3x3 convolution the conv3x3 kernels do not take (16x16 / 8x8 maps): the weights-direct GEMM's conv mode, if it does
        a.Bf = ptr(lin.frag_gemm()) if _hip.lib().pmi_gemm_wd_eligible(C.byref(a)) else None
        conv_gemm = bool(a.Bf)
    if SPLITK_ENABLED:
        sk = _hip.lib().pmi_igemm_splitk(C.byref(a))
        if sk > 1:   # few output tiles, long K: split the reduction over grid.z into fp32 slabs
            ws = _empty((sk, m, lin.n_p), torch.float32, a0.device)
            a.ws, a.splitk = ptr(ws), sk
            if defer_reduce and a.Bf and not conv and _hip.lib().pmi_gemm_wd_eligible(C.byref(a)):
                # the caller fuses the reduction (+ bias / residual) into its next pass (LayerNorm): leave the raw slabs
                a.reserved3 = 1
                call("pmi_igemm", C.byref(a))
                return ("slabs", ws, sk)
    if conv_gemm and (nbias is not None or res_up) and a.splitk <= 1:
        a.Bf = None            # a per-sample bias / up-sampled residual is the split-K reduce kernel's: unsplit, the generic kernel takes the call
    if want_stats:
        rows = _hip.lib().pmi_igemm_stats_rows(C.byref(a))
        if rows > 0:   # fused per-channel (sum, sumsq) of the output for the next GroupNorm
            st = _empty((m // a.hw, rows, lin.n_p, 2), torch.float32, a0.device)
            a.stats, a.stats_p = ptr(st), rows
            out._pmi_stats = (st, rows)
    if DEBUG_WS is not None and a.splitk <= 1:      # (a split-K call's ws is its slab workspace: the phase stamps of the probes are for unsplit calls only)
        a.ws = ptr(DEBUG_WS)
        a.reserved = 77
    if KERNEL_EVENTS is not None and lin.taps == 9 and HALO_ENABLED and _hip.lib().pmi_conv3x3_halo_config(C.byref(a)) >= 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("pmi_igemm", C.byref(a))
        e1.record()
        # algorithmic HBM bytes of the launch: input read once, packed weights, output written once, residual read once
        nbytes = (a0.numel() + (a1.numel() if a1 is not None else 0)) * 2 + lin.w.numel() * 2 + out.numel() * out.element_size()
        if residual is not None:
            nbytes += (residual.numel() if not res_up else residual.numel()) * residual.element_size()
        desc = f"{a.H}x{a.W} {c0}+{c1}->{lin.cout} cfg{_hip.lib().pmi_conv3x3_halo_config(C.byref(a))}" \
               f"{' pro' if prologue is not None else ''}{' res' if residual is not None else ''}{' up' if up else ''}{' stats' if want_stats else ''}"
        KERNEL_EVENTS.append((e0, e1, 2.0 * m * lin.cout * lin.cin * lin.taps, float(nbytes), desc))
        return out
    if GEMM_TRACE is not None:
        kind = "conv" if conv and (lin.taps == 9 or up or stride == 2) else "gemm"
        desc = f"{kind} M={m} N={lin.n_p} K={lin.K} taps={lin.taps}{' up' if up else ''}{' s2' if stride == 2 else ''} splitk={a.splitk}" \
               f" halo={_hip.lib().pmi_conv3x3_halo_config(C.byref(a)) if HALO_ENABLED else -1} wd={int(bool(a.Bf) and bool(_hip.lib().pmi_gemm_wd_eligible(C.byref(a))))}" \
               f"{' res' if residual is not None else ''}{' f32out' if a.out_f32 else ''}{' nbias' if nbias is not None else ''}{' stats' if a.stats else ''}"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("pmi_igemm", C.byref(a))
        e1.record()
        GEMM_TRACE.append((desc, 2.0 * m * lin.n_p * lin.K, e0, e1))
        return out
    call("pmi_igemm", C.byref(a))
    return out


def geglu_linear(x: torch.Tensor, lin: PackedLinear) -> torch.Tensor:
    """x [M, K] @ lin (columns packed as 16 value | 16 gate per 32: interleave_geglu) -> value * gelu(gate) [M, N / 2].
    In the GEMM's epilogue where the weights-direct kernel takes the shape; otherwise the GEMM followed by the gate pass."""
    dt = lin.dt
    a = IgemmArgs()
    a.taps, a.stride, a.M, a.N, a.K, a.C0, a.batch, a.Bf, a.act = 1, 1, x.shape[0], lin.n_p, lin.K, lin.K, 1, 1, ACT_GEGLU
    if GEMM_WD_ENABLED and not lin.split and lin.n_p % 32 == 0 and lin.K % 32 == 0 and _hip.lib().pmi_gemm_wd_eligible(C.byref(a)):
        return igemm(x, lin, act=ACT_GEGLU)
    f = igemm(x, lin)
    out = _empty((x.shape[0], lin.n_p // 2), f.dtype, f.device)
    call("pmi_geglu", ptr(f), ptr(out), x.shape[0], lin.n_p // 2, 1, dt)
    return out


def interleave_geglu(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """Rows of a GEGLU projection [2F, K] = (value F | gate F) reordered to 16 value rows, their 16 gate rows, 16 value rows, ...
    (every 32-column slice a wave of the weights-direct GEMM owns then holds matching value / gate columns)."""
    f = weight.shape[0] // 2
    assert f % 16 == 0
    idx = torch.arange(f).view(-1, 16)
    perm = torch.cat([idx, idx + f], dim=1).flatten()
    return weight[perm].contiguous(), (bias[perm].contiguous() if bias is not None else None)


def fused_mlp_epilogues(lin: PackedLinear, m: int) -> bool:
    """True when a plain 16-bit-output GEMM of m rows with these weights runs in the weights-direct kernel UNSPLIT, whose epilogue can also
    write the pre-activation value (pre_out) and multiply by an activation gradient (act_grad_of).  The library decides (its split-K cost
    model and A/B options included): pmi_igemm rejects D2 / aux on any other route."""
    # the structural part keeps the choice independent of the batch a rank holds (fused and unfused epilogues round differently: a shard must
    # reproduce its slice of the full-batch gradient, tests/test_gpu_clip.py): wide layers only, any m a ViT batch produces
    if not (GEMM_WD_ENABLED and lin.taps == 1 and not lin.split and lin.n_p % 256 == 0 and lin.K % 128 == 0 and m >= 64):
        return False
    a = IgemmArgs()
    a.taps, a.stride, a.M, a.N, a.K, a.C0, a.batch, a.batch_inner, a.hw = 1, 1, m, lin.n_p, lin.K, lin.K, 1, 1, 1
    a.lda0, a.ldb, a.ldd, a.dtype, a.alpha = lin.K, lin.K, lin.n_p, lin.dt, 1.0
    a.Bf, a.D2 = 1, 1                        # non-null markers: eligibility is asked for a call WITH the second output
    if SPLITK_ENABLED and _hip.lib().pmi_igemm_splitk(C.byref(a)) > 1:
        return False
    return bool(_hip.lib().pmi_gemm_wd_eligible(C.byref(a)))


def bgemm(A: torch.Tensor, B: torch.Tensor, D: torch.Tensor, *, M: int, N: int, K: int, lda: int, ldb: int, ldd: int,
          batch: int, batch_inner: int, sA, sB, sD, dt: int, alpha: float = 1.0, a_off: int = 0, b_off: int = 0, d_off: int = 0):
    """Batched D[z] = alpha * A[z] @ B[z]^T with two-level (outer, inner) element strides."""
    a = IgemmArgs()
    es = A.element_size()
    a.A0 = A.data_ptr() + a_off * es
    a.B = B.data_ptr() + b_off * B.element_size()
    a.D = D.data_ptr() + d_off * D.element_size()
    a.M, a.N, a.K, a.C0, a.C1 = M, N, K, K, 0
    a.lda0, a.ldb, a.ldd = lda, ldb, ldd
    a.taps, a.stride, a.hw, a.alpha = 1, 1, 1, alpha
    a.out_f32 = int(D.dtype == torch.float32)
    a.batch, a.batch_inner = batch, batch_inner
    a.sA_o, a.sA_i = sA
    a.sB_o, a.sB_i = sB
    a.sD_o, a.sD_i = sD
    a.dtype = dt
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("pmi_igemm", C.byref(a))
        e1.record()
        GEMM_TRACE.append((f"bgemm M={M} N={N} K={K} batch={batch}", 2.0 * M * N * K * batch, e0, e1))
        return D
    call("pmi_igemm", C.byref(a))
    return D


def _gn_coeffs(x, x1, gamma, beta, groups, dt, film, film_ld, eps):
    n, h, w, _ = x.shape
    c0 = logical_c(x, dt)
    c1 = logical_c(x1, dt) if x1 is not None else 0
    c, hw, dev = c0 + c1, h * w, x.device
    ca = _empty((n, c), torch.float32, dev)
    cb = _empty((n, c), torch.float32, dev)
    st0 = getattr(x, "_pmi_stats", None)
    st1 = getattr(x1, "_pmi_stats", None) if x1 is not None else None
    if st0 is not None and (x1 is None or st1 is not None):
        # statistics came out of the producing kernels' epilogues: no pass over the activations at all
        call("pmi_gn_finalize", ptr(st0[0]), st0[1], c0, ptr(st1[0]) if st1 else None, st1[1] if st1 else 0, c1,
             ptr(gamma), ptr(beta), ptr(film), film_ld, ptr(ca), ptr(cb), n, hw, groups, eps)
        return ca, cb
    # pixel chunks per sample: ~1024 workgroups per launch, at least 8 pixels each (64 left a 16x16 map at batch 8 with 32 workgroups whose
    # threads walked 64 pixels one dependent load after the other: 27 us per launch)
    nchunk = max(1, min(hw // 8, (1024 + n - 1) // n))
    ws = _empty((n, nchunk, c, 2), torch.float32, dev)
    call("pmi_gn_stats", ptr(x), ptr(x1), c0, ptr(ws), n, hw, c, groups, nchunk, dt)
    call("pmi_gn_finalize", ptr(ws), nchunk, c, None, 0, 0, ptr(gamma), ptr(beta), ptr(film), film_ld, ptr(ca), ptr(cb), n, hw, groups, eps)
    return ca, cb


def group_norm_coeffs_train(x: torch.Tensor, gamma, beta, groups: int, dt: int, *, x1: Optional[torch.Tensor] = None,
                            film: Optional[torch.Tensor] = None, film_ld: int = 0, eps: float = 1e-5):
    """As group_norm_coeffs, also returning the per-channel (sum, sumsq) partials the coefficients came from, as the argument tuple
    (s0, P0, C0, s1, P1, C1) of pmi_gn_finalize / pmi_gn_bwd_finalize: the backward recomputes the forward moments from them."""
    n, h, w, _ = x.shape
    c0 = logical_c(x, dt)
    c1 = logical_c(x1, dt) if x1 is not None else 0
    c, hw, dev = c0 + c1, h * w, x.device
    ca, cb = _empty((n, c), torch.float32, dev), _empty((n, c), torch.float32, dev)
    st0 = getattr(x, "_pmi_stats", None)
    st1 = getattr(x1, "_pmi_stats", None) if x1 is not None else None
    if st0 is not None and (x1 is None or st1 is not None):
        parts = (st0[0], st0[1], c0, st1[0] if st1 else None, st1[1] if st1 else 0, c1)
    else:
        nchunk = max(1, min(hw // 8, (1024 + n - 1) // n))
        ws = _empty((n, nchunk, c, 2), torch.float32, dev)
        call("pmi_gn_stats", ptr(x), ptr(x1), c0, ptr(ws), n, hw, c, groups, nchunk, dt)
        parts = (ws, nchunk, c, None, 0, 0)
    call("pmi_gn_finalize", ptr(parts[0]), parts[1], parts[2], ptr(parts[3]), parts[4], parts[5], ptr(gamma), ptr(beta), ptr(film), film_ld,
         ptr(ca), ptr(cb), n, hw, groups, eps)
    return ca, cb, parts


def group_norm_backward(x: torch.Tensor, dy: torch.Tensor, ca, cb, parts, gamma, groups: int, dt: int, *, x1: Optional[torch.Tensor] = None,
                        film: Optional[torch.Tensor] = None, film_ld: int = 0, act: int = ACT_NONE, gadd0: Optional[torch.Tensor] = None,
                        gadd1: Optional[torch.Tensor] = None, eps: float = 1e-5):
    """Gradient wrt x (and x1) of y = act(GroupNorm(cat(x, x1)) * gamma [FiLM] + beta) given dy = d loss / d y [N, H, W, C] (one tensor over the
    concat) -- pmi_gn_bwd_stats / _finalize / _apply; gadd0 / gadd1: gradients arriving over another path, added on the way out.
    Returns (dx, dx1)."""
    n, h, w, c0 = x.shape
    c1 = x1.shape[-1] if x1 is not None else 0
    c, hw, dev = c0 + c1, h * w, x.device
    assert dy.shape[-1] == c and dy.is_contiguous() and x.is_contiguous() and (x1 is None or x1.is_contiguous())
    nchunk = max(1, min(hw // 8, (1024 + n - 1) // n))
    wsb = _empty((n, nchunk, c, 2), torch.float32, dev)
    call("pmi_gn_bwd_stats", ptr(x), ptr(x1), c0, ptr(dy), ptr(ca), ptr(cb), act, ptr(wsb), n, hw, c, nchunk, dt)
    cp, cq = _empty((n, c), torch.float32, dev), _empty((n, c), torch.float32, dev)
    call("pmi_gn_bwd_finalize", ptr(parts[0]), parts[1], parts[2], ptr(parts[3]), parts[4], parts[5], ptr(wsb), nchunk, ptr(gamma), ptr(film), film_ld,
         ptr(cp), ptr(cq), n, hw, groups, eps)
    dx0 = torch.empty_like(x)
    dx1 = torch.empty_like(x1) if x1 is not None else None
    call("pmi_gn_bwd_apply", ptr(x), ptr(x1), c0, ptr(dy), ptr(ca), ptr(cb), ptr(cp), ptr(cq), act, ptr(gadd0), ptr(gadd1), ptr(dx0), ptr(dx1),
         n, hw, c, dt)
    return dx0, dx1


def group_norm_coeffs(x: torch.Tensor, gamma, beta, groups: int, dt: int, *, x1: Optional[torch.Tensor] = None,
                      film: Optional[torch.Tensor] = None, film_ld: int = 0, eps: float = 1e-5):
    """Per-(sample, channel) coefficients (a, b) with norm(x)*gamma+beta[FiLM] = x*a+b (no apply pass)."""
    return _gn_coeffs(x, x1, gamma, beta, groups, dt, film, film_ld, eps)


def group_norm(x: torch.Tensor, gamma, beta, groups: int, dt: int, *, x1: Optional[torch.Tensor] = None,
               film: Optional[torch.Tensor] = None, film_ld: int = 0, residual: Optional[torch.Tensor] = None,
               act: int = ACT_NONE, pool: bool = False, eps: float = 1e-5) -> torch.Tensor:
    """GroupNorm over the channel-concat of x (and x1) -> act(norm * gamma + beta [FiLM]) [-> 2x2 avg pool] [+ residual]."""
    n, h, w, _ = x.shape
    c0 = logical_c(x, dt)
    c = c0 + (logical_c(x1, dt) if x1 is not None else 0)
    ca, cb = _gn_coeffs(x, x1, gamma, beta, groups, dt, film, film_ld, eps)
    cphys = 2 * c if dt == DT_F16X2 else c
    y = _empty((n, h // 2, w // 2, cphys) if pool else (n, h, w, cphys), x.dtype, x.device)
    call("pmi_gn_apply", ptr(x), ptr(x1), c0, ptr(ca), ptr(cb), ptr(residual), ptr(y), n, h, w, c, act, int(pool), dt)
    return y


def group_norm_pool_skip(x: torch.Tensor, gamma, beta, groups: int, dt: int, act: int = ACT_NONE, eps: float = 1e-5):
    """(AvgPool2d(2)(act(GroupNorm(x))), AvgPool2d(2)(x)) from one pass over x: both inputs of a down ResBlock (unet.py:232-243)."""
    n, h, w, _ = x.shape
    c = logical_c(x, dt)
    ca, cb = _gn_coeffs(x, None, gamma, beta, groups, dt, None, 0, eps)
    y = _empty((n, h // 2, w // 2, x.shape[-1]), x.dtype, x.device)
    y_raw = _empty((n, h // 2, w // 2, x.shape[-1]), x.dtype, x.device)
    call("pmi_gn_apply_pool_skip", ptr(x), ptr(ca), ptr(cb), ptr(y), ptr(y_raw), n, h, w, c, act, dt)
    return y, y_raw


def avgpool2(x: torch.Tensor, dt: int) -> torch.Tensor:
    n, h, w, c = x.shape
    y = _empty((n, h // 2, w // 2, c), x.dtype, x.device)
    call("pmi_avgpool2", ptr(x), ptr(y), n, h, w, logical_c(x, dt), dt)
    return y


def upsample_bilinear2(x: torch.Tensor, dt: int) -> torch.Tensor:
    n, h, w, c = x.shape
    y = _empty((n, h * 2, w * 2, c), x.dtype, x.device)
    call("pmi_upsample_bilinear2", ptr(x), ptr(y), n, h, w, logical_c(x, dt), dt)
    return y


def upsample_nearest2(x: torch.Tensor) -> torch.Tensor:
    n, h, w, c = x.shape
    y = _empty((n, h * 2, w * 2, c), x.dtype, x.device)
    call("pmi_upsample_nearest2", ptr(x), ptr(y), n, h, w, c)
    return y


def attention(qkv: torch.Tensor, heads: int, order: int, dt: int, causal: bool = False) -> torch.Tensor:
    """Self-attention over tokens.  qkv: [N, T, 3C] 16-bit -> [N, T, C].
    causal: query i sees keys 0..i (the CLIP text tower's mask, ruclip/model.py:181-185); runs the batched-GEMM path.

    order 0: channels = (head, {q,k,v}, d) (unet.py:332-348); order 1: ({q,k,v}, head, d).
    Head dim 64 runs the fused flash kernel; other head dims use batched MFMA GEMMs + softmax.
    """
    if dt == DT_F16X2:
        if causal:
            raise NotImplementedError("causal attention has no precise-mode path")
        return attention_precise(qkv, heads, order)
    n, t, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    dev = qkv.device
    out = _empty((n, t, c), qkv.dtype, dev)
    scale = float(d) ** -0.5
    if d == 64 and not causal:
        tp = (t + 31) // 32 * 32
        q = _empty((n * heads, tp, 64), qkv.dtype, dev)
        k = _empty((n * heads, tp, 64), qkv.dtype, dev)
        vt = _empty((n * heads, 64, tp), qkv.dtype, dev)
        call("pmi_qkv_split", ptr(qkv), ptr(q), ptr(k), ptr(vt), n, t, heads, order, dt)
        call("pmi_attn_d64", ptr(q), ptr(k), ptr(vt), ptr(out), n, t, heads, scale, dt)
        return out
    assert d % 8 == 0, "head dim must be a multiple of 8"
    if FLASH_ENABLED and order == 1 and not causal and d <= 160:
        return flash_attention(qkv, qkv[..., c:], qkv[..., 2 * c:], heads, d, dt)
    tp = (t + 7) // 8 * 8
    if order == 0:
        qo, ko, vo, hs = 0, d, 2 * d, 3 * d
    else:
        qo, ko, vo, hs = 0, c, 2 * c, d
    s = _empty((n * heads, t, tp), torch.float32, dev)
    bgemm(qkv, qkv, s, M=t, N=t, K=d, lda=c3, ldb=c3, ldd=tp, batch=n * heads, batch_inner=heads,
          sA=(t * c3, hs), sB=(t * c3, hs), sD=(heads * t * tp, t * tp), dt=dt, a_off=qo, b_off=ko)
    p = _empty((n * heads, t, tp), qkv.dtype, dev)
    call("pmi_softmax_causal_fwd" if causal else "pmi_softmax_fwd", ptr(s), ptr(p), n * heads * t, t, tp, tp, scale, dt)
    vt = _empty((n * heads, d, tp), qkv.dtype, dev)
    call("pmi_transpose_16", qkv.data_ptr() + vo * qkv.element_size(), ptr(vt), t, d, c3, t * c3, hs, heads, n * heads)
    bgemm(p, vt, out, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=c, batch=n * heads, batch_inner=heads,
          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * c, d), dt=dt)
    return out


def flash_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, d: int, dt: int) -> torch.Tensor:
    """softmax(q k^T d^-1/2) v through pmi_attn_flash.  q [N, T, >= heads*d] and k, v [N, Tk, >= heads*d] are (views of) 16-bit tensors
    whose last-dim stride is 1 and whose head h sits at channels [h*d, (h+1)*d) of the view; -> [N, T, heads*d]."""
    n, t = q.shape[:2]
    tk = k.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1 and k.stride(1) == v.stride(1)
    assert q.stride(0) == t * q.stride(1) and k.stride(0) == tk * k.stride(1)
    kib = _hip.lib().pmi_attn_flash_workspace(n, t, tk, heads, d)
    if kib < 0:
        raise ValueError(f"flash attention: unsupported head dim {d}")
    ws = _empty((kib * 512,), q.dtype, q.device)
    out = _empty((n, t, heads * d), q.dtype, q.device)
    call("pmi_attn_flash", ptr(q), q.stride(1), ptr(k), ptr(v), k.stride(1), ptr(out), ptr(ws), n, t, tk, heads, d, float(d) ** -0.5, dt)
    return out


def cross_attention(q: torch.Tensor, kv: torch.Tensor, heads: int, dt: int) -> torch.Tensor:
    """softmax(q k^T d^-1/2) v with keys / values from another sequence (stable_diffusion/attention.py:268-298, the fused call at :285).
    q [N, T, C], kv [N, Tc, 2C] = (k | v) x (head, d), 16-bit -> [N, T, C].  Batched MFMA GEMMs + fp32 softmax (Tc = 77 prompt tokens)."""
    n, t, c = q.shape
    tc = kv.shape[1]
    assert kv.shape[0] == n and kv.shape[2] == 2 * c
    d = c // heads
    assert d % 8 == 0, "head dim must be a multiple of 8"
    if FLASH_ENABLED and d <= 160:
        return flash_attention(q, kv, kv[..., c:], heads, d, dt)
    tcp = (tc + 7) // 8 * 8
    dev = q.device
    s = _empty((n * heads, t, tcp), torch.float32, dev)
    bgemm(q, kv, s, M=t, N=tc, K=d, lda=c, ldb=2 * c, ldd=tcp, batch=n * heads, batch_inner=heads,
          sA=(t * c, d), sB=(tc * 2 * c, d), sD=(heads * t * tcp, t * tcp), dt=dt)
    p = _empty((n * heads, t, tcp), q.dtype, dev)
    call("pmi_softmax_fwd", ptr(s), ptr(p), n * heads * t, tc, tcp, tcp, float(d) ** -0.5, dt)
    vt = _transpose16(kv, c, tc, d, 2 * c, tc * 2 * c, d, heads, n * heads)
    out = _empty((n, t, c), q.dtype, dev)
    bgemm(p, vt, out, M=t, N=d, K=tcp, lda=tcp, ldb=tcp, ldd=c, batch=n * heads, batch_inner=heads,
          sA=(heads * t * tcp, t * tcp), sB=(heads * d * tcp, d * tcp), sD=(t * c, d), dt=dt)
    return out


def _transpose16(src: torch.Tensor, off: int, rows: int, cols: int, ld: int, s_o: int, s_i: int, inner: int, batch: int) -> torch.Tensor:
    rp = (rows + 7) // 8 * 8
    out = _empty((batch, cols, rp), src.dtype, src.device)
    call("pmi_transpose_16", src.data_ptr() + off * src.element_size(), ptr(out), rows, cols, ld, s_o, s_i, inner, batch)
    return out


def attention_train(qkv: torch.Tensor, heads: int, dt: int):
    """Self-attention for any head dim, channels (q|k|v, head, d), keeping the softmax for attention_backward:
    qkv [N, T, 3C] 16-bit -> (out [N, T, C], P [N*heads, T, Tp] 16-bit).  Batched MFMA GEMMs + softmax (as `attention`, order 1)."""
    n, t, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    tp = (t + 7) // 8 * 8
    dev = qkv.device
    sc = _empty((n * heads, t, tp), torch.float32, dev)
    bgemm(qkv, qkv, sc, M=t, N=t, K=d, lda=c3, ldb=c3, ldd=tp, batch=n * heads, batch_inner=heads,
          sA=(t * c3, d), sB=(t * c3, d), sD=(heads * t * tp, t * tp), dt=dt, b_off=c)
    p = _empty((n * heads, t, tp), qkv.dtype, dev)
    call("pmi_softmax_fwd", ptr(sc), ptr(p), n * heads * t, t, tp, tp, float(d) ** -0.5, dt)
    vt = _transpose16(qkv, 2 * c, t, d, c3, t * c3, d, heads, n * heads)
    out = _empty((n, t, c), qkv.dtype, dev)
    bgemm(p, vt, out, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=c, batch=n * heads, batch_inner=heads,
          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * c, d), dt=dt)
    return out, p


def attention_backward(qkv: torch.Tensor, p: torch.Tensor, d_out: torch.Tensor, heads: int, dt: int) -> torch.Tensor:
    """d loss / d qkv [N, T, 3C] from d loss / d out [N, T, C] and the saved softmax P (the same five products autograd forms:
    dP = dO V^T, dS = softmax'(P, dP), dV = P^T dO, dQ = dS K, dK = dS^T Q, scale folded into softmax_bwd)."""
    n, t, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    tp = (t + 7) // 8 * 8
    dev = qkv.device
    da = d_out.reshape(n * t, c)
    dqkv = _empty((n, t, c3), qkv.dtype, dev)
    dp = _empty((n * heads, t, tp), torch.float32, dev)
    bgemm(da, qkv, dp, M=t, N=t, K=d, lda=c, ldb=c3, ldd=tp, batch=n * heads, batch_inner=heads,
          sA=(t * c, d), sB=(t * c3, d), sD=(heads * t * tp, t * tp), dt=dt, b_off=2 * c)
    ds = _empty((n * heads, t, tp), qkv.dtype, dev)
    call("pmi_softmax_bwd", ptr(dp), ptr(p), ptr(ds), n * heads * t, t, tp, tp, float(d) ** -0.5, dt)
    pt = _transpose16(p, 0, t, t, tp, heads * t * tp, t * tp, heads, n * heads)
    dot = _transpose16(da, 0, t, d, c, t * c, d, heads, n * heads)
    bgemm(pt, dot, dqkv, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=c3, batch=n * heads, batch_inner=heads,
          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * c3, d), dt=dt, d_off=2 * c)        # dV
    kt = _transpose16(qkv, c, t, d, c3, t * c3, d, heads, n * heads)
    bgemm(ds, kt, dqkv, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=c3, batch=n * heads, batch_inner=heads,
          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * c3, d), dt=dt)                     # dQ
    dst = _transpose16(ds, 0, t, t, tp, heads * t * tp, t * tp, heads, n * heads)
    qt = _transpose16(qkv, 0, t, d, c3, t * c3, d, heads, n * heads)
    bgemm(dst, qt, dqkv, M=t, N=d, K=tp, lda=tp, ldb=tp, ldd=c3, batch=n * heads, batch_inner=heads,
          sA=(heads * t * tp, t * tp), sB=(heads * d * tp, d * tp), sD=(t * c3, d), dt=dt, d_off=c)            # dK
    return dqkv


def gemm_f32(A: torch.Tensor, B: torch.Tensor, D: torch.Tensor, *, M: int, N: int, K: int, lda: int, ldb: int, ldd: int, trans_b: bool = False,
             bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, alpha: float = 1.0, batch: int = 1, batch_inner: int = 1,
             sA=(0, 0), sB=(0, 0), sD=(0, 0), a_off: int = 0, b_off: int = 0, d_off: int = 0, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Exact-fp32 batched GEMM on the f32-input MFMA (csrc/f32gemm.hip): D[z] = act(alpha * A[z] @ B[z]^T + bias)."""
    a = _hip.GemmF32Args()
    a.A, a.B, a.bias, a.D = A.data_ptr() + 4 * a_off, B.data_ptr() + 4 * b_off, ptr(bias), D.data_ptr() + 4 * d_off
    a.M, a.N, a.K, a.lda, a.ldb, a.ldd = M, N, K, lda, ldb, ldd
    a.transB, a.act, a.alpha, a.batch, a.batch_inner = int(trans_b), act, alpha, batch, batch_inner
    a.sA_o, a.sA_i = sA
    a.sB_o, a.sB_i = sB
    a.sD_o, a.sD_i = sD
    a.R = ptr(residual)
    call("pmi_gemm_f32", C.byref(a))
    return D


def linear_f32(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE,
               residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 linear layer x [M, K] @ weight [N, K]^T + bias (the time MLPs in precise mode)."""
    m, k = x.shape
    n = weight.shape[0]
    out = _empty((m, n), torch.float32, x.device)
    return gemm_f32(x, weight, out, M=m, N=n, K=k, lda=x.stride(0), ldb=weight.stride(0), ldd=n, bias=bias, act=act, residual=residual)


def attention_precise(qkv: torch.Tensor, heads: int, order: int) -> torch.Tensor:
    """Self-attention in precise mode: qkv is a precise tensor [N, T, 2*3C]; scores, softmax and values in exact fp32
    (both operands of these products are activations, so the hi + lo weight trick does not apply); returns a precise [N, T, 2C]."""
    n, t, c6 = qkv.shape
    c3 = c6 // 2
    c = c3 // 3
    d = c // heads
    dev = qkv.device
    q32 = _empty((n, t, c3), torch.float32, dev)
    call("pmi_split_to_f32", ptr(qkv), ptr(q32), n * t, c3)
    if order == 0:
        qo, ko, vo, hs = 0, d, 2 * d, 3 * d
    else:
        qo, ko, vo, hs = 0, c, 2 * c, d
    s = _empty((n * heads, t, t), torch.float32, dev)
    gemm_f32(q32, q32, s, M=t, N=t, K=d, lda=c3, ldb=c3, ldd=t, batch=n * heads, batch_inner=heads,
             sA=(t * c3, hs), sB=(t * c3, hs), sD=(heads * t * t, t * t), a_off=qo, b_off=ko)
    call("pmi_softmax_f32", ptr(s), n * heads * t, t, t, float(d) ** -0.5)
    o32 = _empty((n, t, c), torch.float32, dev)
    gemm_f32(s, q32, o32, M=t, N=d, K=t, lda=t, ldb=c3, ldd=c, trans_b=True, batch=n * heads, batch_inner=heads,
             sA=(heads * t * t, t * t), sB=(t * c3, hs), sD=(t * c, d), b_off=vo)
    out = _empty((n, t, 2 * c), torch.float16, dev)
    call("pmi_split_from_f32", ptr(o32), c, ptr(out), n * t, c)
    return out
