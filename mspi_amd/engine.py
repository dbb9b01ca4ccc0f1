"""Host-side plumbing between the reference-shaped nn.Modules and the C ABI.

* `CL`   -- a channels-last activation: M = N*T*H*W rows x C columns, row stride `ld`
            (floats, multiple of 4) inside a flat fp32 torch buffer.  Slicing channels is free,
            which is how concats are eliminated (producers write into their slice).
* pack_* -- weight packing done once per parameter version: eval-mode BatchNorm folded into
            the conv, taps made channel-minor, channel counts padded to a multiple of 4.
* op wrappers -- conv / dwconv / layernorm / ... : fill the POD descriptor, pass raw device
            pointers and torch's current hipStream_t.  No CPU fallback anywhere.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH, PREC_F16X3, PREC_F32, AttnDesc, ConvDesc,
                   DwConvDesc, MspiError, MvitAugDesc, check)

# GEMM arithmetic for every dense conv / Linear: "f16x3" (default; fp32-accurate split product on the f16 matrix
# pipe, see include/mspi_hip.h) or "f32" (v_mfma_f32_32x32x2_f32).  Read when weights are packed.
import os as _os
DEFAULT_PREC = {"f32": PREC_F32, "f16x3": PREC_F16X3}[_os.environ.get("MSPI_GEMM_PREC", "f16x3")]

__all__ = ["CL", "SP", "alloc", "alloc_sp", "pack_conv", "pack_dwconv", "PackedConv", "PackedDw", "conv", "dwconv", "maxpool",
           "layernorm", "attention", "upsample", "rowgate", "logsumexp_sub", "mean_rows", "neg_cosine",
           "se_gate", "add", "fold_bn", "ACT_NONE", "ACT_RELU", "ACT_GELU", "ACT_SIGMOID", "ACT_SWISH"]


def rup4(c):
    return (c + 3) // 4 * 4


# ----------------------------------------------------------------------------- conv autotuning
# Like cudnn.benchmark (which the reference switches on, inference.py:189): the first time a conv shape is seen
# with autotuning enabled, every kernel instantiation that applies is timed on the real tensors and the fastest is
# remembered.  Off by default (tile -1 = the library's heuristic); bench.py / inference enable it before the
# hipGraph is captured.  The cache is keyed by the GEMM shape, so it is shared by all layers with that shape.
AUTOTUNE = {"on": _os.environ.get("MSPI_AUTOTUNE", "0") == "1", "cache": {}, "reps": 3}


def autotune(on=True):
    AUTOTUNE["on"] = bool(on)


def save_autotune(path):
    """Write the tile choices made so far (MIOpen's find-db role): {repr(shape key): tile code}."""
    import json
    with open(path, "w") as f:
        json.dump({repr(k): v for k, v in AUTOTUNE["cache"].items()}, f, indent=0, sort_keys=True)


def load_autotune(path):
    """Adopt tile choices saved by save_autotune; shapes not in the file fall back to the library heuristic
    (or are tuned, when autotune is on).  Returns the number of entries."""
    import ast
    import json
    with open(path) as f:
        AUTOTUNE["cache"].update({ast.literal_eval(k): int(v) for k, v in json.load(f).items()})
    return len(AUTOTUNE["cache"])


THIN = 100            # kernel choice "row-stationary thin GEMM" next to MspiConvDesc.tile codes 0..14
SPLITK = 200          # kernel choice SPLITK + S: split-K with S slices (mspi_conv_splitk_fwd)
SPLITK_ENABLED = _os.environ.get("MSPI_SPLITK", "1") != "0"   # A/B switch
THIN_DEFAULT = True   # without autotuning: take the thin kernel wherever it applies
THIN_ENABLED = _os.environ.get("MSPI_THIN", "1") != "0"   # A/B switch


_TUNE_LOG = _os.environ.get("MSPI_TUNE_LOG") == "1"


def _tune_conv(launch, key, candidates):
    best, best_t, times = -1, float("inf"), []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for tile in candidates:
        if launch(tile) != 0:
            continue
        e0.record()
        for _ in range(AUTOTUNE["reps"]):
            launch(tile)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        times.append((tile, t / AUTOTUNE["reps"]))
        if t < best_t:
            best, best_t = tile, t
    AUTOTUNE["cache"][key] = best
    if _TUNE_LOG:       # MSPI_TUNE_LOG=1: every candidate's time (ms per launch, alone on the chip) to stderr
        import sys
        print("[tune] %s -> %d  %s" % (key, best, " ".join("%d:%.4f" % tt for tt in times)), file=sys.stderr)
    return best


# ----------------------------------------------------------------------------- per-launch timing
class Profiler:
    """Per-launch HIP-event timing of the C-ABI calls, on the stream the kernels are launched on
    (torch's current stream).  Used by bench.py for the roofline line; off by default."""
    active = None

    def __init__(self):
        self.records = []   # (kernel_name, flops, bytes, start_event, end_event, detail)

    def __enter__(self):
        Profiler.active = self
        return self

    def __exit__(self, *a):
        Profiler.active = None

    def summary(self):
        """{kernel: dict(calls, ms, flops, bytes)} -- call after torch.cuda.synchronize()."""
        out = {}
        for name, fl, by, e0, e1, _ in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


class _Timed:
    __slots__ = ("name", "flops", "bytes", "e0", "detail")

    def __init__(self, name, flops=0.0, nbytes=0.0, detail=""):
        self.name, self.flops, self.bytes, self.detail = name, flops, nbytes, detail

    def __enter__(self):
        if Profiler.active is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        p = Profiler.active
        if p is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            p.records.append((self.name, self.flops, self.bytes, self.e0, e1, self.detail))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t):
    if not t.is_cuda:
        raise MspiError("mspi_amd runs on the GPU only (tensor on %s); there is no CPU fallback" % t.device)
    if _STATUS["word"] is None:
        _register_status_word()


# ----------------------------------------------------------------------------- f16x3 operand range
# f16x3 splits an fp32 operand into f16 hi + lo halves.  Weights are pre-scaled by a power of two at pack time; activations
# are split as they are, which is exact to 2^-22 relative while 2^-5 <~ |x| < 65504 for the LARGEST entries of the tensor:
# beyond 65504 the hi half is inf; far below, the lo half sinks into f16 subnormals (absolute error 2^-25 per element,
# whatever its size).  Two safeguards:
#  * RANGE GUARD (always on): the GEMM kernels store 1 into a pinned host word when a result is inf / NaN
#    (mspi_set_status_word); range_flag() / check_range() read it -- no device call, the caller synchronises first.
#  * RANGE CHECK on first sight of a pack (the first, autotuning forward -- the same "first input is representative"
#    contract as cudnn.benchmark upstream, inference.py:19): max|x| of the layer's input outside [2^-5, 2^15] moves THAT layer
#    to the fp32 MFMA path (exact fp32 fmaf chain, 5.3x the MFMA time) for good.  MSPI_RANGE_CHECK=0 disables it.
_STATUS = {"word": None}
RANGE_CHECK = {"on": _os.environ.get("MSPI_RANGE_CHECK", "1") != "0", "lo": 2.0 ** -5, "hi": 2.0 ** 15, "moved": []}


def _register_status_word():
    w = torch.zeros(1, dtype=torch.int32).pin_memory()
    check(_lib.load().mspi_set_status_word(w.data_ptr()), "mspi_set_status_word")
    _STATUS["word"] = w


def range_flag(reset=True):
    """True when a GEMM kernel has produced a non-finite result since the last reset.  Host read of a pinned word: valid for
    launches the caller has synchronised with (an event / stream / device sync)."""
    w = _STATUS["word"]
    if w is None:
        return False
    bad = bool(int(w[0]))
    if bad and reset:
        w[0] = 0
    return bad


def check_range(sync=True):
    """Raise MspiError when an f16x3 GEMM has overflowed (or was fed non-finite data) since the last check."""
    if sync and torch.cuda.is_available():
        torch.cuda.synchronize()
    if range_flag():
        raise MspiError("a GEMM produced inf/NaN: an activation left the f16x3 range (|x| >= 65504) or the input was not finite; "
                        "let the first forward see representative data (engine.autotune(True): out-of-range layers move to the "
                        "fp32 path) or run with MSPI_GEMM_PREC=f32")


def _range_check(pk, amax_fn, what):
    """First sight of a pack while tuning: move it to the fp32 path if its input's magnitude is outside the f16x3 window."""
    if pk.checked:
        return
    pk.checked = True
    amax = float(amax_fn())
    if not (amax == amax) or amax >= RANGE_CHECK["hi"] or 0.0 < amax < RANGE_CHECK["lo"]:
        pk.w, pk.ldw, pk.prec, pk.thin, pk.w_scale = pk.w32, pk.ldw32, PREC_F32, None, 1.0
        RANGE_CHECK["moved"].append((what, amax))


class CL:
    """Channels-last activation view: rows (n,t,h,w) x C stored channels, row stride ld."""
    __slots__ = ("buf", "off", "N", "T", "H", "W", "C", "ld", "sN")

    def __init__(self, buf, off, N, T, H, W, Cc, ld, sN=None):
        self.buf, self.off, self.N, self.T, self.H, self.W, self.C, self.ld = buf, off, N, T, H, W, Cc, ld
        self.sN = T * H * W * ld if sN is None else sN  # sample stride (floats); dense unless a token slab

    @property
    def dense(self):
        return self.sN == self.T * self.H * self.W * self.ld

    @property
    def Cs(self):
        """stored channels"""
        return rup4(self.C) if self.C > 1 else 1

    @property
    def M(self):
        return self.N * self.T * self.H * self.W

    @property
    def ptr(self):
        return self.buf.data_ptr() + 4 * self.off

    def slice(self, c0, c):
        assert c0 % 4 == 0 and c0 + c <= self.ld
        return CL(self.buf, self.off + c0, self.N, self.T, self.H, self.W, c, self.ld, self.sN)

    def reshape(self, N, T, H, W):
        assert N * T * H * W == self.M and self.dense
        return CL(self.buf, self.off, N, T, H, W, self.C, self.ld)

    def tokens(self, r0, T, H, W):
        """Rows [r0, r0+T*H*W) of every sample as a [N,T,H,W,C] view (sample stride kept)."""
        assert r0 + T * H * W <= self.T * self.H * self.W
        return CL(self.buf, self.off + r0 * self.ld, self.N, T, H, W, self.C, self.ld, self.sN)

    def as_ncdhw(self, channels=None):
        """Logical [N,C,T,H,W] view (no copy) -- what the reference's modules return."""
        c = self.C if channels is None else channels
        ld, T, H, W = self.ld, self.T, self.H, self.W
        return self.buf.as_strided((self.N, c, T, H, W), (self.sN, 1, H * W * ld, W * ld, ld),
                                   self.buf.storage_offset() + self.off)

    def as_rows(self, channels=None):
        c = self.C if channels is None else channels
        assert self.dense
        return self.buf.as_strided((self.M, c), (self.ld, 1), self.buf.storage_offset() + self.off)


def alloc(N, T, H, W, Cc, device, ld=None):
    if ld is None:
        ld = rup4(Cc) if Cc > 1 else 1
    buf = torch.empty(N * T * H * W * ld, dtype=torch.float32, device=device)
    return CL(buf, 0, N, T, H, W, Cc, ld)


SP_ENABLED = _os.environ.get("MSPI_PRESPLIT", "1") != "0"   # A/B switch: pre-split activations between LN / GEMM / GEMM


class SP:
    """Pre-split activation rows: two f16 planes (hi, lo) in one buffer -- what a producer's epilogue hands to the f16x3 GEMM so
    that neither operand needs conversion work in the loop (include/mspi_hip.h, mspi_gemm_sp_fwd).  Each plane is BLOCKED:
    16 rows x 32 columns = 1 KB contiguous per block, blocks column-fastest, rows allocated to a multiple of 16 (ld == C).
    Same logical shape as a dense CL; only GEMM-shaped consumers (conv on 1x1x1 / Linear packs) accept it."""
    __slots__ = ("buf", "N", "T", "H", "W", "C", "ld")

    def __init__(self, buf, N, T, H, W, Cc, ld):
        self.buf, self.N, self.T, self.H, self.W, self.C, self.ld = buf, N, T, H, W, Cc, ld

    @property
    def M(self):
        return self.N * self.T * self.H * self.W

    @property
    def plane(self):
        return (self.M + 15) // 16 * 16 * self.ld      # blocked planes: rows allocated to a multiple of 16

    @property
    def ptr(self):
        return self.buf.data_ptr()


def sp_supported(c):
    """Channel counts the pre-split GEMM takes as its K: multiples of the 32-deep stage (then ldw == K)."""
    return SP_ENABLED and DEFAULT_PREC == PREC_F16X3 and c % 32 == 0


def alloc_sp(N, T, H, W, Cc, device):
    assert Cc % 32 == 0
    m = N * T * H * W
    mp = (m + 15) // 16 * 16
    buf = torch.empty(2 * mp * Cc, dtype=torch.float16, device=device)
    if mp != m:      # the rows that pad the last 16-row group are read by the GEMM (their outputs are never stored) and by the range check
        buf.view(2, mp * Cc)[:, (mp - 16) * Cc:].zero_()
    return SP(buf, N, T, H, W, Cc, Cc)


def join_planes(sp):
    """SP -> dense CL of fp32 rows (hi + lo)."""
    out = alloc(sp.N, sp.T, sp.H, sp.W, sp.C, sp.buf.device)
    check(_lib.load().mspi_join_planes_fwd(sp.ptr, sp.ld, sp.plane, sp.M, sp.C, out.ptr, out.ld, _stream()), "mspi_join_planes_fwd")
    return out


def from_rows(t2d):
    """Wrap a contiguous [M, C] tensor (C % 4 == 0) as a CL with N=M, T=H=W=1."""
    assert t2d.dim() == 2 and t2d.stride(1) == 1 and t2d.stride(0) % 4 == 0
    return CL(t2d, 0, t2d.shape[0], 1, 1, 1, t2d.shape[1], t2d.stride(0))


# ----------------------------------------------------------------------------- packing
def fold_bn(weight, bias, bn):
    """Fold an eval-mode BatchNorm (running stats) into the preceding conv: returns (w, b)."""
    w = weight.detach().float()
    b = None if bias is None else bias.detach().float()
    if bn is not None:
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * s.view(-1, *([1] * (w.dim() - 1)))
        b0 = bn.bias.detach().float() - bn.running_mean.detach().float() * s
        b = b0 if b is None else b0 + b * s
    return w, b


class PackedConv:
    __slots__ = ("w", "bias", "k", "stride", "pad", "cin", "cin_s", "cout", "cout_s", "ldw", "act", "prec", "w_scale", "thin",
                 "w32", "ldw32", "checked", "wsp")


def pack_conv(weight, bias=None, bn=None, stride=(1, 1, 1), pad=(0, 0, 0), act=ACT_NONE, cin_stored=None,
              out_scale=None, device=None, prec=None):
    """weight: [Co,Ci] (Linear), [Co,Ci,kh,kw] (2-D) or [Co,Ci,kt,kh,kw].  Result rows are
    [Co_s][ldw] with k = (kt,kh,kw,ci) ci fastest, Ci padded to cin_stored, Co to a multiple of 4."""
    w, b = fold_bn(weight, bias, bn)
    if w.dim() == 2:
        w = w[:, :, None, None, None]
    elif w.dim() == 4:
        w = w[:, :, None]
    if out_scale is not None:  # e.g. ConvNeXt layer-scale gamma folded into the producing conv
        s = out_scale.detach().float().view(-1)
        w = w * s.view(-1, 1, 1, 1, 1)
        b = None if b is None else b * s
    co, ci, kt, kh, kw = w.shape
    cin_s = ci if cin_stored is None else cin_stored
    assert cin_s >= ci
    cout_s = rup4(co) if co > 1 else 1
    K = kt * kh * kw * cin_s
    prec = DEFAULT_PREC if prec is None else prec
    ldw = rup4(K) if prec == PREC_F32 else (K + 31) // 32 * 32
    wp = torch.zeros(cout_s, kt, kh, kw, cin_s, dtype=torch.float32, device=w.device)
    wp[:co, :, :, :, :ci] = w.permute(0, 2, 3, 4, 1)
    wf = torch.zeros(cout_s, ldw, dtype=torch.float32, device=w.device)
    wf[:, :K] = wp.reshape(cout_s, K)
    p = PackedConv()
    p.thin = None
    p.wsp = None
    p.checked = False
    dev = w.device if device is None else device
    p.prec, p.w_scale = prec, 1.0
    if prec == PREC_F16X3:
        # power-of-two pre-scale puts max|w| in [2^13, 2^14): the lo halves of typical weights are normal f16
        mx = float(wf.abs().max())
        if not math.isfinite(mx):
            raise MspiError("pack_conv: non-finite weights")
        e = 0 if mx == 0.0 else max(-10, min(24, int(math.floor(math.log2(16384.0 / mx)))))
        p.w_scale = float(2.0 ** e)
        ws = wf * p.w_scale
        hi = ws.to(torch.float16)
        lo = (ws - hi.float()).to(torch.float16)
        p.w = torch.stack([hi, lo]).to(dev).contiguous()
        p.ldw32 = rup4(K)                                   # the fp32 form, for layers the range check moves off f16x3
        p.w32 = wf[:, :p.ldw32].to(dev).contiguous()
        if (kt, kh, kw) == (1, 1, 1) and tuple(stride) == (1, 1, 1) and tuple(pad) == (0, 0, 0):
            p.thin = _pack_rowgemm(ws[:, :K], cin_s, cout_s, dev)
    else:
        p.w = wf.to(dev).contiguous()
        p.w32, p.ldw32 = p.w, ldw
    if b is None:
        p.bias = None
    else:
        bp = torch.zeros(cout_s, dtype=torch.float32, device=w.device)
        bp[:co] = b
        p.bias = bp.to(dev)
    p.k, p.stride, p.pad = (kt, kh, kw), tuple(stride), tuple(pad)
    p.cin, p.cin_s, p.cout, p.cout_s, p.ldw, p.act = ci, cin_s, co, cout_s, ldw, act
    return p


def rowgemm_ksb(k):
    """k-steps of 16 the row-stationary thin GEMM keeps in registers for K stored input columns (0: not covered)."""
    return 2 if k <= 32 else 4 if k <= 64 else 8 if k <= 128 else 14 if k <= 224 else 0


def rowgemm_supported(k, n):
    """Mirror of mspi_rowgemm_supported."""
    return bool(rowgemm_ksb(k)) and 4 <= n <= 1024


def _pack_rowgemm(ws, k_s, n_s, dev):
    """Scaled weights ws [n_s, k_s] -> fragment-order f16 hi/lo planes for mspi_rowgemm_fwd (layout: include/mspi_hip.h);
    None when the shape is outside the kernel's range."""
    if not rowgemm_supported(k_s, n_s):
        return None
    ksb, nch = rowgemm_ksb(k_s), (n_s + 31) // 32
    wp = torch.zeros(nch * 32, ksb * 16, dtype=torch.float32)
    wp[:n_s, :k_s] = ws.cpu()
    hi = wp.to(torch.float16)
    lo = (wp - hi.float()).to(torch.float16)
    planes = [pl.view(nch, 32, ksb, 2, 8).permute(0, 2, 3, 1, 4).reshape(nch, ksb, 64, 8) for pl in (hi, lo)]
    return torch.stack(planes, 2).contiguous().to(dev)      # [nch, ksb, plane, lane, 8]


class PackedMlp:
    __slots__ = ("w", "b1", "b2", "c", "hidden", "s1", "s2", "act", "src", "checked", "fallback")


def _f16_scale(w):
    mx = float(w.abs().max())
    if not math.isfinite(mx):
        raise MspiError("pack: non-finite weights")
    e = 0 if mx == 0.0 else max(-10, min(24, int(math.floor(math.log2(16384.0 / mx)))))
    return float(2.0 ** e)


def mlp_supported(c, hidden):
    """Shapes mspi_mlp_fwd covers (the rows stay in registers as MFMA fragments: C <= 192)."""
    if _os.environ.get("MSPI_MLP_FUSED", "1") == "0":   # A/B switch
        return False
    return DEFAULT_PREC == PREC_F16X3 and c in (96, 192) and hidden % 32 == 0 and hidden <= 1024


def pack_mlp(fc1_w, fc1_b, fc2_w, fc2_b, out_scale=None, act=ACT_GELU, device=None):
    """Fragment-order f16 hi/lo packing of a Linear(C, hidden) -> act -> Linear(hidden, C) pair for mspi_mlp_fwd
    (layout: include/mspi_hip.h).  out_scale (ConvNeXt layer-scale gamma) is folded into the second layer."""
    w1 = fc1_w.detach().float().cpu()
    w2 = fc2_w.detach().float().cpu()
    b2 = fc2_b.detach().float().cpu()
    if out_scale is not None:
        g = out_scale.detach().float().cpu().view(-1)
        w2, b2 = w2 * g[:, None], b2 * g
    hidden, c = w1.shape
    if w2.shape != (c, hidden) or c % 32 or hidden % 32:
        raise MspiError("pack_mlp: shapes %s / %s" % (tuple(w1.shape), tuple(w2.shape)))
    nch, ks, ct = hidden // 32, c // 16, c // 32
    p = PackedMlp()
    p.s1, p.s2 = _f16_scale(w1), _f16_scale(w2)

    def planes(w, s):
        ws = w * s
        hi = ws.to(torch.float16)
        return hi, (ws - hi.float()).to(torch.float16)

    parts1, parts2 = [], []
    for pl in planes(w1, p.s1):   # [hidden, C] -> [nch, n32, ks, g2, e8] -> [nch, ks, (g, n) = lane, e]
        parts1.append(pl.view(nch, 32, ks, 2, 8).permute(0, 2, 3, 1, 4).reshape(nch, ks, 64, 8))
    for pl in planes(w2, p.s2):   # [C, hidden] -> [ct, c32, nch, s2, eh2, g2, r4] -> [nch, s, ct, (g, c) = lane, (eh, r) = e]
        parts2.append(pl.view(ct, 32, nch, 2, 2, 2, 4).permute(2, 3, 0, 5, 1, 4, 6).reshape(nch, 2, ct, 64, 8))
    w1p = torch.stack(parts1, 2).reshape(nch, -1)      # [nch, ks, plane, 64, 8]
    w2p = torch.stack(parts2, 3).reshape(nch, -1)      # [nch, s, ct, plane, 64, 8]
    dev = fc1_w.device if device is None else device
    p.w = torch.cat([w1p, w2p], 1).contiguous().to(dev)
    p.b1 = fc1_b.detach().float().contiguous().to(dev)
    p.b2 = b2.contiguous().to(dev)
    p.c, p.hidden, p.act = c, hidden, act
    # first-sight range check (mlp()): the layer pair as two GEMM packs, built from these when the check needs them
    p.src, p.checked, p.fallback = (fc1_w, fc1_b, fc2_w, fc2_b, out_scale), False, None
    return p


def _tuning():
    return AUTOTUNE["on"] and RANGE_CHECK["on"] and DEFAULT_PREC == PREC_F16X3 and not torch.cuda.is_current_stream_capturing()


def range_check_input(pk, x):
    """First-sight range check of PackedConv `pk` on its input `x` (CL), for callers that bypass conv() with a fused kernel
    built from the pack's f16x3 planes (X3D a + b).  Returns True while the pack is on the f16x3 path."""
    if _tuning() and pk.prec == PREC_F16X3:
        _range_check(pk, lambda: (x.as_rows()[:, : x.C] if x.dense else x.buf).abs().max(), "conv %s %d -> %d" % (pk.k, pk.cin, pk.cout))
    return pk.prec == PREC_F16X3


class PackedDw:
    __slots__ = ("w", "bias", "k", "stride", "pad", "c", "c_s", "act")


def pack_dwconv(weight, bias=None, bn=None, stride=(1, 1, 1), pad=(0, 0, 0), act=ACT_NONE, device=None):
    """weight: [C,1,kt,kh,kw] or [C,1,kh,kw] depthwise.  Packed as [taps][C_s]."""
    w, b = fold_bn(weight, bias, bn)
    if w.dim() == 4:
        w = w[:, :, None]
    c, one, kt, kh, kw = w.shape
    assert one == 1
    c_s = rup4(c)
    wp = torch.zeros(kt * kh * kw, c_s, dtype=torch.float32, device=w.device)
    wp[:, :c] = w.reshape(c, kt * kh * kw).t()
    bp = torch.zeros(c_s, dtype=torch.float32, device=w.device)
    if b is not None:
        bp[:c] = b
    p = PackedDw()
    dev = w.device if device is None else device
    p.w, p.bias = wp.to(dev).contiguous(), bp.to(dev)
    p.k, p.stride, p.pad, p.c, p.c_s, p.act = (kt, kh, kw), tuple(stride), tuple(pad), c, c_s, act
    return p


class PackedX3dAb:
    __slots__ = ("wa", "ba", "wb", "bb", "wa_scale", "cin_s", "cmid", "cmid_s")


def pack_x3d_ab(pa, pb):
    """Operands of the fused X3D `a` + `b` kernel (mspi_x3d_ab_fwd) from the packed 1x1x1 conv `pa` (PackedConv, f16x3: BN
    folded, scaled hi/lo planes) and the packed 3x3x3 depthwise conv `pb` (PackedDw); None when the layer pair is outside
    the kernel's range.  Fragment order of wa: include/mspi_hip.h."""
    if pa.prec != PREC_F16X3 or pa.k != (1, 1, 1) or pa.stride != (1, 1, 1) or pb.k != (3, 3, 3) or pb.stride != (1, 1, 1) \
            or pb.pad != (1, 1, 1) or pa.cout_s != pb.c_s or pa.cin_s % 8 or pa.act != ACT_RELU or pa.bias is None:
        return None
    ks = (pa.cin_s + 31) // 32
    if ks not in (1, 2, 3, 6):
        return None
    nch = (pa.cout_s + 31) // 32
    dev = pa.w.device
    ws = torch.zeros(nch * 32, ks * 32, dtype=torch.float32)
    ws[:pa.cout_s, :pa.cin_s] = (pa.w[0].float() + pa.w[1].float()).cpu()[:, :pa.cin_s]     # hi + lo = the scaled fp32 weight's 22 bits
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    planes = [pl.view(nch, 2, 16, ks, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(nch, ks, 2, 64, 8) for pl in (hi, lo)]
    p = PackedX3dAb()
    p.wa = torch.stack(planes, 3).contiguous().to(dev)        # [nch, ks, half, plane, lane, 8]
    p.ba, p.wb, p.bb = pa.bias, pb.w, pb.bias
    p.wa_scale, p.cin_s, p.cmid, p.cmid_s = pa.w_scale, pa.cin_s, pb.c, pb.c_s
    return p


def _x3d_ab_desc(x, pk, out_ld, act):
    d = _lib.X3dAbDesc()
    d.N, d.T, d.H, d.W = x.N, x.T, x.H, x.W
    d.Cin, d.Cmid, d.ldx, d.ldu, d.act, d.wa_scale = pk.cin_s, pk.cmid_s, x.ld, out_ld, act, pk.wa_scale
    return d


def x3d_ab_supported(x, pk):
    return pk is not None and x.dense and x.Cs == pk.cin_s and bool(_lib.load().mspi_x3d_ab_supported(C.byref(_x3d_ab_desc(x, pk, pk.cmid_s, ACT_NONE))))


def x3d_ab(x, pk, pool=False):
    """u = act(b_bn(dw3x3x3(relu(a_bn(a(x)))))) in one launch (csrc/x3d_block.hip); pool=True: no activation, also returns
    the [N, rows, C] partial sums of u for the squeeze-excite gate (X3DTransform with SE); otherwise act = Swish."""
    lib = _lib.load()
    _need_gpu(x.buf)
    out = alloc(x.N, x.T, x.H, x.W, pk.cmid, x.buf.device)
    d = _x3d_ab_desc(x, pk, out.ld, ACT_NONE if pool else ACT_SWISH)
    part = None
    if pool:
        rows = lib.mspi_x3d_ab_pool_rows(C.byref(d))
        part = torch.empty(x.N, rows, pk.cmid_s, dtype=torch.float32, device=x.buf.device)
    with _Timed("x3d_ab_pool" if pool else "x3d_ab", 2.0 * x.M * (pk.cin_s + 27) * pk.cmid, 4.0 * x.M * (x.C + pk.cmid),
                "in=%s Cin=%d Cmid=%d" % ((x.N, x.T, x.H, x.W), x.C, pk.cmid)):
        check(lib.mspi_x3d_ab_fwd(C.byref(d), x.ptr, pk.wa.data_ptr(), pk.ba.data_ptr(), pk.wb.data_ptr(), pk.bb.data_ptr(),
                                  out.ptr, part.data_ptr() if pool else None, _stream()), "mspi_x3d_ab_fwd")
    return (out, part) if pool else out


class PackedX3dCa:
    __slots__ = ("w", "bc", "ba", "d", "cx", "cx_s", "d_s", "wc_scale", "wa_scale")


def x3d_ca_supported(d_s, cx_s):
    """Mirror of mspi_x3d_ca_supported."""
    return 4 <= d_s <= 224 and d_s % 4 == 0 and 4 <= cx_s <= 256 and cx_s % 4 == 0


def pack_x3d_ca(pc, pa):
    """Operands of the fused X3D block seam (mspi_x3d_ca_fwd): this block's `c` conv `pc` and the next block's `a` conv `pa`
    (both PackedConv, 1x1x1, f16x3, ReLU); None when the pair is outside the kernel's range."""
    ok = all(q.prec == PREC_F16X3 and q.k == (1, 1, 1) and q.stride == (1, 1, 1) and q.pad == (0, 0, 0) and q.act == ACT_RELU
             and q.bias is not None for q in (pc, pa))
    if not ok or pc.cout_s != pa.cin_s or pc.cin_s != pa.cout_s or not x3d_ca_supported(pc.cin_s, pc.cout_s):
        return None
    d_s, cx_s = pc.cin_s, pc.cout_s
    c = 128 if d_s <= 128 else 224
    hid = (cx_s + 31) // 32 * 32
    nch, ks, ct = hid // 32, c // 16, c // 32
    w1 = torch.zeros(hid, c, dtype=torch.float32)
    w1[:cx_s, :d_s] = (pc.w[0].float() + pc.w[1].float()).cpu()[:, :d_s]      # hi + lo = the scaled fp32 weight's 22 bits
    w2 = torch.zeros(c, hid, dtype=torch.float32)
    w2[:d_s, :cx_s] = (pa.w[0].float() + pa.w[1].float()).cpu()[:, :cx_s]

    def planes(ws):
        hi = ws.to(torch.float16)
        return hi, (ws - hi.float()).to(torch.float16)

    parts1 = [pl.view(nch, 32, ks, 2, 8).permute(0, 2, 3, 1, 4).reshape(nch, ks, 64, 8) for pl in planes(w1)]
    parts2 = [pl.view(ct, 32, nch, 2, 2, 2, 4).permute(2, 3, 0, 5, 1, 4, 6).reshape(nch, 2, ct, 64, 8) for pl in planes(w2)]
    p = PackedX3dCa()
    p.w = torch.cat([torch.stack(parts1, 2).reshape(nch, -1), torch.stack(parts2, 3).reshape(nch, -1)], 1).contiguous().to(pc.w.device)
    p.bc, p.ba = pc.bias, pa.bias
    p.d, p.d_s, p.cx, p.cx_s, p.wc_scale, p.wa_scale = pc.cin, d_s, pc.cout, cx_s, pc.w_scale, pa.w_scale
    return p


def x3d_ca(u, pk, res, gate=None):
    """(y, t) = (relu(c(u') + res), relu(a_next(y))) in one launch (csrc/mlp_fused.hip); u' = swish(u * gate) with a gate."""
    lib = _lib.load()
    _need_gpu(u.buf)
    if not (u.dense and res.dense) or u.Cs != pk.d_s or res.Cs != pk.cx_s or res.M != u.M:
        raise MspiError("x3d_ca: u %s / res %s do not match the pack (%d, %d)" % ((u.M, u.Cs), (res.M, res.Cs), pk.d_s, pk.cx_s))
    if pk.w.device != u.buf.device:
        raise MspiError("x3d_ca: packed weights live on %s, the input on %s" % (pk.w.device, u.buf.device))
    y = alloc(u.N, u.T, u.H, u.W, pk.cx, u.buf.device)
    t = alloc(u.N, u.T, u.H, u.W, pk.d, u.buf.device)
    d = _lib.X3dCaDesc()
    d.M, d.D, d.Cx = u.M, pk.d_s, pk.cx_s
    d.ldu, d.ldr, d.ldy, d.ldt, d.ldg = u.ld, res.ld, y.ld, t.ld, pk.d_s
    d.rows_per_sample, d.wc_scale, d.wa_scale = u.T * u.H * u.W, pk.wc_scale, pk.wa_scale
    with _Timed("x3d_ca", 4.0 * u.M * pk.d * pk.cx, 4.0 * u.M * (2 * pk.d + 2 * pk.cx),
                "M=%d D=%d Cx=%d%s" % (u.M, pk.d, pk.cx, " +gate" if gate is not None else "")):
        check(lib.mspi_x3d_ca_fwd(C.byref(d), u.ptr, gate.data_ptr() if gate is not None else None, pk.w.data_ptr(),
                                  pk.bc.data_ptr(), pk.ba.data_ptr(), res.ptr, y.ptr, t.ptr, _stream()), "mspi_x3d_ca_fwd")
    return y, t


def _pad_vec(v, n):
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v.detach().float().view(-1)
    return out


# ----------------------------------------------------------------------------- op wrappers
def _out_extent(T, H, W, k, s, p):
    return ((T + 2 * p[0] - k[0]) // s[0] + 1, (H + 2 * p[1] - k[1]) // s[1] + 1, (W + 2 * p[2] - k[2]) // s[2] + 1)


# mspi_gemm_sp_fwd: 128 x {128,64,96,192,256}, 256 x {256,192,128}
SP_TILES = (6, 7, 9, 10, 11, 12, 13, 14)


W_BLOCKED = _os.environ.get("MSPI_W_BLOCKED", "1") != "0"      # A/B switch: blocked weights for the LDS-DMA kernels on fp32 activations


def sp_weights(pk):
    """The weights of an f16x3 pack in the form mspi_gemm_sp_fwd takes: blocked like the activation planes (16 rows x 32 k =
    1 KB contiguous, k-fastest, rows zero-padded to a multiple of 16), so that every LDS-DMA piece of a stage is 8 full cache
    lines.  Built on first use, kept on the pack."""
    if pk.wsp is None:
        npad = (pk.cout_s + 15) // 16 * 16
        w = torch.zeros(2, npad, pk.ldw, dtype=torch.float16, device=pk.w.device)
        w[:, : pk.cout_s] = pk.w
        pk.wsp = w.view(2, npad // 16, 16, pk.ldw // 32, 32).permute(0, 1, 3, 2, 4).contiguous()
    return pk.wsp


def _conv_sp(x, pk, out, res, act, tile, sp_out):
    """Dense GEMM on pre-split activation planes; result as fp32 rows (CL) or, sp_out, as planes for the next GEMM."""
    lib = _lib.load()
    _need_gpu(x.buf)
    if pk.k != (1, 1, 1) or pk.stride != (1, 1, 1) or pk.pad != (0, 0, 0) or pk.prec != PREC_F16X3:
        raise MspiError("conv: split-plane activations feed 1x1x1 / Linear f16x3 layers only")
    if x.C != pk.cin_s or pk.ldw != x.C:
        raise MspiError("conv: split-plane input has %d channels, weights were packed for %d (ldw %d)" % (x.C, pk.cin_s, pk.ldw))
    dev = x.buf.device
    if pk.w.device != dev:
        raise MspiError("conv: packed weights live on %s, the input on %s" % (pk.w.device, dev))
    M = x.M
    if sp_out:
        if res is not None or pk.cout_s % 32:
            raise MspiError("conv: split-plane output takes no residual and needs Cout %% 32 == 0")
        out = alloc_sp(x.N, x.T, x.H, x.W, pk.cout_s, dev)
    else:
        if out is None:
            out = alloc(x.N, x.T, x.H, x.W, pk.cout, dev)
        if out.M != M or out.Cs != pk.cout_s or not out.dense:
            raise MspiError("conv: output CL does not match %d rows x %d channels" % (M, pk.cout))
    if res is not None and (res.M != M or not res.dense):
        raise MspiError("conv: residual rows %d != output rows %d (or residual not dense)" % (res.M, M))
    d = ConvDesc()
    d.N, d.T, d.H, d.W, d.C = x.N, x.T, x.H, x.W, x.C
    d.kT = d.kH = d.kW = d.strT = d.strH = d.strW = 1
    d.To, d.Ho, d.Wo, d.Cout = x.T, x.H, x.W, pk.cout_s
    d.ldy = 0 if sp_out else out.ld
    d.ldw, d.ldr = pk.ldw, (res.ld if res is not None else 0)
    d.act = pk.act if act is None else act
    d.prec, d.w_scale = pk.prec, pk.w_scale
    args = (x.ptr, x.ld, x.plane, sp_weights(pk).data_ptr(), pk.bias.data_ptr() if pk.bias is not None else None,
            res.ptr if res is not None else None, None if sp_out else out.ptr, out.ptr if sp_out else None,
            out.ld if sp_out else 0, out.plane if sp_out else 0, _stream())

    def launch(t):
        d.tile = t
        return lib.mspi_gemm_sp_fwd(C.byref(d), *args)

    key = ("sp", M, x.C, pk.cout_s, res is not None, bool(sp_out))
    choice = -1
    if tile is not None:
        choice = tile
    elif AUTOTUNE["on"] and not torch.cuda.is_current_stream_capturing():
        choice = AUTOTUNE["cache"].get(key)
        if choice is None:
            choice = _tune_conv(launch, key, [t for t in SP_TILES if t < 12 or M >= 4096])
    elif key in AUTOTUNE["cache"]:
        choice = AUTOTUNE["cache"][key]
    with _Timed("conv_gemm", 2.0 * M * pk.cin * pk.cout, 4.0 * (M * pk.cin + M * pk.cout * (2 if res is not None else 1) + pk.cout * pk.cin),
                "M=%d K=%d(1x%d) N=%d pre-split%s%s" % (M, pk.cin, pk.cin, pk.cout, " +res" if res is not None else "", " ->planes" if sp_out else "")) as tm:
        check(launch(choice), "mspi_gemm_sp_fwd")
        if Profiler.active is not None:
            c = lib.mspi_conv_last_config()
            tm.name = "conv_gemm<%d,%d,dma-presplit,f16x3>" % (c >> 16, (c >> 4) & 0xFFF)
    return out


def conv(x, pk, out=None, res=None, gate=None, act=None, tile=None, sp_out=False):
    """x: CL, or a raw 5-D [N,C,T,H,W] / 4-D [N,C,H,W] torch tensor with arbitrary strides.
    tile: force a kernel instantiation (MspiConvDesc.tile); None = autotune cache / library heuristic."""
    lib = _lib.load()
    tuning = AUTOTUNE["on"] and RANGE_CHECK["on"] and pk.prec == PREC_F16X3 and not torch.cuda.is_current_stream_capturing()
    if isinstance(x, SP):
        if tuning:   # planes: the hi plane carries the magnitude (an out-of-range layer is fixed from the NEXT forward on: its
            #          producer stops emitting planes once this pack is fp32)
            _range_check(pk, lambda: x.buf[: x.plane].abs().max(), "planes -> %dx%d" % (pk.cin, pk.cout))      # the hi plane (pad rows are zero)
        if pk.prec == PREC_F16X3:
            return _conv_sp(x, pk, out, res, act, tile, sp_out)
        # The range check has moved this layer to the fp32 path while its producer had already emitted planes (from the next
        # forward on the producer hands over fp32 rows: mlp_tail / layernorm_for_gemm look at the pack's precision): rebuild
        # the rows (hi + lo = 22 bits of the value) and take the fp32 path for this one call.
        x, sp_out = join_planes(x), False
    if tuning:
        _range_check(pk, (lambda: (x.as_rows()[:, : x.C] if x.dense else x.buf).abs().max()) if isinstance(x, CL) else (lambda: x.abs().max()),
                     "conv %s %d -> %d" % (pk.k, pk.cin, pk.cout))
    if sp_out:
        raise MspiError("conv: split-plane output needs a split-plane input (mspi_gemm_sp_fwd)")
    d = ConvDesc()
    if isinstance(x, CL):
        _need_gpu(x.buf)
        N, T, H, W, Cin = x.N, x.T, x.H, x.W, x.Cs
        d.sN, d.sT, d.sH, d.sW, d.sC = x.sN, H * W * x.ld, W * x.ld, x.ld, 1
        xptr, dev = x.ptr, x.buf.device
    else:
        _need_gpu(x)
        if x.dim() == 4:
            x = x[:, :, None]
        if x.dtype != torch.float32:
            raise MspiError("conv: input must be fp32")
        N, Cin, T, H, W = x.shape
        sN, sC, sT, sH, sW = x.stride()
        d.sN, d.sT, d.sH, d.sW, d.sC = sN, sT, sH, sW, sC
        xptr, dev = x.data_ptr(), x.device
    if Cin != pk.cin_s:
        raise MspiError("conv: input has %d stored channels, weights were packed for %d" % (Cin, pk.cin_s))
    if pk.w.device != dev:   # a host pointer handed to a kernel is a GPU memory fault, not an exception
        raise MspiError("conv: packed weights live on %s, the input on %s" % (pk.w.device, dev))
    To, Ho, Wo = _out_extent(T, H, W, pk.k, pk.stride, pk.pad)
    if out is None:
        out = alloc(N, To, Ho, Wo, pk.cout, dev)
    if (out.N, out.T, out.H, out.W) != (N, To, Ho, Wo) or out.Cs != pk.cout_s or not out.dense:
        raise MspiError("conv: output CL %s does not match %s" % ((out.N, out.T, out.H, out.W, out.C), (N, To, Ho, Wo, pk.cout)))
    d.N, d.T, d.H, d.W, d.C = N, T, H, W, Cin
    d.kT, d.kH, d.kW = pk.k
    d.strT, d.strH, d.strW = pk.stride
    d.padT, d.padH, d.padW = pk.pad
    d.To, d.Ho, d.Wo = To, Ho, Wo
    d.Cout = pk.cout_s
    d.ldy, d.ldw = out.ld, pk.ldw
    d.ldr = res.ld if res is not None else 0
    d.act = pk.act if act is None else act
    d.prec, d.w_scale = pk.prec, pk.w_scale
    d.w_blocked = sp_weights(pk).data_ptr() if (pk.prec == PREC_F16X3 and W_BLOCKED) else None      # the LDS-DMA kernels' weight source
    if res is not None and (res.M != out.M or not res.dense):
        raise MspiError("conv: residual rows %d != output rows %d (or residual not dense)" % (res.M, out.M))
    M = N * To * Ho * Wo
    taps = pk.k[0] * pk.k[1] * pk.k[2]
    tm = _Timed("conv_gemm", 2.0 * M * taps * pk.cin * pk.cout,
                4.0 * (N * T * H * W * pk.cin + M * pk.cout * (2 if res is not None else 1) + pk.cout * taps * pk.cin),
                "M=%d K=%d(%dx%d) N=%d s=%s%s%s" % (M, taps * pk.cin, taps, pk.cin, pk.cout, pk.stride,
                                                  " +res" if res is not None else "", " +gate" if gate is not None else ""))
    args = (xptr, pk.w.data_ptr(), pk.bias.data_ptr() if pk.bias is not None else None,
            res.ptr if res is not None else None, gate.data_ptr() if gate is not None else None, out.ptr, _stream())
    # the row-stationary thin GEMM (mspi_rowgemm_fwd) is a second implementation of dense 1x1x1 layers with K <= 224:
    # kernel choice THIN competes with the tile codes of mspi_conv_fwd in the autotuner
    rg = None
    if THIN_ENABLED and pk.thin is not None and isinstance(x, CL) and x.dense:
        rg = _lib.RowGemmDesc()
        rg.M, rg.K, rg.N = M, pk.cin_s, pk.cout_s
        rg.ldx, rg.ldy, rg.ldr, rg.ldg = x.ld, out.ld, d.ldr, pk.cin_s
        rg.act, rg.rows_per_sample, rg.w_scale = d.act, To * Ho * Wo, pk.w_scale
        rg_args = (xptr, pk.thin.data_ptr(), args[2], args[3], args[4], out.ptr, args[6])

    def launch(t):
        if t == THIN:
            return lib.mspi_rowgemm_fwd(C.byref(rg), *rg_args)
        if t >= SPLITK:      # split-K: t - SPLITK slices of the contraction, partial sums through a scratch buffer
            S = t - SPLITK
            ws = torch.empty(S * M * pk.cout_s, dtype=torch.float32, device=dev)   # stream-ordered: safe to drop after the launch
            d.tile = 3
            return lib.mspi_conv_splitk_fwd(C.byref(d), args[0], args[1], args[2], args[3], args[5], ws.data_ptr(), S, args[6])
        d.tile = t
        return lib.mspi_conv_fwd(C.byref(d), *args)

    choice = -1
    key = (M, taps * pk.cin_s, pk.cout_s, pk.k, pk.stride, pk.prec, d.sC == 1, res is not None, gate is not None, rg is not None)
    if tile is not None:
        choice = tile
    elif AUTOTUNE["on"] and not torch.cuda.is_current_stream_capturing():
        choice = AUTOTUNE["cache"].get(key)
        if choice is None:
            cands = [1, 2, 3, 4]
            if pk.prec == PREC_F16X3 and d.sC == 1 and Cin % 4 == 0:
                cands += [6, 7, 9, 10] + ([8] if pk.cout_s <= 256 else [])
                if M >= 16384:
                    cands += [12, 13, 14]        # 256-row / 8-wave form of the LDS-DMA kernel
            if rg is not None:
                cands.append(THIN)
            nk = pk.ldw // 32
            if SPLITK_ENABLED and gate is None and -(-M // 64) * -(-pk.cout_s // 64) <= 384 and nk >= 32:
                # few output tiles, long contraction: K slices across workgroups (mspi_conv_splitk_fwd)
                cands += [SPLITK + S for S in (2, 4, 8) if nk >= 8 * S]
            choice = _tune_conv(launch, key, cands)
    elif key in AUTOTUNE["cache"]:
        choice = AUTOTUNE["cache"][key]
    elif rg is not None and THIN_DEFAULT:
        choice = THIN
    if choice == THIN and rg is None:
        raise MspiError("conv: the thin-GEMM kernel does not cover this call")
    with tm:
        check(launch(choice), "mspi_rowgemm_fwd" if choice == THIN else "mspi_conv_splitk_fwd" if choice >= SPLITK else "mspi_conv_fwd")
        if Profiler.active is not None:
            if choice == THIN:
                tm.name = "rowgemm<%d,f16x3>" % rowgemm_ksb(pk.cin_s)
            elif choice >= SPLITK:
                tm.name = "conv_gemm<64,64,splitk%d>" % (choice - SPLITK)
            else:
                c = lib.mspi_conv_last_config()
                tm.name = "conv_gemm<%d,%d,%s,%s>" % (c >> 16, (c >> 4) & 0xFFF,
                                                      "dma" if c & 4 else ("s" if c & 1 else "v4") + ("w8" if c & 8 else ""),
                                                      "f16x3" if (c >> 1) & 1 else "f32")
    return out


def _dw_desc(x, k, s, p, out_ld):
    d = DwConvDesc()
    assert x.dense
    d.N, d.T, d.H, d.W, d.C = x.N, x.T, x.H, x.W, x.Cs
    d.ldx, d.ldy = x.ld, out_ld
    d.kT, d.kH, d.kW = k
    d.strT, d.strH, d.strW = s
    d.padT, d.padH, d.padW = p
    d.To, d.Ho, d.Wo = _out_extent(x.T, x.H, x.W, k, s, p)
    return d


def dwconv(x, pk, out=None, pool=False, act=None):
    """pool=True (X3D squeeze-excite): also returns the [N, rows, C] partial sums of the pre-activation output."""
    lib = _lib.load()
    _need_gpu(x.buf)
    if x.Cs != pk.c_s:
        raise MspiError("dwconv: input has %d channels, weights packed for %d" % (x.C, pk.c_s))
    if pk.w.device != x.buf.device:
        raise MspiError("dwconv: packed weights live on %s, the input on %s" % (pk.w.device, x.buf.device))
    To, Ho, Wo = _out_extent(x.T, x.H, x.W, pk.k, pk.stride, pk.pad)
    if out is None:
        out = alloc(x.N, To, Ho, Wo, x.C, x.buf.device)
    d = _dw_desc(x, pk.k, pk.stride, pk.pad, out.ld)
    d.act = pk.act if act is None else act
    part = None
    if pool:
        rows = lib.mspi_dwconv_pool_rows(C.byref(d))
        if rows <= 0:
            raise MspiError("dwconv: squeeze-excite pooling is not supported for kernel %s stride %s" % (pk.k, pk.stride))
        part = torch.empty(x.N, rows, pk.c_s, dtype=torch.float32, device=x.buf.device)
    taps = pk.k[0] * pk.k[1] * pk.k[2]
    with _Timed("dwconv_pool" if pool else "dwconv", 2.0 * out.M * taps * pk.c, 4.0 * (x.M + out.M) * pk.c,
                "in=%s C=%d k=%s s=%s" % ((x.N, x.T, x.H, x.W), pk.c, pk.k, pk.stride)):
        check(lib.mspi_dwconv_fwd(C.byref(d), x.ptr, pk.w.data_ptr(), pk.bias.data_ptr(), out.ptr,
                                  part.data_ptr() if pool else None, _stream()), "mspi_dwconv_fwd")
    return (out, part) if pool else out


def maxpool(x, k, s, p, out=None):
    lib = _lib.load()
    _need_gpu(x.buf)
    To, Ho, Wo = _out_extent(x.T, x.H, x.W, k, s, p)
    if out is None:
        out = alloc(x.N, To, Ho, Wo, x.C, x.buf.device)
    d = _dw_desc(x, k, s, p, out.ld)
    d.act = ACT_NONE
    with _Timed("maxpool", 0.0, 4.0 * (x.M + x.N * To * Ho * Wo) * x.C):
        check(lib.mspi_maxpool_fwd(C.byref(d), x.ptr, out.ptr, _stream()), "mspi_maxpool_fwd")
    return out


def se_gate(pool, inv_count, w1, b1, w2, b2, gate=None):
    """pool: [N, rows, C] partial sums from dwconv(pool=True) -> gate [N, C]."""
    lib = _lib.load()
    N, rows, Cc = pool.shape
    if gate is None:
        gate = torch.empty(N, Cc, dtype=torch.float32, device=pool.device)
    check(lib.mspi_se_gate(pool.data_ptr(), rows, float(inv_count), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                           b2.data_ptr(), gate.data_ptr(), N, Cc, w1.shape[0], _stream()), "mspi_se_gate")
    return gate


def layernorm(x, gamma, beta, eps, out=None, act=ACT_NONE, table=None, sp=False):
    """Rows of x -> rows of out; x and out may be token slabs (sample stride != dense).
    sp=True: the result as pre-split f16 planes (SP) for a following GEMM."""
    lib = _lib.load()
    _need_gpu(x.buf)
    if sp:
        assert out is None and table is None and x.C % 32 == 0
        o = alloc_sp(x.N, x.T, x.H, x.W, x.C, x.buf.device)
        with _Timed("layernorm", 8.0 * x.M * x.C, 8.0 * x.M * x.C, "M=%d C=%d ->planes" % (x.M, x.C)):
            check(lib.mspi_layernorm_sp_fwd(x.ptr, x.ld, x.sN, o.ptr, o.ld, o.plane, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                            x.N, x.T * x.H * x.W, x.C, act, _stream()), "mspi_layernorm_sp_fwd")
        return o
    if out is None:
        out = alloc(x.N, x.T, x.H, x.W, x.C, x.buf.device)
    R = x.T * x.H * x.W
    assert out.N == x.N and out.T * out.H * out.W == R and out.C == x.C
    P = 0 if table is None else table.shape[0]
    assert table is None or P == R
    with _Timed("layernorm", 8.0 * x.M * x.C, 8.0 * x.M * x.C, "M=%d C=%d" % (x.M, x.C)):
        check(lib.mspi_layernorm_fwd(x.ptr, x.ld, x.sN, out.ptr, out.ld, out.sN, gamma.data_ptr(), beta.data_ptr(),
                                     float(eps), x.N, R, x.C, act, table.data_ptr() if table is not None else None,
                                     _stream()), "mspi_layernorm_fwd")
    return out


def attention(qkv, B, Ntok, heads, hd, scale, out=None, biasT=None, maskT=None, tok_idx=None, rows_per_sample=None):
    """qkv: CL with rows (b, token) and 3*heads*hd columns laid out [3][heads][hd]
    (what `qkv.reshape(B,N,3,h,hd)` means, model/model_utils.py:100).  B sequences of Ntok tokens.
    biasT [heads][Ntok][Ntok] / maskT [nmask][Ntok][Ntok]: key-major additive terms (Swin).
    tok_idx int32 [nwin][Ntok]: the B = samples*nwin sequences are windows whose token t sits at row
    tok_idx[win][t] of its sample (shifted-window attention without gather/scatter passes).
    rows_per_sample: rows of qkv (and of the output) per sample when that is not nwin*Ntok -- Swin on a grid that is
    not a multiple of the window keeps ONE extra row per sample for all padding tokens (tok_idx points there)."""
    lib = _lib.load()
    Cc = heads * hd
    assert qkv.C == 3 * Cc and qkv.dense and (rows_per_sample is not None or qkv.M == B * Ntok)
    if out is None:
        out = alloc(qkv.N, qkv.T, qkv.H, qkv.W, Cc, qkv.buf.device)
    d = AttnDesc()
    d.B, d.Hh, d.Nq, d.Nk, d.D, d.Dv = B, heads, Ntok, Ntok, hd, hd
    d.nmask = 0 if maskT is None else maskT.shape[0]
    d.nwin = 0 if tok_idx is None else tok_idx.shape[0]
    if rows_per_sample is None:
        rows_per_sample = Ntok * max(d.nwin, 1)
    d.q_sB = d.k_sB = d.v_sB = rows_per_sample * qkv.ld
    d.q_sH = d.k_sH = d.v_sH = hd
    d.q_sT = d.k_sT = d.v_sT = qkv.ld
    d.o_sB, d.o_sH, d.o_sT = rows_per_sample * out.ld, hd, out.ld
    d.scale = float(scale)
    d.prec = _attn_prec(("qkv", heads, hd, Ntok, d.nwin), lambda: qkv.as_rows()[:, :Cc].abs().max() * abs(float(scale)),
                        lambda: qkv.as_rows()[:, Cc:].abs().max())
    base = qkv.ptr
    with _Timed("attention", 4.0 * B * heads * Ntok * Ntok * hd, 16.0 * B * Ntok * Cc, "B=%d h=%d N=%d d=%d" % (B, heads, Ntok, hd)):
        _attn_launch(lib, d, base, base + 4 * Cc, base + 8 * Cc, None,
                     biasT.data_ptr() if biasT is not None else None, maskT.data_ptr() if maskT is not None else None,
                     tok_idx.data_ptr() if tok_idx is not None else None, out.ptr, qkv.buf.device)
    return out


ATTN_PLANES = _os.environ.get("MSPI_ATTN_PLANES", "1") != "0"      # A/B switch: K / V split once per head (mspi_attn_fwd_ws)
# f16x3 attention scales q by 64 and k, v by 16 before the split (csrc/attn.hip): |q * scale| >= 1023 or |k|, |v| >= 4094 is inf
# in the hi half.  First sight of an attention shape while tuning: operands beyond a quarter of that move the SHAPE to the fp32
# MFMA kernel for good (same contract as the GEMM packs' range check).
ATTN_PREC = {}


def _attn_prec(key, amax_q, amax_kv):
    if DEFAULT_PREC != PREC_F16X3:
        return DEFAULT_PREC
    prec = ATTN_PREC.get(key)
    if prec is None:
        if not _tuning():
            return PREC_F16X3
        aq, akv = float(amax_q()), float(amax_kv())
        bad = not (aq == aq and akv == akv) or aq >= 256.0 or akv >= 1024.0
        prec = ATTN_PREC[key] = PREC_F32 if bad else PREC_F16X3
        if bad:
            RANGE_CHECK["moved"].append(("attention %s" % (key,), max(aq, akv)))
    return prec


def _attn_launch(lib, d, q, k, v, res, biasT, maskT, tok_idx, o, dev):
    nbytes = lib.mspi_attn_ws_bytes(C.byref(d)) if ATTN_PLANES else 0
    if nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # stream-ordered: safe to drop after the launch
        check(lib.mspi_attn_fwd_ws(C.byref(d), q, k, v, res, biasT, maskT, tok_idx, o, ws.data_ptr(), _stream()), "mspi_attn_fwd_ws")
    else:
        check(lib.mspi_attn_fwd(C.byref(d), q, k, v, res, biasT, maskT, tok_idx, o, _stream()), "mspi_attn_fwd")


def mlp(x, pk, res=None, ln=None, eps=1e-6, out=None, split=None):
    """out = res + fc2(act(fc1(LN(x)))) in one launch (mspi_mlp_fwd); ln = (gamma, beta) or None.
    split: the same pair as two PackedConv (fc1 with the activation, fc2), if the caller has them -- used by the first-sight
    range check (else built from the fused pack's sources)."""
    lib = _lib.load()
    _need_gpu(x.buf)
    if pk.fallback is None and not pk.checked and _tuning():
        # First sight of this fused pair while tuning: run it ONCE as LayerNorm + two GEMMs, whose packs go through conv()'s
        # range check on their real inputs (fc1: the normalised rows, fc2: the activations).  If either leaves the f16x3
        # window the pair stays unfused on the fp32 path for good; otherwise the fused kernel takes over from the next call.
        pk.checked = True
        if split is None:
            w1, b1, w2, b2, osc = pk.src
            split = (pack_conv(w1, b1, act=pk.act, device=pk.w.device), pack_conv(w2, b2, out_scale=osc, device=pk.w.device))
        y = conv(conv(layernorm(x, ln[0], ln[1], eps) if ln is not None else x, split[0]), split[1], res=res, out=out)
        if split[0].prec != PREC_F16X3 or split[1].prec != PREC_F16X3:
            pk.fallback = split
        return y
    if pk.fallback is not None:
        return conv(conv(layernorm(x, ln[0], ln[1], eps) if ln is not None else x, pk.fallback[0]), pk.fallback[1], res=res, out=out)
    if x.C != pk.c or not x.dense:
        raise MspiError("mlp: input has %d channels (dense=%s), packed for %d" % (x.C, x.dense, pk.c))
    if pk.w.device != x.buf.device:
        raise MspiError("mlp: packed weights live on %s, the input on %s" % (pk.w.device, x.buf.device))
    if out is None:
        out = alloc(x.N, x.T, x.H, x.W, pk.c, x.buf.device)
    if res is not None and (res.M != x.M or not res.dense):
        raise MspiError("mlp: residual rows %d != rows %d (or residual not dense)" % (res.M, x.M))
    d = _lib.MlpDesc()
    d.M, d.C, d.hidden = x.M, pk.c, pk.hidden
    d.ldx, d.ldy, d.ldr = x.ld, out.ld, (res.ld if res is not None else 0)
    d.ln, d.act, d.eps = (1 if ln is not None else 0), pk.act, eps
    d.w1_scale, d.w2_scale = pk.s1, pk.s2
    with _Timed("mlp_fused", 4.0 * x.M * pk.c * pk.hidden, 4.0 * x.M * pk.c * (3 if res is not None else 2),
                "M=%d C=%d hidden=%d" % (x.M, pk.c, pk.hidden)):
        check(lib.mspi_mlp_fwd(C.byref(d), x.ptr, ln[0].data_ptr() if ln is not None else None,
                               ln[1].data_ptr() if ln is not None else None, pk.w.data_ptr(), pk.b1.data_ptr(),
                               pk.b2.data_ptr(), res.ptr if res is not None else None, out.ptr, _stream()), "mspi_mlp_fwd")
    return out


def layernorm_for_gemm(x, gamma, beta, eps, *packs):
    """LayerNorm whose result is consumed ONLY by the dense GEMMs `packs`: handed over as pre-split planes when every
    consumer can take them (K a multiple of 32, K >= 256: below that the thin-GEMM kernels on fp32 rows are faster)."""
    if sp_supported(x.C) and x.C >= 256 and all(p.k == (1, 1, 1) and p.stride == (1, 1, 1) and p.ldw == x.C and p.prec == PREC_F16X3
                                                 for p in packs):
        return layernorm(x, gamma, beta, eps, sp=True)
    return layernorm(x, gamma, beta, eps)


def pack_mlp_tail(fc1, fc2, out_scale=None):
    """LN -> Linear -> GELU -> Linear (+ residual) tail of a ConvNeXt / Swin / MViT block: the fused kernel where it
    applies (C in {96, 192}, f16x3), else the two GEMM packs.  Use with mlp_tail()."""
    if mlp_supported(fc1.in_features, fc1.out_features) and fc2.out_features == fc1.in_features:
        return ("fused", pack_mlp(fc1.weight, fc1.bias, fc2.weight, fc2.bias, out_scale=out_scale))
    return ("split", pack_conv(fc1.weight, fc1.bias, act=ACT_GELU), pack_conv(fc2.weight, fc2.bias, out_scale=out_scale))


def mlp_tail(x, packed, ln, eps, res):
    """res + fc2(GELU(fc1(LayerNorm(x)))) with packed = pack_mlp_tail(...), ln = (gamma, beta)."""
    if packed[0] == "fused":
        return mlp(x, packed[1], res=res, ln=ln, eps=eps)
    if sp_supported(x.C) and sp_supported(packed[1].cout_s) and packed[1].ldw == x.C and packed[2].ldw == packed[1].cout_s \
            and packed[1].prec == PREC_F16X3 and packed[2].prec == PREC_F16X3:
        # LN -> planes, fc1 + GELU -> planes, fc2 (+res) -> fp32 rows: no operand is converted inside a GEMM loop
        return conv(conv(layernorm(x, ln[0], ln[1], eps, sp=True), packed[1], sp_out=True), packed[2], res=res)
    return conv(conv(layernorm(x, ln[0], ln[1], eps), packed[1]), packed[2], res=res)


def space_to_depth(x, out=None):
    """Swin PatchMerging gather: [N,T,H,W,C] -> [N,T,H/2,W/2,4C], quadrant order (0,0),(1,0),(0,1),(1,1)."""
    lib = _lib.load()
    assert x.dense and x.C % 4 == 0
    if out is None:
        out = alloc(x.N, x.T, x.H // 2, x.W // 2, 4 * x.C, x.buf.device)
    with _Timed("space_to_depth", 0.0, 8.0 * x.M * x.C):
        check(lib.mspi_space_to_depth(x.ptr, x.ld, out.ptr, out.ld, x.N * x.T, x.H, x.W, x.C, _stream()), "mspi_space_to_depth")
    return out


def mvit_attention(q, k, v, B, heads, hd, scale, q_thw, k_thw, Rh, Rw, Rt, out=None, rel_gemm=None):
    """MViTv2 pooled attention with decomposed relative positions and residual pooling (backbones/MViT.py:1261-1301).
    q: CL rows (b, tq,hq,wq) x heads*hd (pooled + normed); k, v likewise over the pooled key grid.
    Rh / Rw / Rt: gathered tables [q_size][k_size][hd].  rel_gemm = (packed_tables, idx_h, idx_w, idx_t): the q . R dot
    products as ONE thin GEMM of the q rows against all distinct table rows (pack_conv of their stack) followed by a gather
    (mspi_mvit_qk_augment_p) instead of the per-(row, j) dot-product kernel.
    Returns CL rows (b, q token) x heads*hd = softmax(...) v + q."""
    lib = _lib.load()
    Nq, Nk = q_thw[0] * q_thw[1] * q_thw[2], k_thw[0] * k_thw[1] * k_thw[2]
    J = k_thw[0] + k_thw[1] + k_thw[2]
    DA = 128 if hd + J <= 128 else (144 if hd + J <= 144 else 160)      # k16 steps of S: 8 / 9 / 10
    if hd != 96 or hd + J > DA:
        raise MspiError("mvit_attention: head_dim %d with %d relative-position columns is not instantiated" % (hd, J))
    dev = q.buf.device
    qa = torch.empty(B * heads * Nq * DA, dtype=torch.float32, device=dev)
    ka = torch.empty(B * heads * Nk * DA, dtype=torch.float32, device=dev)
    a = MvitAugDesc()
    a.B, a.heads, a.Dh, a.DA = B, heads, hd, DA
    a.qT, a.qH, a.qW = q_thw
    a.kT, a.kH, a.kW = k_thw
    a.ldq, a.ldk, a.scale = q.ld, k.ld, float(scale)
    if rel_gemm is not None and q.dense and q.ld == heads * hd:
        pkT, idx_h, idx_w, idx_t = rel_gemm
        rows = CL(q.buf, q.off, q.M * heads, 1, 1, 1, hd, hd)           # (b, token, head) rows of hd channels
        P = conv(rows, pkT)
        with _Timed("mvit_qk_augment", 0.0, 4.0 * B * heads * (Nq + Nk) * (hd + DA)):
            check(lib.mspi_mvit_qk_augment_p(C.byref(a), q.ptr, k.ptr, P.ptr, P.ld, idx_h.data_ptr(), idx_w.data_ptr(),
                                             idx_t.data_ptr(), qa.data_ptr(), ka.data_ptr(), _stream()), "mspi_mvit_qk_augment_p")
    else:
        with _Timed("mvit_qk_augment", 2.0 * B * heads * Nq * J * hd, 4.0 * B * heads * (Nq + Nk) * (hd + DA)):
            check(lib.mspi_mvit_qk_augment(C.byref(a), q.ptr, k.ptr, Rh.data_ptr(), Rw.data_ptr(), Rt.data_ptr(),
                                           qa.data_ptr(), ka.data_ptr(), _stream()), "mspi_mvit_qk_augment")
    if out is None:
        out = alloc(q.N, q.T, q.H, q.W, heads * hd, dev)
    d = AttnDesc()
    d.B, d.Hh, d.Nq, d.Nk, d.D, d.Dv, d.nmask = B, heads, Nq, Nk, DA, hd, 0
    d.q_sB, d.q_sH, d.q_sT = heads * Nq * DA, Nq * DA, DA
    d.k_sB, d.k_sH, d.k_sT = heads * Nk * DA, Nk * DA, DA
    d.v_sB, d.v_sH, d.v_sT = Nk * v.ld, hd, v.ld
    d.o_sB, d.o_sH, d.o_sT = Nq * out.ld, hd, out.ld
    d.scale = 1.0
    d.prec = _attn_prec(("mvit", heads, hd, Nq, Nk), lambda: qa.abs().max(), lambda: torch.maximum(ka.abs().max(), v.buf.abs().max()))
    assert q.ld == out.ld and q.dense and out.dense   # residual pooling reads q with o's strides
    with _Timed("attention", 2.0 * B * heads * Nq * Nk * (DA + hd), 4.0 * B * heads * (Nq * (DA + 2 * hd) + Nk * (DA + hd)),
                "B=%d h=%d Nq=%d Nk=%d d=%d+%d" % (B, heads, Nq, Nk, DA, hd)):
        _attn_launch(lib, d, qa.data_ptr(), ka.data_ptr(), v.ptr, q.ptr, None, None, None, out.ptr, dev)
    return out


def upsample(src, factor, dst=None, accumulate=False, act=ACT_NONE):
    lib = _lib.load()
    assert src.dense
    if dst is None:
        assert not accumulate
        dst = alloc(src.N, src.T, src.H * factor, src.W * factor, src.C, src.buf.device)
    assert (dst.N, dst.T, dst.H, dst.W) == (src.N, src.T, src.H * factor, src.W * factor) and dst.C == src.C
    assert dst.dense
    with _Timed("upsample", 0.0, 4.0 * (src.M + dst.M * (2 if accumulate else 1)) * src.C):
        check(lib.mspi_upsample_fwd(src.ptr, src.ld, dst.ptr, dst.ld, src.N * src.T, src.H, src.W, src.Cs, factor,
                                    1 if accumulate else 0, act, _stream()), "mspi_upsample_fwd")
    return dst


def rowgate(x, mask):
    lib = _lib.load()
    assert mask.M == x.M and mask.ld == 1
    assert x.dense and mask.dense
    with _Timed("rowgate", 0.0, 8.0 * x.M * x.C):
        check(lib.mspi_rowgate(x.ptr, x.ld, mask.ptr, x.M, x.Cs, _stream()), "mspi_rowgate")
    return x


def logsumexp_sub(t, N, L):
    lib = _lib.load()
    check(lib.mspi_logsumexp_sub(t.data_ptr(), N, L, _stream()), "mspi_logsumexp_sub")
    return t


def mean_rows(x, N, R, out):
    """x: CL with R = T*H*W rows per sample (token slabs allowed); out [N, C] tensor."""
    lib = _lib.load()
    assert x.N == N and x.T * x.H * x.W == R
    S = lib.mspi_mean_rows_slices(R)
    if S:      # long samples: two deterministic stages through a workspace (stream-ordered: safe to drop after the launch)
        ws = torch.empty(N * S * x.C, dtype=torch.float32, device=x.buf.device)
        check(lib.mspi_mean_rows_ws(x.ptr, x.ld, x.sN, out.data_ptr(), ws.data_ptr(), N, R, x.C, _stream()), "mspi_mean_rows_ws")
    else:
        check(lib.mspi_mean_rows(x.ptr, x.ld, x.sN, out.data_ptr(), N, R, x.C, _stream()), "mspi_mean_rows")
    return out


def neg_cosine(p, z, out, scale, accumulate):
    lib = _lib.load()
    assert p.ld == p.C and z.ld == z.C and p.dense and z.dense
    check(lib.mspi_neg_cosine(p.ptr, z.ptr, out.data_ptr(), p.M, p.C, float(scale), 1 if accumulate else 0, _stream()),
          "mspi_neg_cosine")
    return out


def add(a, b, y):
    lib = _lib.load()
    check(lib.mspi_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "mspi_add")
    return y


def permute(src, dims, strides, out=None):
    """Strided gather (mspi_permute_fwd): dense fp32 tensor of shape `dims` (<= 6, innermost contiguous on both sides)
    with out[i0..] = src.flat[sum_k i_k * strides[k]].  src: a CL (its buffer from .ptr on) or a flat torch tensor."""
    lib = _lib.load()
    if isinstance(src, CL):
        _need_gpu(src.buf)
        ptr, n_src, dev = src.ptr, src.buf.numel() - src.off, src.buf.device
    else:
        _need_gpu(src)
        ptr, n_src, dev = src.data_ptr(), src.numel(), src.device
    dims, strides = list(dims), list(strides)
    assert len(dims) == len(strides) <= 6
    pad = 6 - len(dims)
    d = _lib.PermuteDesc()
    d.dims[:] = [1] * pad + dims
    d.strides[:] = [0] * pad + strides
    d.src_elems = n_src
    n = 1
    for v in dims:
        n *= v
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=dev)
    assert out.numel() == n
    with _Timed("permute", 0.0, 8.0 * n):
        check(lib.mspi_permute_fwd(C.byref(d), ptr, out.data_ptr(), _stream()), "mspi_permute_fwd")
    return out


def gated_sum(srcs, logit, out=None):
    """sum_j softmax_j(logit[n, c*J+j]) * srcs[j][n,r,c] over dense CLs of one shape (J = len(srcs) in {2, 3})."""
    lib = _lib.load()
    a = srcs[0]
    for s_ in srcs:
        _need_gpu(s_.buf)
        assert s_.dense and s_.ld == a.C and (s_.N, s_.M, s_.C) == (a.N, a.M, a.C)
    if out is None:
        out = alloc(a.N, a.T, a.H, a.W, a.C, a.buf.device)
    J = len(srcs)
    assert logit.shape == (a.N, a.C * J) and logit.is_contiguous() and out.ld == a.C
    with _Timed("gated_sum", 0.0, 4.0 * (J + 1) * a.M * a.C):
        check(lib.mspi_gated_sum_fwd(srcs[0].ptr, srcs[1].ptr, srcs[2].ptr if J == 3 else None, logit.data_ptr(), out.ptr,
                                     a.N, a.M // a.N, a.C, J, _stream()), "mspi_gated_sum_fwd")
    return out


def postprocess_u8(logmap, out_hw, out=None):
    """[N,H,W] log-probability maps (GPU) -> uint8 [N,Ho,Wo] grey maps: blur, exp, resize, min-max, round.
    out: where the last kernel writes -- a device tensor (default: a new one) or a PINNED host tensor, which the kernel
    addresses directly (no D2H copy is queued; the caller synchronises on an event behind the launch before reading)."""
    lib = _lib.load()
    _need_gpu(logmap)
    N, H, W = logmap.shape
    Ho, Wo = out_hw
    logmap = logmap.contiguous()
    ws = torch.empty(lib.mspi_postprocess_workspace(N, H, W, Ho, Wo), dtype=torch.uint8, device=logmap.device)
    if out is None:
        out = torch.empty(N, Ho, Wo, dtype=torch.uint8, device=logmap.device)
    elif not (out.dtype == torch.uint8 and tuple(out.shape) == (N, Ho, Wo) and out.is_contiguous()
              and (out.is_cuda or out.is_pinned())):
        raise MspiError("postprocess_u8: out must be a contiguous uint8 [N,Ho,Wo] device tensor or pinned host tensor")
    check(lib.mspi_postprocess_u8(logmap.data_ptr(), out.data_ptr(), ws.data_ptr(), N, H, W, Ho, Wo, _stream()),
          "mspi_postprocess_u8")
    return out
