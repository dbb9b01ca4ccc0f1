"""Clip-loop pre-processing on the GPU: log-spectrogram windows and frame
resize + normalise through the C ABI (csrc/preproc.hip).  Host code here only builds small tables."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import MspiError, check

PRECISION_BITS = 32 - 8 - 2          # PIL's fixed point for 8-bit resampling


def pil_bilinear_coeffs(in_size, out_size):
    """PIL's precompute_coeffs (bilinear, support 1 scaled by the shrink factor = antialiasing) followed by
    normalize_coeffs_8bpc: (bounds int32 [out][2] = (first input index, taps), kk int32 [out][ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = np.zeros(xmax, np.float64)
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            k[x] = 1.0 - t if t < 1.0 else 0.0
        ww = k.sum()
        if ww != 0.0:
            k = k / ww
        bounds[xx] = (xmin, xmax)
        kk[xx, :xmax] = [int(v * (1 << PRECISION_BITS) + (0.5 if v >= 0 else -0.5)) for v in k]
    return bounds, kk, ksize


_COEFFS = {}


def _coeffs(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _COEFFS:
        b, k, ks = pil_bilinear_coeffs(in_size, out_size)
        _COEFFS[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return _COEFFS[key]


def resize_normalize(rgb_u8, out_hw, mean, std, out=None):
    """rgb_u8: uint8 [Hin, Win, 3] on the GPU -> fp32 [3, Hout, Wout]: PIL bilinear resize (bit-exact), /255, -mean, /std."""
    lib = _lib.load()
    if not rgb_u8.is_cuda:
        raise MspiError("resize_normalize runs on the GPU only (tensor on %s)" % rgb_u8.device)
    assert rgb_u8.dtype == torch.uint8 and rgb_u8.dim() == 3 and rgb_u8.shape[2] == 3 and rgb_u8.is_contiguous()
    Hin, Win = rgb_u8.shape[:2]
    Hout, Wout = out_hw
    dev = rgb_u8.device
    hb, hk, hks = _coeffs(Win, Wout, dev)
    vb, vk, vks = _coeffs(Hin, Hout, dev)
    tmp = torch.empty(Hin * Wout * 3, dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty(3, Hout, Wout, dtype=torch.float32, device=dev)
    assert out.shape == (3, Hout, Wout) and out.stride(2) == 1 and out.stride(1) == Wout
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(lib.mspi_resize_norm_fwd(rgb_u8.data_ptr(), Hin, Win, tmp.data_ptr(), out.data_ptr(), out.stride(0), Hout, Wout,
                                   hb.data_ptr(), hk.data_ptr(), hks, vb.data_ptr(), vk.data_ptr(), vks, m, s,
                                   torch.cuda.current_stream().cuda_stream), "mspi_resize_norm_fwd")
    return out


_WINDOW = {}


def log_spectrogram(wave, segments, Wa=111):
    """wave: fp32 [n] 16 kHz mono on the GPU; segments: list of (start, length, reversed) -> [B, 1, 257, Wa] standardised
    log-spectrogram windows (inference.py:24-63 upstream), padded with 0.02."""
    lib = _lib.load()
    if not wave.is_cuda:
        raise MspiError("log_spectrogram runs on the GPU only (tensor on %s)" % wave.device)
    wave = wave.reshape(-1).contiguous()
    seg_host = np.ascontiguousarray(np.asarray(segments, dtype=np.int32).reshape(-1, 3))
    B = seg_host.shape[0]
    dev = wave.device
    if str(dev) not in _WINDOW:
        _WINDOW[str(dev)] = torch.hann_window(512, dtype=torch.float32).to(dev)
    seg = torch.from_numpy(seg_host).to(dev)
    out = torch.empty(B, 1, 257, Wa, dtype=torch.float32, device=dev)
    check(lib.mspi_logspec_fwd(wave.data_ptr(), wave.numel(), seg.data_ptr(), seg_host.ctypes.data, B, _WINDOW[str(dev)].data_ptr(),
                               out.data_ptr(), Wa, torch.cuda.current_stream().cuda_stream), "mspi_logspec_fwd")
    return out
