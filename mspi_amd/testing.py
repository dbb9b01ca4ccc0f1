"""Deterministic synthetic weights / inputs shared by tests, bench.py and the golden generator.

There is no network, so every measurement uses random-init weights of the real architecture
(reference initialisers under a seed) with *non-trivial* BatchNorm statistics, norm affines,
biases and layer-scale values -- otherwise eval-mode BN is the identity and ConvNeXt blocks
are x + 1e-6*f(x), which would make parity tests blind to most of the arithmetic.
"""
import zlib

import numpy as np
import torch
import torch.nn as nn


def randomize_(model, seed=0):
    """In-place, CPU generator, order = named_modules() order (stable across machines)."""
    g = torch.Generator().manual_seed(seed + 1000)

    def U(t, lo, hi):
        t.copy_(torch.rand(t.shape, generator=g) * (hi - lo) + lo)

    def Nrm(t, std):
        t.copy_(torch.randn(t.shape, generator=g) * std)

    with torch.no_grad():
        for _, m in model.named_modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                U(m.weight, 0.5, 1.5)
                Nrm(m.bias, 0.1)
                Nrm(m.running_mean, 0.1)
                U(m.running_var, 0.5, 1.5)
            elif isinstance(m, nn.LayerNorm):
                U(m.weight, 0.5, 1.5)
                Nrm(m.bias, 0.1)
            elif isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)) and m.bias is not None:
                Nrm(m.bias, 0.05)
        for name, p in model.named_parameters():
            if name.endswith("gamma"):        # ConvNeXt layer scale (1e-6 at init)
                U(p, 0.1, 0.4)
            elif name.endswith("rel_pos_h") or name.endswith("rel_pos_w") or name.endswith("rel_pos_t") \
                    or name.endswith("relative_position_bias_table"):
                Nrm(p, 0.2)
            elif name == "readout.12.weight":
                # random-init maps are almost flat (std ~0.008 nat); a trained model's log-probability map spans
                # several nats.  Scale the last conv so the synthetic map has that dynamic range and a 1e-3
                # max-abs parity bound actually constrains the arithmetic.
                p.mul_(100.0)
    return model


def sd_checksum(sd):
    """Order-independent fingerprint of a state dict (float tensors only)."""
    acc = 0
    for k in sorted(sd):
        v = sd[k]
        if torch.is_tensor(v) and v.is_floating_point():
            acc = zlib.crc32(np.ascontiguousarray(v.detach().cpu().float().numpy()).tobytes(), acc)
    return acc


def make_cfg(name, num_aud_tokens=36, num_vis_tokens=None, swin_depths=None):
    """A private clone of mspi_amd.config.cfg pointed at motion encoder `name`."""
    from .config import cfg, select_model
    c = cfg.clone()
    select_model(name, c)
    c.MODEL.NUM_AUD_TOKENS = num_aud_tokens
    if num_vis_tokens is not None:
        c.MODEL.NUM_VIS_TOKENS = dict(c.MODEL.NUM_VIS_TOKENS)
        c.MODEL.NUM_VIS_TOKENS[name] = num_vis_tokens
    if swin_depths is not None:
        c.MODEL.SWIN.DEPTHS = list(swin_depths)
    return c


def seeded(build, seed=0):
    """build() under torch.manual_seed(seed) (reference initialisers), then randomize_, eval()."""
    torch.manual_seed(seed)
    m = build()
    randomize_(m, seed)
    return m.eval()


def synth_inputs(B, T=16, H=224, W=224, Wa=111, seed=0, device="cpu"):
    """clips ~ N(0,1) (ImageNet-normalised range), audio ~ N(0,1) (per-column standardised spectrograms)."""
    g = torch.Generator().manual_seed(seed + 77)
    clips = torch.randn(B, 3, T, H, W, generator=g)
    audio = torch.randn(B, 1, 257, Wa, generator=g)
    return clips.to(device), audio.to(device)
