"""Deterministic synthetic weights / inputs shared by tests, bench.py and the golden generator.

There is no network, so every measurement uses random weights of the real architecture
(variance-preserving uniform draws under a seed) with *non-trivial* BatchNorm statistics, norm affines,
biases and layer-scale values -- otherwise eval-mode BN is the identity and ConvNeXt blocks
are x + 1e-6*f(x), which would make parity tests blind to most of the arithmetic.
"""
import zlib

import numpy as np
import torch
import torch.nn as nn


def randomize_(model, seed=0):
    """Re-draw EVERY parameter and buffer in place from a seeded CPU generator using uniform draws
    only: torch's uniform is an exact integer->float map, so the weights are bit-identical on every
    host (normal_/trunc_normal_ go through vectorised log/erfinv whose last bit depends on the CPU's
    ISA -- observed between the build container and the GPU box).  Order = named_modules() order."""
    g = torch.Generator().manual_seed(seed + 1000)

    def U(t, lo, hi):
        t.copy_(torch.rand(t.shape, generator=g) * (hi - lo) + lo)

    with torch.no_grad():
        for _, m in model.named_modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                U(m.weight, 0.5, 1.5)
                U(m.bias, -0.2, 0.2)
                U(m.running_mean, -0.2, 0.2)
                U(m.running_var, 0.5, 1.5)
            elif isinstance(m, nn.LayerNorm):
                U(m.weight, 0.5, 1.5)
                U(m.bias, -0.2, 0.2)
            elif isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)):
                fan_in = m.weight[0].numel()
                b = (3.0 / fan_in) ** 0.5            # variance 1/fan_in
                U(m.weight, -b, b)
                if m.bias is not None:
                    U(m.bias, -0.1, 0.1)
        for name, p in model.named_parameters():
            if name.endswith("gamma"):        # ConvNeXt layer scale (1e-6 at init)
                U(p, 0.1, 0.4)
            elif name.endswith("rel_pos_h") or name.endswith("rel_pos_w") or name.endswith("rel_pos_t") \
                    or name.endswith("relative_position_bias_table"):
                U(p, -0.3, 0.3)
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


def golden_hw(g):
    """Frame size of a model-level fixture (older fixtures are square and only carry `size`)."""
    return (int(g["H"]), int(g["W"])) if "H" in g.files else (int(g["size"]), int(g["size"]))


def golden_cfg(g, name):
    """The config a model-level fixture (oracle/gen_golden.py::_model_case) was generated with."""
    depths = [int(v) for v in g["swin_depths"]] if "swin_depths" in g.files else []
    return make_cfg(name, num_aud_tokens=int(g["num_aud_tokens"]), num_vis_tokens=int(g["num_vis_tokens"]),
                    swin_depths=depths or None)


def seeded(build, seed=0):
    """build() under torch.manual_seed(seed), then randomize_ (which re-draws everything), eval()."""
    torch.manual_seed(seed)
    m = build()
    randomize_(m, seed)
    return m.eval()


def synth_inputs(B, T=16, H=224, W=224, Wa=111, seed=0, device="cpu"):
    """clips / audio with zero mean, unit variance (ImageNet-normalised frames, per-column standardised
    spectrograms).  Irwin-Hall sums of 4 uniforms instead of randn: bit-identical on every host."""
    g = torch.Generator().manual_seed(seed + 77)

    def approx_normal(*shape):
        u = torch.rand(4, *shape, generator=g)
        return ((u[0] + u[1]) + (u[2] + u[3]) - 2.0) * (3.0 ** 0.5)

    clips = approx_normal(B, 3, T, H, W)
    audio = approx_normal(B, 1, 257, Wa)
    return clips.to(device), audio.to(device)


def feature_error(feat, g, key):
    """Relative max error of a feature map against a strided-sample golden (oracle/gen_golden.py::_feat_fixture);
    also checks shape and whole-tensor mean."""
    f = feat.detach().float().cpu()
    assert tuple(f.shape) == tuple(int(v) for v in g[key + "_shape"]), (tuple(f.shape), g[key + "_shape"])
    flat = f.reshape(-1)
    ref = torch.as_tensor(g[key + "_sample"])
    scale = float(g[key + "_absmax"])
    err = (flat[::int(g[key + "_stride"])] - ref).abs().max().item() / max(scale, 1e-6)
    merr = abs(flat.double().mean().item() - float(g[key + "_mean"])) / max(scale, 1e-6)
    return max(err, merr)


def damp_(model, suffixes, scale=0.25, prefix=""):
    """Scale the parameters whose names end with one of `suffixes` (and start with `prefix`) by a power of two (exact
    on every host).  Used for deep pre-norm residual stacks (UniFormer: 40 blocks) whose variance-preserving random
    branches would otherwise grow the activations to ~1e4, where fp32 rounding alone is 1e-3 absolute."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.startswith(prefix) and name.endswith(tuple(suffixes)):
                p.mul_(scale)
    return model


UNIFORMER_BRANCH_OUT = (".conv2.weight", ".conv2.bias", ".mlp.fc2.weight", ".mlp.fc2.bias", ".attn.proj.weight", ".attn.proj.bias",
                        ".pos_embed.weight", ".pos_embed.bias")


def condition_(model, name):
    """Per-encoder conditioning of the synthetic weights, applied after seeded(): today only UniFormer's residual
    branches are damped (see damp_).  `model` is the bare backbone or a saliency model holding it as `.visnet`."""
    if name == "uniformerb":
        damp_(model, UNIFORMER_BRANCH_OUT, 0.25, "visnet." if hasattr(model, "visnet") else "")
    return model
