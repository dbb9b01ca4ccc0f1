"""Saliency metrics: oracle vs the reference's functions (golden), HIP kernel vs oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import restate as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold():
    z = np.load(os.path.join(GOLD, "saliency_metrics.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_oracle_metrics_match_reference():
    """fixture: kldiv / cc / similarity / nss of utils/compute_saliency_metrics.py run on the reference itself."""
    g = _gold()
    ora = R.saliency_metrics(g["pred"], g["gt"], g["fix"])
    assert torch.equal(ora, g["per_sample"])
    assert (ora.mean(0) - g["ref_means"]).abs().max().item() < 1e-6


@pytest.mark.gpu
def test_hip_metrics_vs_golden(dev):
    from mspi_amd import metrics as M
    g = _gold()
    got = M.per_sample(g["pred"].to(dev), g["gt"].to(dev), g["fix"].to(dev)).cpu()
    ref = g["per_sample"]
    assert ((got - ref).abs() / ref.abs().clamp_min(1e-3)).max().item() < 2e-5
    # the four reference-named entry points return the batch means
    for fn, col, args in ((M.kldiv, 0, ("pred", "gt")), (M.cc, 1, ("pred", "gt")), (M.similarity, 2, ("pred", "gt")),
                          (M.nss, 3, ("pred", "fix"))):
        v = fn(g[args[0]].to(dev), g[args[1]].to(dev)).item()
        assert abs(v - g["ref_means"][col].item()) < 2e-5 * max(1.0, abs(g["ref_means"][col].item()))


@pytest.mark.gpu
def test_hip_metrics_full_size_log_input_and_loss(dev):
    """Model-sized maps (8 x 224 x 224) given as log-probabilities, as SalLoss feeds them; bitwise repeatable."""
    from mspi_amd import metrics as M
    gen = torch.Generator().manual_seed(3)
    B, H, W = 8, 224, 224
    logmap = torch.log_softmax((torch.rand(B, H * W, generator=gen) * 8), 1).view(B, H, W)
    gt = torch.rand(B, H, W, generator=gen) ** 6
    fix = (torch.rand(B, H, W, generator=gen) > 0.999).float()
    ref = R.saliency_metrics(logmap.double().exp().float(), gt, fix)
    a = M.per_sample(logmap.to(dev), gt.to(dev), fix.to(dev), pred_is_log=True)
    b = M.per_sample(logmap.to(dev), gt.to(dev), fix.to(dev), pred_is_log=True)
    assert torch.equal(a, b)
    assert ((a.cpu() - ref).abs() / ref.abs().clamp_min(1e-3)).max().item() < 1e-4
    crit = M.SalLoss()
    loss = crit(logmap.to(dev), gt.to(dev), fix.to(dev)).item()
    m = ref.mean(0)
    assert abs(loss - (m[0] - m[1] - 0.1 * m[3]).item()) < 1e-4
    assert abs(crit.log["sim"].avg - m[2].item()) < 1e-4
