"""Saliency metrics on the GPU: the reference's `utils/compute_saliency_metrics.py`
functions `kldiv`, `cc`, `similarity`, `nss` (same names, same [B,H,W] arguments, same batch-mean results) and the
bookkeeping of `utils/loss.py:SalLoss` -- all four metrics of a batch come from ONE launch of mspi_saliency_metrics.
Evaluation only (no autograd); there is no CPU fallback."""
import ctypes as C

import torch

from . import _lib
from ._lib import MspiError, check


def per_sample(pred, gt, fix=None, pred_is_log=False):
    """[B,4] tensor of (KL, CC, SIM, NSS) per sample; pred / gt / fix: [B,H,W] fp32 CUDA tensors."""
    lib = _lib.load()
    if not pred.is_cuda:
        raise MspiError("mspi_amd.metrics runs on the GPU only; there is no CPU fallback")
    if pred.shape != gt.shape or (fix is not None and fix.shape != pred.shape) or pred.dim() != 3:
        raise MspiError("metrics: pred %s, gt %s, fix %s must be equal [B,H,W] shapes" % (
            tuple(pred.shape), tuple(gt.shape), None if fix is None else tuple(fix.shape)))
    p, g = pred.float().contiguous(), gt.float().contiguous()
    f = None if fix is None else fix.float().contiguous()
    B, L = p.shape[0], p.shape[1] * p.shape[2]
    out = torch.empty(B, 4, dtype=torch.float32, device=p.device)
    check(lib.mspi_saliency_metrics(p.data_ptr(), g.data_ptr(), None if f is None else f.data_ptr(), out.data_ptr(), B, L,
                                    1 if pred_is_log else 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "mspi_saliency_metrics")
    return out


def kldiv(s_map, gt):
    return per_sample(s_map, gt)[:, 0].mean()


def cc(s_map, gt):
    return per_sample(s_map, gt)[:, 1].mean()


def similarity(s_map, gt):
    return per_sample(s_map, gt)[:, 2].mean()


def nss(s_map, gt):
    """gt is the fixation map here (compute_saliency_metrics.py:93-107)."""
    return per_sample(s_map, gt, fix=gt)[:, 3].mean()


class _Avg:
    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, v, n=1):
        self.sum += float(v) * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


class SalLoss:
    """utils/loss.py:6-49 for evaluation: forward(log_map, density[, fixations]) -> kl - cc [- 0.1 nss], with running
    averages of every term in .log (timm's AverageMeter upstream)."""

    def __init__(self):
        self.reset_records()

    def reset_records(self):
        self.log = {k: _Avg() for k in ("kl", "cc", "sim", "nss", "loss")}

    def forward(self, inputs, targets, fixations=None, targets2=None):
        m = per_sample(inputs, targets, fix=fixations, pred_is_log=True).mean(0)
        kl, c, sim, ns = (float(v) for v in m.tolist())
        loss = kl - c - (0.1 * ns if fixations is not None else 0.0)
        self.log["kl"].update(kl)
        self.log["cc"].update(c)
        self.log["sim"].update(sim)
        if fixations is not None:
            self.log["nss"].update(ns)
        self.log["loss"].update(loss)
        return torch.tensor(loss, device=inputs.device)

    __call__ = forward
