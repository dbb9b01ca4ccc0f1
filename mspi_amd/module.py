"""Base class of every HIP-backed nn.Module: torch layers are kept only as named parameter
holders (so state-dict keys equal the reference's), the arithmetic runs from a packed plan
that is rebuilt whenever the parameters may have changed."""
import torch
import torch.nn as nn

from ._lib import MspiError


class HipModule(nn.Module):
    def _invalidate(self):
        for m in self.modules():
            m.__dict__.pop("_pk", None)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._invalidate()
        return r

    def _pack(self):
        raise NotImplementedError

    @property
    def pk(self):
        """Packed weights (BN folded, channel-minor taps), built lazily on the parameters' device."""
        d = self.__dict__
        if "_pk" not in d:
            with torch.no_grad():
                d["_pk"] = self._pack()
        return d["_pk"]

    def _check_eval(self):
        if self.training:
            raise MspiError("%s is an inference engine (BatchNorm is folded into the convolutions): call .eval()"
                            % type(self).__name__)


def to_cl(x):
    """Accept a CL or an NCDHW tensor (e.g. from a user-supplied backbone) as a head input."""
    from .engine import CL, rup4
    if isinstance(x, CL):
        return x
    N, Cc, T, H, W = x.shape
    if x.stride(1) == 1 and x.stride(4) % 4 == 0 and x.stride(3) == W * x.stride(4) and x.stride(2) == H * x.stride(3):
        return CL(x, 0, N, T, H, W, Cc, x.stride(4), x.stride(0))
    ld = rup4(Cc)
    buf = torch.zeros(N, T, H, W, ld, dtype=torch.float32, device=x.device)
    buf[..., :Cc] = x.permute(0, 2, 3, 4, 1)
    return CL(buf.view(-1), 0, N, T, H, W, Cc, ld)
