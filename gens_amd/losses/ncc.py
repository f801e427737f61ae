"""compute_LNCC on the device in one kernel (gens_lncc_fwd / gens_lncc_bwd) instead of five grouped 11x11 convolutions.

Mirrors /root/reference/models/losses/ncc.py:7-55: same name, arguments and result, differentiable w.r.t. both inputs
(first order, which is all loss.py:36-38 needs)."""
import torch

from .. import lib as L

_f32 = torch.float32


class _LNCC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref_gray, src_grays):
        assert ref_gray.ndim == 4 and src_grays.ndim == 4 and ref_gray.shape[0] == 1 and ref_gray.shape[1:] == src_grays.shape[1:], \
            "ref_gray (1,B,P,C) and src_grays (S,B,P,C) expected (ncc.py:8-9)"
        ref = ref_gray.detach().to(_f32).contiguous()
        src = src_grays.detach().to(_f32).contiguous()
        s, b, p, c = src.shape
        if s < 2:                                                  # torch.topk(ncc, 2, dim=1) of ncc.py:54 raises the same way
            raise RuntimeError("compute_LNCC: selected index k out of range (two source views are needed, ncc.py:54)")
        ncc = torch.empty(b, 1, device=ref.device, dtype=_f32)
        sel = torch.empty(b, 2, device=ref.device, dtype=torch.int32)
        L.call("gens_lncc_fwd", L.ptr(ref), L.ptr(src), b, s, p, c, L.ptr(ncc), L.ptr(sel, torch.int32), L.stream(),
               nbytes=b * p * c * (s + 1) * 4 + b * 12)
        ctx.save_for_backward(ref, src, sel)
        return ncc

    @staticmethod
    def backward(ctx, g_ncc):
        ref, src, sel = ctx.saved_tensors
        s, b, p, c = src.shape
        g_ref, g_src = torch.empty_like(ref), torch.empty_like(src)
        L.call("gens_lncc_bwd", L.ptr(ref), L.ptr(src), L.ptr(g_ncc.to(_f32).contiguous()), L.ptr(sel, torch.int32), b, s, p, c,
               L.ptr(g_ref), L.ptr(g_src), L.stream(), nbytes=2 * b * p * c * (s + 1) * 4 + b * 12)
        return g_ref, g_src


def compute_LNCC(ref_gray, src_grays):
    """ref_gray (1, B, 121, n), src_grays (nsrc, B, 121, n) -> (B, 1)   (ncc.py:7-55)."""
    return _LNCC.apply(ref_gray, src_grays)
