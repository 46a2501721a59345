"""Mini-batch OT pairing: host-side mirror of ``flocoder/ot.py`` (ot.py:63-84).  The POT / torchcfm variants upstream are
dead code (the wrapper hard-wires the greedy matcher, ot.py:80-84); only that matcher exists here."""
import torch

from . import _binding as B


def compute_ot_pairing_approximate(source, target):
    """ot.py:63-78 on the GPU: L2 distance matrix, then the sequential greedy sweep (first minimum among unused targets).
    Returns an int64 permutation on the inputs' device."""
    if not source.is_cuda:
        raise RuntimeError("flocoder_amd.compute_ot_pairing runs on MI355X (gfx950) only; there is no CPU path")
    bsz = source.shape[0]
    s = source.reshape(bsz, -1).contiguous().float()
    t = target.reshape(bsz, -1).contiguous().float()
    dist = torch.empty(bsz, bsz, device=s.device, dtype=torch.float32)
    perm = torch.empty(bsz, device=s.device, dtype=torch.int64)
    B.check(B.lib().fc_ot_pairing(B.ptr(s), B.ptr(t), bsz, s.shape[1], B.ptr(dist), B.ptr(perm), B.current_stream(s.device)))
    return perm


def compute_ot_pairing(source, target, debug=False):
    """ot.py:80-84."""
    return compute_ot_pairing_approximate(source, target)
