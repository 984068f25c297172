"""One process per GPU: photon-id slabs dealt round-robin to ranks, one all-reduce of the tallies.

Photons and atmospheres are independent, so the walk shards without any exchange while it runs.
Every atmosphere's photon-id range is cut into ``n_slabs`` fixed slabs (the cut does not depend on
the number of ranks); rank r walks the slabs s with s % world == r.  Each slab's partial sums are
produced by exactly one rank and are zero elsewhere, so a sum all-reduce (RCCL on GPUs, gloo in the
CPU tests) returns them unchanged -- adding zeros is exact -- and the final sum over the slab axis is
done in slab order on every rank.  The result is bit-for-bit the same for 1, 2, 4 or 8 ranks.
torch is used for device buffers and the collective only.
"""
import numpy as np

from . import _abi as abi
from .engine import SweepResult, make_params


def all_reduce_tallies(tensors, group=None):
    """In-place sum all-reduce of the tally tensors (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            if t is not None:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def result_from_arrays(slab_sums, counts, mu_hist, depth, n_atm, n_mu_bins, n_depth_bins, n_slabs):
    res = SweepResult(n_atm, n_mu_bins, n_depth_bins, n_slabs)
    res.slab_sums[...] = np.asarray(slab_sums, dtype=np.float64).reshape(n_atm, n_slabs, abi.NSUM)
    res.counts[...] = np.asarray(counts).reshape(n_atm, abi.NCNT).astype(np.uint64)
    res.mu_hist[...] = np.asarray(mu_hist).reshape(n_atm, n_mu_bins + 1).astype(np.uint64)
    if n_depth_bins:
        d = np.asarray(depth, dtype=np.float64).reshape(2, n_atm, n_depth_bins)
        res.depth_momentum[...] = d[0]
        res.depth_energy[...] = d[1]
    res.finalize_from_slabs()
    return res


class DistributedSweep:
    """Runs sweeps on this rank's GPU and combines the tallies over the process group."""

    def __init__(self, engine, group=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device("cuda", engine.device)
        self._bufs = {}

    def _buffers(self, n_atm, n_slabs, n_mu_bins, n_depth_bins):
        key = (n_atm, n_slabs, n_mu_bins, n_depth_bins)
        if key not in self._bufs:
            t = self.torch
            self._bufs[key] = (t.zeros(n_atm * n_slabs * abi.NSUM, dtype=t.float64, device=self.device),
                               t.zeros(n_atm * abi.NCNT, dtype=t.int64, device=self.device),
                               t.zeros(n_atm * (n_mu_bins + 1), dtype=t.int64, device=self.device),
                               t.zeros(max(2 * n_atm * n_depth_bins, 1), dtype=t.float64, device=self.device))
        return self._bufs[key]

    def enqueue(self, tau, n_photons, n_slabs=8, **kw):
        """Launch the sweep for this rank's slabs and the all-reduce; returns the device tensors."""
        if n_slabs % self.world:
            raise ValueError(f"n_slabs ({n_slabs}) must be a multiple of the number of ranks ({self.world})")
        prm = make_params(tau, n_photons, n_slabs=n_slabs, rank=self.rank, world=self.world, **kw)
        slab, counts, hist, depth = self._buffers(prm.n_atm, n_slabs, prm.n_mu_bins, prm.n_depth_bins)
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        self.engine.run_device(prm, slab.data_ptr(), counts.data_ptr(), hist.data_ptr(),
                               depth.data_ptr() if prm.n_depth_bins else None, stream)
        all_reduce_tallies([slab, counts, hist, depth if prm.n_depth_bins else None], self.group)
        return prm, (slab, counts, hist, depth)

    def run(self, tau, n_photons, n_slabs=8, **kw):
        prm, (slab, counts, hist, depth) = self.enqueue(tau, n_photons, n_slabs, **kw)
        self.torch.cuda.synchronize(self.device)
        return result_from_arrays(slab.cpu().numpy(), counts.cpu().numpy(), hist.cpu().numpy(), depth.cpu().numpy(),
                                  prm.n_atm, prm.n_mu_bins, prm.n_depth_bins, n_slabs)
