"""One process per GPU: photon-id slabs dealt round-robin to ranks, ONE all-reduce of the tallies.

Photons and atmospheres are independent, so the walk shards without any exchange while it runs.
Every atmosphere's photon-id range is cut into ``n_slabs`` fixed slabs (the cut does not depend on
the number of ranks); rank r walks the slabs s with s % world == r.  The sweep writes its slab
partial sums (fp64), integer tallies and escape-cosine histogram into ONE device buffer
(``TallyBuffer``), which is all-reduced once as 64-bit integers (RCCL on GPUs, gloo in the CPU
tests): the integer tallies add as integers, and every slab entry is non-zero on exactly one rank
and all-zero bits elsewhere, so adding the bit patterns returns that rank's double unchanged.  The
final sum over the slab axis is done in slab order on every rank.  The result is bit-for-bit the
same for 1, 2, 4 or 8 ranks.  Depth-resolved deposition (``n_depth_bins > 0``) travels the same way:
per slab, in the same buffer, as bit patterns.
torch is used for device buffers and the collective only.
"""
import numpy as np

from . import _abi as abi
from .engine import SweepResult, make_params


class TallyBuffer:
    """Slab sums, counts, escape-cosine histogram and per-slab depth bins of one sweep in one int64 tensor.

    Layout (64-bit words): [n_atm][n_slabs][NSUM] doubles | [n_atm][NCNT] u64 | [n_atm][n_mu_bins+1] u64 |
    [n_atm][n_slabs][2][n_depth_bins] doubles.
    ``words`` is what the collective sees; ``slab``, ``counts``, ``hist`` and ``depth`` are views of it."""

    def __init__(self, torch, n_atm, n_slabs, n_mu_bins, device, n_depth_bins=0):
        self.shape = (n_atm, n_slabs, n_mu_bins, n_depth_bins)
        n_slab, n_cnt, n_hist = n_atm * n_slabs * abi.NSUM, n_atm * abi.NCNT, n_atm * (n_mu_bins + 1)
        n_depth = n_atm * n_slabs * 2 * n_depth_bins
        self.words = torch.zeros(n_slab + n_cnt + n_hist + n_depth, dtype=torch.int64, device=device)
        self.slab = self.words[:n_slab].view(torch.float64)
        self.counts = self.words[n_slab:n_slab + n_cnt]
        self.hist = self.words[n_slab + n_cnt:n_slab + n_cnt + n_hist]
        self.depth = self.words[n_slab + n_cnt + n_hist:].view(torch.float64) if n_depth else None

    def pointers(self):
        """Device addresses of the four parts, as rt_run_device wants them (the last is None without depth bins)."""
        return (self.slab.data_ptr(), self.counts.data_ptr(), self.hist.data_ptr(),
                None if self.depth is None else self.depth.data_ptr())

    def all_reduce(self, group=None):
        """The run's only collective (no-op for a single process).  With the gloo backend (CPU rehearsal of
        N > 1) the words are staged through host memory."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return
        if self.words.is_cuda and dist.get_backend(group) == "gloo":
            host = self.words.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            self.words.copy_(host)
        else:
            dist.all_reduce(self.words, op=dist.ReduceOp.SUM, group=group)

    def result(self):
        n_atm, n_slabs, n_mu_bins, n_depth_bins = self.shape
        return result_from_arrays(self.slab.cpu().numpy(), self.counts.cpu().numpy(), self.hist.cpu().numpy(),
                                  None if self.depth is None else self.depth.cpu().numpy(), n_atm, n_mu_bins,
                                  n_depth_bins, n_slabs)


def result_from_arrays(slab_sums, counts, mu_hist, depth, n_atm, n_mu_bins, n_depth_bins, n_slabs):
    res = SweepResult(n_atm, n_mu_bins, n_depth_bins, n_slabs)
    res.slab_sums[...] = np.asarray(slab_sums, dtype=np.float64).reshape(n_atm, n_slabs, abi.NSUM)
    res.counts[...] = np.asarray(counts).reshape(n_atm, abi.NCNT).astype(np.uint64)
    res.mu_hist[...] = np.asarray(mu_hist).reshape(n_atm, n_mu_bins + 1).astype(np.uint64)
    if n_depth_bins:
        res.slab_depth[...] = np.asarray(depth, dtype=np.float64).reshape(n_atm, n_slabs, 2, n_depth_bins)
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
            self._bufs[key] = TallyBuffer(self.torch, n_atm, n_slabs, n_mu_bins, self.device, n_depth_bins)
        return self._bufs[key]

    def enqueue(self, tau, n_photons, n_slabs=8, **kw):
        """Launch the sweep for this rank's slabs and the all-reduce; returns the device buffer."""
        if n_slabs % self.world:
            raise ValueError(f"n_slabs ({n_slabs}) must be a multiple of the number of ranks ({self.world})")
        prm = make_params(tau, n_photons, n_slabs=n_slabs, rank=self.rank, world=self.world, **kw)
        buf = self._buffers(prm.n_atm, n_slabs, prm.n_mu_bins, prm.n_depth_bins)
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        self.engine.run_device(prm, *buf.pointers(), stream)
        buf.all_reduce(self.group)
        return prm, buf

    def run(self, tau, n_photons, n_slabs=8, **kw):
        prm, buf = self.enqueue(tau, n_photons, n_slabs, **kw)
        self.torch.cuda.synchronize(self.device)
        return buf.result()
