"""Multi-GPU sharding of the manifold grid: contiguous index ranges per rank, one
exchange step -- the arg-max -- done as an integer all-reduce(MAX) over packed keys.

key = (float32 score bits << 32) | (2^32-1 - global index): scores are >= 0, so the IEEE bits are
monotone and the key fits a non-negative int64; the larger score wins and equal scores resolve to the
SMALLER global index, i.e. thrust::max_element's first maximum (batchcorrmanifold.cu:2589-2590),
independent of the number of ranks.  The same packing is produced on-device by bcm_scan_kernel.
"""
import numpy as np


def shard_range(G_global, rank, world):
    """[begin, end) of rank's contiguous slice; keeps the reference's index order (t fastest)."""
    base, rem = divmod(int(G_global), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_keys(scores, index_offset=0):
    """numpy twin of the device packing: float32 scores [n] -> int64 keys [n]."""
    s = np.ascontiguousarray(scores, dtype=np.float32)
    bits = s.view(np.uint32).astype(np.uint64)
    idx = np.uint64(0xFFFFFFFF) - (np.arange(s.size, dtype=np.uint64) + np.uint64(index_offset))
    return ((bits << np.uint64(32)) | idx).view(np.int64)


def unpack_key(key):
    key = np.uint64(np.int64(key).view(np.uint64)) if not isinstance(key, np.uint64) else key
    score = np.array([int(key) >> 32], dtype=np.uint32).view(np.float32)[0]
    return float(score), int(0xFFFFFFFF - (int(key) & 0xFFFFFFFF))


def allreduce_argmax(local_keys, dist, device=None, async_op=False):
    """local_keys: int64 array/tensor of this rank's best key per (window, manifold).  Returns the
    global best keys on every rank (one collective per call).  async_op=True returns (tensor, work): the caller
    waits on `work` only before the keys are needed or their buffer is reused -- with the two alternating key
    sets of BatchCorrManifold that is just before the NEXT manifold scan, so the collective hides behind the next
    step's correlator kernels."""
    import torch
    t = local_keys if isinstance(local_keys, torch.Tensor) else torch.as_tensor(np.asarray(local_keys), device=device)
    if async_op:
        return t, dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def allreduce_scores(local_scores, begin, G_global, dist, device=None):
    """North-star-literal exchange: zero-initialised global vector, own slice filled, all-reduce(SUM)."""
    import torch
    loc = local_scores if isinstance(local_scores, torch.Tensor) else torch.as_tensor(np.asarray(local_scores), device=device)
    glob = torch.zeros(loc.shape[:-1] + (int(G_global),), dtype=loc.dtype, device=loc.device)
    glob[..., begin:begin + loc.shape[-1]] = loc
    dist.all_reduce(glob, op=dist.ReduceOp.SUM)
    return glob
