"""Data-parallel training step: the caption batch shards over videos, one process per GPU,
one RCCL all-reduce (torch.distributed backend "nccl" IS RCCL on ROCm) of the flat gradient
buffer over xGMI per step.

The reference is single-process; the exactness rule that makes N ranks
reproduce its global-batch gradient (model_attention.py:1129-1147) is:
    NLL term      mean over the GLOBAL batch  -> every rank scales its rows by 1/B_global, ranks SUM
    alpha reg.    a SUM over the batch        -> ranks SUM (scale 1)
    L2 decay      batch independent           -> added ONCE, after the reduce (stattn_update)
    clip          on the GLOBAL gradient norm -> after the reduce (stattn_update)
Adadelta is deterministic, so replicas that start equal stay bit-identical."""
import numpy as np


def shard_rows(batch, rank, world):
    """Contiguous row shard of a prepare_data() 8-tuple dict (x/mask are (t, m): shard axis 1)."""
    m = batch['x'].shape[1]
    lo = (m * rank) // world
    hi = (m * (rank + 1)) // world
    out = {}
    for k, v in batch.items():
        out[k] = np.ascontiguousarray(v[:, lo:hi] if k in ('x', 'mask') else v[lo:hi])
    return out


class DeviceVector(object):
    """Zero-copy view of a device buffer of the library for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr='<f4', data=(int(ptr), False), version=2)


def grad_tensor(decoder):
    """torch tensor aliasing the decoder's flat gradient buffer (no copy)."""
    import torch
    ptr, n = decoder.grad_buffer_dev()
    return torch.as_tensor(DeviceVector(ptr, n), device='cuda')


def allreduce_sum(flat, group=None):
    """SUM all-reduce of one flat gradient vector (torch tensor, any backend: nccl on GPU, gloo in the
    CPU tests).  One collective for the whole parameter set: 171 MB at the MSVD config; over xGMI
    (7 point-to-point links x ~153 GB/s per GPU) a ring is per-link bound at ~2 ms."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class DataParallelStep(object):
    """f_grad_shared + f_update (model_attention.py:1259, 1278) for one rank."""

    def __init__(self, decoder, global_batch, alpha_c=0.70602, decay_c=1e-4, clip_c=10.0, group=None):
        self.dec = decoder
        self.global_batch = int(global_batch)
        self.alpha_c, self.decay_c, self.clip_c = float(alpha_c), float(decay_c), float(clip_c)
        self.group = group
        self._gt = None

    def __call__(self):
        """One optimisation step on the batch staged with decoder.set_batch()."""
        d = self.dec
        d.forward_train()
        d.backward(nll_scale=1.0 / self.global_batch, alpha_c=self.alpha_c)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if self._gt is None:
                self._gt = grad_tensor(d)
            allreduce_sum(self._gt, self.group)
        d.update(decay_c=self.decay_c, clip_c=self.clip_c)
