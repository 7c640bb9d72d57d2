"""Data-parallel training step: the caption batch shards over videos, one process per GPU, the flat
gradient buffer is summed over the ranks by RCCL over xGMI *inside libstattn.so*
(stattn_comm_init / stattn_allreduce_grads, csrc/comm.cpp) -- the host side only has to hand every
rank the same 128-byte rendezvous token.

The reference is single-process; the exactness rule that makes N ranks
reproduce its global-batch gradient (model_attention.py:1129-1147) is:
    NLL term      mean over the GLOBAL batch  -> every rank scales its rows by 1/B_global, ranks SUM
    alpha reg.    a SUM over the batch        -> ranks SUM (scale 1)
    L2 decay      batch independent           -> added ONCE, after the reduce (stattn_update)
    clip          on the GLOBAL gradient norm -> after the reduce (stattn_update)
Adadelta is deterministic, so replicas that start equal stay bit-identical.

Stream ordering: the collective runs on the library's own streams -- the regions of the gradient
buffer are reduced on a side stream behind events recorded on the compute stream while backward is
still running, and stattn_allreduce_grads makes the compute stream wait for them -- so it is ordered
with backward and update whatever stream the handle was created on (no torch stream involved)."""
import os

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


def allreduce_sum(flat, group=None):
    """SUM all-reduce of a flat torch tensor through torch.distributed (any backend).  Only the CPU tests of
    the host logic (gloo) use it; on the GPU the reduce is the library's own (GradReducer)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def exchange_token(make_token, rank, world, group=None, path=None):
    """Hand rank 0's rendezvous token to every rank.  Through torch.distributed when a process group exists (any
    backend), else through a file `path` that all ranks can see (a plain-C launcher would use its own means)."""
    if world == 1:
        return make_token()
    try:
        import torch.distributed as dist
        have = dist.is_available() and dist.is_initialized()
    except ImportError:
        have = False
    if have:
        box = [make_token() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return box[0]
    if path is None:
        raise RuntimeError("exchange_token needs an initialised torch.distributed group or a shared file path")
    # File rendezvous.  Rank 0 writes the token atomically, every other rank acknowledges with `<path>.ack<rank>`, and
    # rank 0 removes all files once every rank has read: a finished run leaves nothing behind that a later run could
    # mistake for its own token.  A run that CRASHED in between can: give every run a fresh `path`, or set
    # STATTN_RENDEZVOUS_ID (folded into the file name) in the launcher.
    import time
    run_id = os.environ.get("STATTN_RENDEZVOUS_ID")
    if run_id:
        path = "%s.%s" % (path, run_id)
    if rank == 0:
        for f in ["%s.ack%d" % (path, r) for r in range(1, world)]:
            if os.path.exists(f):
                os.remove(f)
        tok = make_token()
        with open(path + ".tmp", "wb") as f:
            f.write(tok)
        os.replace(path + ".tmp", path)
        for _ in range(6000):
            if all(os.path.exists("%s.ack%d" % (path, r)) for r in range(1, world)):
                break
            time.sleep(0.01)
        else:
            raise RuntimeError("rendezvous: not every rank picked up %s" % path)
        for f in [path] + ["%s.ack%d" % (path, r) for r in range(1, world)]:
            os.remove(f)
        return tok
    for _ in range(6000):
        if os.path.exists(path):
            with open(path, "rb") as f:
                tok = f.read()
            if len(tok) == 128:
                with open("%s.ack%d" % (path, rank), "wb") as f:
                    f.write(b"1")
                return tok
        time.sleep(0.01)
    raise RuntimeError("rendezvous token %s never appeared" % path)


def init_comm(decoder, rank=None, world=None, group=None, path=None, seed=1234):
    """Bind `decoder` to an RCCL communicator over all ranks, broadcast rank 0's parameters so the replicas start
    identical, and give every rank its own dropout stream (the reference draws one mask per global batch; ranks
    that shared a seed would repeat the same mask on every shard).  seed=None leaves a single process's seed alone."""
    if rank is None or world is None:
        try:
            import torch.distributed as dist
            have = dist.is_available() and dist.is_initialized()
        except ImportError:
            have = False
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if have else (0, 1)   # no group: a single process
    if world > 1:
        decoder.comm_init(rank, world, exchange_token(decoder.comm_unique_id, rank, world, group, path))
        decoder.broadcast_params(0)
    if world > 1 or seed is not None:
        decoder.set_seed((1234 if seed is None else seed) + rank)
    return rank, world


class GradReducer(object):
    """The all-reduce half of f_grad_shared for one rank (model_attention.build_train_functions)."""

    def __init__(self, decoder, group=None, rank=None, world=None, path=None):
        self.dec = decoder
        _, n = decoder.comm_info()
        if n == 0:
            init_comm(decoder, rank, world, group, path, seed=None)      # a seed the caller set on one process stays

    def allreduce(self):
        self.dec.allreduce_grads()

    def global_loss(self, local_loss, decoder, decay_c):
        """Loss of the GLOBAL batch from the per-rank values: the NLL and regulariser parts add up over the ranks,
        the L2 term is counted once."""
        _, n = decoder.comm_info()
        if n < 2:
            return local_loss
        l2 = decoder.get_loss(decay_c) - decoder.get_loss(0.0) if decay_c else 0.0
        return float(decoder.allreduce_scalars([local_loss - l2])[0]) + l2


class DataParallelStep(object):
    """f_grad_shared + f_update (model_attention.py:1259, 1278) for one rank: forward, backward (regions of the
    gradient buffer start their RCCL sum while backward still runs), finish the sum, clip + Adadelta."""

    def __init__(self, decoder, global_batch, alpha_c=0.70602, decay_c=1e-4, clip_c=10.0, group=None):
        self.dec = decoder
        self.global_batch = int(global_batch)
        self.alpha_c, self.decay_c, self.clip_c = float(alpha_c), float(decay_c), float(clip_c)
        self.group = group

    def __call__(self):
        """One optimisation step on the batch staged with decoder.set_batch()."""
        d = self.dec
        d.forward_train()
        d.backward(nll_scale=1.0 / self.global_batch, alpha_c=self.alpha_c)
        d.allreduce_grads()
        d.update(decay_c=self.decay_c, clip_c=self.clip_c)
