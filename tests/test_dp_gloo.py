"""World-size-2 test of the data-parallel exactness rule on CPU (gloo): the product's shard /
all-reduce host logic (stattn.dp) applied to oracle-computed shard gradients must reproduce the
single-process full-batch gradient and Adadelta step."""
import os
import socket
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TINY = dict(dim=16, dim_word=8, n_words=11, ctxg_dim=16, ctxl_dim=12, ctxm_dim=10, ctxglm_dim=16)
ALPHA_C, DECAY_C, CLIP_C = 0.70602, 1e-3, 0.5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stattn
    from stattn import dp
    from oracle import stattn_oracle as O
    from oracle import stattn_oracle_grad as OG
    opt = O.default_options(**TINY)
    P = O.random_params(opt, seed=4, dtype=np.float64)
    batch = O.synthetic_batch(opt, B=5, T=4, K=3, t=5, seed=8, dtype=np.float64)   # 5 rows: uneven shards 2 / 3
    shard = dp.shard_rows(batch, rank, world)
    B_global = batch['x'].shape[1]
    g = OG.loss_and_grads(P, opt, shard, alpha_c=ALPHA_C, nll_scale=1.0 / B_global)['grads']
    flat = torch.from_numpy(np.concatenate([np.asarray(v).reshape(-1) for v in g.values()]))
    dp.allreduce_sum(flat)                                   # the product's collective wrapper
    flat = flat.numpy()
    # decay once, after the reduce; clip on the global norm; Adadelta
    off = 0
    grads = OrderedDict()
    for k, v in P.items():
        n = v.size
        grads[k] = flat[off:off + n].reshape(v.shape) + 2 * DECAY_C * v
        off += n
    grads = O.clip_grads(grads, CLIP_C)
    rg2 = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
    ru2 = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
    O.adadelta_update(P, grads, rg2, ru2)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **{k: np.asarray(v) for k, v in P.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step(tmp_path):
    from oracle import stattn_oracle as O
    from oracle import stattn_oracle_grad as OG
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    opt = O.default_options(**TINY)
    P = O.random_params(opt, seed=4, dtype=np.float64)
    batch = O.synthetic_batch(opt, B=5, T=4, K=3, t=5, seed=8, dtype=np.float64)
    g = OG.loss_and_grads(P, opt, batch, alpha_c=ALPHA_C, decay_c=DECAY_C)['grads']      # single process, full batch
    g2 = sum(float((v ** 2).sum()) for v in g.values())
    assert g2 > CLIP_C ** 2                                                              # the clip is active
    rg2 = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
    ru2 = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
    O.adadelta_update(P, O.clip_grads(g, CLIP_C), rg2, ru2)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    for k in P:
        np.testing.assert_array_equal(r0[k], r1[k])                 # replicas stay bit-identical
        np.testing.assert_allclose(r0[k], P[k], rtol=1e-9, atol=1e-12)


def _token_worker(rank, world, port, outdir, use_group):
    from stattn import dp
    if use_group:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    made = []

    def make():                      # stands in for Decoder.comm_unique_id (needs librccl + a GPU): rank 0 only
        made.append(1)
        return bytes(range(128))
    tok = dp.exchange_token(make, rank, world, path=None if use_group else os.path.join(outdir, "token"))
    assert tok == bytes(range(128)) and len(made) == (1 if rank == 0 else 0)
    open(os.path.join(outdir, "ok%d_%d" % (use_group, rank)), "w").close()
    if use_group:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("use_group", [1, 0])
def test_rendezvous_token_reaches_every_rank(tmp_path, use_group):
    """The only thing the host has to do for the in-library RCCL communicator: hand rank 0's 128-byte token to all
    ranks -- through a torch.distributed group when one exists, else through a shared file."""
    mp.spawn(_token_worker, args=(2, _free_port(), str(tmp_path), use_group), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d_%d" % (use_group, r))) for r in range(2))


def test_shard_rows_partitions_every_row_once():
    from stattn import dp
    from oracle import stattn_oracle as O
    opt = O.default_options(**TINY)
    b = O.synthetic_batch(opt, B=7, T=3, K=2, t=4, seed=1)
    for world in (1, 2, 3, 4, 8):
        rows = 0
        xs = []
        for r in range(world):
            s = dp.shard_rows(b, r, world)
            assert s['x'].shape[0] == 4 and s['ctxl'].shape[0] == s['x'].shape[1] == s['mask'].shape[1]
            assert s['x'].flags['C_CONTIGUOUS']
            rows += s['x'].shape[1]
            xs.append(s['x'])
        assert rows == 7
        np.testing.assert_array_equal(np.concatenate(xs, 1), b['x'])


def test_all_reduce_regions_tile_the_gradient_buffer_for_every_option_variant():
    """The five regions stattn_backward hands to the overlapped all-reduce (csrc/handle.h GRAD_REGIONS, summed by comm.cpp as each
    becomes final) must cover every float of the flat gradient buffer exactly once, whatever optional parameters the configuration
    has (selector / ctx2out / prev2out off, odd vocabulary sizes that the device layout pads) -- until round 6 only a runtime check
    inside stattn_allreduce_grads (comm_covered != nflat) guarded this.  Host-only: the table is evaluated without a device."""
    import itertools
    from stattn import _native
    from oracle import stattn_oracle as O
    seen = set()
    for sel, c2o, p2o, V, D, E in itertools.product((1, 0), (1, 0), (1, 0), (12000, 211, 128), (1024, 64), (512, 64)):
        opt = dict(dim=D, dim_word=E, n_words=V, ctxg_dim=D, ctxl_dim=96, ctxm_dim=64, selector=sel, use_dropout=1, prev2out=p2o, ctx2out=c2o)
        regions, nflat = _native.grad_regions(opt)
        assert len(regions) == 5
        ordered = sorted(regions)
        assert ordered[0][0] == 0
        for (o0, l0), (o1, _) in zip(ordered, ordered[1:]):
            assert l0 > 0 and o0 + l0 == o1, (opt, regions)          # no gap, no overlap
        assert ordered[-1][0] + ordered[-1][1] == nflat, (opt, regions, nflat)
        # the buffer is at least as long as the parameters the oracle's table lists for this variant (padding only adds)
        shapes = O.param_shapes(O.default_options(dim=D, dim_word=E, n_words=V, ctxg_dim=D, ctxl_dim=96, ctxm_dim=64, ctxglm_dim=D,
                                                  selector=bool(sel), ctx2out=bool(c2o), prev2out=bool(p2o)))
        assert nflat >= sum(int(np.prod(s)) if len(s) else 1 for s in shapes.values())
        # completion order: the readout first (final before the reverse scan), the embedding last
        assert regions[0][0] + regions[0][1] == nflat and regions[-1][0] == 0
        seen.add(nflat)
    assert len(seen) > 8
