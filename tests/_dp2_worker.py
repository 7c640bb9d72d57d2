"""One rank of tests/test_gpu_dp2.py (not a test module).  argv: rank world outdir overlap_mode.

Runs the product's data-parallel step (stattn.dp: file rendezvous -> stattn_comm_init -> broadcast -> forward,
backward with the gradient regions handed to ncclAllReduce, stattn_allreduce_grads, clip + Adadelta) on this rank's
row shard and stores the summed gradient, the updated parameters and stattn_comm_stats."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SMALL = dict(dim=128, dim_word=64, n_words=211, ctxg_dim=128, ctxl_dim=96, ctxm_dim=64, ctxglm_dim=128)
ALPHA_C, DECAY_C, CLIP_C = 0.5, 1e-4, 0.05
B_GLOBAL = 5                                                      # uneven shards: 2 / 3 rows


def problem():
    from oracle import stattn_oracle as O
    opt = O.default_options(**SMALL)
    P = O.random_params(opt, seed=3, dtype=np.float32)
    batch = O.synthetic_batch(opt, B=B_GLOBAL, T=4, K=3, t=4, seed=3)
    return O, opt, P, batch


def main():
    rank, world, outdir, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=False)          # a rank that hangs says where (the caller's time limit is longer)
    import stattn
    from stattn import dp
    O, opt, P, batch = problem()
    dec = stattn.Decoder(opt, lt_mode=1)
    if rank == 0:
        dec.set_params(P)
    else:                                                         # other ranks start from garbage: the broadcast must fix it
        dec.set_params(O.random_params(opt, seed=77 + rank, dtype=np.float32))
    try:
        dp.init_comm(dec, rank=rank, world=world, path=os.path.join(outdir, "token"), seed=None)
    except Exception:
        if os.environ.get("STATTN_DP2_MAPS"):
            libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("hip", "hsa", "rccl", "stattn"))})
            print("loaded:", libs, "rccl:", dec.comm_library_path(), file=sys.stderr)
        raise
    dec.comm_set_overlap(mode)
    p_start = dec.get_params()
    dec.set_batch(**dp.shard_rows(batch, rank, world))
    dec.forward_train()
    dec.backward(nll_scale=1.0 / B_GLOBAL, alpha_c=ALPHA_C)
    dec.allreduce_grads()
    grads = dec.get_grads()
    stats = dec.comm_stats()
    loss = dp.GradReducer(dec).global_loss(dec.get_loss(DECAY_C), dec, DECAY_C)
    dec.update(decay_c=DECAY_C, clip_c=CLIP_C)
    # a second step through the step object (regions of step 2 must wait for nothing of step 1)
    dec.set_batch(**dp.shard_rows(batch, rank, world))
    dp.DataParallelStep(dec, global_batch=B_GLOBAL, alpha_c=ALPHA_C, decay_c=DECAY_C, clip_c=CLIP_C)()
    params = dec.get_params()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **{"g_" + k: v for k, v in grads.items()},
             **{"p_" + k: v for k, v in params.items()}, **{"s_" + k: v for k, v in p_start.items()})
    with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
        json.dump(dict(stats=stats, comm_info=list(dec.comm_info()), loss=float(loss), library=dec.comm_library_path()), f)
    dec.comm_destroy()
    dec.close()


if __name__ == "__main__":
    main()
