"""CPU, world_size 2, gloo: the N>1 path -- grid sharding, packed-key all-reduce(MAX) and the
north-star-literal score all-reduce(SUM) -- reproduces the single-process arg-max (first maximum)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import navlab_dpe_sdr_amd as dpe
from tests import helpers


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scores, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, G = scores.shape
    b, e = dpe.sharding.shard_range(G, rank, world)
    local = scores[:, b:e]
    # what bcm_scan_kernel leaves in dpe_bcm_keys(): best key of the local shard per window
    keys = np.stack([dpe.sharding.pack_keys(local[w], b).max() for w in range(W)])
    best = dpe.sharding.allreduce_argmax(keys, dist).numpy()
    # the overlapped form bench.py uses: collective in flight while the next step's stage 1 is enqueued
    t, work = dpe.sharding.allreduce_argmax(keys.copy(), dist, async_op=True)
    work.wait()
    assert np.array_equal(t.numpy(), best)
    glob = dpe.sharding.allreduce_scores(local, b, G, dist).numpy()
    if rank == 0:
        out["idx"] = [dpe.sharding.unpack_key(k)[1] for k in best]
        out["score"] = [dpe.sharding.unpack_key(k)[0] for k in best]
        out["glob"] = glob
    dist.destroy_process_group()


def test_shard_ranges_cover_grid():
    for G, n in ((390625, 8), (10, 3), (7, 8), (1000000, 8)):
        r = [dpe.sharding.shard_range(G, k, n) for k in range(n)]
        assert r[0][0] == 0 and r[-1][1] == G
        assert all(r[k][1] == r[k + 1][0] for k in range(n - 1))
        assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def test_key_order_is_score_then_first_index():
    s = np.array([1.0, 3.5, 3.5, 0.0, 2.0], dtype=np.float32)
    k = dpe.sharding.pack_keys(s, 100)
    assert dpe.sharding.unpack_key(k.max()) == (3.5, 101)          # tie -> smaller index
    assert np.all(k >= 0)


@pytest.mark.timeout(120)
def test_two_rank_argmax_matches_single_process():
    case = helpers.make_case(seed=11, S=12500, K=4, G=3001, amp=200.0, W=2)   # odd size: ragged shards
    ref = helpers.run_oracle(case, 8, 32)
    scores = np.stack([p.astype(np.float32) for p in ref["pos"]])
    scores[1, 5] = scores[1, 2999] = scores[1].max() * 2                      # a tie across the two shards
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), scores, out), nprocs=2, join=True)
    for w in range(2):
        assert out["idx"][w] == int(np.argmax(scores[w]))                     # numpy argmax = first maximum
        assert out["score"][w] == float(scores[w].max())
    assert out["idx"][1] == 5
    assert np.array_equal(out["glob"], scores)                                # disjoint slices: the sum is exact
    assert [int(np.argmax(g)) for g in out["glob"]] == list(out["idx"])
