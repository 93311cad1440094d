"""The N>1 host logic with two gloo ranks on the CPU: user sharding, all-reduced evaluation sums
against the single-process metrics, and the combination of per-rank item tables."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_py as orc
from recommendation_amd.dist_util import combine_item_tables, metrics_from_sums, shard_range, sums_from_counts
from tests.golden_util import load


def _worker(rank, world, port, counts, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    U = len(counts)
    ub, ue = shard_range(U, rank, world)
    # evaluation: each rank reduces its own users, the sums are all-reduced (what bench.py does with rec_eval_sums)
    s = torch.tensor(sums_from_counts(counts[ub:ue]))
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    # training: every rank moves a replicated item table by its own delta
    base = torch.arange(12, dtype=torch.float64).reshape(3, 4).clone()
    table = base + (rank + 1) * 0.5
    combine_item_tables([table], [base], world, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    if rank == 0:
        out.put((s.numpy().copy(), table.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_eval_sums_and_table_combination(world):
    counts = load("ml5_BPRMF")["counts"]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29600 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, out)) for r in range(world)]
    for p in procs:
        p.start()
    sums, table = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = orc.metrics(counts)
    got = metrics_from_sums(sums)
    assert int(sums[7]) == int((counts[:, 2] >= 0).sum())
    assert np.allclose(got[:6], ref[:6], rtol=0, atol=1e-12)
    assert abs(got[6] - ref[6]) < 1e-12
    base = np.arange(12, dtype=np.float64).reshape(3, 4)
    assert np.allclose(table, base + 0.5 * sum(r + 1 for r in range(world)))


def test_shards_cover_all_users_once():
    for U in (1, 7, 6040, 1_000_000):
        for world in (1, 2, 3, 8):
            r = [shard_range(U, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == U
            assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
