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


class _FakeEngine:
    """What ShardedTrainer needs from an engine, on CPU tensors: two tables (users x 2, items x 2) in the flat order
    [user table][item table]; a 'round' adds rank-dependent amounts to the rank's own user rows and to every item row."""

    def __init__(self, U, I, ub, ue, rank):
        self.n_users, self.n_items, self.user_begin, self.user_end, self.rank = U, I, ub, ue, rank
        self.steps_per_epoch = 8 * I
        self.users = torch.zeros(U, 2, dtype=torch.float64)
        self.items = torch.arange(I * 2, dtype=torch.float64).reshape(I, 2).clone()
        self.calls = []

    def tables(self):
        return [(0, self.n_users, 2, True), (0, self.n_items, 2, False)]

    def item_tensors(self, device):
        return [self.items.view(-1)]

    def sgd_epoch(self, lr, n_steps, seed, epoch):
        self.calls.append((n_steps, epoch))
        self.users[self.user_begin:self.user_end] += (self.rank + 1)
        self.items += (self.rank + 1) * 0.25

    def sync(self):
        pass

    def download(self):
        return np.concatenate([self.users.numpy().ravel(), self.items.numpy().ravel()])


def _trainer_worker(rank, world, port, out):
    from recommendation_amd.sharded import ShardedTrainer, default_rounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    U, I = 10, 6
    ub, ue = shard_range(U, rank, world)
    eng = _FakeEngine(U, I, ub, ue, rank)
    tr = ShardedTrainer(eng, rank, world, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), torch.device("cpu"))
    assert tr.rounds == default_rounds(eng.steps_per_epoch, I) == 4
    tr.epoch(0.05, seed=1, epoch=3)

    def all_gather_host(a):
        parts = [None] * world
        dist.all_gather_object(parts, a)
        return parts

    W = tr.gather_params(all_gather_host)
    if rank == 0:
        out.put((W, eng.calls))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_trainer_rounds_combination_and_gather():
    """Two gloo ranks drive ShardedTrainer: rounds per epoch from the default rule, distinct tuple streams per (epoch,
    round, rank), item tables = base + sum of the ranks' deltas after every round, user rows gathered from their owners."""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, 29650, out)) for r in range(world)]
    for p in procs:
        p.start()
    W, calls = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    U, I, rounds = 10, 6, 4
    users, items = W[:U * 2].reshape(U, 2), W[U * 2:].reshape(I, 2)
    # every round adds 0.25 (rank 0) + 0.5 (rank 1) to every item value
    assert np.allclose(items, np.arange(I * 2).reshape(I, 2) + rounds * 0.75)
    # rows of rank 0's users moved by 1 per round, rank 1's by 2
    assert np.allclose(users[:5], rounds * 1.0) and np.allclose(users[5:], rounds * 2.0)
    assert [c[0] for c in calls] == [8 * I // (world * rounds)] * rounds
    assert [c[1] for c in calls] == [(3 * rounds + k) * world + 0 for k in range(rounds)]
