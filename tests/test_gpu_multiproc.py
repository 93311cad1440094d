"""User-sharded training and evaluation with REAL engines in SEPARATE processes (one rank per process, here both on the one
GPU, collectives over gloo through host memory - what `bench.py --gpus 2 --backend gloo --single-device` runs): the
processes' result must be what the in-process emulation of the same ranks (recommendation_amd.sharded.ShardGroup) computes,
bit for bit where the local steps are deterministic, and the all-reduced evaluation sums must be those of one engine
holding the gathered parameters.  Needs an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODELS = ["BPRMF", "TransRec"]
EPOCHS, HOGWILD_EPOCHS = 2, 3


def _setup(model):
    import recommendation_amd as ra
    from oracle import oracle_py as orc
    from recommendation_amd.capi import F32
    from tests.golden_util import fixture_case
    g, _, co, ci, labels = fixture_case(model, "ml5")
    seq_off, seq_item, test, val = orc.split(co, ci)      # input preparation only (model.hpp:22-45)
    U, I = len(seq_off) - 1, int(g["n_items"])
    kw = dict(K=10, lam=0.05, bias_reg=0.01, relation_reg=0.05, precision=F32)

    def make(ub, ue):
        return ra.Engine(model, I, seq_off, seq_item, test, val, user_begin=ub, user_end=ue, **kw)
    return make, U, I


def _deterministic_round(trainer):
    """The round's tuples drawn by the engine's sampler and applied strictly in order (batch-1 replay): every rank's local
    step is then a pure function of its inputs."""
    from recommendation_amd.capi import SEQUENTIAL

    def local_round(lr, seed, epoch, k):
        e = trainer.eng
        T = e.sample_tuples(trainer.steps_per_round(), seed=seed, epoch=(epoch * trainer.rounds + k) * trainer.world + trainer.rank)
        e.sgd_replay(lr, T, SEQUENTIAL)
        if trainer.projects:
            pass        # the replay applies the reference's per-step projection itself
    return local_round


def _rank_main(rank, world, port, model, out):
    import torch
    import torch.distributed as dist
    from recommendation_amd.dist_util import shard_range
    from recommendation_amd.sharded import ShardedTrainer, item_table_tensors
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    make, U, I = _setup(model)
    eng = make(*shard_range(U, rank, world))
    eng.init_params(0)

    def all_reduce_sum(t):           # staged through host memory, as bench.py --backend gloo does
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)

    def all_gather_host(a):
        parts = [None] * world
        dist.all_gather_object(parts, a)
        return parts

    tr = ShardedTrainer(eng, rank, world, all_reduce_sum, dev)
    hog = tr.local_round
    tr.local_round = _deterministic_round(tr)
    for ep in range(EPOCHS):
        tr.epoch(0.05, seed=0, epoch=ep)
    sums = tr.eval_sums(50)
    W = tr.gather_params(all_gather_host)
    # then Hogwild rounds (float atomics: not reproducible bit for bit), after which the replicas must still agree exactly
    tr.local_round = hog
    for ep in range(EPOCHS, EPOCHS + HOGWILD_EPOCHS):
        tr.epoch(0.05, seed=0, epoch=ep)
    torch.cuda.synchronize()
    items = [t.cpu().numpy().copy() for t in item_table_tensors(eng, dev)]
    sums2 = tr.eval_sums(50)
    all_items = all_gather_host(items)
    if rank == 0:
        out.put((W, sums, all_items, sums2, tr.rounds, dist.get_world_size(), dist.get_backend()))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("model", MODELS)
def test_two_processes_with_real_engines_equal_the_in_process_shards(model):
    import torch
    import torch.multiprocessing as mp
    from recommendation_amd.dist_util import metrics_from_sums
    from recommendation_amd.sharded import ShardGroup
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, 29710 + MODELS.index(model), model, out)) for r in range(world)]
    for p in procs:
        p.start()
    W, sums, all_items, sums2, rounds, ws, backend = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ws == world and backend == "gloo"
    # the same ranks inside this process
    make, U, I = _setup(model)
    grp = ShardGroup(make, U, world, torch.device("cuda", 0))
    assert grp.rounds == rounds
    grp.init_params(0)
    for t in grp.trainers:
        t.local_round = _deterministic_round(t)
    for ep in range(EPOCHS):
        grp.epoch(0.05, seed=0, epoch=ep)
    Wg = grp.gather_params()
    sg = sum(np.asarray(e.eval_sums(50)) for e in grp.engines)
    grp.close()
    assert np.array_equal(W, Wg), np.abs(W - Wg).max()             # tables combined over gloo == combined in process, bit for bit
    assert np.allclose(sums, sg, rtol=0, atol=1e-9) and sums[7] == sg[7]
    # the gathered parameters in ONE engine score like the ranks' all-reduced sums
    one = make(0, 0)
    one.upload(W)
    s1 = np.asarray(one.eval_sums(50))
    one.close()
    assert s1[7] == sums[7] and np.allclose(s1, sums, rtol=0, atol=1e-9), (s1, sums)
    assert 0.5 < metrics_from_sums(sums)[1] < 1.0
    # after the Hogwild rounds: both processes hold identical item tables, finite, and the model has kept training
    for a, b in zip(all_items[0], all_items[1]):
        assert np.array_equal(a, b) and np.isfinite(a).all()
    assert metrics_from_sums(sums2)[1] > 0.5
