"""User-sharded training and evaluation over N engines (one per GPU, one process each), and the same logic for N
engines inside one process on one device (tests and rehearsal on a single GPU).

Partitioning: users are split into N contiguous ranges; rank r samples and evaluates only its own users (MC: its own
range of source items), so rows indexed by user are written by exactly one rank and never exchanged during training.
Tables indexed by item are replicated.  An epoch is cut into `rounds` rounds; in a round every rank applies its share
of the round's tuples to its replica, then the replicas are combined:


    rule "sum":   table = base + sum_r (table_r - base)        every rank's steps are kept at full size: what sequential
                                                               SGD would have done had the steps not interacted
    rule "mean":  table = base + sum_r (table_r - base) / N    the average of the replicas (local SGD)

where `base` is the common value at the start of the round.  One all-reduce (sum) of every item table per round:
S = n_items * K * 4 bytes per table and round.

How many rounds: the steps of one round do not see each other across ranks, and their deltas are added at full size,
so a round must stay short enough that an item row receives only a few updates in it.  Measured on a 12k-user x
4k-item corpus against the reference's own trajectories (scripts/exp_sharded.py; K = 64, 2 / 4 / 8 shards): with
about 4.5 updates per item row and round (64 rounds per epoch there) test AUC and HIT@50 stay within 0.002 of the
reference for every shard count; with 18 per round HIT@50 drops by 0.03 at 4 shards and training diverges at 8;
one round per epoch (290 updates per row) diverges from 2 shards on.  Averaging instead of summing never diverges
but moves the item tables N times more slowly (test AUC -0.023 at 8 shards after the same 40 epochs).
Default: rule "sum", rounds = ceil(steps_per_epoch / (2 * n_items)), i.e. at most 4 updates per item row and round.
On the 1M x 200k x 100M corpus that is 245 rounds of 400k tuples per epoch, each followed by an all-reduce of the
51 MB item table: the exchange (113 bytes per tuple over xGMI) then costs more than the tuples' own HBM traffic, so
at that shape user-sharded training does not run faster than one GPU; it is provided for corpora whose user side
does not fit one GPU, and its accuracy is what is tested.

Metrics: every rank reduces its users to the 8 sums of rec_eval_sums; one all-reduce of 8 doubles."""
import numpy as np

from .dist_util import metrics_from_sums, shard_range


class DevArray:
    """Engine-owned device memory seen through the CUDA array interface (for torch)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def item_table_tensors(eng, device):
    """The engine's item-indexed tables as flat torch tensors over the engine's own memory."""
    import torch
    from .capi import F32
    out = []
    for ptr, rows, cols, is_user in eng.tables():
        if not is_user:
            out.append(torch.as_tensor(DevArray(ptr, (rows * cols,), "<f4" if eng.precision == F32 else "<f8"), device=device))
    return out


def user_table_slices(eng):
    """(offset, length) in the flat parameter vector of the rows this engine owns in every user-indexed table."""
    out, off = [], 0
    for ptr, rows, cols, is_user in eng.tables():
        if is_user:
            out.append((off + eng.user_begin * cols, (eng.user_end - eng.user_begin) * cols))
        off += rows * cols
    return out


def default_rounds(steps_per_epoch, n_items):
    """Rounds per epoch that keep an item row at about 4 updates per round (every step updates a positive's and a negative's row)."""
    return max(1, -(-steps_per_epoch // (2 * n_items)))


class ShardedTrainer:
    """One rank of an N-rank job.  all_reduce_sum(tensor) sums a device tensor over the ranks in place (RCCL through
    torch.distributed in production; any callable in tests)."""

    def __init__(self, eng, rank, world, all_reduce_sum, device, rounds=None, rule="sum"):
        assert rule in ("sum", "mean")
        if rounds is None:
            rounds = default_rounds(eng.steps_per_epoch, eng.n_items) if world > 1 else 1
        self.eng, self.rank, self.world, self.all_reduce_sum, self.rounds, self.rule = eng, rank, world, all_reduce_sum, rounds, rule
        # an engine may bring its own view of its item tables (tests without a GPU)
        self.tabs = [] if world == 1 else eng.item_tensors(device) if hasattr(eng, "item_tensors") else item_table_tensors(eng, device)
        self.bases = [t.clone() for t in self.tabs]
        self.device = device
        # TransRec's projection of the item rows onto the unit ball is not additive: done by every rank after its launches
        # it would enter the sum of deltas once per rank.  The ranks' launches only add gradient steps (rows are projected
        # as they are read); the combined table is projected once, after the all-reduce.
        self.projects = world > 1 and getattr(eng, "model", "") in ("TransRec", "TransRec_L1")
        if self.projects:
            eng.set_option("project_after_launch", 0)

    def steps_per_round(self):
        return self.eng.steps_per_epoch // (self.world * self.rounds)

    def _torch_done(self):
        # the engine launches on its own stream: what torch has queued on the tables must have finished before it goes on
        if self.tabs and self.tabs[0].is_cuda:
            import torch
            torch.cuda.current_stream(self.tabs[0].device).synchronize()

    def begin_round(self):
        for t, b in zip(self.tabs, self.bases):
            b.copy_(t)
        self._torch_done()

    def local_round(self, lr, seed, epoch, k):
        # a different tuple stream for every (epoch, round, rank)
        self.eng.sgd_epoch(lr, self.steps_per_round(), seed=seed, epoch=(epoch * self.rounds + k) * self.world + self.rank)
        self.eng.sync()

    def combine(self):
        scale = 1.0 if self.rule == "sum" else 1.0 / self.world
        for t, b in zip(self.tabs, self.bases):
            # sum_r table_r, then base + scale * (sum - N * base)
            self.all_reduce_sum(t)
            t.sub_(b, alpha=float(self.world)).mul_(scale).add_(b)
        self._torch_done()
        if self.projects:
            self.eng.project_rows()

    def epoch(self, lr, seed, epoch):
        for k in range(self.rounds):
            self.begin_round()
            self.local_round(lr, seed, epoch, k)
            if self.world > 1:
                self.combine()

    def eval_sums(self, topn=50):
        import torch
        s = torch.tensor(self.eng.eval_sums(topn), dtype=torch.float64, device=self.device)
        if self.world > 1:
            self.all_reduce_sum(s)
        return s.cpu().numpy()

    def metrics(self, topn=50):
        return metrics_from_sums(self.eval_sums(topn))

    def gather_params(self, all_gather_host):
        """The whole parameter vector in the reference's flat order (what saveModel writes): item tables from this rank's
        replica, user rows from their owners.  all_gather_host(numpy array) -> list of every rank's array."""
        W = self.eng.download()
        mine = np.concatenate([W[o:o + n] for o, n in user_table_slices(self.eng)]) if user_table_slices(self.eng) else np.zeros(0)
        parts = all_gather_host(mine)
        U = self.eng.n_users
        for r, part in enumerate(parts):
            ub, ue = shard_range(U, r, self.world)
            pos, off = 0, 0
            for ptr, rows, cols, is_user in self.eng.tables():
                if is_user:
                    n = (ue - ub) * cols
                    W[off + ub * cols: off + ub * cols + n] = part[pos:pos + n]
                    pos += n
                off += rows * cols
        return W


class ShardGroup:
    """N ranks inside one process on one device: the ranks' rounds run one after the other, the all-reduce is a sum
    over the engines' tensors.  Same arithmetic as N processes; used by the tests and to rehearse on one GPU."""

    def __init__(self, make_engine, n_users, world, device, rounds=None, rule="sum"):
        import torch
        self.world = world
        self.engines = [make_engine(*shard_range(n_users, r, world)) for r in range(world)]
        self.trainers = [ShardedTrainer(e, r, world, None, device, rounds, rule) for r, e in enumerate(self.engines)]
        self.rounds = self.trainers[0].rounds
        self.torch = torch

    def epoch(self, lr, seed, epoch):
        tr = self.trainers
        for k in range(self.rounds):
            for t in tr:
                t.begin_round()
                t.local_round(lr, seed, epoch, k)
            if self.world > 1:
                scale = 1.0 if tr[0].rule == "sum" else 1.0 / self.world
                for ti in range(len(tr[0].tabs)):
                    total = tr[0].tabs[ti].clone()
                    for t in tr[1:]:
                        total.add_(t.tabs[ti])
                    new = total.sub_(tr[0].bases[ti], alpha=float(self.world)).mul_(scale).add_(tr[0].bases[ti])
                    for t in tr:
                        t.tabs[ti].copy_(new)
                self.torch.cuda.synchronize()
                for t in tr:
                    if t.projects:
                        t.eng.project_rows()

    def metrics(self, topn=50):
        s = sum(np.asarray(e.eval_sums(topn)) for e in self.engines)
        return metrics_from_sums(s)

    def init_params(self, seed):
        # every replica starts from the same values; a rank's user rows are the ones the single-engine run would hold
        for e in self.engines:
            e.init_params(seed)

    def gather_params(self):
        parts = []
        for t in self.trainers:
            W = t.eng.download()
            sl = user_table_slices(t.eng)
            parts.append(np.concatenate([W[o:o + n] for o, n in sl]) if sl else np.zeros(0))
        return self.trainers[0].gather_params(lambda mine: parts)

    def close(self):
        for e in self.engines:
            e.close()
