"""Multi-GPU host logic: cars are independent, so the batch is cut into contiguous blocks of
N/P cars, one process per GPU, with no collective inside a tick.  The only exchange is the learner-side one the
reference's RL loop implies: gather of the [N/P, 26] observation/reward/flag block and scatter of the [N/P, 2]
actions (RCCL over xGMI on the GPU box; the same code runs on gloo/CPU tensors in tests/test_sharding.py).
Per-car inputs are keyed by the GLOBAL car index so results do not depend on the number of ranks."""
import numpy as np

OUT_COLS = 26   # pdb_step_out: obs[24], reward, flags (include/pdb_types.h)


def shard_bounds(n_global, world, rank):
    """[first, last) of this rank's contiguous block; the first n_global % world ranks hold one extra car."""
    if not (0 <= rank < world):
        raise ValueError('rank %d outside world %d' % (rank, world))
    q, r = divmod(n_global, world)
    first = rank * q + min(rank, r)
    return first, first + q + (1 if rank < r else 0)


def global_actions(n_global, seed, lo=-0.3, hi=0.3):
    """BASELINE configs[1]/[3]: per-car constant action, steer ~ U(lo,hi), a1 ~ U(-1,1), numpy RandomState(seed)
    indexed by global car id."""
    rng = np.random.RandomState(seed)
    a = np.empty((n_global, 2), dtype=np.float32)
    a[:, 0] = rng.uniform(lo, hi, n_global)
    a[:, 1] = rng.uniform(-1.0, 1.0, n_global)
    return a


class ObsGather:
    """Per-tick all-gather of every rank's [n_local, 26] output block into one [world * n_local, 26] tensor
    (equal block sizes: weak scaling).  Buffers are allocated once; the collective is enqueued on torch's current
    stream, i.e. behind the step kernel when the batch was attached to that stream with pdb_set_stream."""

    def __init__(self, n_local, world, device, dist=None):
        import torch
        self.dist = dist
        self.world = world
        self.n_local = n_local
        self.gathered = torch.empty((world * n_local, OUT_COLS), dtype=torch.float32, device=device) if world > 1 else None

    def __call__(self, out_block):
        if self.world == 1:
            return out_block
        if out_block.is_cuda and self.dist.get_backend() == 'gloo':   # single-GPU test of the multi-rank path: stage through the host
            host = self.gathered.cpu()
            self.dist.all_gather_into_tensor(host, out_block.cpu())
            self.gathered.copy_(host)
            return self.gathered
        self.dist.all_gather_into_tensor(self.gathered, out_block)
        return self.gathered


def scatter_actions(all_actions, n_local, world, rank, device, dist=None):
    """Learner (rank 0) -> every rank: this rank's [n_local, 2] slice of the global action tensor."""
    import torch
    mine = torch.empty((n_local, 2), dtype=torch.float32, device=device)
    if world == 1:
        mine.copy_(all_actions)
        return mine
    chunks = list(all_actions.reshape(world, n_local, 2).unbind(0)) if rank == 0 else None
    dist.scatter(mine, [c.contiguous() for c in chunks] if chunks is not None else None, src=0)
    return mine


def max_over_ranks(value, device, dist=None, world=1):
    """bench.py timing rule: the job's time is the slowest rank's."""
    if world == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=('cpu' if dist.get_backend() == 'gloo' else device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
