"""Multi-GPU host logic: cars are independent, so the batch is cut into contiguous blocks of
N/P cars, one process per GPU, with no collective inside a tick.  The only exchange is the learner-side one the
reference's RL loop implies: gather of the [N/P, 26] observation/reward/flag block and scatter of the [N/P, 2]
actions (RCCL over xGMI on the GPU box; the same code runs on gloo/CPU tensors in tests/test_sharding.py).
The output blocks are gathered as k-tick trajectory rings (TrajectoryGather) so that the collective overlaps compute.
Per-car inputs are keyed by the GLOBAL car index so results do not depend on the number of ranks."""
import os
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


class _StreamEvent:
    """what TrajectoryGather keeps of a collective it enqueued on a stream of its choosing: an event behind it; wait() = the caller's current stream waits for it"""
    def __init__(self, stream):
        import torch
        self.e = torch.cuda.Event(); self.e.record(stream)

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.e)


class TrajectoryGather:
    """Learner-side exchange of the per-tick output blocks ("per tick (or per k-tick macro-step)").

    The step kernel writes tick t's [n_local, 26] block straight into slot t % k of a trajectory ring that this object
    owns (pdb_set_out_device: no staging copy); when a ring of k ticks is full it is all-gathered to every rank
    ([world, k, n_local, 26]) while the kernel already fills the second ring.  On the GPU (RCCL) the collective runs on a
    side stream, ordered after the ring's last kernel by an event, so its latency over xGMI overlaps compute; a ring is
    not rewritten before its previous gather has completed.  k = 1 is the plain per-tick gather.  On CPU tensors (gloo,
    tests) everything is synchronous and `write(t, block)` stands in for the kernel."""

    def __init__(self, n_local, world, device, dist=None, k=8, force=False, producer_wait=None, producer_streams=None):
        import torch
        self.dist, self.world, self.n_local, self.k = dist, world, n_local, max(1, int(k))
        self.active = (world > 1 or force) and dist is not None
        self.cuda = str(device).startswith('cuda')
        self.rings = [torch.zeros((self.k, n_local, OUT_COLS), dtype=torch.float32, device=device) for _ in range(2)]
        self.gathered = [torch.empty((world, self.k, n_local, OUT_COLS), dtype=torch.float32, device=device) for _ in range(2)] if self.active else None
        self.overlap = self.active and self.cuda and dist.get_backend() == 'nccl'
        self.work = [None, None]
        self.last = None
        self.producer_streams = producer_streams   # callable() -> the streams whose kernels fill the rings (free-running partitions): a ring's previous gather is then waited
                                                   # for on THOSE streams, and the caller need not hold them back behind the current stream (pdb_step_ring with fork=False)
        self.producer_wait = producer_wait   # callable(stream_ptr): make that stream wait for the kernels that fill the rings when they
                                             # do not run on the current stream (free-running partitions: pdbatch.Batch.wait_partitions)
        if self.overlap:
            # the gather's stream.  With producers that run on streams of their own and are not held back behind the current stream (producer_streams), the
            # current stream itself: it is otherwise idle, and one more stream would share a hardware queue with one of the producers' (four per process) --
            # a 1 ms kernel on such a stream held a partition up for 1 ms per ring (measured with a sleep kernel: 66.4 -> 48.2 M)
            self.comm = torch.cuda.current_stream(device) if producer_streams is not None else torch.cuda.Stream(device=device)
            self.e_full = torch.cuda.Event()

    def slot(self, t):
        """tensor [n_local, 26] the kernel of tick t must write (its data_ptr() goes to pdb_set_out_device)"""
        r = (t // self.k) & 1
        if self.overlap and t % self.k == 0 and self.work[r] is not None:
            ps = self.producer_streams() if self.producer_streams is not None else None
            if ps:
                import torch
                for st in ps:
                    with torch.cuda.stream(st):
                        self.work[r].wait()   # the producers' own streams wait: this ring's previous gather still reads it
            else:
                self.work[r].wait()            # current stream waits: this ring's previous gather still reads it
            self.work[r] = None
        return self.rings[r][t % self.k]

    def warm(self):
        """one throw-away collective on the gather buffers (communicator set-up happens on first use: keep it out of a timed loop
        whose warm-up is shorter than a ring)"""
        if not self.active:
            return
        flat_in = self.rings[1].view(-1, OUT_COLS); flat_out = self.gathered[1].view(-1, OUT_COLS)
        if self.cuda and self.dist.get_backend() == 'gloo':
            host = flat_out.cpu(); self.dist.all_gather_into_tensor(host, flat_in.cpu())
        else:
            self.dist.all_gather_into_tensor(flat_out, flat_in)

    def ring(self, t):
        """the whole ring [k, n_local, 26] tick t belongs to (for pdb_step_ring: the kernels of up to k ticks are enqueued at once)"""
        self.slot(t - t % self.k)          # same guard as slot(): a ring is not rewritten before its previous gather is done
        return self.rings[(t // self.k) & 1]

    def write(self, t, block):
        self.slot(t).copy_(block)

    def after_tick(self, t):
        """call once tick t's kernel is enqueued; starts the gather when the ring is full.  Returns the gathered tensor
        ([world, k, n_local, 26]) of that ring, else None."""
        if not self.active or (t + 1) % self.k != 0:
            return None
        import torch
        r = (t // self.k) & 1
        flat_in = self.rings[r].view(-1, OUT_COLS)
        flat_out = self.gathered[r].view(-1, OUT_COLS)
        if self.overlap:
            self.e_full.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.e_full)
                if self.producer_wait is not None:
                    self.producer_wait(self.comm.cuda_stream)
                if self.producer_streams is not None and not os.environ.get('PDB_GATHER_ASYNC'):
                    # the "synchronous" form: torch then enqueues the collective on the CURRENT stream (here the otherwise idle one the producers no longer
                    # wait for) instead of the process group's own side stream, which with three partitions shares a hardware queue with one of them --
                    # one rank, 4096 cars: 67.7 M against 65.5 M with 32-tick rings, 66.4 against 57.4 M with 8-tick rings (no gather at all: 68.6 M).
                    # Nothing blocks on the host; completion is an event of our own on that stream
                    self.dist.all_gather_into_tensor(flat_out, flat_in, async_op=False)
                    self.work[r] = _StreamEvent(self.comm)
                else:
                    self.work[r] = self.dist.all_gather_into_tensor(flat_out, flat_in, async_op=True)
                if os.environ.get('PDB_EXP_COMM_SLEEP'):   # experiment: a long, CU-free kernel on the gather's stream (does a partition's stream share its hardware queue?)
                    if os.environ.get('PDB_EXP_LATE_STREAM'):   # ... or on a stream created now, as the process group's own collective stream is: after the partitions' streams
                        if not hasattr(self, '_late'):
                            self._late = torch.cuda.Stream(device=self.rings[0].device)
                        with torch.cuda.stream(self._late):
                            torch.cuda._sleep(int(os.environ['PDB_EXP_COMM_SLEEP']))
                    else:
                        torch.cuda._sleep(int(os.environ['PDB_EXP_COMM_SLEEP']))
        elif self.cuda and self.dist.get_backend() == 'gloo':   # single-GPU test of the multi-rank path: through the host
            if self.producer_wait is not None:
                self.producer_wait(None)   # the batch's (= current) stream waits for the ring's kernels before the copy to the host
            host = flat_out.cpu()
            self.dist.all_gather_into_tensor(host, flat_in.cpu())
            flat_out.copy_(host)
        else:
            self.dist.all_gather_into_tensor(flat_out, flat_in)
        self.last = r
        return self.gathered[r]

    def finish(self):
        """make the caller's current stream wait for every outstanding gather"""
        for r in (0, 1):
            if self.work[r] is not None:
                self.work[r].wait(); self.work[r] = None
        return self.gathered[self.last] if (self.active and self.last is not None) else None


class PartitionExchange:
    """BASELINE configs[3] as SURVEY section 8d words it -- EVERY tick the [n, 26] output block goes to the learner and the learner's
    [n, 2] actions come back -- without joining the free-running partitions of a GPU: each partition has its own process group
    (its own RCCL communicator and stream) and, per tick, its own three steps in the order of its own HIP stream:
        scatter of its action rows  ->  the partition's tick  ->  all-gather of its output rows.
    Nothing orders one partition's collectives against another partition's kernels, so the partitions keep drifting against each
    other as they do without any exchange.  The learner (rank 0) fills scatter_src[p] ([world, cars of p, 2]) and reads
    gathered[p] ([world, cars of p, 26]).  Collectives are enqueued under `with torch.cuda.stream(partition stream)`: the
    partition's stream waits for them (stream-side, the host never does)."""

    def __init__(self, part_ranges, world, rank, device, dist, backend=None, strict=False):
        import torch
        self.dist, self.world, self.rank, self.ranges = dist, world, rank, list(part_ranges)
        self.strict = strict and dist is not None   # one rank takes no shortcut: its scatter is a collective too (the exact calls N > 1 makes)
        # every rank creates the groups in the same order
        self.groups = [dist.new_group(ranks=list(range(world)), backend=backend) for _ in self.ranges] if dist is not None else [None] * len(self.ranges)
        self.gathered2 = [[torch.empty((world, c, OUT_COLS), dtype=torch.float32, device=device) for (f, c) in self.ranges] for _ in range(2)]
        self.gathered = self.gathered2[0]
        self.work = [[None] * len(self.ranges) for _ in range(2)]
        self.scatter_src = [torch.zeros((world, c, 2), dtype=torch.float32, device=device) for (f, c) in self.ranges] if rank == 0 else None

    def load_actions(self, all_actions):
        """learner: all_actions [world * n_local, 2] (global car order) -> the per-partition scatter sources"""
        a = all_actions.reshape(self.world, -1, 2)
        for p, (f, c) in enumerate(self.ranges):
            self.scatter_src[p].copy_(a[:, f:f + c])

    def _via_host(self, t):
        # gloo moves host memory: device rows go through the host, synchronously (the single-GPU test of the multi-rank path; RCCL takes them as they are)
        return t.is_cuda and self.world > 1 and self.dist.get_backend(self.groups[0]) == 'gloo'

    def scatter(self, p, act_rows):
        """this rank's action rows of partition p <- the learner's (act_rows: the [c, 2] view of the batch's action block)"""
        if self.world == 1 and not self.strict:
            act_rows.copy_(self.scatter_src[p][0])
        elif self._via_host(act_rows):
            import torch
            host = torch.empty(act_rows.shape, dtype=act_rows.dtype)
            self.dist.scatter(host, [x.cpu() for x in self.scatter_src[p].unbind(0)] if self.rank == 0 else None, src=0, group=self.groups[p])
            act_rows.copy_(host)
        else:
            self.dist.scatter(act_rows, list(self.scatter_src[p].unbind(0)) if self.rank == 0 else None, src=0, group=self.groups[p])

    def gather(self, p, out_rows, slot=0, wait=True):
        """partition p's output rows of every rank -> gathered2[slot & 1][p] (out_rows: the contiguous [c, 26] view of this rank's output
        block of that tick).  wait=True: the caller's current stream (the partition's) waits for the collective -- stream-side, never the
        host -- so the rows may be rewritten by the partition's next tick.  wait=False: asynchronous; the caller alternates between two
        output blocks and calls wait(p, slot) before the block of that parity is written again.  (Measured on one rank, 8192 cars: the
        asynchronous form costs ~10 us more host time per partition and tick than it saves on the stream -- the loop is bound by the
        host's ~25 us per torch.distributed call -- so bench.py uses the waiting form.)"""
        r = slot & 1
        if self.world > 1 and self._via_host(out_rows):
            import torch
            torch.cuda.current_stream().synchronize()      # the partition's tick has written the rows
            host = torch.empty((self.world * out_rows.shape[0], OUT_COLS), dtype=out_rows.dtype)
            self.dist.all_gather_into_tensor(host, out_rows.cpu(), group=self.groups[p])
            self.gathered2[r][p].view(-1, OUT_COLS).copy_(host)
            self.work[r][p] = None
            return
        w = self.dist.all_gather_into_tensor(self.gathered2[r][p].view(-1, OUT_COLS), out_rows, group=self.groups[p], async_op=not wait)
        self.work[r][p] = None if wait else w

    def wait(self, p, slot):
        """the caller's current stream waits for the asynchronous gather of partition p issued with this slot parity (if one is outstanding)"""
        r = slot & 1
        if self.work[r][p] is not None:
            self.work[r][p].wait()
            self.work[r][p] = None


class LibraryExchange:
    """PartitionExchange's three steps per partition and tick -- the learner's action rows in, the partition's tick, its output rows of every
    rank out -- issued by the LIBRARY on the partition's stream through its own RCCL communicators (pdb_comm_init /
    pdb_step_exchange_partition): three enqueues from C, no torch.distributed call in the loop.  The 128-byte communicator ids travel over the
    torch process group once, at construction (which is collective).  Raises RuntimeError on every rank if any rank cannot set it up (no RCCL in
    the process, two ranks on one GPU, ...): the caller falls back to PartitionExchange."""

    def __init__(self, batch, part_ranges, world, rank, device, dist, action_stride=2, strict=False):
        import torch
        self.batch, self.world, self.rank, self.ranges = batch, world, rank, list(part_ranges)
        multi = world > 1 or (strict and dist is not None and dist.is_initialized())   # strict: one rank goes through the id broadcast and the agreement like N ranks
        if dist is not None and dist.is_initialized() and dist.get_backend() == 'nccl' and os.environ.get('PDB_EXCHANGE_NO_PREWARM') is None:
            # torch's own RCCL communicator first: created AFTER the library's (at the first torch collective) it left every later kernel of the process
            # 25 % slower, for good (measured: tools/exchange_residue2.py)
            dist.all_reduce(torch.zeros(1, device=device))
            torch.cuda.synchronize()
        ok, err, ids = 1, None, None
        try:
            if rank == 0:
                ids = batch.comm_unique_ids(len(self.ranges))
        except RuntimeError as e:
            ok, err = 0, e
        if multi:
            box = [ids if ok else None]
            dist.broadcast_object_list(box, src=0)
            ids = box[0]
            ok = 0 if ids is None else ok
        if ok:
            try:
                batch.comm_init(world, rank, ids)
            except RuntimeError as e:
                ok, err = 0, e
        if multi:   # agree before anybody enters a collective the others will not
            flag = torch.tensor([ok], dtype=torch.int32, device=(device if dist.get_backend() == 'nccl' else 'cpu'))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            raise RuntimeError('LibraryExchange unavailable: %s' % (err if err is not None else 'another rank could not set it up'))
        self.gathered = [torch.empty((world, c, OUT_COLS), dtype=torch.float32, device=device) for (f, c) in self.ranges]
        self.scatter_src = [torch.zeros((world, c, action_stride), dtype=torch.float32, device=device) for (f, c) in self.ranges] if rank == 0 else None

    def load_actions(self, all_actions):
        """learner: all_actions [world * n_local, stride] (global car order) -> the per-partition scatter sources"""
        a = all_actions.reshape(self.world, -1, all_actions.shape[-1])
        for p, (f, c) in enumerate(self.ranges):
            self.scatter_src[p].copy_(a[:, f:f + c])

    def step(self, p):
        """enqueue partition p's scatter -> tick -> gather on its stream (nothing waited for)"""
        self.batch.step_exchange_partition(p, self.scatter_src[p].data_ptr() if self.rank == 0 else 0, self.gathered[p].data_ptr())


def scatter_actions(all_actions, n_local, world, rank, device, dist=None, strict=False):
    """Learner (rank 0) -> every rank: this rank's [n_local, 2] slice of the global action tensor.  strict: one rank scatters to itself through the backend."""
    import torch
    mine = torch.empty((n_local, 2), dtype=torch.float32, device=device)
    if world == 1 and not (strict and dist is not None):
        mine.copy_(all_actions)
        return mine
    chunks = list(all_actions.reshape(world, n_local, 2).unbind(0)) if rank == 0 else None
    dist.scatter(mine, [c.contiguous() for c in chunks] if chunks is not None else None, src=0)
    return mine


def max_over_ranks(value, device, dist=None, world=1, strict=False):
    """bench.py timing rule: the job's time is the slowest rank's."""
    if world == 1 and not (strict and dist is not None):
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=('cpu' if dist.get_backend() == 'gloo' else device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
