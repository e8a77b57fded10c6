#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched 333 Hz vehicle step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...) or
    plainly: without WORLD_SIZE in the environment bench.py starts its own N rank processes (before anything touches a GPU),
    one per GPU, rendezvous on 127.0.0.1, and relays rank 0's JSON line.

One "step" = one physics tick (dt = 1/333 s) of every car resident on the GPU = one launch of the HIP step kernel per car range.
Workload = BASELINE.json configs[1]: 4096 AE86 cars per GPU on the synthetic flat-plane track, per-car constant random
actions (steer ~ U(-0.3,0.3), a1 ~ U(-1,1), numpy RandomState(1234) indexed by GLOBAL car id).  State, actions and
outputs are resident in HBM before the timed region; the cars have settled on their springs and are driving (--settle ticks
of state preparation, outside warm-up and timing).  On each GPU the cars step as --partitions free-running ranges (one HIP
stream each: cars are independent, the ranges' kernels overlap).  Multi-GPU: cars are sharded contiguously (weak scaling,
4096 per GPU); the only collective is the RCCL all-gather of k-tick trajectory rings of the [N,26] observation/reward/flag
block to the learner, on a side stream.

Timing: W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides, max over ranks.  A
region shorter than 0.2 s is repeated (same K steps each time, same bracketing) until 0.25 s of timed work has accumulated;
`ms_per_step` / `value` are then the median region's, `repeats` and `timed_region_s` say what was measured.

Prints ONE JSON line on rank 0; at N = 1 it also carries `cpu_baseline` and an `extra` block: the other BASELINE configs'
shapes measured the same way in the same process (16384 cars on the dense mountain-road spline, guard rails + MLP policy,
the 8192-car shard of configs[3], per-tick gather + action scatter, episodes with terminations and resets).
"""
import argparse, ctypes as C, json, os, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

CARS_PER_GPU = 4096
# algorithmic bytes per car-tick (DESIGN.md "Algorithmic bytes"): record read + record write + action + output row
B_ALG = 2352 + 2352 + 8 + 104   # sizeof(pdb_dyn_state) in and out, float[2] action, pdb_step_out (checked against the ctypes mirrors in measure)
HBM_PEAK_GBS = 8000.0
PROFILE_TAG = 'r04'


def cpu_baseline(P, trk, S0, actions, seconds_target=15.0):
    """oracle (CPU restatement, glibc build) timed on this host: 1 core, bounded sample of the same workload"""
    import numpy as np
    import oracle_ctypes   # test infrastructure: only this cpu_baseline leg touches the oracle
    orc = oracle_ctypes.load_oracle(portable_math=False)
    h = orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0))
    n, ticks = 16, 333
    a = np.ascontiguousarray(actions[:n], dtype=np.float32)
    t = orc.cpuref_bench(h, n, ticks, a.ctypes.data_as(C.c_void_p), 1, None)
    rate1 = n * ticks / t
    # size the sample for ~seconds_target of CPU work
    ticks2 = 3330   # configs[1] runs 3330 ticks per car
    n2 = int(max(16, min(len(actions), rate1 * seconds_target / ticks2)))
    a2 = np.ascontiguousarray(actions[:n2], dtype=np.float32)
    t2 = orc.cpuref_bench(h, n2, ticks2, a2.ctypes.data_as(C.c_void_p), 1, None)
    # all hardware threads (context only): every car of the bench workload (cars >> threads), as many ticks as make >= 5 s of wall time
    # at an assumed half-linear speed-up, re-run once if the first estimate was short
    ncores = os.cpu_count() or 1
    n3 = len(actions)
    a3 = np.ascontiguousarray(actions[:n3], dtype=np.float32)
    rate_est = (n2 * ticks2 / t2) * max(1.0, 0.5 * ncores)
    t3 = 0.0; ticks3 = 0
    for _ in range(2):
        ticks3 = int(min(3330, max(50, 6.0 * rate_est / n3)))
        t3 = orc.cpuref_bench(h, n3, ticks3, a3.ctypes.data_as(C.c_void_p), ncores, None)
        if t3 >= 5.0 or ticks3 >= 3330:
            break
        rate_est = n3 * ticks3 / t3
    orc.cpuref_destroy(h)
    return {"value": n2 * ticks2 / t2, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d cars x %d ticks of the bench workload (its first %d cars, %.0f s of CPU work), CPU restatement of Car::step + ODE-equivalent solve, single thread" % (n2, ticks2, n2, t2),
            "all_cores_value": n3 * ticks3 / t3, "all_cores": ncores,
            "all_cores_sample": "%d cars x %d ticks on %d OpenMP threads, %.1f s of wall time" % (n3, ticks3, ncores, t3)}


class _Arr:   # zero-copy torch view of a library-owned device block
    def __init__(self, ptr, shape, typestr='<f4'):
        self.__cuda_array_interface__ = {'shape': shape, 'typestr': typestr, 'data': (ptr, False), 'version': 2}


def measure(args, world, rank, local_rank, dist, want_cpu=False):
    """one configuration, measured by the contract's rule; returns the result dict on rank 0 (None elsewhere)"""
    if args.partitions is None:
        args.partitions = 3 if world == 1 else 2
    import numpy as np
    import torch
    import pdbatch, pdb_ctypes as pc, sharding

    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # one rank per GPU on the driver's node; ranks share devices only in the single-GPU gloo test
    torch.cuda.set_device(dev_index)
    dev = 'cuda:%d' % dev_index
    n = args.cars
    policy = args.policy or ('feedback' if (args.workload != 'flat' or args.episodes) else 'constant')
    assert B_ALG == 2 * C.sizeof(pc.DynState) + 8 + C.sizeof(pc.StepOut), 'B_ALG is stale: update it with the record layout'
    P = pdbatch.packed_params(os.environ['PDB_BENCH_CAR'] + '.env') if os.environ.get('PDB_BENCH_CAR') else pdbatch.packed_params()   # (diagnostic: another packed car; the bench line is the AE86)
    if args.no_body_contacts:
        P.collider.enabled = 0
    gen_args = {}
    if args.spline_step:
        gen_args['step'] = args.spline_step
    if args.walls:
        gen_args['walls'] = True
    is_ref = args.workload in pdbatch.REFERENCE_TRACKS   # one of the reference's own tracks, packed in the build container (tools/pack_tracks.py)
    trk = pdbatch.reference_track(args.workload) if is_ref else pdbatch.synthetic_track(args.workload, **gen_args)
    lib = pc.load_product()
    S0 = pc.DynState()
    assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    all_actions = sharding.global_actions(n * world, 1234)          # indexed by global car id => invariant to the sharding
    first, last = sharding.shard_bounds(n * world, world, rank)
    actions = all_actions[first:last]
    b = pdbatch.Batch(n, P, trk, device=dev_index, action_mode=1)
    stream = torch.cuda.current_stream()
    b.set_stream(stream.cuda_stream)
    b.upload_actions(actions)
    if args.workload in ('playground', 'nordring') or is_ref:   # reference-scale meshes: every car to its own random point of the lap, on the device
        b.set_seed(np.arange(first, first + n, dtype=np.uint32) * 2654435761 % 4294967291 + 1)   # Car::teleportByMode(Random) draws from the car's own rand()
        b.reset(mode=2)
    if args.workload == 'touge':   # spread the cars around the lap (host-side teleports, once)
        st = (pc.DynState * n)()
        for i in range(n):
            s = pc.DynState.from_buffer_copy(bytes(S0))
            lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(((first + i) % 4096) / 4096.0), C.byref(s))
            C.memmove(C.byref(st[i]), C.byref(s), C.sizeof(s))
        b.set_state(st)

    out_t = torch.as_tensor(_Arr(b.out_device_ptr(), (n, 26)), device=dev)
    ring_streams = []
    gather = sharding.TrajectoryGather(n, world, dev, dist, k=args.gather_ticks, force=args.force_gather, producer_wait=b.wait_partitions,
                                       producer_streams=(lambda: ring_streams) if not args.ring_fork else None)
    act_t = torch.as_tensor(_Arr(b.actions_device_ptr(), (n, 2)), device=dev)
    do_scatter = bool(args.scatter_actions and gather.active)   # configs[3]: the learner (rank 0) scatters every tick's actions back
    scatter_src = torch.from_numpy(all_actions).to(dev) if (do_scatter and rank == 0) else None

    if policy == 'mlp':
        g = torch.Generator(device='cpu'); g.manual_seed(4567)
        w1 = (torch.randn(24, 256, generator=g) / 24 ** 0.5).to(dev); b1 = torch.zeros(256, device=dev)
        w2 = (torch.randn(256, 256, generator=g) / 16.0).to(dev); b2 = torch.zeros(256, device=dev)
        w3 = (torch.randn(256, 2, generator=g) / 16.0).to(dev); b3 = torch.tensor([0.0, 0.5], device=dev)
        import projectd_env
        obs_scale = (1.0 / torch.from_numpy(projectd_env.obs_bounds(projectd_env.EnvConfig())[1])).to(dev)
        w1 = (obs_scale[:, None] * w1).contiguous()   # the observation's normalisation folded into the first layer (one launch less per tick)
        h1_buf, h2_buf = {}, {}
    if policy == 'scripted':   # gas = 0.6 + 0.4 sin(2 pi t / 7 s + phi_i), phi_i from seed 2345 by GLOBAL car id; steer = a P-law on
        # lookAhead[0] (the road's bend 10 m ahead; bodyVsTrack is a cosine and carries no sign) + the side probes' centring + yaw damping, clipped to +-1
        phi = torch.from_numpy(np.random.RandomState(2345).uniform(0.0, 2.0 * np.pi, n * world).astype(np.float32)[first:last]).to(dev)
    if policy in ('host_mlp',):   # the SAC-sized actor on the HOST (24 -> 256 -> 256 -> 2, fixed random weights, seed 4567), torch on the CPU cores
        g = torch.Generator(device='cpu'); g.manual_seed(4567)
        hw1 = torch.randn(24, 256, generator=g) / 24 ** 0.5; hb1 = torch.zeros(256)
        hw2 = torch.randn(256, 256, generator=g) / 16.0; hb2 = torch.zeros(256)
        hw3 = torch.randn(256, 2, generator=g) / 16.0; hb3 = torch.tensor([0.0, 0.5])
        import projectd_env
        h_scale = 1.0 / torch.from_numpy(projectd_env.obs_bounds(projectd_env.EnvConfig())[1])
        torch.set_num_threads(int(os.environ.get('PDB_HOST_MLP_THREADS', max(1, min(16, (os.cpu_count() or 2) // 2)))))   # 2.7 k rows x 256 x 256: a few cores' worth; torch's default (every hardware thread of a 256-thread host) spends the tick waking its pool
    if policy == 'feedback':
        fw = np.zeros((24, 2), np.float32)
        fw[21, 0] = 0.03; fw[20, 0] = -0.03; fw[19, 0] = 0.015; fw[18, 0] = -0.015; fw[4, 0] = 0.15; fw[2, 1] = -0.3
        fb_w = torch.from_numpy(fw).to(dev); fb_b = torch.tensor([0.0, 0.3 * 12.0], device=dev)
    tick_id = [0]

    # --episodes: the env loop -- rewards with penalties, terminations (hit / off track / stuck / low reward) and the reset tick
    # (teleport + zero action) are evaluated inside the step kernel (pdb_set_env = projectd_env.py:173-227 per car), so a tick of
    # the env is the same launches as a tick of the bare stepper
    if args.episodes:
        import projectd_env as E
        b.set_env(E.EnvConfig(teleport_mode=args.teleport_mode))

    def host_policy(o, a):
        """obs rows [m, 24] -> action rows [m, 2], on the host, in place"""
        if policy == 'host_mlp':
            x = torch.from_numpy(o) * h_scale
            h2 = torch.relu(torch.relu(x @ hw1 + hb1) @ hw2 + hb2)
            torch.tanh(h2 @ hw3 + hb3, out=torch.from_numpy(a))
        else:   # the probe-feedback law in numpy
            np.clip(0.03 * (o[:, 21] - o[:, 20]) + 0.015 * (o[:, 19] - o[:, 18]) + 0.15 * o[:, 4], -1.0, 1.0, out=a[:, 0])
            np.clip(0.3 * (12.0 - o[:, 2]), -1.0, 1.0, out=a[:, 1])

    def policy_step(o, a, t=0, f=0):
        if policy == 'scripted':
            c = a.shape[0]
            # four columns by hand (no GEMM dispatch for a [c, 24] x [24, 1] product)
            torch.add(o[:, 21], o[:, 20], alpha=-1.0, out=a[:, 0]).mul_(0.03).add_(o[:, 12], alpha=-1.0).add_(o[:, 4], alpha=0.15)
            a[:, 0].clamp_(-1.0, 1.0)
            # env action -> gas is linscale(a1, -1, 1, 0.1, 1.0) (projectd_env.py:160): a1 = (gas - 0.1) / 0.45 - 1
            torch.sin(phi[f:f + c] + (2.0 * np.pi / 7.0) * (t / 333.0), out=a[:, 1])
            a[:, 1].mul_(0.4 / 0.45).add_(0.5 / 0.45 - 1.0)
            return
        if policy == 'feedback':   # oracle/scenarios.h scenarioFeedback (a linear law of the observation, clamped) as one addmm + one clamp
            torch.addmm(fb_b, o[:, :24], fb_w, out=a)
            a.clamp_(-1.0, 1.0)
        elif policy == 'mlp':      # obs -> normalise -> 256 -> 256 -> 2, tanh-squashed like SAC's actor mean (hyperparams/sac.yml net_arch)
            # six launches: three GEMMs with their bias (addmm), two ReLUs and the tanh in place, hidden layers in buffers kept per row count
            c = a.shape[0]
            if c not in h1_buf:
                h1_buf[c] = torch.empty(c, 256, device=dev); h2_buf[c] = torch.empty(c, 256, device=dev)
            h1 = torch.addmm(b1, o[:, :24], w1, out=h1_buf[c]).relu_()
            h2 = torch.addmm(b2, h1, w2, out=h2_buf[c]).relu_()
            torch.addmm(b3, h2, w3, out=a).tanh_()
        elif policy == 'random':   # fresh uniform actions every tick (an untrained agent: ends episodes quickly)
            a.uniform_(-1.0, 1.0)

    use_ring = policy == 'constant' and args.partitions > 1 and not do_scatter
    # a per-tick policy: per-partition closed loops, unless a gather has to see whole ticks (N > 1)
    # and only where a tick is long enough to hide the doubled number of (small) policy launches: the host enqueues ~10 per partition and tick
    part_loops = policy not in ('constant', 'host', 'host_sync', 'host_mlp') and args.partitions > 1 and not gather.active and n >= args.part_loop_min
    host_pipe = policy in ('host', 'host_mlp') and args.partitions > 1
    # configs[3] as worded (a gather and an action scatter EVERY tick) over free-running partitions: one set of collectives per partition
    part_exchange = do_scatter and args.partitions > 1 and policy == 'constant' and n >= args.part_loop_min and (dist is not None)
    split = use_ring or part_loops or host_pipe or part_exchange   # the cars step as free-running partitions
    if split:
        b.set_partitions(args.partitions)
    if use_ring and not args.ring_fork:   # the gather orders a ring's reuse on the partitions' own streams: the rings then start without waiting for the batch's stream
        ring_streams.extend(torch.cuda.ExternalStream(b.partition_stream(p), device=dev) for p in range(args.partitions))
    lib_exchange = False
    part_graph = None; part_graph_whole = False
    if part_exchange:
        part_st = [torch.cuda.ExternalStream(b.partition_stream(p), device=dev) for p in range(args.partitions)]
        part_rng = [b.partition_range(p) for p in range(args.partitions)]
        # the library's own RCCL path has only ever run with ONE rank (two ranks cannot share the build's single GPU under RCCL): with more ranks it is opt-in,
        # so that an untried code path can never cost the driver's multi-GPU line; torch's PartitionExchange (held on gloo at world sizes 2 and 4) is the default there
        if not args.torch_exchange and (world == 1 or (args.library_exchange and dist.get_backend() == 'nccl')):
            try:     # the three steps issued by the library through its own RCCL communicators (collective: every rank succeeds or none)
                exch = sharding.LibraryExchange(b, part_rng, world, rank, dev, dist)
                lib_exchange = True
            except RuntimeError as e:
                sys.stderr.write('bench: %s; per-partition exchange through torch.distributed instead\n' % e)
        if not lib_exchange:
            exch = sharding.PartitionExchange(part_rng, world, rank, dev, dist)
        if rank == 0:
            exch.load_actions(torch.from_numpy(all_actions).to(dev))
    if host_pipe:
        part_rng = [b.partition_range(p) for p in range(args.partitions)]
        h_act, h_out = b.host_mirrors()
        h_act[:] = actions
        host_primed = [False] * args.partitions
    if part_loops:
        part_st = [torch.cuda.ExternalStream(b.partition_stream(p), device=dev) for p in range(args.partitions)]
        part_rng = [b.partition_range(p) for p in range(args.partitions)]
        # the policy's handful of small launches per partition and tick as ONE graph launch (the loop is launch-bound on the host below ~8192 cars: twelve torch
        # dispatches + six kernel launches per 60-100 us tick).  The kernels then write the library's own output block (a fixed address, which the graph holds)
        # instead of the trajectory ring.  The scripted law reads the tick number: it stays uncaptured.
        part_graph = None
        if not args.no_graph_policy and policy in ('feedback', 'mlp') and n < 8192:   # (measured: 4096 cars 45.0 -> 46.5 M with the env loop, 32.5 -> 35.6 M reset-free; at 16384 cars the loop is GPU-bound and the graph loses: 47.4 against 52.2 M)
            try:
                for p in range(args.partitions):   # (first use of the library's GEMM kernels, of the partition's launch path and its buffers outside a capture)
                    f, c = part_rng[p]
                    with torch.cuda.stream(part_st[p]):
                        b.step_partition(p, out_t.data_ptr())
                        policy_step(out_t[f:f + c], act_t[f:f + c], 0, f)
                torch.cuda.synchronize()
                act_t.copy_(torch.from_numpy(actions).to(dev)); torch.cuda.synchronize()
                gs = []
                whole = not args.graph_policy_only
                if whole:   # the partition's tick itself goes into the graph too (two more launches): the contact pass's grid is then fixed
                    b.set_contact_grid(args.graph_contact_grid)
                for p in range(args.partitions):
                    f, c = part_rng[p]
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=part_st[p]):
                        if whole:
                            b.step_partition(p, out_t.data_ptr())
                        policy_step(out_t[f:f + c], act_t[f:f + c], 0, f)
                    gs.append(g)
                part_graph = gs; part_graph_whole = whole
            except Exception as e:
                sys.stderr.write('bench: the policy could not be captured (%r): plain launches\n' % (e,))
                part_graph = None

    host_act = [np.ascontiguousarray(actions, dtype=np.float32).copy()]

    def tick():
        t = tick_id[0]; tick_id[0] = t + 1
        if host_pipe:   # configs[4] as SURVEY 8d words it, pipelined over the partitions: while the host works out
            # partition p's actions from the rows that have just come down, the other partitions' ticks and copies are in flight
            for p in range(args.partitions):
                f, c = part_rng[p]
                if host_primed[p]:
                    b.wait_host_partition(p)
                    host_policy(h_out['obs'][f:f + c], h_act[f:f + c])
                b.step_host_partition(p)
                host_primed[p] = True
            return
        if policy in ('host', 'host_sync', 'host_mlp'):   # the same loop as one synchronous round trip per tick (pdb_step_host): actions up, the tick,
            # observations down, then the host computes the next actions (the probe-feedback law in numpy)
            o = b.step_host(host_act[0])['obs']
            host_policy(np.ascontiguousarray(o), host_act[0])
            return
        if part_exchange and lib_exchange:   # per partition, on its own stream: actions from the learner, the tick, outputs to the learner -- three enqueues inside the library
            for p in range(args.partitions):
                exch.step(p)
            return
        if part_exchange:   # the same through torch.distributed (gloo, or --torch-exchange)
            for p in range(args.partitions):
                f, c = part_rng[p]
                with torch.cuda.stream(part_st[p]):
                    exch.scatter(p, act_t[f:f + c])
                    b.step_partition(p, out_t.data_ptr())
                    exch.gather(p, out_t[f:f + c], wait=True)        # the partition's stream waits for it (the host does not): its next tick rewrites these rows
            return
        o = gather.slot(t)                  # the kernel writes tick t straight into its trajectory-ring slot
        if part_loops:     # every partition runs its own closed loop (kernel, then the policy on its rows) on its own stream
            for p in range(args.partitions):
                f, c = part_rng[p]
                with torch.cuda.stream(part_st[p]):
                    if part_graph is not None:
                        if not part_graph_whole:
                            b.step_partition(p, out_t.data_ptr())
                        part_graph[p].replay()
                    else:
                        b.step_partition(p, o.data_ptr())
                        policy_step(o[f:f + c], act_t[f:f + c], t, f)
            return
        if do_scatter:
            act_t.copy_(sharding.scatter_actions(scatter_src, n, world, rank, dev, dist))
        b.set_out_device_ptr(o.data_ptr())
        b.step_async()
        policy_step(o, act_t, t, 0)
        gather.after_tick(t)

    def run(nsteps):
        """enqueue nsteps ticks: one by one, or (free-running partitions) a trajectory ring at a time"""
        if not use_ring:
            for _ in range(nsteps):
                tick()
            return
        k = gather.k
        t = tick_id[0]; end = t + nsteps
        while t < end:
            m = min(k - t % k, end - t)
            b.step_ring(m, gather.ring(t).data_ptr(), k, t % k, join=False, fork=bool(args.ring_fork))   # every partition's kernels of these m ticks, written straight into the ring;
                                                                              # the partitions are never joined inside the loop: only the gather waits for a ring
            t += m
            gather.after_tick(t - 1)                              # a full ring starts its all-gather (N > 1)
        tick_id[0] = end

    def fence():
        b.wait_partitions()                   # the batch's stream (= torch's current one) waits for every partition's last kernel
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # bring the GPU out of its idle power state before anything is timed
    _spin = torch.randn(2048, 2048, device=dev)
    _t = time.perf_counter()
    while time.perf_counter() - _t < 0.4:
        for _ in range(20):
            _spin = torch.tanh(_spin @ _spin * 1e-3)
        torch.cuda.synchronize()
    del _spin
    torch.cuda.synchronize()              # set-up done before any partition stream starts
    gather.warm()
    run(args.settle)                      # state preparation: off the springs and rolling (not warm-up, not timed)
    fence()
    run(args.warmup)
    fence()

    regions = []          # (wall seconds (max over ranks), event ms on the batch stream, partition-0 event ms, cars per launch)
    total = 0.0
    while True:
        b.event_record(0)
        if split:
            b.partition_mark()
        t0 = time.perf_counter()
        run(args.steps)
        b.wait_partitions()
        b.event_record(1)
        gather.finish()                       # outstanding gathers belong to the timed region

        fence()
        elapsed = sharding.max_over_ranks(time.perf_counter() - t0, dev, dist, world)
        region_ms = b.event_elapsed_ms()
        # (a partition's whole tick replayed from a graph never passes the library's event marks: the wall time of the region stands in for them)
        part = b.partition_elapsed_ms(0) if (split and not part_graph_whole) else ((elapsed * 1000.0, b.partition_range(0)[1]) if split else (None, n))
        regions.append((elapsed, region_ms, part[0], part[1]))
        total += elapsed
        if regions[0][0] >= 0.2 or total >= 0.25 or len(regions) >= 400:   # the same decision on every rank (max-over-ranks times)
            break
    regions.sort(key=lambda r: r[0])
    elapsed, region_ms, part_ms, launch_cars = regions[len(regions) // 2]

    res = None
    if rank == 0:
        traffic = None; valu_busy = None; prof = None
        for tag in (PROFILE_TAG, 'r01'):
            try:   # memory-side bytes per launch from the committed PMC passes of this same command (profiles/, tools/profile_round.sh)
                pm = json.load(open(os.path.join(ROOT, 'profiles', tag + '_pmc.json')))
                pc_cfg = pm.get('bench', {}).get('config', {})
                if args.workload == 'flat' and n == CARS_PER_GPU and pc_cfg.get('cars_per_gpu') == n and pc_cfg.get('partitions', 1) == (args.partitions if split else 1):
                    traffic = pm.get('traffic_bytes_per_launch'); valu_busy = pm.get('valu_issue_busy_frac', pm.get('valu_busy_frac_approx')); prof = tag
                    break
            except Exception:
                continue
        conc = 1
        kernel_us = region_ms * 1000.0 / args.steps          # HIP events on the kernel's stream around the timed region
        if split:   # one launch = one partition's cars; HIP events on that partition's own stream
            kernel_us = part_ms * 1000.0 / args.steps
            conc = args.partitions
        else:
            launch_cars = n
        achieved = B_ALG * launch_cars / (kernel_us * 1e-6) / 1e9
        hdr = pc.TrackHeader.from_buffer_copy(trk[:C.sizeof(pc.TrackHeader)])
        wl = ("configs[4] shape: %d cars/GPU, AE86, %s (%d surfaces, %d triangles, %d spline points, %.1f MB track blob), policy=%s, dt=1/333 s" %
              (n, {'playground': "synthetic paddock of the reference's driftplayground scale: barriers, tyre stacks, cones, islands as separate WALL meshes",
                   'nordring': "synthetic open ribbon of the reference's ks_nordschleife scale with guard rails"}[args.workload],
               hdr.numSurfaces, hdr.numTris, hdr.numFat, len(trk) / 1e6, policy)) if args.workload in ('playground', 'nordring') else \
             ("%s: %d cars/GPU, AE86, the reference's %s (%d surfaces, %d triangles, %d spline points, %.1f MB track blob), policy=%s, dt=1/333 s" %
              ("configs[2]" if args.workload == 'ek_akina' else "configs[4]", n,
               {'ek_akina': "ek_akina spline (shipped) with the road generated around it as a ribbon (its surfaces.bin is a missing blob)",
                'ks_nordschleife': "ks_nordschleife spline (shipped) with the road generated around it as a ribbon",
                'ks_nordschleife_walls': "ks_nordschleife spline (shipped) with the road and guard rails (WALL surfaces) generated around it"}.get(args.workload, args.workload + " as shipped (surfaces.bin, spline.bin, spline.cache)"),
               hdr.numSurfaces, hdr.numTris, hdr.numFat, len(trk) / 1e6,
               {'host': "probe-feedback law in numpy on the HOST, pipelined over the partitions (actions up / observations down every tick)",
                'host_mlp': "SAC-sized 24-256-256-2 MLP with fixed random weights evaluated by torch on the HOST's cores, pipelined over the partitions (actions up / observations down every tick)",
                'scripted': "gas = 0.6 + 0.4 sin(2 pi t / 7 s + phi_i), P-steer on lookAhead[0] + side probes, on the GPU"}.get(policy, policy))) if is_ref else \
             ("configs[1]: %d cars/GPU, AE86, flat-plane track, %s, dt=1/333 s" % (n, "per-car constant random actions" if policy == "constant" else "policy=" + policy)) if args.workload == 'flat' else \
             ("configs[2] shape: %d cars/GPU, AE86, synthetic closed mountain road (%s%s), policy=%s on the GPU, dt=1/333 s" %
              (n, "spline point every %.1f m" % args.spline_step if args.spline_step else "1782 triangles, 891 spline points", ", guard rails (WALL surfaces) along both edges" if args.walls else "", policy))
        if args.episodes:
            wl += "; episodes: the env's rewards, terminations (hit / off track / stuck / low reward) and reset ticks inside the step kernel (pdb_set_env)"
        res = {
            "metric": "env-steps/sec (333 Hz tick, 4-wheel car)",
            "value": n * world * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1000.0 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (+f64 drivetrain)", "data": "synthetic",
            "repeats": len(regions), "timed_region_s": total,
            "config": {"workload": wl, "cars_per_gpu": n, "partitions": (args.partitions if split else 1), "settle_ticks": args.settle,
                       "collective": (("per partition and tick, on the partition's own stream and RCCL communicator: scatter of its [n,2] action rows from rank 0 -> tick -> all-gather of its [n,26] output rows (the partitions are never joined); issued by %s" % ("the library (pdb_step_exchange_partition: three enqueues from C per partition and tick)" if lib_exchange else "torch.distributed (six calls per tick)")) if part_exchange else
                                      "RCCL all-gather of %d-tick trajectory rings [k,N,26] obs/reward/flags on a side stream, kernel writes the ring in place%s" %
                                      (args.gather_ticks, "; actions scattered from rank 0 every tick" if args.scatter_actions else "")) if (world > 1 or args.force_gather) else "none",
                       "parity": "bit-exact vs CPU oracle (tests/, -m gpu); rigid-body solver and contact generation unpinned (ODE absent from the reference tree)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": ("profiles/%s_pmc.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, rocprofv3 --pmc passes of this command" % prof if traffic else None),
                         "kernel": "pdb_step_kernel", "kernel_avg_us": kernel_us, "alg_bytes_per_car_tick": B_ALG, "cars_per_launch": launch_cars,
                         "concurrent_launches": conc, "device_achieved": B_ALG * n * args.steps / elapsed / 1e9, "device_frac": B_ALG * n * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,   # whole device, from the wall time of the timed region
                         "valu_issue_busy_frac": valu_busy,   # the resource that actually bounds the kernel (profiles/*_pmc.json; definition in DESIGN.md section 3)
                         "note": ("one launch = one partition (%d cars); %d partitions step concurrently on their own streams, device_achieved = this GPU's algorithmic bytes over the wall time of the timed region" % (launch_cars, conc)) if conc > 1 else None},
        }
        if hasattr(lib, 'pdb_contact_pass_load'):
            res["contact_pass_cars"] = [int(lib.pdb_contact_pass_load(b.h, q)) for q in (list(range(args.partitions)) if split else [4])]   # cars the last contact passes held (diagnostic)
        if args.episodes and not policy.startswith('host'):   # how often episodes end in this workload: counted over 300 more ticks, outside the timed region
            ends = torch.zeros((), dtype=torch.int64, device=dev)
            for _ in range(300):
                tick()
                ends += (((out_t if (part_loops and part_graph is not None) else gather.slot(tick_id[0] - 1))[:, 25].view(torch.int32) & 8) != 0).sum()
            res["episode_ends_per_tick"] = float(ends.item()) / 300.0
        if want_cpu:   # the CPU leg is timed at N = 1 only
            res["cpu_baseline"] = cpu_baseline(P, trk, S0, all_actions)
    b.close()
    return res


EXTRA = [   # (key, argv) -- the other BASELINE configs' shapes, each measured by measure() in this process at N = 1
    ("configs2_16384_dense_spline", ['--workload', 'touge', '--cars', '16384', '--spline-step', '0.9', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs2_16384", ['--workload', 'touge', '--cars', '16384', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs4_shape_16384_walls_mlp", ['--workload', 'touge', '--cars', '16384', '--walls', '--policy', 'mlp', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs3_shard_8192", ['--cars', '8192', '--steps', '600', '--warmup', '100']),
    ("configs3_shard_8192_gather_k1_scatter", ['--cars', '8192', '--steps', '300', '--warmup', '50', '--force-gather', '--gather-ticks', '1', '--scatter-actions']),
    ("configs3_shard_8192_gather_k1_scatter_torch", ['--cars', '8192', '--steps', '300', '--warmup', '50', '--force-gather', '--gather-ticks', '1', '--scatter-actions', '--torch-exchange']),
    ("configs3_shard_8192_gather_k32", ['--cars', '8192', '--steps', '600', '--warmup', '100', '--force-gather', '--gather-ticks', '32']),
    ("configs4_shape_16384_walls_host_policy", ['--workload', 'touge', '--cars', '16384', '--walls', '--policy', 'host', '--steps', '200', '--warmup', '30', '--settle', '100']),
    ("configs4_shape_16384_walls_host_policy_sync", ['--workload', 'touge', '--cars', '16384', '--walls', '--policy', 'host_sync', '--steps', '200', '--warmup', '30', '--settle', '100']),
    ("configs4_playground_16384_mlp", ['--workload', 'playground', '--cars', '16384', '--policy', 'mlp', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs4_playground_16384_episodes", ['--workload', 'playground', '--cars', '16384', '--episodes', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs4_nordring_16384_mlp", ['--workload', 'nordring', '--cars', '16384', '--policy', 'mlp', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs4_nordring_16384_feedback", ['--workload', 'nordring', '--cars', '16384', '--steps', '300', '--warmup', '50', '--settle', '200']),
    # the reference's own tracks: configs[2] on the Akina ribbon with SURVEY 8d's scripted gas / P-steer (env loop: a car that leaves the
    # road is put back at a random point of the lap); configs[4] as worded -- the policy on the HOST, pipelined -- at the 8192-car shard size
    ("configs2_akina_16384_scripted", ['--workload', 'ek_akina', '--cars', '16384', '--policy', 'scripted', '--episodes', '--teleport-mode', '2', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs4_nordschleife_8192_host_policy", ['--workload', 'ks_nordschleife_walls', '--cars', '8192', '--policy', 'host', '--episodes', '--teleport-mode', '2', '--steps', '200', '--warmup', '30', '--settle', '100']),
    ("configs4_nordschleife_8192_host_mlp", ['--workload', 'ks_nordschleife_walls', '--cars', '8192', '--policy', 'host_mlp', '--episodes', '--teleport-mode', '2', '--steps', '100', '--warmup', '20', '--settle', '100']),
    ("configs4_driftplayground_8192_host_policy", ['--workload', 'driftplayground', '--cars', '8192', '--policy', 'host', '--episodes', '--teleport-mode', '2', '--steps', '200', '--warmup', '30', '--settle', '100']),
    ("configs4_driftplayground_8192_host_mlp", ['--workload', 'driftplayground', '--cars', '8192', '--policy', 'host_mlp', '--episodes', '--teleport-mode', '2', '--steps', '100', '--warmup', '20', '--settle', '100']),
    ("configs4_driftplayground_16384_mlp", ['--workload', 'driftplayground', '--cars', '16384', '--policy', 'mlp', '--episodes', '--teleport-mode', '2', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("episodes_4096", ['--workload', 'touge', '--walls', '--cars', '4096', '--episodes', '--steps', '600', '--warmup', '100', '--settle', '200']),
    ("episodes_4096_reset_free", ['--workload', 'touge', '--walls', '--cars', '4096', '--policy', 'feedback', '--steps', '600', '--warmup', '100', '--settle', '200']),
    ("episodes_16384", ['--workload', 'touge', '--walls', '--cars', '16384', '--episodes', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("episodes_16384_reset_free", ['--workload', 'touge', '--walls', '--cars', '16384', '--policy', 'feedback', '--steps', '300', '--warmup', '50', '--settle', '200']),
]


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--warmup', type=int, default=333)
    ap.add_argument('--settle', type=int, default=333, help='ticks of state preparation before warm-up (cars come off their springs and get rolling); neither warm-up nor timed')
    ap.add_argument('--cars', type=int, default=CARS_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--part-loop-min', type=int, default=4096, help='per-tick policies: from this many cars up every partition runs its own closed loop on its own stream')
    ap.add_argument('--no-extra', action='store_true', help='skip the `extra` block (the other configs measured in the same run)')
    ap.add_argument('--partitions', type=int, default=None, choices=[1, 2, 3, 4],
                    help='free-running car ranges per GPU, one HIP stream each (pdb_set_partitions / pdb_step_ring); 1 = one launch per tick.  Default: 3 on one rank; 2 with more ranks -- a process has four hardware queues, three partitions and the null stream use them up, and the process group\'s collective stream would then share one with a partition and hold it up for as long as a gather runs (tools/hwqueue_probe.py: 47 M against 69 M with a 1 ms kernel per ring on a fifth stream; with two partitions 63-67 M against 67 M)')
    ap.add_argument('--walls', action='store_true', help='touge workload: line both edges of the road with WALL surfaces (configs[4] shape: hull-vs-wall narrow phase next to the guard rails)')
    ap.add_argument('--spline-step', type=float, default=0.0, help='touge workload: metres between spline points (default 5 m = 891 points; 0.9 = 4.9 k points, the density of the reference tracks)')
    ap.add_argument('--no-body-contacts', action='store_true', help='diagnostic A/B: switch the collision pass off in the car block (never the bench line)')
    ap.add_argument('--policy', choices=['constant', 'feedback', 'mlp', 'random', 'host', 'host_sync', 'host_mlp', 'scripted'], default=None,
                    help='where actions come from each tick: constant (configs[1]), feedback (probe controller on the GPU, default for touge), '
                         'mlp (configs[4] shape: a SAC-sized 24-256-256-2 tanh MLP with fixed random weights, evaluated with torch on the GPU from the observation block), random, '
                         'host (the feedback law in numpy on the HOST, actions up / observations down every tick, pipelined over the partitions), host_sync (the same through the synchronous pdb_step_host)')
    ap.add_argument('--episodes', action='store_true', help='run the env loop: terminations with penalties like projectd_env.py, resets through the device reset mask')
    ap.add_argument('--gather-ticks', type=int, default=32, help='ticks per trajectory ring gathered to the learner (N > 1): 1 = plain per-tick gather')
    ap.add_argument('--scatter-actions', action='store_true', help='with a gather: rank 0 scatters the [N,2] action block back every tick (configs[3] as SURVEY 8d words it)')
    ap.add_argument('--library-exchange', action='store_true', help='N > 1: the per-partition exchange through the library\'s own RCCL communicators (pdb_step_exchange_partition); default with one rank, opt-in with more')
    ap.add_argument('--graph-policy-only', action='store_true', help='per-partition loops below 8192 cars: only the policy in the captured graph, the tick as plain launches (A/B)')
    ap.add_argument('--graph-contact-grid', type=int, default=32, help='workgroups of the contact pass inside a captured per-partition tick')
    ap.add_argument('--no-graph-policy', action='store_true', help='per-partition policy loops: the policy as plain torch launches instead of one captured graph per partition (A/B)')
    ap.add_argument('--ring-fork', action='store_true', help='ring mode: every ring starts behind the batch stream (the older form; A/B)')
    ap.add_argument('--torch-exchange', action='store_true', help='the per-partition exchange (--scatter-actions with --gather-ticks 1) through torch.distributed instead of the library\'s own RCCL communicators (A/B)')
    ap.add_argument('--force-gather', action='store_true', help='run the observation all-gather even with one rank (exercises the RCCL + side-stream path on a single GPU)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the multi-rank path on a single-GPU box)')
    ap.add_argument('--teleport-mode', type=int, default=0, choices=[0, 1, 2], help='--episodes: where a reset puts the car (projectd_env.py teleport_mode: 0 start, 1 nearest, 2 random point of the lap)')
    ap.add_argument('--workload', choices=['flat', 'touge', 'playground', 'nordring', 'driftplayground', 'ebisu_touge', 'yamanashi_short', 'euphoria_hillside_park', 'ek_akina', 'ks_nordschleife', 'ks_nordschleife_walls'], default='flat',
                    help='flat = BASELINE configs[1] (the bench line); touge = configs[2] shape: closed hilly road, cars spread around the lap, probe-feedback steering computed on the GPU each tick')
    return ap


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N rank processes of this same command, started before this process has
    touched a GPU (it never does); rank 0's stdout is relayed, the others' goes to stderr."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(subprocess.PIPE if r == 0 else sys.stderr), stderr=sys.stderr))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode() if out else '')
    sys.stdout.flush()
    sys.exit(rc)


def main():
    ap = parser()
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args.gpus)

    # stdout carries the one JSON line and nothing else: whatever a library writes to file descriptor 1 (RCCL prints its version
    # banner there, through C stdio, flushed at exit -- i.e. AFTER the line) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU fallback for the product path')
    dist = None
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(args.backend, init_method='env://')

    res = measure(args, world, rank, local_rank, dist, want_cpu=(not args.no_cpu_baseline and world == 1))
    is_headline = world == 1 and not args.no_extra and args.workload == 'flat' and args.cars == CARS_PER_GPU and not args.episodes and not args.force_gather
    if is_headline:
        extra = {}
        only = [k for k in os.environ.get('PDB_BENCH_EXTRA', '').split(',') if k]   # diagnostic: a subset of the legs, in the order given
        for key, argv in ([(k, dict(EXTRA)[k]) for k in only] if only else EXTRA):
            a = parser().parse_args(argv + ['--no-cpu-baseline', '--no-extra'])
            d2 = None
            try:
                if a.force_gather:
                    import torch.distributed as d2
                    if not d2.is_initialized():
                        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29512')
                        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
                        d2.init_process_group(args.backend, init_method='env://')
                r = measure(a, 1, 0, 0, d2)
                extra[key] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "repeats": r["repeats"], "workload": r["config"]["workload"],
                              "partitions": r["config"]["partitions"], "collective": r["config"]["collective"], "kernel_avg_us": r["roofline"]["kernel_avg_us"],
                              "roofline": {k: r["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_avg_us", "cars_per_launch", "concurrent_launches", "device_achieved", "device_frac")}}
                if "contact_pass_cars" in r:
                    extra[key]["contact_pass_cars"] = r["contact_pass_cars"]
                if "episode_ends_per_tick" in r:
                    extra[key]["episode_ends_per_tick"] = r["episode_ends_per_tick"]
            except Exception as e:   # an extra line must never cost the headline
                extra[key] = {"error": repr(e)[:200]}
        res["extra"] = extra
    if world > 1 and not args.scatter_actions and not args.no_extra:   # the other collective variant SURVEY 8d words (a gather + an action scatter every tick), same process, same rule
        a2 = parser().parse_args(sys.argv[1:] + ['--gather-ticks', '1', '--scatter-actions', '--no-cpu-baseline', '--no-extra'])
        # this optional second measurement must never cost the headline already measured: if it does not come back (a rank that failed alone
        # leaves the others inside a collective), rank 0 prints the line it has and every rank leaves
        import threading

        def _give_up():
            if rank == 0:
                res.setdefault("extra", {})["configs3_gather_k1_scatter"] = {"error": "did not finish within 240 s"}
                json_out.write(json.dumps(res) + '\n'); json_out.flush()
            os._exit(0)
        watchdog = threading.Timer(240.0, _give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            r2 = measure(a2, world, rank, local_rank, dist)
            if rank == 0:
                res.setdefault("extra", {})["configs3_gather_k1_scatter"] = {"value": r2["value"], "ms_per_step": r2["ms_per_step"], "steps": r2["steps"], "repeats": r2["repeats"],
                                                                             "partitions": r2["config"]["partitions"], "collective": r2["config"]["collective"]}
        except Exception as e:
            if rank == 0:
                res.setdefault("extra", {})["configs3_gather_k1_scatter"] = {"error": repr(e)[:200]}
        watchdog.cancel()
    if rank == 0:
        json_out.write(json.dumps(res) + '\n')
        json_out.flush()
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()
    elif is_headline:
        import torch.distributed as d3
        if d3.is_initialized():
            d3.destroy_process_group()


if __name__ == '__main__':
    main()
