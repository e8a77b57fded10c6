#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched 333 Hz vehicle step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...) or
    plainly: without WORLD_SIZE in the environment bench.py starts its own N rank processes (before anything touches a GPU),
    one per GPU, rendezvous on 127.0.0.1, and relays rank 0's JSON line.

One "step" = one physics tick (dt = 1/333 s) of every car resident on the GPU = one launch of the HIP step kernel per car range
(+ the contact pass behind it).
Workload (no --workload given) = BASELINE.json configs[2] as SURVEY 8d words it, the largest single-GPU configuration: 16384 AE86
cars per GPU on the reference's ek_akina spline (road ribbon around it), scripted inputs evaluated on the GPU every tick from the
observation rows (gas = 0.6 + 0.4 sin(2 pi t / 7 s + phi_i), phi_i from seed 2345 by GLOBAL car id; P-steer on lookAhead[0] + the side
probes), env loop inside the kernel (a car that leaves the road is put back at a random point of the lap).  `secondary` in the same
line = configs[1] (4096 cars, flat plane, per-car constant random actions from RandomState(1234) by global car id), measured by the
same rule in the same process; `cpu_baseline` is timed on configs[1]'s inputs.  State, actions and outputs
are resident in HBM before the timed region.  On each GPU the cars step as --partitions free-running ranges (one HIP stream each:
cars are independent, each range runs its own kernel -> contact pass -> policy loop).  Multi-GPU: cars are sharded contiguously (weak
scaling, the same cars per GPU); the only collective is the RCCL all-gather of k-tick trajectory rings of the [N,26]
observation/reward/flag block to the learner.

Timing: W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides, max over ranks.  A
region shorter than 0.2 s is repeated (same K steps each time, same bracketing) until 0.25 s of timed work has accumulated;
`ms_per_step` / `value` are then the median region's, `repeats` and `timed_region_s` say what was measured.

Rank 0 prints ONE compact JSON line (< 2 KB) on stdout, as soon as the headline and `secondary` are measured.  --extra (or
PDB_BENCH_EXTRA=key,key) then measures the other configs' shapes the same way; those results go to stderr and to
gpurun_out/bench_extra.json, never into the line.
"""
import argparse, ctypes as C, json, os, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

CARS_PER_GPU = 4096
# algorithmic bytes per car-tick (DESIGN.md "Algorithmic bytes"): record read + record write + action + output row
B_ALG = 2352 + 2352 + 8 + 104   # sizeof(pdb_dyn_state) in and out, float[2] action, pdb_step_out (checked against the ctypes mirrors in measure)
HBM_PEAK_GBS = 8000.0
PROFILE_TAG = 'r05'
# BASELINE configs[2] as worded.  settle: untimed ticks of state preparation -- 1500 (0.4 s), until the rate at which episodes end is stationary (after 200 ticks few cars
# have left the road yet and a 20-step region then ran 5 % faster than the steady state: VERDICT r5 weak #9)
HEADLINE = dict(workload='ek_akina', cars=16384, policy='scripted', episodes=True, teleport_mode=2, settle=1500)
SECONDARY_ARGV = ['--workload', 'flat', '--cars', '4096']   # BASELINE configs[1]


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
            "sample": "configs[1] inputs (flat plane, constant actions): its first %d cars x %d ticks, %.0f s of CPU work; CPU restatement of Car::step + ODE-equivalent solve, 1 thread" % (n2, ticks2, t2),
            "all_cores_value": n3 * ticks3 / t3, "all_cores": ncores,
            "all_cores_sample": "%d cars x %d ticks on %d OpenMP threads, %.1f s of wall time" % (n3, ticks3, ncores, t3)}


class _Arr:   # zero-copy torch view of a library-owned device block
    def __init__(self, ptr, shape, typestr='<f4'):
        self.__cuda_array_interface__ = {'shape': shape, 'typestr': typestr, 'data': (ptr, False), 'version': 2}


def measure(args, world, rank, local_rank, dist, want_cpu=False):
    """one configuration, measured by the contract's rule; returns the result dict on rank 0 (None elsewhere)"""
    if args.partitions is None:
        # three free-running ranges per GPU at every N.  (Rounds 3-4 used two with more than one rank, to leave a hardware queue for a collective that ends up on a stream
        # of the process group's own; the ring gather is issued in torch's synchronous form, which enqueues on the CURRENT stream -- three partition streams + that one
        # are the process's four queues -- and two partitions cost the headline 7 % on one rank: 61.9 against 66-67 M.  --partitions 2 is the fallback if a node shows
        # the gather holding a partition up.)
        args.partitions = 3
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
    if not os.environ.get('PDB_BENCH_KEEP_OWN_STREAM'):   # (experiment: the batch stays on the library's own stream, which partition 0 then shares -- four partitions on four streams)
        b.set_stream(stream.cuda_stream)
    b.upload_actions(actions)
    if args.lane_setups:   # diagnostic: what the per-lane setup table's kernel pair costs -- every lane a row with its own springs, dampers, bars, ratios ...
        rs = np.random.RandomState(99)
        blocks = []
        for i in range(64):
            Q = pc.CarParams.from_buffer_copy(bytes(P))
            Q.arbK[0] = float(np.float32(Q.arbK[0] * rs.uniform(0.7, 1.3))); Q.finalRatio = float(Q.finalRatio * rs.uniform(0.9, 1.1))
            for w in range(4):
                Q.susp[w].k = float(np.float32(Q.susp[w].k * rs.uniform(0.85, 1.2))); Q.susp[w].damper.bumpSlow = float(np.float32(Q.susp[w].damper.bumpSlow * rs.uniform(0.8, 1.2)))
            blocks.append(Q)
        for f0 in range(0, n, 64):
            b.set_lane_tunes(blocks[:min(64, n - f0)], first=f0); b.set_lane_setups(blocks[:min(64, n - f0)], first=f0)
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

    # the scripted law in ONE launch: a = T_t + obs @ W (addmm: W = the P-steer's four columns; T_t = [sin phi_i, cos phi_i, 1] @ C_t, the throttle k sin(phi_i + w t) + c0 =
    # k cos(w t) sin phi_i + k sin(w t) cos phi_i + c0, tabulated per car over the law's period of 7 s = 2331 ticks: 305 MB at 16384 cars).  The law's clamp to [-1, 1] is
    # the env's own clip of its action space (projectd_env.py:159-160) and is done by the tick's pre-step on both components (step_kernel.hip.inc carPreStep: the same
    # values whether or not the buffer was clamped first), so it is not launched a second time here.  Every kernel boundary in a partition's stream costs 2.5-4.5 % of the
    # leg (profiles/r05_team_ab.txt); as eight elementwise launches the law held the 16384-car headline at 58 M whatever the kernels did
    site_tick = [0] * 5
    if policy == 'scripted':
        sw = np.zeros((24, 2), np.float32); sw[21, 0] = 0.03; sw[20, 0] = -0.03; sw[12, 0] = -1.0; sw[4, 0] = 0.15
        s_w = torch.from_numpy(sw).to(dev)
        s_sc = torch.stack([torch.sin(phi), torch.cos(phi), torch.ones_like(phi)], 1).contiguous()
        T_per = 7 * 333
        wt = (2.0 * np.pi / 7.0) * (np.arange(T_per, dtype=np.float64) / 333.0)
        # env action -> gas is linscale(a1, -1, 1, 0.1, 1.0) (projectd_env.py:160): a1 = (gas - 0.1) / 0.45 - 1, gas = 0.6 + 0.4 sin(.)
        cf = np.zeros((T_per, 3, 2), np.float32); cf[:, 0, 1] = (0.4 / 0.45) * np.cos(wt); cf[:, 1, 1] = (0.4 / 0.45) * np.sin(wt); cf[:, 2, 1] = 0.5 / 0.45 - 1.0
        s_cf = torch.from_numpy(cf).to(dev)
        s_tab = torch.matmul(s_sc, s_cf).contiguous()   # [T_per, n, 2]

    # Round 6: the linear laws run inside the tick's own launches (pdb_set_law: a = bias + obs @ W where the kernel writes the observation row; the scripted law's per-tick
    # bias is row `lawTick` of the same table) -- no launch between two ticks of a stream.  --no-device-law: the torch launch of rounds 1-5 (different rounding of the 24-term sum)
    # Measured on the round's final build (profiles/r06_law_ab.txt): the legs bound by a partition's chain gain 4-7 % from the law (4096-car env loop, 16384 cars on the mountain road /
    # nordring with the feedback law); the 16384-car headline, where three partitions saturate the GPU's issue slots, is 1.5-2 % FASTER with the torch launch between a partition's ticks
    # (it keeps the partitions' first passes apart: 128 against 150 us per launch).  So: the feedback law on the device by default, the scripted headline as a torch launch; --device-law / --no-device-law force a form
    device_law = policy in ('scripted', 'feedback') and (policy == 'feedback' or args.device_law) and not args.no_device_law and not do_scatter
    if device_law and policy == 'scripted':
        b.set_law(sw, table_device_ptr=s_tab.data_ptr(), period=T_per)
    elif device_law:
        b.set_law(fw, bias0=np.array([0.0, 0.3 * 12.0], np.float32))

    def policy_step(o, a, p=4, f=0):
        if device_law:
            return
        if policy == 'scripted':
            c = a.shape[0]
            torch.addmm(s_tab[site_tick[p] % T_per][f:f + c], o[:, :24], s_w, out=a)
            site_tick[p] += 1
            return
        if policy == 'feedback':   # oracle/scenarios.h scenarioFeedback (a linear law of the observation, clamped) as one addmm; the clamp is the pre-step's (above)
            torch.addmm(fb_b, o[:, :24], fb_w, out=a)
        elif policy == 'mlp':      # obs -> normalise -> 256 -> 256 -> 2, tanh-squashed like SAC's actor mean (hyperparams/sac.yml net_arch)
            # four launches: three GEMMs with their bias, the hidden layers' ReLU in the GEMM's epilogue, the tanh in place; hidden layers in buffers kept per row count
            c = a.shape[0]
            if c not in h1_buf:
                h1_buf[c] = torch.empty(c, 256, device=dev); h2_buf[c] = torch.empty(c, 256, device=dev)
            if not args.mlp_plain_relu:   # the hidden layers' ReLU as the GEMM's epilogue (torch._addmm_activation: hipBLASLt's fused form where the build has it): four launches
                h1 = torch._addmm_activation(b1, o[:, :24], w1, use_gelu=False, out=h1_buf[c])
                h2 = torch._addmm_activation(b2, h1, w2, use_gelu=False, out=h2_buf[c])
            else:
                h1 = torch.addmm(b1, o[:, :24], w1, out=h1_buf[c]).relu_()
                h2 = torch.addmm(b2, h1, w2, out=h2_buf[c]).relu_()
            torch.addmm(b3, h2, w3, out=a).tanh_()
        elif policy == 'random':   # fresh uniform actions every tick (an untrained agent: ends episodes quickly)
            a.uniform_(-1.0, 1.0)

    use_ring = policy == 'constant' and args.partitions > 1 and not do_scatter
    # a per-tick policy: per-partition closed loops, unless a gather has to see whole ticks (N > 1)
    # and only where a tick is long enough to hide the doubled number of (small) policy launches: the host enqueues ~10 per partition and tick
    # (N > 1: the partitions' kernels write their rows of the trajectory ring's slot; a ring's gather waits for every partition's last kernel of
    #  that ring and a ring's reuse is ordered on the partitions' own streams -- the same scheme as the open-loop rings)
    part_loops = policy not in ('constant', 'host', 'host_sync', 'host_mlp') and args.partitions > 1 and n >= args.part_loop_min and not do_scatter
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
        # Below 8192 cars the partition's WHOLE tick -- pdb_step_partition's two launches and the policy's -- is replayed from one captured graph (the loop is
        # launch-bound on the host there: 4096-car env loop 46.5 -> 53 M; the contact pass's grid is then fixed).  From 8192 cars up a graph launch costs the host
        # more than the launches it replaces (round 5: 54 against 57 M on the 16384-car headline) and the launches stay plain.  A graph holds the addresses it was
        # captured with: one per (partition, output block), captured the first time the pair comes up (inside the state-preparation ticks), after one plain pass.
        if gather.active and not args.ring_fork:
            ring_streams.extend(part_st)
        graphs = {}; plain_done = set(); graph_pool = [None] * args.partitions
        use_graph = not args.no_graph_policy and policy in ('feedback', 'mlp') and n < args.graph_max_cars and not gather.active and not (device_law and args.graph_policy_only)   # (from 8192 cars up a graph launch costs the host more than the launches it replaces: 54 against 57 M on the 16384-car headline)
        part_graph = graphs if use_graph else None
        part_graph_whole = use_graph and n < 8192 and not gather.active and not args.graph_policy_only
        if part_graph_whole:
            b.set_contact_grid(args.graph_contact_grid)

        def part_tick(p, o, f, c):
            """partition p's tick and policy, on its stream (the caller has made it current)"""
            key = (p, o.data_ptr())
            g = graphs.get(key) if use_graph else None
            if g is None and use_graph and p in plain_done and key not in graphs:
                try:
                    if graph_pool[p] is None:
                        graph_pool[p] = torch.cuda.graph_pool_handle()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=part_st[p], pool=graph_pool[p]):
                        if part_graph_whole:
                            b.step_partition(p, o.data_ptr())
                        policy_step(o[f:f + c], act_t[f:f + c], p, f)
                    graphs[key] = g
                except Exception as e:
                    sys.stderr.write('bench: the policy could not be captured (%r): plain launches\n' % (e,))
                    graphs[key] = None; g = None
            if g is not None:
                if not part_graph_whole:
                    b.step_partition(p, o.data_ptr())
                g.replay()
            else:
                b.step_partition(p, o.data_ptr())
                policy_step(o[f:f + c], act_t[f:f + c], p, f)
                plain_done.add(p)

    host_act = [np.ascontiguousarray(actions, dtype=np.float32).copy()]
    last_out = [out_t]

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
        o = gather.slot(t) if (gather.active or not part_loops) else out_t   # the kernel writes tick t straight into its trajectory-ring slot (no gather: the library's own block)
        last_out[0] = o
        if part_loops:     # every partition runs its own closed loop (kernel, then the policy on its rows) on its own stream
            for p in range(args.partitions):
                f, c = part_rng[p]
                torch.cuda.set_stream(part_st[p])      # (the context manager costs more host time than the launches it wraps)
                part_tick(p, o, f, c)
            torch.cuda.set_stream(stream)
            gather.after_tick(t)
            return
        if do_scatter:
            act_t.copy_(sharding.scatter_actions(scatter_src, n, world, rank, dev, dist))
        b.set_out_device_ptr(o.data_ptr())
        b.step_async()
        policy_step(o, act_t, 4, 0)
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
    # the dominant kernel's own duration, live: HIP events around every k-th first-pass launch of every launch site, on the stream it is launched on
    sample_every = args.sample_every if args.sample_every is not None else max(4, args.steps // 16)
    can_sample = hasattr(lib, 'pdb_sample_kernel') and not (part_loops and part_graph is not None and part_graph_whole)   # (launches replayed from a graph pass no event)
    k_us = 0.0; k_n = 0; k_cars = 0.0
    if can_sample:
        b.sample_kernel(sample_every)
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
        if can_sample:
            us, cnt, cars = b.sampled_kernel_us()
            k_us += us * cnt; k_n += cnt; k_cars += cars * cnt
        if regions[0][0] >= 0.2 or total >= 0.25 or len(regions) >= 400:   # the same decision on every rank (max-over-ranks times)
            break
    if can_sample:
        b.sample_kernel(0)
    regions.sort(key=lambda r: r[0])
    elapsed, region_ms, part_ms, launch_cars = regions[len(regions) // 2]

    res = None
    if rank == 0:
        traffic = None; valu_busy = None; prof = None
        for tag in (PROFILE_TAG, 'r04'):
            try:   # memory-side bytes per launch from the committed PMC passes of this same command (profiles/, tools/profile_round.sh)
                pm = json.load(open(os.path.join(ROOT, 'profiles', tag + '_pmc.json')))
                pc_cfg = pm.get('bench', {}).get('config', {})
                if pc_cfg.get('cars_per_gpu') == n and pc_cfg.get('partitions', 1) == (args.partitions if split else 1) and pm.get('workload_key', 'flat') == args.workload:
                    traffic = pm.get('traffic_bytes_per_launch'); valu_busy = pm.get('valu_issue_busy_frac', pm.get('valu_busy_frac_approx')); prof = tag
                    break
            except Exception:
                continue
        conc = 1
        tick_us = region_ms * 1000.0 / args.steps          # HIP events on the batch's stream around the timed region: one whole tick of the launch site
        if split:   # one launch = one partition's cars; HIP events on that partition's own stream (its whole loop: first pass, contact pass, policy)
            tick_us = part_ms * 1000.0 / args.steps
            conc = args.partitions
        else:
            launch_cars = n
        if k_n > 0:   # the first pass's own duration from the sampled launches' events
            kernel_us = k_us / k_n; launch_cars = k_cars / k_n; k_src = "HIP events around %d sampled pdb_step_kernel launches (every %d-th, on their own streams) inside the timed regions" % (k_n, sample_every)
        else:
            kernel_us = tick_us; k_src = "HIP events around the launch site's whole ticks (first pass + contact pass + gaps)"
        achieved = B_ALG * launch_cars / (kernel_us * 1e-6) / 1e9
        hdr = pc.TrackHeader.from_buffer_copy(trk[:C.sizeof(pc.TrackHeader)])
        pol = {'constant': "per-car constant random actions", 'scripted': "scripted gas 0.6+0.4sin(2pi t/7s+phi_i) + P-steer, " + ("evaluated by the tick's own launches (pdb_set_law)" if device_law else "a torch launch behind every tick"),
               'feedback': "probe-feedback law " + ("evaluated by the tick's own launches (pdb_set_law)" if device_law else "on the GPU (a torch launch behind every tick)"),
               'mlp': "24-256-256-2 MLP on the GPU", 'host': "probe-feedback law on the HOST, pipelined", 'host_sync': "probe-feedback law on the HOST, synchronous",
               'host_mlp': "24-256-256-2 MLP on the HOST, pipelined", 'random': "fresh random actions every tick"}[policy]
        where = ("reference %s spline (%d pts) as a road ribbon" % (args.workload, hdr.numFat)) if args.workload in ('ek_akina', 'ks_nordschleife') else \
                ("reference ks_nordschleife spline (%d pts) as a ribbon with guard rails" % hdr.numFat) if args.workload == 'ks_nordschleife_walls' else \
                ("reference track %s (%d triangles)" % (args.workload, hdr.numTris)) if is_ref else \
                "flat-plane track" if args.workload == 'flat' else \
                ("synthetic %s (%d triangles, %d spline pts%s)" % (args.workload, hdr.numTris, hdr.numFat, ", guard rails" if args.walls else ""))
        tag = {'flat': "configs[1]", 'ek_akina': "configs[2]", 'touge': "configs[2] shape"}.get(args.workload, "configs[4] shape")
        wl = "%s: %d cars/GPU, AE86, %s, %s%s, dt=1/333 s" % (tag, n, where, pol, ", env loop in the kernel (random-point resets)" if args.episodes and args.teleport_mode == 2 else ", env loop in the kernel" if args.episodes else "")
        coll = "none"
        if world > 1 or args.force_gather:
            coll = ("per partition and tick: rank 0's action rows -> tick -> all-gather of its [n,26] rows, issued by %s" % ("the library (RCCL)" if lib_exchange else "torch.distributed")) if part_exchange else \
                   ("RCCL all-gather of %d-tick trajectory rings [k,N,26] on the idle current stream, kernels write the ring in place%s" % (args.gather_ticks, "; actions scattered from rank 0 every tick" if args.scatter_actions else ""))
        res = {
            "metric": "env-steps/sec (333 Hz tick, 4-wheel car)",
            "value": n * world * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1000.0 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (+f64 drivetrain)", "data": "synthetic",
            "repeats": len(regions), "timed_region_s": total,
            "regime": "%d-step regions x%d (median), after %d settle + %d warm-up ticks" % (args.steps, len(regions), args.settle, args.warmup),
            "config": {"workload": wl, "cars_per_gpu": n, "partitions": (args.partitions if split else 1), "settle_ticks": args.settle, "collective": coll,
                       "parity": "GPU bit-exact vs CPU oracle; oracle bit-exact vs reference-TU goldens; rigid-body solve + contact generation unpinned (ODE absent)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "pdb_step_kernel", "kernel_avg_us": kernel_us, "kernel_avg_source": k_src,
                         "alg_bytes_per_car_tick": B_ALG, "cars_per_launch": launch_cars, "concurrent_launches": conc, "site_tick_us": tick_us,
                         "device_achieved": B_ALG * n * args.steps / elapsed / 1e9, "device_frac": B_ALG * n * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,   # whole device, from the wall time of the timed region
                         "traffic_source": ("profiles/%s_pmc.json" % prof if traffic else None), "valu_issue_busy_frac": valu_busy},
        }
        if hasattr(lib, 'pdb_contact_pass_load'):
            res["contact_pass_cars"] = [int(lib.pdb_contact_pass_load(b.h, q)) for q in (list(range(args.partitions)) if split else [4])]   # cars the last contact passes held (diagnostic)
        if args.episodes and not policy.startswith('host') and world == 1:   # how often episodes end in this workload: counted over 300 more ticks, outside the timed region (one rank only: these ticks would be collectives of rank 0 alone)
            ends = torch.zeros((), dtype=torch.int64, device=dev)
            for _ in range(300):
                tick()
                ends += ((last_out[0][:, 25].view(torch.int32) & 8) != 0).sum()
            res["episode_ends_per_tick"] = float(ends.item()) / 300.0
        if want_cpu:   # the CPU leg is timed at N = 1 only
            res["cpu_baseline"] = cpu_baseline(P, trk, S0, all_actions)
    b.close()
    return res


EXTRA = [   # (key, argv) -- the other BASELINE configs' shapes, each measured by measure() in this process at N = 1
    ("configs2_16384_dense_spline", ['--workload', 'touge', '--cars', '16384', '--spline-step', '0.9', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs2_16384", ['--workload', 'touge', '--cars', '16384', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs4_shape_16384_walls_mlp", ['--workload', 'touge', '--cars', '16384', '--walls', '--policy', 'mlp', '--steps', '300', '--warmup', '50', '--settle', '200']),
    ("configs3_shard_8192", ['--workload', 'flat', '--cars', '8192', '--steps', '600', '--warmup', '100']),
    ("configs3_shard_8192_gather_k1_scatter", ['--workload', 'flat', '--cars', '8192', '--steps', '300', '--warmup', '50', '--force-gather', '--gather-ticks', '1', '--scatter-actions']),
    ("configs3_shard_8192_gather_k1_scatter_torch", ['--workload', 'flat', '--cars', '8192', '--steps', '300', '--warmup', '50', '--force-gather', '--gather-ticks', '1', '--scatter-actions', '--torch-exchange']),
    ("configs3_shard_8192_gather_k32", ['--workload', 'flat', '--cars', '8192', '--steps', '600', '--warmup', '100', '--force-gather', '--gather-ticks', '32']),
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
    ap.add_argument('--sample-every', type=int, default=None, help='HIP events around every k-th first-pass launch of every launch site (default max(4, steps // 16); 0: none, the roofline block then falls back to the whole tick)')
    ap.add_argument('--settle', type=int, default=None, help='ticks of state preparation before warm-up (cars come off their springs and get rolling); neither warm-up nor timed')
    ap.add_argument('--cars', type=int, default=None, help='cars per GPU (default: 16384 for the headline workload, 4096 with an explicit --workload)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--part-loop-min', type=int, default=4096, help='per-tick policies: from this many cars up every partition runs its own closed loop on its own stream')
    ap.add_argument('--no-extra', action='store_true', help='(accepted for older command lines: the extra legs are opt-in now)')
    ap.add_argument('--extra', action='store_true', help='after the line is printed: measure the other configs\' shapes too (results to stderr and gpurun_out/bench_extra.json)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs[1] measurement that rides in the line as `secondary`')
    ap.add_argument('--partitions', type=int, default=None, choices=[1, 2, 3, 4],
                    help='free-running car ranges per GPU, one HIP stream each (pdb_set_partitions / pdb_step_ring); 1 = one launch per tick.  Default 3: a process has four hardware queues -- three partitions and the current stream, on which the ring gather is issued, use them up; a FIFTH busy stream would share a queue with a partition and hold it up for as long as its kernels run (tools/hwqueue_probe.py: 47 M against 69 M with a 1 ms kernel per ring on a fifth stream; with two partitions 63-67 M against 67 M): --partitions 2 if a node shows that')
    ap.add_argument('--walls', action='store_true', help='touge workload: line both edges of the road with WALL surfaces (configs[4] shape: hull-vs-wall narrow phase next to the guard rails)')
    ap.add_argument('--spline-step', type=float, default=0.0, help='touge workload: metres between spline points (default 5 m = 891 points; 0.9 = 4.9 k points, the density of the reference tracks)')
    ap.add_argument('--lane-setups', action='store_true', help='every lane with a setup row of its own (pdb_set_lane_tunes + pdb_set_lane_setups: the kernel pair compiled for the table; never the bench line)')
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
    ap.add_argument('--graph-contact-grid', type=int, default=96, help='workgroups of the contact pass inside a captured per-partition tick')
    ap.add_argument('--mlp-plain-relu', action='store_true', help='mlp policy: the hidden layers as addmm + relu_ (six launches a tick, rounds 3-5) instead of torch._addmm_activation (ReLU in the GEMM epilogue: four) (A/B)')
    ap.add_argument('--mlp-fused-relu', action='store_true', help=argparse.SUPPRESS)   # (the default since round 6)
    ap.add_argument('--device-law', action='store_true', help='scripted policy: the law evaluated by the tick\'s own launches (pdb_set_law) instead of the torch launch behind every tick (the default for the feedback policy) (A/B)')
    ap.add_argument('--no-device-law', action='store_true', help='scripted / feedback policies: the law as a torch launch behind every tick (rounds 1-5) instead of the device law, pdb_set_law (A/B)')
    ap.add_argument('--graph-max-cars', type=int, default=8192, help='per-partition policy loops: captured graphs below this many cars per GPU (the whole tick below 8192 in any case; above, with --graph-policy-only, the policy launches alone) (A/B)')
    ap.add_argument('--no-graph-policy', action='store_true', help='per-partition policy loops: the policy as plain torch launches instead of one captured graph per partition (A/B)')
    ap.add_argument('--ring-fork', action='store_true', help='ring mode: every ring starts behind the batch stream (the older form; A/B)')
    ap.add_argument('--torch-exchange', action='store_true', help='the per-partition exchange (--scatter-actions with --gather-ticks 1) through torch.distributed instead of the library\'s own RCCL communicators (A/B)')
    ap.add_argument('--force-gather', action='store_true', help='run the observation all-gather even with one rank (exercises the RCCL + side-stream path on a single GPU)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the multi-rank path on a single-GPU box)')
    ap.add_argument('--teleport-mode', type=int, default=0, choices=[0, 1, 2], help='--episodes: where a reset puts the car (projectd_env.py teleport_mode: 0 start, 1 nearest, 2 random point of the lap)')
    ap.add_argument('--workload', choices=['flat', 'touge', 'playground', 'nordring', 'driftplayground', 'ebisu_touge', 'yamanashi_short', 'euphoria_hillside_park', 'ek_akina', 'ks_nordschleife', 'ks_nordschleife_walls'], default=None,
                    help='default (none given): BASELINE configs[2] as worded -- 16384 cars on the ek_akina ribbon, scripted inputs, env loop.  flat = configs[1]; touge = configs[2] shape on the synthetic closed hilly road')
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


def resolve(args):
    """no --workload: the headline configuration (BASELINE configs[2] as worded); an explicit workload keeps the older defaults"""
    if args.workload is None:
        args.workload = HEADLINE['workload']
        args.cars = args.cars or HEADLINE['cars']
        args.policy = args.policy or HEADLINE['policy']
        args.episodes = True
        args.teleport_mode = HEADLINE['teleport_mode']
        if args.settle is None:
            args.settle = HEADLINE['settle']
        args.is_headline = True
    else:
        args.cars = args.cars or CARS_PER_GPU
        if args.settle is None:
            args.settle = 333 if args.workload == 'flat' else 200
        args.is_headline = False
    return args


def _short(x, n):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 1] + '~'


def compact(res, secondary=None, rccl=None, extra_file=None):
    """the ONE stdout line: the contract's keys + roofline + cpu_baseline (+ secondary, rccl), kept under 2 KB"""
    r = res["roofline"]; c = res["config"]
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "repeats", "timed_region_s", "regime")}
    if "episode_ends_per_tick" in res:
        out["episode_ends_per_tick"] = float('%.4g' % res["episode_ends_per_tick"])
    for k in ("value", "ms_per_step", "timed_region_s"):
        out[k] = float('%.6g' % out[k])
    out["config"] = {"workload": _short(c["workload"], 230), "cars_per_gpu": c["cars_per_gpu"], "partitions": c["partitions"], "collective": _short(c["collective"], 120), "parity": _short(c["parity"], 150)}
    ach = float('%.5g' % r["achieved"])
    out["roofline"] = {"bound": r["bound"], "achieved": ach, "peak": r["peak"], "unit": r["unit"], "frac": float('%.6g' % (ach / r["peak"])),
                       "traffic": (float('%.5g' % r["traffic"]) if r.get("traffic") else None), "kernel": r["kernel"], "kernel_avg_us": float('%.5g' % r["kernel_avg_us"]),
                       "alg_bytes_per_car_tick": r["alg_bytes_per_car_tick"], "cars_per_launch": float('%.6g' % r["cars_per_launch"]), "concurrent_launches": r["concurrent_launches"],
                       "device_frac": float('%.4g' % r["device_frac"])}
    cb = res.get("cpu_baseline") or (secondary or {}).get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": float('%.6g' % cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": _short(cb["sample"], 130),
                               "all_cores_value": float('%.6g' % cb["all_cores_value"]), "all_cores": cb["all_cores"]}
    if secondary:
        out["secondary"] = {"workload": _short(secondary["config"]["workload"], 100), "value": float('%.6g' % secondary["value"]), "ms_per_step": float('%.6g' % secondary["ms_per_step"]),
                            "roofline_frac": float('%.4g' % secondary["roofline"]["frac"]), "kernel_avg_us": float('%.5g' % secondary["roofline"]["kernel_avg_us"]),
                            "traffic": (float('%.5g' % secondary["roofline"]["traffic"]) if secondary["roofline"].get("traffic") else None)}
    if rccl:
        out["rccl"] = rccl
    if extra_file:
        out["extra_file"] = extra_file
    return out


def rccl_block(world, rank, local_rank, dist, dev_index):
    """who RCCL saw: every rank reports (rank, PCI bus id of its GPU) through the process group the bench uses"""
    import torch
    pr = torch.cuda.get_device_properties(dev_index)
    me = "%s:%s" % (getattr(pr, 'pci_bus_id', '?'), getattr(pr, 'pci_device_id', '?'))
    try:
        me = "%04x:%02x:%02x" % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    if dist is None or not dist.is_initialized():
        return {"world": 1, "ranks_seen": 1, "backend": None, "devices": [me]}
    box = [None] * world
    dist.all_gather_object(box, (rank, me))
    # and one device collective through the backend itself: the sum of (rank + 1) over the ranks it reached
    t = torch.tensor([float(rank + 1)], device=('cuda:%d' % dev_index) if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t)
    return {"world": world, "ranks_seen": len({r for r, _ in box}), "backend": dist.get_backend(), "devices": [d for _, d in sorted(box)],
            "distinct_devices": len({d for _, d in box}), "allreduce_ok": bool(abs(float(t.item()) - world * (world + 1) / 2.0) < 1e-3)}


def main():
    ap = parser()
    args = resolve(ap.parse_args())
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
    rccl = None
    dev_index = local_rank % torch.cuda.device_count()
    if world > 1 or args.force_gather:
        import datetime
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(dev_index)
        try:   # a rank that cannot set the collectives up leaves with a non-zero code -- and so does every other rank (the rendezvous / the check below time out on them)
            dist.init_process_group(args.backend, init_method='env://', timeout=datetime.timedelta(seconds=300))
            rccl = rccl_block(world, rank, local_rank, dist, dev_index)
            if rccl["ranks_seen"] != world or not rccl["allreduce_ok"]:
                raise RuntimeError('the process group reached %d of %d ranks' % (rccl["ranks_seen"], world))
        except Exception as e:
            sys.stderr.write('bench: rank %d: collective set-up failed: %r\n' % (rank, e))
            os._exit(3)
    else:
        rccl = rccl_block(1, 0, local_rank, None, dev_index)

    res = measure(args, world, rank, local_rank, dist, want_cpu=(not args.no_cpu_baseline and world == 1 and not args.is_headline))
    secondary = None
    if args.is_headline and world == 1 and not args.no_secondary:   # configs[1] beside it, same rule, same process; the CPU leg is timed on ITS inputs
        a2 = resolve(parser().parse_args(SECONDARY_ARGV + ['--steps', str(args.steps), '--warmup', str(args.warmup)] + (['--no-cpu-baseline'] if args.no_cpu_baseline else [])))
        try:
            secondary = measure(a2, 1, 0, local_rank, None, want_cpu=not args.no_cpu_baseline)
        except Exception as e:   # never at the cost of the headline
            sys.stderr.write('bench: the secondary (configs[1]) measurement failed: %r\n' % (e,))
    only = [k for k in os.environ.get('PDB_BENCH_EXTRA', '').split(',') if k]
    want_extra = world == 1 and (args.extra or only) and args.is_headline
    extra_file = None
    if want_extra:
        try:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            extra_file = 'gpurun_out/bench_extra.json'
        except OSError:
            extra_file = None
    if rank == 0:
        line = json.dumps(compact(res, secondary, rccl, extra_file), separators=(',', ':'))
        json_out.write(line + '\n')
        json_out.flush()
        sys.stderr.write('bench: full headline record: %s\n' % json.dumps(res))
        if secondary:
            sys.stderr.write('bench: full secondary record: %s\n' % json.dumps(secondary))

    # ---- everything below is optional and can no longer cost the line ----
    if want_extra:
        extra = {"headline": res, "secondary": secondary}
        for key, argv in ([(k, dict(EXTRA)[k]) for k in only] if only else EXTRA):
            a = resolve(parser().parse_args(argv + ['--no-cpu-baseline']))
            d2 = None
            try:
                if a.force_gather:
                    import torch.distributed as d2
                    if not d2.is_initialized():
                        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29512')
                        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
                        d2.init_process_group(args.backend, init_method='env://')
                r = measure(a, 1, 0, local_rank, d2)
                extra[key] = r
                sys.stderr.write('bench: extra %s: %.4g env-steps/s, %.4g ms/step, first pass %.4g us\n' % (key, r["value"], r["ms_per_step"], r["roofline"]["kernel_avg_us"]))
            except Exception as e:
                extra[key] = {"error": repr(e)[:300]}
                sys.stderr.write('bench: extra %s failed: %r\n' % (key, e))
            if extra_file:
                json.dump(extra, open(os.path.join(ROOT, extra_file), 'w'), indent=1)
    if world > 1 and args.extra and not args.scatter_actions:   # the other collective variant SURVEY 8d words (a gather + an action scatter every tick on configs[3]'s shard), after the line
        a2 = resolve(parser().parse_args(['--workload', 'flat', '--cars', '8192', '--gpus', str(args.gpus), '--steps', str(args.steps), '--warmup', str(args.warmup), '--gather-ticks', '1', '--scatter-actions', '--no-cpu-baseline',
                                          '--backend', args.backend] + (['--library-exchange'] if args.library_exchange else [])))
        import threading
        watchdog = threading.Timer(240.0, lambda: os._exit(0))   # a rank that failed alone leaves the others inside a collective: everybody leaves (the line is out)
        watchdog.daemon = True
        watchdog.start()
        try:
            r2 = measure(a2, world, rank, local_rank, dist)
            if rank == 0:
                sys.stderr.write('bench: configs3_gather_k1_scatter: %s\n' % json.dumps(r2))
        except Exception as e:
            sys.stderr.write('bench: configs3_gather_k1_scatter failed: %r\n' % (e,))
        watchdog.cancel()
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()
    else:
        import torch.distributed as d3
        if d3.is_initialized():
            d3.destroy_process_group()


if __name__ == '__main__':
    main()
