#!/usr/bin/env python3
"""Diagnostic (GPU box): does creating / using / destroying the library's RCCL communicators leave something behind that slows later batches?
usage: exchange_residue.py [init|steps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import bench
import pdbatch, pdb_ctypes as pc
mode = sys.argv[1] if len(sys.argv) > 1 else 'init'
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if len(sys.argv) > 3 and sys.argv[3] == 'pg':
    import torch.distributed as d2
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29512'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    d2.init_process_group('nccl', init_method='env://')
    x = torch.zeros(8, device='cuda'); d2.all_reduce(x); torch.cuda.synchronize()
def rate(tag):
    a = bench.parser().parse_args(['--cars', '8192', '--steps', '600', '--warmup', '100', '--no-cpu-baseline', '--no-extra'])
    r = bench.measure(a, 1, 0, 0, None)
    print('%-40s %.2f M' % (tag, r['value'] / 1e6), flush=True)
rate('before')
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
b = pdbatch.Batch(8192, P, trk, device=0, action_mode=1)
b.set_partitions(3)
ids = b.comm_unique_ids(3)
b.comm_init(1, 0, ids)
if mode == 'steps':
    g = [torch.empty((1, b.partition_range(p)[1], 26), device='cuda') for p in range(3)]
    s = [torch.zeros((1, b.partition_range(p)[1], 2), device='cuda') for p in range(3)]
    for t in range(nsteps):
        for p in range(3):
            b.step_exchange_partition(p, s[p].data_ptr(), g[p].data_ptr())
    torch.cuda.synchronize()
rate('with the communicators alive')
b.close()
rate('after pdb_destroy (ncclCommDestroy)')
