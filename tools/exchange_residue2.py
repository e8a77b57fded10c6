#!/usr/bin/env python3
"""Diagnostic (GPU box): which bench leg leaves later legs slower?  usage: exchange_residue2.py <first leg args...>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import bench
def run(tag, argv, dist=None):
    a = bench.parser().parse_args(argv + ['--no-cpu-baseline', '--no-extra'])
    r = bench.measure(a, 1, 0, 0, dist)
    print('%-60s %.2f M' % (tag, r['value'] / 1e6), flush=True)
plain = ['--cars', '8192', '--steps', '600', '--warmup', '100']
run('plain, before', plain)
import torch.distributed as d2
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29512'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
d2.init_process_group('nccl', init_method='env://')
run('plain, torch NCCL group initialised', plain)
first = sys.argv[1:]
run('first leg: ' + ' '.join(first), plain + first, d2)
run('plain, after it', plain)
run('k32 ring gather after it', plain + ['--force-gather', '--gather-ticks', '32'], d2)
run('plain again', plain)
