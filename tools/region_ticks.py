"""Diagnostic (GPU box): where a short timed region of the bench workload loses its time -- per-tick end times of every partition inside 20-tick regions
that start from an idle, synchronised device (the driver's --steps 20), from events on the partitions' own streams.  usage: python3 tools/region_ticks.py [cars] [ticks]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import pdbatch, parity_util as pu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
b = pdbatch.Batch(n, P, trk, 0, 1)
if not os.environ.get('PDB_OWN_STREAM'):
    b.set_stream(torch.cuda.current_stream().cuda_stream)   # as bench.py does: the library's own stream is gone, three partition streams + the null stream = the process's four hardware queues
b.upload_actions(pu.make_actions(n, 1234))   # (no kernel ever runs on the batch's own stream here, as in bench.py: which streams share a hardware queue depends on the order of their first use)
out = torch.zeros(n, 26, device=dev)
b.set_partitions(3)
st = [torch.cuda.ExternalStream(b.partition_stream(p), device=dev) for p in range(3)]
def region(record):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] for _ in range(3)] if record else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if record:
        for p in range(3): ev[p][0].record(st[p])
    for i in range(K):
        for p in range(3):
            b.step_partition(p, out.data_ptr())
            if record: ev[p][i + 1].record(st[p])
    b.wait_partitions(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6, ev
for _ in range(40): region(False)   # (settles the cars too)
plain = np.median([region(False)[0] for _ in range(50)])
rows = []
for _ in range(30):
    us, ev = region(True)
    rows.append([[ev[p][0].elapsed_time(ev[p][i + 1]) * 1e3 for i in range(K)] for p in range(3)])
r = np.median(np.array(rows), axis=0)      # [3][K]: end of tick i of partition p, us since the partition's first event
print('%d cars, %d-tick regions from an idle device: wall %.0f us without the events' % (n, K, plain))
for p in range(3):
    d = np.diff(np.concatenate([[0.0], r[p]]))
    print('partition %d: tick durations (us) ' % p + ' '.join('%5.1f' % x for x in d) + '   last tick ends at %.0f' % r[p][-1])
