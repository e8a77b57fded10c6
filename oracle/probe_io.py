"""ORACLE / TEST INFRASTRUCTURE.  Reader for the probe files written by oracle/probe.h."""
import numpy as np, struct
def load(path):
    with open(path, 'rb') as f:
        magic, nf, nr = struct.unpack('<3i', f.read(12))
        assert magic == 0x50444F52
        nl, = struct.unpack('<i', f.read(4))
        names = f.read(nl).decode().split('\n')[:-1]
        ticks = np.frombuffer(f.read(4 * nr), dtype='<i4')
        actions = np.frombuffer(f.read(8 * nr), dtype='<f4').reshape(nr, 2)
        data = np.frombuffer(f.read(8 * nr * nf), dtype='<f8').reshape(nr, nf)
    assert len(names) == nf
    return dict(names=names, ticks=ticks, actions=actions, data=data, idx={n: i for i, n in enumerate(names)})
