import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype); self.high = np.asarray(high, dtype=dtype)
        self.dtype = np.dtype(dtype); self.shape = self.low.shape

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)
