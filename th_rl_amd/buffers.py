"""ReplayBuffer -- host-side mirror of the reference container (th_rl/buffers.py:6-41).

In the fused device path (GameBatch.run) the append -> replay -> empty cycle
lives inside the kernels; this class is the object-level container the
duck-typed protocol (`agent.memory.append(...)`, trainer.py:62) needs.  It
stores and returns data only -- no arithmetic.
"""
from collections import deque

import numpy
import torch


class ReplayBuffer:
    def __init__(self, capacity, experience):
        self.capacity = capacity
        self.buffer = deque(maxlen=capacity)
        self.experience = experience

    def __len__(self):
        return len(self.buffer)

    def append(self, *args):
        self.buffer.append(self.experience(*args))

    def _gather(self, indices, cast, as_array):
        columns = zip(*[self.buffer[i] for i in indices])
        if cast:
            wrap = (lambda t: numpy.array(t)) if as_array else (lambda t: t)
            columns = (torch.tensor(wrap(t), dtype=dt) for t, dt in zip(columns, cast))
        return columns

    def sample(self, batch_size, cast=None):
        """Uniform sample without replacement (dead code in the reference, kept for API surface)."""
        indices = numpy.random.choice(len(self.buffer), batch_size, replace=False)
        return self._gather(indices, cast, as_array=False)

    def replay(self, cast=None, replay_size=0):
        """All entries (or the last `replay_size`) in insertion order, transposed."""
        n = len(self.buffer)
        first = 0 if replay_size == 0 else n - replay_size
        return self._gather(range(first, n), cast, as_array=True)

    def empty(self):
        self.buffer = deque(maxlen=self.capacity)
