"""GPU-resident replay ring — the device counterpart of UniformMemory (reference hirl/utils/buffer.py:11-54).

Rows are 32 fp32 words = one 128-B line: s[13] a[4] s'[13] r done (Transition, buffer.py:8; step_success is kept
in a parallel int8 array because sample() never returns it, buffer.py:47-48).  `total` counts transitions ever
stored; position = total % capacity (buffer.py:36), len = min(total, capacity).
"""
import torch

from .. import _lib


class DeviceReplay:
    def __init__(self, capacity, device="cuda"):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.ring = torch.zeros((self.capacity, _lib.ROW_WORDS), dtype=torch.float32, device=self.device)
        self.success = torch.zeros(self.capacity, dtype=torch.int8, device=self.device)
        self.total = torch.zeros(1, dtype=torch.int64, device=self.device)

    def __len__(self):  # buffer.py:53-54 (host sync: for drivers/tests, not the hot loop)
        return min(int(self.total.item()), self.capacity)

    def fullEnough(self, batchSize):  # buffer.py:50-51
        return len(self) >= batchSize

    def store_rows(self, rows, success=None):
        """Append pre-built rows [k, 32] (expert buffer fill, train_all.py:289-306; tests)."""
        rows = rows.to(self.device, torch.float32).reshape(-1, _lib.ROW_WORDS)
        k = rows.shape[0]
        start = int(self.total.item())
        idx = (torch.arange(k, device=self.device) + start) % self.capacity
        self.ring[idx] = rows
        if success is not None:
            self.success[idx] = success.to(self.device, torch.int8)
        self.total += k
        self.fixed_len = min(start + k, self.capacity)  # host-known live length (tables filled once: expert ring)


# ---- reference-API façade -------------------------------------------------------------------------------------------
import random  # noqa: E402
from collections import namedtuple  # noqa: E402

import numpy as np  # noqa: E402,F401  (the reference's drivers obtain `np`, `torch`, `device` from this module's star-export)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")  # buffer.py:6

Transition = namedtuple("Transition", ("state", "action", "next_state", "reward", "done", "step_success"))  # buffer.py:8


class UniformMemory(DeviceReplay):
    """hirl.utils.buffer.UniformMemory (buffer.py:11-54) over the device ring: store() appends one transition,
    sample() draws without replacement with Python's `random` like the reference and returns the five tuples."""

    def __init__(self, capacity, upsample=False):
        if upsample:
            raise NotImplementedError("priority upsampling is disabled in the reference (HIRL.py:184,189)")
        super().__init__(int(capacity), device)
        self.upsample = False
        self._len = 0
        self.position = 0

    def store(self, state, action, next_state, reward, done, step_success=0):  # buffer.py:20-36
        row = np.zeros(32, np.float32)
        row[0:13], row[13:17], row[17:30] = state, action, next_state
        row[30], row[31] = reward, float(done)
        self.ring[self.position] = torch.from_numpy(row).to(self.device)
        self.success[self.position] = int(step_success)
        self.position = (self.position + 1) % self.capacity
        self._len = min(self._len + 1, self.capacity)
        self.total += 1

    @property
    def memory(self):  # drivers only take len() of it (train_all.py:306,308)
        return range(self._len)

    def __len__(self):
        return self._len

    def fullEnough(self, batchSize):
        return self._len >= batchSize

    def sample_indices(self, batchSize):
        return random.sample(range(self._len), batchSize)  # buffer.py:45

    def sample(self, batchSize):  # buffer.py:38-48
        rows = self.ring[torch.as_tensor(self.sample_indices(batchSize), device=self.device)].cpu().numpy()
        return (tuple(rows[:, 0:13]), tuple(rows[:, 13:17]), tuple(rows[:, 17:30]), tuple(rows[:, 30]), tuple(rows[:, 31]))
