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
