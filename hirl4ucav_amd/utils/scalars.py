"""Scalar logging with the reference's tags (hirl/train_all.py:97-100,362-367,383-388): a TensorBoard SummaryWriter when the
`tensorboard` package is present (it is not in every ROCm image), otherwise the same add_scalar calls land in
`<summary_dir>/scalars.jsonl`, one {"tag", "value", "step"} object per line — nothing is silently dropped."""
import json
import os


class JsonlWriter:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self._f = open(self.path, "a")

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def make_writer(summary_dir):
    """SummaryWriter(summary_dir) (train_all.py:236) or its JSONL stand-in."""
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(summary_dir)
    except Exception:  # tensorboard missing: ImportError here, ModuleNotFoundError inside torch's shim
        return JsonlWriter(summary_dir)
