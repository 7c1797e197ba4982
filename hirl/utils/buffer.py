"""hirl/utils/buffer.py of the reference -> hirl4ucav_amd.utils.buffer.  The drivers star-import this module and rely on it for
`torch`, `np` and `device` (train_all.py:112, HIRL.py:153): all three are re-exported."""
from hirl4ucav_amd.utils.buffer import *  # noqa: F401,F403
from hirl4ucav_amd.utils.buffer import UniformMemory, Transition, device, np, torch  # noqa: F401
