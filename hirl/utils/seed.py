"""hirl/utils/seed.py of the reference -> hirl4ucav_amd.utils.seed (set_seed)."""
from hirl4ucav_amd.utils.seed import *  # noqa: F401,F403
from hirl4ucav_amd.utils.seed import set_seed  # noqa: F401
