"""hirl/environments/HarfangEnv_GYM.py of the reference -> hirl4ucav_amd.environments.HarfangEnv_GYM (HarfangEnv, HarfangSerpentineEnv,
HarfangCircularEnv, HarfangSerpentineInfiniteEnv: reset / random_reset / step / step_test, one env per instance like the reference)."""
from hirl4ucav_amd.environments.HarfangEnv_GYM import *  # noqa: F401,F403
from hirl4ucav_amd.environments.HarfangEnv_GYM import HarfangCircularEnv, HarfangEnv, HarfangSerpentineEnv, HarfangSerpentineInfiniteEnv  # noqa: F401
