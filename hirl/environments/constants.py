"""hirl/environments/constants.py of the reference -> hirl4ucav_amd.environments.constants (NormStates)."""
from hirl4ucav_amd.environments.constants import *  # noqa: F401,F403
