"""hirl/agents/TD3.py of the reference -> hirl4ucav_amd.agents.TD3 (same class name, constructor and methods)."""
from hirl4ucav_amd.agents.TD3 import *  # noqa: F401,F403
from hirl4ucav_amd.agents.TD3 import Agent  # noqa: F401
