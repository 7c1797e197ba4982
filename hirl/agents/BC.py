"""hirl/agents/BC.py of the reference -> hirl4ucav_amd.agents.BC (same class name, constructor and methods)."""
from hirl4ucav_amd.agents.BC import *  # noqa: F401,F403
from hirl4ucav_amd.agents.BC import Agent  # noqa: F401
