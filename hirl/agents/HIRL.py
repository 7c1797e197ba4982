"""hirl/agents/HIRL.py of the reference -> hirl4ucav_amd.agents.HIRL (same class name, constructor and methods)."""
from hirl4ucav_amd.agents.HIRL import *  # noqa: F401,F403
from hirl4ucav_amd.agents.HIRL import Agent  # noqa: F401
