"""hirl/agents/SAC/agent.py of the reference -> hirl4ucav_amd.agents.SAC.agent (SacAgent: explore / exploit / learn / save_models, .memory)."""
from hirl4ucav_amd.agents.SAC.agent import *  # noqa: F401,F403
from hirl4ucav_amd.agents.SAC.agent import SacAgent  # noqa: F401
