"""hirl/environments/dogfight_client.py of the reference -> the no-socket shim (connect / disable_log / set_renderless_mode /
set_client_update_mode are accepted and ignored: the simulator is in-process)."""
from hirl4ucav_amd.environments.dogfight_client import *  # noqa: F401,F403
