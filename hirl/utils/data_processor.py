"""hirl/utils/data_processor.py of the reference -> hirl4ucav_amd.utils.data_processor (read_data / write_data: the two-row CSV)."""
from hirl4ucav_amd.utils.data_processor import *  # noqa: F401,F403
from hirl4ucav_amd.utils.data_processor import read_data, write_data  # noqa: F401
