"""`hirl` — alias package: the reference's module paths (hirl/agents/HIRL.py, hirl/utils/buffer.py, hirl/environments/HarfangEnv_GYM.py
...) re-exporting the MI355X-native implementations of `hirl4ucav_amd`.

The reference's drivers import script-relative modules (`from agents.HIRL import Agent`, hirl/train_all.py:1-9) while its agents
import package-absolute ones (`from hirl.utils.buffer import *`, hirl/agents/HIRL.py:7).  Both forms resolve here with ZERO edits:
copy (or symlink) the reference's own `train_all.py` / `train_sac.py` / `validate_all.py` into this directory and run them from
it — `python hirl/train_all.py --agent HIRL --type soft --env straight_line --random` — with the repository root on PYTHONPATH.
`local_config.yaml` beside this file is the sample the drivers read (train_all.py:143-149); no Harfang process is needed: `df.connect`
and friends are accepted and ignored (hirl4ucav_amd/environments/dogfight_client.py).  See INTEGRATION.md, level 1."""
