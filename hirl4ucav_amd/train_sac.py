"""hirl/train_sac.py's command line on the vectorised driver: `python -m hirl4ucav_amd.train_sac --type SAC|ESAC --env serpentine --random --num_envs 16384`
is `python -m hirl4ucav_amd.train_all --agent SAC ...` (train_all.py holds the SAC / E-SAC loop of train_sac.py:224-437 beside the HIRL / TD3 / BC ones).
Flags and defaults: train_sac.py:441-457 (`--type` defaults to ESAC there), plus train_all's extras."""
from . import train_all as T


def parser():
    p = T.parser()
    p.set_defaults(agent="SAC", type="ESAC")  # train_sac.py:448
    return p


if __name__ == "__main__":
    T.main(parser().parse_args())
