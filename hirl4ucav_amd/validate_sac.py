"""hirl/validate_sac.py's command line: `python -m hirl4ucav_amd.validate_sac --model_dir <run>/model --model_name <tag> --random [--infinite]` is
`python -m hirl4ucav_amd.validate_all --agent SAC ...` (validate_all.py runs both reference drivers' rounds as batches).  Flags: validate_sac.py:191-201."""
from . import validate_all as V


def parser():
    p = V.parser()
    p.set_defaults(agent="SAC", type="ESAC")  # validate_sac.py:195
    return p


if __name__ == "__main__":
    V.main(parser().parse_args())
