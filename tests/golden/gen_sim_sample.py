#!/usr/bin/env python3
"""The ONE sample of the external simulator's plane state the reference holds -> tests/golden/sim_state_sample.npz.

hirl/data/straight_line/ai_env.py:18 keeps, in a comment, a `get_plane_state` reply of the Harfang sandbox (the simulator whose
source is not in the reference).  It is the only data point that ties the re-derived model's sign and unit conventions
(docs/DYNAMICS.md "Conventions") to Harfang.  This script parses that literal where it lies (development container only:
/root/reference never travels) and commits its numeric fields as a fixture.

    python tests/golden/gen_sim_sample.py
"""
import ast
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/hirl/data/straight_line/ai_env.py"


def main():
    line = open(REF_FILE).read().split("\n")[17]  # line 18
    d = ast.literal_eval(re.search(r"\{.*\}", line).group(0))
    assert d["type"] == "AICRAFT" and d["target_id"] == "ennemy_2"
    np.savez(os.path.join(HERE, "sim_state_sample.npz"),
             source=np.array("hirl/data/straight_line/ai_env.py:18 (get_plane_state reply, comment)"),
             timestep=np.float64(d["timestep"]), position=np.asarray(d["position"], np.float64),
             euler_angles=np.asarray(d["Euler_angles"], np.float64), move_vector=np.asarray(d["move_vector"], np.float64),
             horizontal_speed=np.float64(d["horizontal_speed"]), vertical_speed=np.float64(d["vertical_speed"]),
             linear_speed=np.float64(d["linear_speed"]), altitude=np.float64(d["altitude"]), heading_deg=np.float64(d["heading"]),
             pitch_attitude_deg=np.float64(d["pitch_attitude"]), roll_attitude_deg=np.float64(d["roll_attitude"]),
             target_angle_deg=np.float64(d["target_angle"]), thrust_level=np.float64(d["thrust_level"]),
             target_locked=np.bool_(d["target_locked"]), target_out_of_range=np.bool_(d["target_out_of_range"]),
             user_levels=np.asarray([d["user_pitch_level"], d["user_roll_level"], d["user_yaw_level"]], np.float64))
    print("wrote sim_state_sample.npz:", {k: d[k] for k in ("position", "Euler_angles", "move_vector", "heading", "pitch_attitude", "roll_attitude")})


if __name__ == "__main__":
    main()
