"""E5 conventions against the ONE plane-state sample of the external simulator the reference holds
(hirl/data/straight_line/ai_env.py:18 -> tests/golden/sim_state_sample.npz, generator: tests/golden/gen_sim_sample.py).

Dynamics parity with Harfang stays unpinned (no source, no trajectories).  What this sample CAN pin — and these tests do, on the CPU
oracle and through the C ABI on the GPU — are the sign and unit conventions of docs/DYNAMICS.md "Conventions":
  * Euler_angles = (pitch, heading, roll) in radians, R = Ry(heading) Rx(pitch) Rz(roll): the model's read-back returns the sample's
    three angles from the quaternion built with that composition;
  * pitch is POSITIVE NOSE DOWN: the sample climbs (vertical_speed +97.6 m/s) with Euler pitch -0.785 rad and pitch_attitude +45 deg; the
    model's nose axis from that attitude lies within 8 deg of the sample's velocity vector — with the opposite sign it would be ~86 deg off;
  * pitch_attitude = -deg(pitch), roll_attitude = +deg(roll), heading = deg(heading) in [0, 360);
  * Y is up: altitude = position[1], vertical_speed = move_vector[1], horizontal_speed = |(vx, vz)|, linear_speed = |v|;
  * one tick = 1/60 s.
"""
import os

import numpy as np
import pytest

from tests import _oracle as ox

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sim_state_sample.npz")


def quat_from_euler(pitch, heading, roll):
    """body -> world quaternion (w, x, y, z) of R = Ry(heading) Rx(pitch) Rz(roll) (docs/DYNAMICS.md)"""
    def mul(a, b):
        aw, ax, ay, az = a
        bw, bx, by, bz = b
        return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                         aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])
    qx = np.array([np.cos(pitch / 2), np.sin(pitch / 2), 0, 0])
    qy = np.array([np.cos(heading / 2), 0, np.sin(heading / 2), 0])
    qz = np.array([np.cos(roll / 2), 0, 0, np.sin(roll / 2)])
    return mul(qy, mul(qx, qz))


def sample_state(g):
    """an env state (oracle layout, 37 words) whose ALLY is the sample's aircraft; the opponent stays at its reset pose"""
    envs, _ = ox.reset_batch(1, 0, 0, seed=0)
    e = envs[0]
    e[0:3] = g["position"]
    e[3:6] = g["move_vector"]
    e[6:10] = quat_from_euler(*g["euler_angles"])
    return envs


def readback_oracle(envs):
    rb = np.zeros(16, np.float32)
    ox.lib().ox_sim_readback(envs[0].ctypes.data, rb.ctypes.data)
    return rb


def readback_gpu(envs):
    import ctypes

    import torch

    from hirl4ucav_amd import _lib

    _lib.register("hx_sim_readback", [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p])
    st = torch.from_numpy(np.ascontiguousarray(envs.T)).cuda()  # SoA: word w of env i at [w * stride + i]
    out = torch.zeros((1, 16), dtype=torch.float32, device="cuda")
    _lib.call("hx_sim_readback", st.data_ptr(), 1, 1, out.data_ptr(), _lib.stream_ptr())
    return out[0].cpu().numpy()


BACKENDS = [pytest.param(readback_oracle, id="oracle"), pytest.param(readback_gpu, id="gpu", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("readback", BACKENDS)
def test_euler_convention_matches_the_reference_sample(readback):
    if readback is readback_gpu:
        torch = pytest.importorskip("torch")
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
    g = np.load(GOLD)
    envs = sample_state(g)
    rb = readback(envs)
    pos, eul = rb[0:3].astype(np.float64), rb[3:6].astype(np.float64)
    # (pitch, heading, roll) come back as the sample gives them: composition order and signs of the read-back
    np.testing.assert_allclose(eul, g["euler_angles"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(pos, g["position"], rtol=1e-7)
    # Harfang's degree fields follow from the model's conventions
    assert abs(-np.degrees(eul[0]) - g["pitch_attitude_deg"]) < 1e-3
    assert abs(np.degrees(eul[2]) - g["roll_attitude_deg"]) < 1e-3
    assert abs(np.degrees(eul[1]) % 360.0 - g["heading_deg"]) < 2e-3
    assert pos[1] == pytest.approx(float(g["altitude"]), rel=1e-7)
    # pitch positive = nose DOWN: the nose axis the model derives from this attitude points along the sample's (climbing) velocity
    p, h = eul[0], eul[1]
    nose = np.array([np.sin(h) * np.cos(p), -np.sin(p), np.cos(h) * np.cos(p)])  # aZ = R (0, 0, 1)
    v = g["move_vector"] / np.linalg.norm(g["move_vector"])
    assert np.degrees(np.arccos(nose @ v)) < 8.0
    flipped = np.array([np.sin(h) * np.cos(-p), -np.sin(-p), np.cos(h) * np.cos(-p)])
    assert np.degrees(np.arccos(flipped @ v)) > 80.0  # the opposite sign convention is ruled out by this one sample


def test_speed_fields_and_tick_of_the_sample():
    """Y up; the speed fields are norms of move_vector; dt = 1/60 s (the model's DT)."""
    g = np.load(GOLD)
    v = g["move_vector"]
    assert np.hypot(v[0], v[2]) == pytest.approx(float(g["horizontal_speed"]), rel=1e-6)
    assert v[1] == pytest.approx(float(g["vertical_speed"]), rel=1e-7)
    assert np.linalg.norm(v) == pytest.approx(float(g["linear_speed"]), rel=1e-6)
    assert float(g["timestep"]) == pytest.approx(1.0 / 60.0, rel=1e-12)
    # one tick of the model moves the aircraft by v * dt (semi-implicit Euler on a state whose acceleration is bounded by ~3 g)
    envs = sample_state(g)
    before = envs[0, 0:3].astype(np.float64).copy()
    zero = np.zeros(3, np.float32)
    ox.lib().ox_sim_tick(envs[0].ctypes.data, zero.ctypes.data, zero.ctypes.data, 0)
    moved = envs[0, 0:3].astype(np.float64) - before
    np.testing.assert_allclose(moved, v / 60.0, atol=0.02)  # |a| dt^2 <= 40 m/s^2 / 3600


def test_wire_plane_state_fields_follow_the_sample():
    """The GET_PLANE_STATE reply of the wire server (what an unmodified reference client parses) derives heading / pitch_attitude /
    roll_attitude / altitude from the read-back with the sample's conventions."""
    from hirl4ucav_amd.environments.wire import ALLY, plane_state

    g = np.load(GOLD)
    rb = readback_oracle(sample_state(g))
    st = plane_state(rb, ALLY, [0.0, 0.0, 0.0], 1.0)
    assert st["pitch_attitude"] == pytest.approx(float(g["pitch_attitude_deg"]), abs=1e-3)
    assert st["roll_attitude"] == pytest.approx(float(g["roll_attitude_deg"]), abs=1e-3)
    assert st["heading"] == pytest.approx(float(g["heading_deg"]), abs=2e-3) and 0.0 <= st["heading"] < 360.0
    assert st["altitude"] == pytest.approx(float(g["altitude"]), rel=1e-7)
    assert st["Euler_angles"] == pytest.approx(list(g["euler_angles"]), abs=2e-6)
    # the sample's aircraft is ~5.8 km from the opponent's reset pose: outside the 100 m .. 3 km lock envelope, as the sample reports
    assert st["target_out_of_range"] is True and st["target_locked"] is False
