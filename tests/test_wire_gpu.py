"""The product backend of the wire server (one env of the HIP simulator behind hx_sim_tick / hx_sim_readback) against the oracle
backend, both driven through the TCP framing by the same command stream: the reset sequence the reference's wrapper sends, then
400 ticks of a steering + launch script (lock, launch, missile flight, hit)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from hirl4ucav_amd.environments import wire  # noqa: E402
from tests._wire_backend import ALLY, OPPO, OracleSimBackend, RawClient  # noqa: E402


def drive(backend, thrust_opp, speed_opp, steps=1200, replay=None):
    """replay=None: steer from the read-backs and record the command stream; else: send that recorded stream (the Euler read-backs of
    the two backends differ in the last bits — libm vs ocml — so a controller fed by them would not send identical commands)."""
    srv = wire.WireServer(backend).start()
    out, sent = [], []
    try:
        c = RawClient(srv.port)
        for pid in (OPPO, ALLY):
            c.send("RESET_MACHINE", machine_id=pid)
        c.send("SET_HEALTH", machine_id=OPPO, health_level=0.2)
        c.send("RESET_MACHINE_MATRIX", machine_id=OPPO, position=[0, 4200, 0], rotation=[0, 0, 0])
        c.send("RESET_MACHINE_MATRIX", machine_id=ALLY, position=[37, 3460, -4055], rotation=[0, 0, 0])
        c.send("SET_PLANE_THRUST", plane_id=ALLY, thrust_level=1); c.send("SET_PLANE_THRUST", plane_id=OPPO, thrust_level=thrust_opp)
        c.send("SET_PLANE_LINEAR_SPEED", plane_id=ALLY, linear_speed=300); c.send("SET_PLANE_LINEAR_SPEED", plane_id=OPPO, linear_speed=speed_opp)
        c.send("REARM_MACHINE", machine_id=ALLY)
        fired = False
        for t in range(steps):
            a, o = c.ask("GET_PLANE_STATE", plane_id=ALLY), c.ask("GET_PLANE_STATE", plane_id=OPPO)
            h, m = c.ask("GET_HEALTH", machine_id=OPPO)["health_level"], c.ask("GET_MISSILESDEVICE_SLOTS_STATE", machine_id=ALLY)["missiles_slots"][0]
            out.append(a["position"] + a["Euler_angles"] + o["position"] + o["Euler_angles"] + [a["target_angle"], h, float(a["target_locked"]), float(m)])
            # steer the nose onto the target with the read-backs only (pitch > 0 = nose down); launch on lock
            if replay is None:
                d = np.array(o["position"]) - np.array(a["position"])
                want_pitch = -np.arctan2(d[1], np.hypot(d[0], d[2]))
                want_head = np.arctan2(d[0], d[2])
                cmd = (float(np.clip(4 * (want_pitch - a["Euler_angles"][0]), -1, 1)), float(np.clip(4 * (want_head - a["Euler_angles"][1]), -1, 1)),
                       bool(a["target_locked"] and m and not fired))
                sent.append(cmd)
            else:
                cmd = replay[t]
            c.send("SET_PLANE_PITCH", plane_id=ALLY, pitch_level=cmd[0])
            c.send("SET_PLANE_ROLL", plane_id=ALLY, roll_level=0.0)
            c.send("SET_PLANE_YAW", plane_id=ALLY, yaw_level=cmd[1])
            c.send("SET_PLANE_YAW", plane_id=OPPO, yaw_level=0.1 if (t // 250) % 2 else -0.1)
            if cmd[2]:
                c.send("FIRE_MISSILE", machine_id=ALLY, slot_id=0)
                fired = True
            c.send("UPDATE_SCENE")
            if h <= 0 or (replay is not None and t + 1 == len(replay)):
                break
        c.close()
    finally:
        srv.close()
    return np.asarray(out, np.float64), fired, sent


@pytest.mark.parametrize("thrust_opp,speed_opp", [(0.6, 200), (0.8, 290)])
def test_gpu_backend_equals_oracle_backend_over_the_wire(thrust_opp, speed_opp):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ref, fired_ref, script = drive(OracleSimBackend(), thrust_opp, speed_opp)
    got, fired, _ = drive(wire.GpuSimBackend(), thrust_opp, speed_opp, replay=script)
    assert fired == fired_ref and got.shape == ref.shape and len(ref) > 300
    assert fired or thrust_opp > 0.7  # the slow opponent is caught, locked and shot; the fast one need not be
    np.testing.assert_array_equal(got[:, [0, 1, 2, 6, 7, 8, 13, 14, 15]], ref[:, [0, 1, 2, 6, 7, 8, 13, 14, 15]])  # positions, health, lock, slot
    np.testing.assert_allclose(got[:, [3, 4, 5, 9, 10, 11]], ref[:, [3, 4, 5, 9, 10, 11]], rtol=0, atol=2e-6)       # Euler angles (libm vs ocml)
    np.testing.assert_allclose(got[:, 12], ref[:, 12], rtol=0, atol=2e-4)                                            # target angle in degrees
    assert not fired or ref[-1, 13] <= 0 or len(ref) == 1200                                                         # a launch ends in a kill
