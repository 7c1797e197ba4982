"""Wire-protocol server: the GPU simulator behind the reference's own TCP/JSON protocol.

The reference's `HarfangEnv` talks to an external simulator process through `dogfight_client.py` over `socket_lib.py`: every
message is a 4-byte big-endian length followed by a JSON object `{"command": NAME, "args": {...}}` (`socket_lib.py:86-91`), and
the `GET_* / IS_*` commands are answered with one JSON document framed the same way (`socket_lib.py:118-139`).  This module
serves that protocol from a one-env instance of this project's simulator, so that the UNMODIFIED reference client and wrapper
(`hirl/environments/dogfight_client.py`, `HarfangEnv_GYM.py`) can drive it:

    python -m hirl4ucav_amd.environments.wire --port 50888          # then, in the reference: df.connect("127.0.0.1", 50888)

Division of labour exactly as in the reference: the client applies its action with SET_PLANE_PITCH / ROLL / YAW and FIRE_MISSILE,
scripts the opponent the same way, advances time with UPDATE_SCENE, and reads GET_PLANE_STATE / GET_HEALTH /
GET_MISSILESDEVICE_SLOTS_STATE back; observation, latches, reward and termination stay in the client's wrapper.  The server side is
the simulator tick alone (`hx_sim_tick` / `hx_sim_readback`, include/hirl4ucav.h).

The backend is duck-typed (tests run the same server over other backends); `GpuSimBackend` is the product one.
"""
import argparse
import json
import socket
import threading

import logging

log = logging.getLogger("hirl4ucav_amd.wire")
MAX_MESSAGE = 1 << 20  # bytes: the reference's messages are a few hundred bytes; a peer cannot make the server buffer gigabytes

ALLY, OPPO = "ally_1", "ennemy_2"
PLANES = [ALLY, "ally_2", "ennemy_1", OPPO]  # the collectors take planes[0] and planes[3] (hirl/data/*/ai_data_col.py:20-22)

# commands the client waits on (every dogfight_client function that calls socket_lib.get_answer)
ANSWERED = {"GET_HEALTH", "GET_MACHINE_CUSTOM_PHYSICS_MODE", "GET_MACHINE_GUN_STATE", "GET_MACHINE_MISSILES_LIST", "GET_MISSILESDEVICE_SLOTS_STATE",
            "GET_MISSILESLIST", "GET_MISSILE_LAUNCHERS_LIST", "GET_MISSILE_LAUNCHER_STATE", "GET_MISSILE_STATE", "GET_MISSILE_TARGETS_LIST",
            "GET_MOBILE_PARTS_LIST", "GET_PLANESLIST", "GET_PLANE_STATE", "GET_PLANE_THRUST", "GET_RUNNING", "GET_TARGETS_LIST", "GET_TARGET_IDX",
            "GET_TIMESTEP", "IS_AUTOPILOT_ACTIVATED", "IS_IA_ACTIVATED", "IS_USER_CONTROL_ACTIVATED"}


class GpuSimBackend:
    """One env of the HIP simulator: state words in HBM, controls latched on the host until UPDATE_SCENE."""

    def __init__(self, device="cuda"):
        import ctypes

        import torch

        from .. import _lib

        self.torch, self.lib, self.ct = torch, _lib, ctypes
        _lib.load()
        _lib.register("hx_sim_tick", [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
        _lib.register("hx_sim_readback", [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p])
        d = torch.device(device)
        self.state = torch.zeros((_lib.ENV_WORDS, 1), dtype=torch.float32, device=d)
        self.cmd = {ALLY: [0.0, 0.0, 0.0], OPPO: [0.0, 0.0, 0.0]}
        self.fire = False
        self.thrust = {ALLY: 1.0, OPPO: 0.6}
        self._cmd_dev = torch.zeros((2, 3), dtype=torch.float32, device=d)
        self._fire_dev = torch.zeros(1, dtype=torch.uint8, device=d)
        self._rb = torch.zeros((1, 16), dtype=torch.float32, device=d)
        self.reset_env(0)

    # ---- simulator-side state edits the reset sequence of the wrapper performs (HarfangEnv_GYM.py:171-188) ----------------
    def reset_env(self, scenario):
        self.lib.call("hx_env_reset", self.state.data_ptr(), 1, 1, None, None, int(scenario), 0, 0, 0, None, None, self.lib.stream_ptr())

    def reset_machine(self, pid):
        self.cmd[pid] = [0.0, 0.0, 0.0]

    def reset_matrix(self, pid, pos, rot):
        # RESET_MACHINE_MATRIX: position + Euler rotation; the wrapper only ever sends zero rotations (:73-74,379-380,461-462)
        if any(abs(float(r)) > 0 for r in rot):
            raise ValueError("only the level attitudes the reference sends are supported")
        w0 = 0 if pid == ALLY else 13
        s = self.state
        s[w0:w0 + 3, 0] = self.torch.tensor([float(v) for v in pos], device=s.device)
        speed = float(self.torch.linalg.vector_norm(s[w0 + 3:w0 + 6, 0]))
        s[w0 + 3:w0 + 6, 0] = self.torch.tensor([0.0, 0.0, speed], device=s.device)
        s[w0 + 6:w0 + 10, 0] = self.torch.tensor([1.0, 0.0, 0.0, 0.0], device=s.device)
        s[w0 + 10:w0 + 13, 0] = 0.0
        if pid == ALLY:  # a fresh sortie: no missile in flight, lock timer cleared (REARM_MACHINE follows, :186-188)
            s[26:32, 0] = 0.0
            s[33:35, 0] = 0.0
            flags = s[35, 0:1].view(self.torch.int32)
            flags[0] = int(flags[0]) & ~(self.lib.F_M_ACTIVE | self.lib.F_M_GUIDED)

    def set_speed(self, pid, v):
        w0 = 0 if pid == ALLY else 13
        self.state[w0 + 3:w0 + 6, 0] = self.torch.tensor([0.0, 0.0, float(v)], device=self.state.device)

    def set_thrust(self, pid, v):
        self.thrust[pid] = float(v)
        if pid == OPPO:  # the model derives the opponent's thrust from the scenario id: 0.8 = circular (:472), else 0.6
            flags = self.state[35, 0:1].view(self.torch.int32)
            scen = 2 if float(v) > 0.7 else 0
            flags[0] = (int(flags[0]) & ~(3 << self.lib.F_SCEN_SHIFT)) | (scen << self.lib.F_SCEN_SHIFT)

    def set_health(self, pid, v):
        if pid == OPPO:
            self.state[32, 0] = float(v)

    def rearm(self, pid):
        if pid == ALLY:
            self.lib.call("hx_env_rearm", self.state.data_ptr(), 1, 1, None, self.lib.stream_ptr())

    # ---- per-step ---------------------------------------------------------------------------------------------------------
    def set_control(self, pid, axis, level):
        self.cmd[pid][axis] = float(level)

    def fire_missile(self, pid, slot):
        if pid == ALLY and int(slot) == 0:
            self.fire = True

    def tick(self):
        self._cmd_dev.copy_(self.torch.tensor([self.cmd[ALLY], self.cmd[OPPO]], dtype=self.torch.float32))
        self._fire_dev.fill_(1 if self.fire else 0)
        self.lib.call("hx_sim_tick", self.state.data_ptr(), 1, 1, self._cmd_dev[0].data_ptr(), self._cmd_dev[1].data_ptr(),
                      self._fire_dev.data_ptr(), self.lib.stream_ptr())
        self.fire = False

    def readback(self):
        self.lib.call("hx_sim_readback", self.state.data_ptr(), 1, 1, self._rb.data_ptr(), self.lib.stream_ptr())
        return self._rb[0].tolist()


def plane_state(rb, pid, cmd, thrust):
    """GET_PLANE_STATE reply from a 16-float read-back (the fields the reference's wrapper and collectors read; constants for the rest)."""
    ally = pid == ALLY
    pos = rb[0:3] if ally else rb[6:9]
    eul = rb[3:6] if ally else rb[9:12]
    st = {"position": [float(v) for v in pos], "Euler_angles": [float(v) for v in eul], "easy_steering": True,
          "health_level": 1.0 if ally else float(rb[13]), "destroyed": False, "wreck": False, "crashed": False, "active": True,
          "type": "AICRAFT", "nationality": 1 if ally else 2, "thrust_level": thrust, "brake_level": 0, "flaps_level": 0,
          "altitude": float(pos[1]), "heading": (float(eul[1]) * 57.29577951308232) % 360.0, "pitch_attitude": -float(eul[0]) * 57.29577951308232,
          "roll_attitude": float(eul[2]) * 57.29577951308232, "post_combustion": True, "user_pitch_level": cmd[0], "user_roll_level": cmd[1],
          "user_yaw_level": cmd[2], "gear": False, "ia": False, "autopilot": False, "target_id": OPPO if ally else ALLY}
    if ally:
        st["target_locked"] = bool(rb[14] > 0.5)
        # the lock envelope of docs/DYNAMICS.md "Targeting device" (100 m .. 3 km); hirl/data/straight_line/ai_env.py:18 shows the field
        dist = sum((float(rb[c]) - float(rb[6 + c])) ** 2 for c in range(3)) ** 0.5
        st["target_out_of_range"] = not (100.0 < dist < 3000.0)
        st["target_angle"] = float(rb[12])
    return st


class WireServer:
    """Serves ONE client connection at a time on `host:port` (0 = any free port; see `.port`)."""

    def __init__(self, backend, host="127.0.0.1", port=0):
        self.b = backend
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.bind((host, port))
        self.sock.listen(1)
        self.port = self.sock.getsockname()[1]
        self.messages = 0
        self._thread = None
        self._stop = False

    # ---- framing (socket_lib.py:86-91,106-139) ------------------------------------------------------------------------------
    @staticmethod
    def _recv_exact(conn, n):
        buf = b""
        while len(buf) < n:
            chunk = conn.recv(n - len(buf))
            if not chunk:
                return None
            buf += chunk
        return buf

    def _recv_message(self, conn):
        head = self._recv_exact(conn, 4)
        if head is None:
            return None
        n = int.from_bytes(head, "big")
        if n > MAX_MESSAGE:
            raise ValueError(f"message of {n} bytes exceeds the {MAX_MESSAGE}-byte limit")
        body = self._recv_exact(conn, n)
        return None if body is None else json.loads(body.decode())

    @staticmethod
    def _send(conn, obj):
        body = json.dumps(obj).encode()
        conn.sendall(len(body).to_bytes(4, "big") + body)

    # ---- commands -----------------------------------------------------------------------------------------------------------
    def handle(self, command, args):
        """-> reply object for the ANSWERED commands, None otherwise."""
        b = self.b
        pid = args.get("plane_id", args.get("machine_id"))
        if pid is not None and pid not in PLANES and not str(pid).startswith((ALLY, OPPO)):
            raise ValueError(f"unknown machine id {pid!r}")
        if command in ("SET_PLANE_PITCH", "SET_PLANE_ROLL", "SET_PLANE_YAW", "FIRE_MISSILE", "RESET_MACHINE", "RESET_MACHINE_MATRIX",
                       "SET_PLANE_THRUST", "SET_PLANE_LINEAR_SPEED", "SET_HEALTH", "REARM_MACHINE") and pid not in (ALLY, OPPO):
            return None  # the other two aircraft of the sandbox's mission exist in name only
        if command == "UPDATE_SCENE":
            b.tick()
        elif command == "SET_PLANE_PITCH":
            b.set_control(pid, 0, args["pitch_level"])
        elif command == "SET_PLANE_ROLL":
            b.set_control(pid, 1, args["roll_level"])
        elif command == "SET_PLANE_YAW":
            b.set_control(pid, 2, args["yaw_level"])
        elif command == "FIRE_MISSILE":
            b.fire_missile(pid, args["slot_id"])
        elif command == "RESET_MACHINE":
            b.reset_machine(pid)
        elif command == "RESET_MACHINE_MATRIX":
            b.reset_matrix(pid, args["position"], args["rotation"])
        elif command == "SET_PLANE_THRUST":
            b.set_thrust(pid, args["thrust_level"])
        elif command == "SET_PLANE_LINEAR_SPEED":
            b.set_speed(pid, args["linear_speed"])
        elif command == "SET_HEALTH":
            b.set_health(pid, args["health_level"])
        elif command == "REARM_MACHINE":
            b.rearm(pid)
        elif command == "GET_PLANE_STATE":
            return plane_state(b.readback(), pid, list(b.cmd[pid]) if pid in b.cmd else [0.0, 0.0, 0.0], b.thrust.get(pid, 0.0))
        elif command == "GET_HEALTH":
            return {"health_level": float(b.readback()[13]) if pid == OPPO else 1.0}
        elif command == "GET_MISSILESDEVICE_SLOTS_STATE":
            return {"missiles_slots": [bool(b.readback()[15] > 0.5)] if pid == ALLY else [True]}
        elif command == "GET_PLANESLIST":
            return list(PLANES)
        elif command == "GET_MACHINE_MISSILES_LIST":
            return [f"{pid}Meteor0"]
        elif command == "GET_MISSILESLIST":
            return [f"{ALLY}Meteor0"]
        elif command == "GET_TARGETS_LIST":
            return [OPPO] if pid == ALLY else [ALLY]
        elif command == "GET_TARGET_IDX":
            return {"target_idx": 0}
        elif command == "GET_PLANE_THRUST":
            return {"thrust_level": b.thrust.get(pid, 0.0)}
        elif command == "GET_RUNNING":
            return {"running": True}
        elif command == "GET_TIMESTEP":
            return {"timestep": 1.0 / 60.0}
        elif command in ANSWERED:
            return {}
        # everything else the client may send (logging, render modes, gear, target id, autopilot switches ...) changes nothing here
        return None

    def serve_connection(self, conn):
        with conn:
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            while not self._stop:
                try:  # a malformed message ends THIS connection, not the server (JSON errors, missing keys, unknown ids, oversize frames)
                    msg = self._recv_message(conn)
                    if msg is None:
                        return
                    self.messages += 1
                    command = msg["command"]
                    reply = self.handle(command, msg.get("args") or {})
                except (ValueError, KeyError, TypeError, AttributeError, UnicodeDecodeError) as err:
                    log.warning("closing connection after a bad message: %s", err)
                    return
                if command in ANSWERED:
                    self._send(conn, reply)

    def serve_forever(self):
        while not self._stop:
            try:
                conn, _ = self.sock.accept()
            except OSError:
                return
            self.serve_connection(conn)

    def start(self):
        self._thread = threading.Thread(target=self.serve_forever, daemon=True)
        self._thread.start()
        return self

    def close(self):
        self._stop = True
        try:
            self.sock.close()
        except OSError:
            pass


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--host", default="127.0.0.1")
    p.add_argument("--port", type=int, default=50888)  # the reference's local_config.yaml names the port; 50888 is Harfang's default
    c = p.parse_args(argv)
    srv = WireServer(GpuSimBackend(), c.host, c.port)
    print(f"serving the dogfight wire protocol on {c.host}:{srv.port}", flush=True)
    srv.serve_forever()


if __name__ == "__main__":
    main()
