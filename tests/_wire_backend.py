"""The wire server's backend interface over the CPU oracle (test infrastructure): same methods as
hirl4ucav_amd.environments.wire.GpuSimBackend, state in the oracle's array-of-structs layout."""
import json
import socket

import numpy as np

from tests import _oracle as ox

ALLY, OPPO = "ally_1", "ennemy_2"


class OracleSimBackend:
    def __init__(self):
        self.L = ox.lib()
        self.env = np.zeros(37, np.float32)
        self.cmd = {ALLY: [0.0, 0.0, 0.0], OPPO: [0.0, 0.0, 0.0]}
        self.fire = False
        self.thrust = {ALLY: 1.0, OPPO: 0.6}
        self.reset_env(0)

    def _flags(self):
        return self.env[35:36].view(np.uint32)

    def reset_env(self, scenario):
        self.L.ox_env_reset(self.env.ctypes.data, int(scenario), 0, 0, 0, 0)

    def reset_machine(self, pid):
        self.cmd[pid] = [0.0, 0.0, 0.0]

    def reset_matrix(self, pid, pos, rot):
        assert not any(abs(float(r)) > 0 for r in rot)
        w0 = 0 if pid == ALLY else 13
        e = self.env
        e[w0:w0 + 3] = [float(v) for v in pos]
        speed = float(np.sqrt(np.float32(e[w0 + 3] * e[w0 + 3] + e[w0 + 4] * e[w0 + 4]) + e[w0 + 5] * e[w0 + 5]))
        e[w0 + 3:w0 + 6] = (0.0, 0.0, speed)
        e[w0 + 6:w0 + 10] = (1.0, 0.0, 0.0, 0.0)
        e[w0 + 10:w0 + 13] = 0.0
        if pid == ALLY:
            e[26:32] = 0.0
            e[33:35] = 0.0
            self._flags()[0] &= np.uint32(~(ox.F_M_ACTIVE | ox.F_M_GUIDED) & 0xFFFFFFFF)

    def set_speed(self, pid, v):
        w0 = 0 if pid == ALLY else 13
        self.env[w0 + 3:w0 + 6] = (0.0, 0.0, float(v))

    def set_thrust(self, pid, v):
        self.thrust[pid] = float(v)
        if pid == OPPO:
            f = self._flags()
            f[0] = (int(f[0]) & ~(3 << ox.F_SCEN_SHIFT)) | ((2 if float(v) > 0.7 else 0) << ox.F_SCEN_SHIFT)

    def set_health(self, pid, v):
        if pid == OPPO:
            self.env[32] = float(v)

    def rearm(self, pid):
        if pid == ALLY:
            self.L.ox_env_rearm(self.env.ctypes.data)

    def set_control(self, pid, axis, level):
        self.cmd[pid][axis] = float(level)

    def fire_missile(self, pid, slot):
        if pid == ALLY and int(slot) == 0:
            self.fire = True

    def tick(self):
        a, o = np.asarray(self.cmd[ALLY], np.float32), np.asarray(self.cmd[OPPO], np.float32)
        self.L.ox_sim_tick(self.env.ctypes.data, a.ctypes.data, o.ctypes.data, int(self.fire))
        self.fire = False

    def readback(self):
        rb = np.zeros(16, np.float32)
        self.L.ox_sim_readback(self.env.ctypes.data, rb.ctypes.data)
        out = [float(v) for v in rb[:14]]
        out += [float(rb[14:15].view(np.int32)[0]), float(rb[15:16].view(np.int32)[0])]
        return out


class RawClient:
    """The client side of the framing (socket_lib.py:86-91,106-139) in a dozen lines, for tests that must run without the reference."""

    def __init__(self, port):
        self.s = socket.create_connection(("127.0.0.1", port))
        self.s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def send(self, command, **args):
        body = json.dumps({"command": command, "args": args}).encode()
        self.s.sendall(len(body).to_bytes(4, "big") + body)

    def ask(self, command, **args):
        self.send(command, **args)
        n = int.from_bytes(self._exact(4), "big")
        return json.loads(self._exact(n).decode())

    def _exact(self, n):
        buf = b""
        while len(buf) < n:
            buf += self.s.recv(n - len(buf))
        return buf

    def close(self):
        self.s.close()
