"""Shim of the reference's RPC client (hirl/environments/dogfight_client.py).  The simulator now lives in HBM, so the
calls the training drivers make on the connection itself (train_all.py:149-152,187-188) are accepted and ignored; no
socket is opened.  Nothing else of the ~75 wire commands is re-created: the env classes drive the HIP kernel directly."""


def connect(_host, _port):  # dogfight_client.py:5
    return None


def disconnect():  # :9
    return None


def disable_log():  # :15
    return None


def enable_log():  # :19
    return None


def set_renderless_mode(flag: bool):  # :37
    return None


def set_client_update_mode(flag: bool):  # :41 — one tick per UPDATE_SCENE is the only mode the batched integrator has
    return None
