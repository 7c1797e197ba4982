"""hirl/utils/plot.py of the reference: trajectory / distance figures of a validation episode (train_all.py:71-72 with --plot).
Plotting is outside the accelerated path; these keep the call signatures so that the drivers run unchanged, and draw with
matplotlib (Agg) when it is installed — otherwise they say so once and return."""
import os

_warned = False


def _plt():
    global _warned
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        return plt
    except Exception:
        if not _warned:
            print("hirl.utils.plot: matplotlib is not available, figures are skipped")
            _warned = True
        return None


def plot_3d_trajectories(self_pos, oppo_pos, fire, lock, dir, file_name):
    """ally / opponent tracks in 3-D (x, z horizontal, y = altitude up), launch and lock steps marked"""
    plt = _plt()
    if plt is None:
        return
    import numpy as np

    a, o = np.asarray(self_pos, float).reshape(-1, 3), np.asarray(oppo_pos, float).reshape(-1, 3)
    fig = plt.figure(figsize=(7, 6))
    ax = fig.add_subplot(projection="3d")
    ax.plot(a[:, 0], a[:, 2], a[:, 1], label="ally")
    ax.plot(o[:, 0], o[:, 2], o[:, 1], label="opponent")
    for steps, marker, name in ((fire, "^", "fire"), (lock, ".", "lock")):
        idx = [int(i) for i in steps if 0 <= int(i) < len(a)] if steps is not None else []
        if idx:
            ax.scatter(a[idx, 0], a[idx, 2], a[idx, 1], marker=marker, label=name)
    ax.set_xlabel("x [m]"); ax.set_ylabel("z [m]"); ax.set_zlabel("altitude [m]"); ax.legend()  # noqa: E702
    os.makedirs(dir, exist_ok=True)
    fig.savefig(os.path.join(dir, file_name), dpi=120)
    plt.close(fig)


def plot_distance(distance, lock, missile, fire, dir, file_name):
    """ally-opponent distance per step with the lock / missile-on-rail / launch steps marked"""
    plt = _plt()
    if plt is None:
        return
    import numpy as np

    d = np.asarray(distance, float).ravel()
    fig, ax = plt.subplots(figsize=(7, 4))
    ax.plot(d, label="distance [m]")
    for steps, style, name in ((lock, "g.", "lock"), (missile, "y.", "missile"), (fire, "r^", "fire")):
        idx = [int(i) for i in steps if 0 <= int(i) < len(d)] if steps is not None else []
        if idx:
            ax.plot(idx, d[idx], style, label=name)
    ax.set_xlabel("step"); ax.legend()  # noqa: E702
    os.makedirs(dir, exist_ok=True)
    fig.savefig(os.path.join(dir, file_name), dpi=120)
    plt.close(fig)


def plot_2d_trajectories(ally_pos, enemy_pos, save_path=None):
    """top view (x, z) of both tracks"""
    plt = _plt()
    if plt is None:
        return
    import numpy as np

    a, o = np.asarray(ally_pos, float).reshape(-1, 3), np.asarray(enemy_pos, float).reshape(-1, 3)
    fig, ax = plt.subplots(figsize=(6, 6))
    ax.plot(a[:, 0], a[:, 2], label="ally")
    ax.plot(o[:, 0], o[:, 2], label="opponent")
    ax.set_xlabel("x [m]"); ax.set_ylabel("z [m]"); ax.legend()  # noqa: E702
    if save_path:
        fig.savefig(save_path, dpi=120)
    plt.close(fig)
