"""A scripted pursuit pilot for the build's own simulator.

The reference's expert data comes from Harfang's built-in IA autopilot (`df.activate_IA`, hirl/data/*/ai_data_col.py:44),
which lives in the external simulator and is not available here; the Drive files with its recordings are not either
(README.md:6,30).  This pilot plays the same role for docs/DYNAMICS.md's model: point the nose at the opponent, launch
when the targeting device reports a lock.  It is tooling beside the hot path (a handful of elementwise torch ops on the
state words), not part of it.
"""
import torch

# state words of include/hirl4ucav.h (struct-of-arrays [HX_ENV_WORDS][N])
ALLY_POS, ALLY_QUAT, OPPO_POS = 0, 6, 13


def pursuit_actions(state, obs, noise_std=0.0, generator=None):
    """state [37, N] fp32, obs [N, 13] -> actions [N, 4]: (pitch, roll, yaw) levels steering the body-frame bearing of the
    opponent to the nose, fire (+1) iff locked and the missile is still on the rail (obs[7] > 0 and obs[8] > 0, the rule of
    ai_data_col.py:62-63), else -1."""
    w, x, y, z = (state[ALLY_QUAT + i] for i in range(4))
    d = state[OPPO_POS:OPPO_POS + 3] - state[ALLY_POS:ALLY_POS + 3]
    # rows of R^T = columns of the body->world rotation: aX (right), aY (up), aZ (nose)
    ax = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)])
    ay = torch.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)])
    az = torch.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])
    b = torch.stack([(ax * d).sum(0), (ay * d).sum(0), (az * d).sum(0)])
    b = b / b.norm(dim=0).clamp_min(1e-6)
    a = torch.stack([-4.0 * b[1], -2.0 * b[0], 4.0 * b[0]], 1)
    if noise_std > 0:
        a = a + noise_std * torch.randn(a.shape, device=a.device, generator=generator)
    fire = torch.where((obs[:, 7] > 0) & (obs[:, 8] > 0), 1.0, -1.0)
    return torch.cat([a.clamp(-1.0, 1.0), fire[:, None]], 1).contiguous()
