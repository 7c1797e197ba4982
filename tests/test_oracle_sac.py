"""CPU test that PINS the SAC oracle's loss math / update order (oracle/sac_oracle.py) against golden vectors recorded from the
reference's SacAgent.learn (tests/golden/gen_sac_golden.py).  rltorch's builder and memories are un-vendored: initialisation and
sampling order stay UNPINNED (the vectors inject weights, minibatches and noise)."""
import os

import numpy as np
import torch

from oracle import sac_oracle as S
from tests import _hirl_data as D

torch.set_num_threads(1)


def sac_params():
    rng = np.random.default_rng(31)
    return {"policy": S.init_mlp(rng, 13, 8), "q1": S.init_mlp(rng, 17, 1), "q2": S.init_mlp(rng, 17, 1)}


def test_sac_learn_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "sac_learn.npz"))
    params, data = sac_params(), D.make_data(D.DATA_SEED)
    assert D.checksum(params) == str(g["param_checksum"]) and D.checksum(data) == str(g["data_checksum"])
    assert sum(v.size for v in params["policy"].values()) == 139272 and 2 * sum(v.size for v in params["q1"].values()) == 273410
    o = S.SacOracle(params["policy"], params["q1"], params["q2"])
    for k in range(g["out"].shape[0]):
        rows = data["replay"][g["idx"][k]]
        out = o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 30], rows[:, 17:30], rows[:, 31]), g["eps"][k, 0], g["eps"][k, 1])
        np.testing.assert_allclose(out, g["out"][k], rtol=1e-5, atol=1e-6, err_msg=f"sac call {k}")
        for j, net in enumerate((o.policy, o.q1, o.q2, o.q1_t, o.q2_t)):
            s, a, v = D.net_probe(S.flatten(net))
            np.testing.assert_allclose(v, g["probe_val"][k][j], rtol=1e-5, atol=2e-6, err_msg=f"call {k} net {j}")
            np.testing.assert_allclose(a, g["probe_abs"][k][j], rtol=1e-6)
    assert o.learning_steps == 8  # targets moved at calls 3 and 6 (before the update)


def test_sac_sample_entropy_formula():
    """entropy = -sum(log N(x; mean, std) - log(1 - tanh(x)^2 + 1e-6))   SAC/model.py:69-82, against torch.distributions."""
    rng = np.random.default_rng(0)
    p = S.to_t(sac_params()["policy"])
    s = torch.tensor(rng.uniform(-1, 1, (64, 13)).astype(np.float32))
    eps = torch.tensor(rng.normal(0, 1, (64, 4)).astype(np.float32))
    a, h, m = S.sample(p, s, eps)
    mean, log_std = S.policy_forward(p, s)
    n = torch.distributions.Normal(mean, log_std.exp())
    x = mean + log_std.exp() * eps
    ref = -(n.log_prob(x) - torch.log(1 - torch.tanh(x).pow(2) + 1e-6)).sum(1, keepdim=True)
    np.testing.assert_allclose(h.detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    assert torch.all(a.abs() <= 1) and torch.equal(m, torch.tanh(mean))
