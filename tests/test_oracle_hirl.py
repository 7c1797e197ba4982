"""CPU tests that PIN the update oracle (oracle/hirl_oracle.py) against golden vectors recorded from the reference's
own hirl.agents.HIRL.Agent.learn / agents.TD3.Agent.learn / chooseAction* (tests/golden/gen_hirl_golden.py).

Tolerances (fp32, different summation order than the reference's modules): returned losses rtol 1e-5; parameter
probes after k Adam steps atol 2e-6 + rtol 1e-5 (Adam divides by sqrt(v): an entry whose gradient is ~1e-8 can move by
a visible fraction of lr = 1e-3 for a last-bit change of the gradient; the probes bound that)."""
import os

import numpy as np
import pytest
import torch

from oracle import hirl_oracle as H
from tests import _hirl_data as D

torch.set_num_threads(1)
MODES = ["soft_e0", "soft_e64", "fixed_e32", "linear_e0"]


def batches_for_call(g, data, k):
    rows = data["replay"][g["idx_buf"][k]]
    if g["idx_exp"].shape[1]:
        rows = np.concatenate([rows, data["expert_rows"][g["idx_exp"][k]]], 0)  # buffer rows first, HIRL.py:229
    batch = (rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31])
    bc = (data["expert_s"][g["idx_bc"][k]], data["expert_a"][g["idx_bc"][k]])
    return batch, bc


def check_probes(g, k, nets, atol=2e-6):
    for j, net in enumerate(nets):
        flat = H.flatten(net[0], net[1])
        s, a, v = D.net_probe(flat)
        np.testing.assert_allclose(v, g["probe_val"][k][j], rtol=1e-5, atol=atol, err_msg=f"call {k} net {j}")
        np.testing.assert_allclose(a, g["probe_abs"][k][j], rtol=1e-6, err_msg=f"call {k} net {j} abs-sum")


@pytest.mark.parametrize("mode", MODES)
def test_hirl_learn_matches_reference(mode, golden_dir):
    g = np.load(os.path.join(golden_dir, f"hirl_learn_{mode}.npz"))
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    assert D.checksum(params) == str(g["param_checksum"]) and D.checksum(data) == str(g["data_checksum"])
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    for k in range(g["out"].shape[0]):
        batch, bc = batches_for_call(g, data, k)
        w_in = g["bc_w_in"][k]
        ret = o.learn(batch, bc, g["noise"][k], 100 if w_in == 100 else float(w_in), float(g["warm_in"][k]))
        np.testing.assert_allclose(ret, g["out"][k], rtol=1e-5, atol=1e-6, err_msg=f"{mode} call {k}")
        check_probes(g, k, [(o.actor, H.ACTOR_KEYS), (o.critic, H.CRITIC_KEYS), (o.target_actor, H.ACTOR_KEYS),
                            (o.target_critic, H.CRITIC_KEYS)])
    # quirks: actor every 2nd call, targets every 3rd actor update (6th call)
    assert o.opt_critic.t == 10 and o.opt_actor.t == 5 and o.update_count == 5
    assert g["out"][5][5] == g["out"][4][5]  # stale weight on the non-actor first call of "episode 2"
    if mode.startswith("soft"):
        assert g["out"][8][5] != g["out"][7][5]  # "episode 3" starts on an actor call: fresh estimate


def test_td3_learn_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "td3_learn.npz"))
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    o = H.HirlOracle(params["actor"], params["critic"], None, slope=0.01, use_bc=False)
    for k in range(g["out"].shape[0]):
        rows = data["replay"][g["idx_buf"][k]]
        batch = (rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31])
        ret = o.learn(batch, None, g["noise"][k])
        np.testing.assert_allclose(ret[:2], g["out"][k], rtol=1e-5, atol=1e-6, err_msg=f"td3 call {k}")
        check_probes(g, k, [(o.actor, H.ACTOR_KEYS), (o.critic, H.CRITIC_KEYS), (o.target_actor, H.ACTOR_KEYS),
                            (o.target_critic, H.CRITIC_KEYS)])


def test_choose_action_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "hirl_choose_action.npz"))
    params = D.make_params(D.PARAM_SEED)
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    for i, s in enumerate(g["states"]):
        np.testing.assert_allclose(o.choose_action(s, g["noise"][i]), g["action"][i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o.choose_action(s, g["small_noise"][i]), g["action_small"][i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o.choose_action(s), g["action_clean"][i], rtol=1e-5, atol=1e-6)


def test_initialisation_bounds():
    """G8: the reference's init is U(+-sqrt(6/fan_in)) for hidden weights (kaiming_uniform_ a=0.01 'relu', HIRL.py:26)."""
    p = D.make_params(5)
    for net, fan in ((p["actor"], 13), (p["critic"], 17)):
        w = net["full1.weight"]
        assert np.abs(w).max() <= np.sqrt(6 / fan) and np.abs(w).max() > 0.95 * np.sqrt(6 / fan)
    n_actor = sum(v.size for v in p["actor"].values())
    n_critic = sum(v.size for v in p["critic"].values())
    assert (n_actor, n_critic) == (138756, 276482)


def test_bc_train_actor_matches_reference(golden_dir):
    """BC.Agent.train_actor (BC.py:160-185), 6 consecutive calls with the LeakyReLU actor."""
    g = np.load(os.path.join(golden_dir, "bc_train.npz"))
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    o = H.HirlOracle(params["actor"], params["critic"], None, slope=0.01)
    for k in range(g["out"].shape[0]):
        idx = g["idx_bc"][k]
        loss = H.bc_train_actor(o, (data["expert_s"][idx], data["expert_a"][idx]))
        np.testing.assert_allclose(loss, g["out"][k], rtol=1e-5, atol=1e-6, err_msg=f"bc call {k}")
        s, a, v = D.net_probe(H.flatten(o.actor, H.ACTOR_KEYS))
        np.testing.assert_allclose(v, g["probe_val"][k], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(a, g["probe_abs"][k], rtol=1e-6)


def test_hirl_learn_without_layernorm_matches_reference(golden_dir):
    """Agent(..., layerNorm=False, ...): the `else` branches of HIRL.py:58-80,82-97,126-140 (Linear -> ReLU -> Linear -> ReLU -> final),
    10 consecutive learn() calls and chooseActionNoNoise afterwards, recorded from the reference; the LayerNorm modules stay at (1, 0)."""
    g = np.load(os.path.join(golden_dir, "hirl_learn_soft_noln.npz"))
    params, data = D.plain_layernorm(D.make_params(D.PARAM_SEED)), D.make_data(D.DATA_SEED)
    assert D.checksum(params) == str(g["param_checksum"]) and D.checksum(data) == str(g["data_checksum"])
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"], layer_norm=False)
    for k in range(g["out"].shape[0]):
        batch, bc = batches_for_call(g, data, k)
        w_in = g["bc_w_in"][k]
        ret = o.learn(batch, bc, g["noise"][k], 100 if w_in == 100 else float(w_in), float(g["warm_in"][k]))
        np.testing.assert_allclose(ret, g["out"][k], rtol=1e-5, atol=1e-6, err_msg=f"noln call {k}")
        check_probes(g, k, [(o.actor, H.ACTOR_KEYS), (o.critic, H.CRITIC_KEYS), (o.target_actor, H.ACTOR_KEYS),
                            (o.target_critic, H.CRITIC_KEYS)])
    for key in ("layernorm1.weight", "layernorm2.bias"):  # never used, never trained
        assert torch.equal(o.actor[key].detach(), torch.as_tensor(params["actor"][key]))
    for i, s in enumerate(g["states"]):
        np.testing.assert_allclose(o.choose_action(s), g["action_clean_after"][i], rtol=1e-5, atol=2e-6)
    # the LayerNorm really is off: the same call with it on gives another loss
    o2 = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    batch, bc = batches_for_call(g, data, 0)
    assert abs(o2.learn(batch, bc, g["noise"][0], 100, float(g["warm_in"][0]))[0] - g["out"][0][0]) > 1e-2


def test_rounded_operand_switch_is_scoped_and_close_to_fp32():
    """Bf16Layer2 (the bf16 update path's checker) changes the 256 <-> 512 products only inside the `with` block, and only by bf16 rounding."""
    params, data = D.make_params(3), D.make_data(4)
    rows = data["replay"][:128]
    batch = (rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31])
    bc = (data["expert_s"][:128], data["expert_a"][:128])
    noise = np.zeros(4, np.float32)
    a, b, c = (H.HirlOracle(params["actor"], params["critic"], params["bc_actor"]) for _ in range(3))
    la = a.learn(batch, bc, noise, 0.5)
    with H.Bf16Layer2():
        lb = b.learn(batch, bc, noise, 0.5)
    lc = c.learn(batch, bc, noise, 0.5)
    assert la == lc and la != lb
    np.testing.assert_allclose(lb[:4], la[:4], rtol=3e-2)
    g32, g16 = a.last_grads["critic"]["full2.weight"], b.last_grads["critic"]["full2.weight"]
    assert float((g32 - g16).abs().max()) < 0.15 * float(g32.abs().max())  # (bf16 rounding moves a few units across their ReLU kink)
