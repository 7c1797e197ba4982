#!/usr/bin/env python3
"""Generate golden vectors for the HIRL / TD3 update by running the REFERENCE agents themselves.

Development container only (imports /root/reference/hirl/agents/{HIRL,TD3}.py with a `torchvision` stub).

    python tests/golden/gen_hirl_golden.py

Inputs are reproducible from seeds (tests/_hirl_data.py builds the same arrays on the GPU box), so each fixture
holds only: the minibatch indices the reference drew (random.sample / np.random.choice are wrapped and recorded),
the (4,) target-smoothing noise it drew (torch.normal wrapped and recorded), the arguments passed to learn(), the
returned tuples for K consecutive calls, and per-call probes of every network (sum, sum of |x|, 128 fixed entries).

  hirl_learn_<mode>.npz   G4  hirl.agents.HIRL.Agent.learn, 10 calls = three "episodes" starting at calls 0, 5, 8 (the
                              second starts on a non-actor call -> the stale soft weight of SURVEY.md quirk 2,
                              the third on an actor call -> a fresh estimate)
                              modes: soft_e0, soft_e64, fixed_e32, linear_e0
  td3_learn.npz           G5  agents.TD3.Agent.learn (leaky_relu nets), 8 calls
  bc_train.npz                agents.BC.Agent.train_actor (leaky_relu actor), 6 calls
  hirl_choose_action.npz  G7  chooseAction / chooseActionSmallNoise / chooseActionNoNoise on fixed states
  hirl_learn_soft_noln.npz    the same Agent built with layerNorm=False (the `else` branches of HIRL.py:58-80,82-97,126-140), soft schedule,
                              10 calls + chooseActionNoNoise on 32 states; its LayerNorm modules are loaded at (1, 0) (they exist in the
                              state_dict, HIRL.py:28,33,114,119, and are never used or trained)   [python gen_hirl_golden.py noln]
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "hirl"))  # agents/TD3.py imports script-relative `utils.buffer`

tv = types.ModuleType("torchvision")
tv.transforms = types.ModuleType("torchvision.transforms")
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tv.transforms

from tests import _hirl_data as D  # noqa: E402

import hirl.agents.HIRL as ref_hirl  # noqa: E402
import hirl.utils.buffer as ref_buf  # noqa: E402

torch.set_num_threads(1)  # deterministic summation order in the reference's CPU matmuls


EPISODE_STARTS = (0, 5, 8)  # call 5 is a non-actor call (stale weight, quirk 2), call 8 an actor call (re-estimate)


class Recorder:
    """Wraps the three sampling primitives the reference's learn() uses and records what they returned."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.samples, self.choices, self.normals = [], [], []

    def sample(self, population, k):  # stands in for random.sample (buffer.py:45)
        idx = self.rng.choice(len(population), k, replace=False)
        self.samples.append(idx.astype(np.int32))
        return [population[i] for i in idx]

    def choice(self, n, size=None, replace=True, p=None):  # np.random.choice (HIRL.py:249)
        idx = self.rng.choice(n, size, replace=replace)
        self.choices.append(np.asarray(idx, np.int32))
        return idx

    def normal(self, mean, std):  # torch.normal (HIRL.py:196,265)
        v = torch.tensor(self.rng.normal(0, 1, tuple(mean.shape)).astype(np.float32)) * std + mean
        self.normals.append(v.numpy().copy())
        return v


def install(rec, *buffer_modules):
    fake_random = types.SimpleNamespace(sample=rec.sample)
    for m in buffer_modules:
        m.random = fake_random
    np.random.choice = rec.choice
    torch.normal = rec.normal


def load_nets(agent, params, with_bc):
    sd = lambda p: {k: torch.tensor(v) for k, v in p.items()}  # noqa: E731
    agent.actor.load_state_dict(sd(params["actor"]))
    agent.targetActor.load_state_dict(sd(params["actor"]))
    agent.critic.load_state_dict(sd(params["critic"]))
    agent.targetCritic.load_state_dict(sd(params["critic"]))
    if with_bc:
        agent.bc_actor.load_state_dict(sd(params["bc_actor"]))


def probes(agent):
    out = []
    for net in (agent.actor, agent.critic, agent.targetActor, agent.targetCritic):
        flat = np.concatenate([v.detach().numpy().ravel() for v in net.state_dict().values()]).astype(np.float64)
        out.append((flat.sum(), np.abs(flat).sum(), flat[D.probe_index(flat.size)]))
    return out


def fill_buffers(agent, data, with_expert):
    for row in data["replay"]:
        agent.buffer.store(row[0:13], row[13:17], row[17:30], row[30], row[31], 0)
    if with_expert:
        for row in data["expert_rows"]:
            agent.expert_buffer.store(row[0:13], row[13:17], row[17:30], row[30], row[31], 0)


def run_hirl(mode, expert_num, schedule, layer_norm=True):
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    if not layer_norm:
        params = D.plain_layernorm(params)
    agent = ref_hirl.Agent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 100000, 128, layer_norm, "g", data["expert_s"], data["expert_a"], 0.5, True)
    load_nets(agent, params, True)
    fill_buffers(agent, data, True)
    rec = Recorder(1234)
    install(rec, ref_buf)
    K = 10
    outs, prb, w_in, warm_in = [], [], [], []
    bc_weight_now = None
    for k in range(K):
        if k in EPISODE_STARTS:  # episode start, train_all.py:328-339
            bc_weight_now, warm = schedule(EPISODE_STARTS.index(k))
        w_in.append(bc_weight_now)
        warm_in.append(warm)
        ret = agent.learn(bc_weight_now, expert_num, warm)
        outs.append([float(x) for x in ret])
        bc_weight_now = ret[5]  # train_all.py:361
        prb.append(probes(agent))
    nb = 128 - expert_num
    idx_buf = np.asarray([s for s in rec.samples if len(s) == nb][:K])
    idx_exp = np.asarray([s for s in rec.samples if len(s) == expert_num and expert_num != nb][:K]) if expert_num else np.zeros((K, 0), np.int32)
    if expert_num == nb:  # 64/64: samples alternate buffer, expert
        idx_buf, idx_exp = np.asarray(rec.samples[0::2]), np.asarray(rec.samples[1::2])
    np.savez_compressed(
        os.path.join(HERE, f"hirl_learn_{mode}.npz"), expert_num=np.int32(expert_num),
        idx_buf=idx_buf, idx_exp=idx_exp, idx_bc=np.asarray(rec.choices), noise=np.asarray(rec.normals, np.float32),
        bc_w_in=np.asarray(w_in, np.float64), warm_in=np.asarray(warm_in, np.float64), out=np.asarray(outs, np.float64),
        probe_sum=np.asarray([[p[0] for p in row] for row in prb]), probe_abs=np.asarray([[p[1] for p in row] for row in prb]),
        probe_val=np.asarray([[p[2] for p in row] for row in prb], np.float32),
        data_checksum=D.checksum(data), param_checksum=D.checksum(params),
        **({} if layer_norm else {"states": data["replay"][:32, 0:13].astype(np.float32),
                                  "action_clean_after": np.asarray([agent.chooseActionNoNoise(st) for st in data["replay"][:32, 0:13].astype(np.float64)])}))
    print(mode, "bc_weight out:", [round(o[5], 4) for o in outs], "critic loss", [round(o[0], 3) for o in outs[:4]])


def run_td3():
    import agents.TD3 as ref_td3
    import utils.buffer as td3_buf

    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    agent = ref_td3.Agent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 100000, 128, True, "g")
    load_nets(agent, params, False)
    fill_buffers(agent, data, False)
    rec = Recorder(4321)
    install(rec, td3_buf)
    K = 8
    outs, prb = [], []
    for k in range(K):
        ret = agent.learn()
        outs.append([float(x) for x in ret])
        prb.append(probes(agent))
    np.savez_compressed(
        os.path.join(HERE, "td3_learn.npz"), idx_buf=np.asarray(rec.samples), noise=np.asarray(rec.normals, np.float32),
        out=np.asarray(outs, np.float64), probe_sum=np.asarray([[p[0] for p in row] for row in prb]),
        probe_abs=np.asarray([[p[1] for p in row] for row in prb]), probe_val=np.asarray([[p[2] for p in row] for row in prb], np.float32),
        data_checksum=D.checksum(data), param_checksum=D.checksum(params))
    print("td3", [round(o[0], 3) for o in outs[:4]], [round(o[1], 3) for o in outs[:4]])


def run_bc():
    import agents.BC as ref_bc

    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    agent = ref_bc.Agent(1e-3, 13, 4, 256, 512, True, "g", 128, data["expert_s"], data["expert_a"])
    agent.actor.load_state_dict({k: torch.tensor(v) for k, v in params["actor"].items()})
    rec = Recorder(777)
    install(rec, ref_buf)
    outs, prb = [], []
    for k in range(6):
        outs.append(float(agent.train_actor()))
        flat = np.concatenate([v.detach().numpy().ravel() for v in agent.actor.state_dict().values()]).astype(np.float64)
        prb.append((np.abs(flat).sum(), flat[D.probe_index(flat.size)]))
    np.savez_compressed(os.path.join(HERE, "bc_train.npz"), idx_bc=np.asarray(rec.choices), out=np.asarray(outs, np.float64),
                        probe_abs=np.asarray([p[0] for p in prb]), probe_val=np.asarray([p[1] for p in prb], np.float32),
                        data_checksum=D.checksum(data), param_checksum=D.checksum(params))
    print("bc", [round(o, 5) for o in outs])


def run_choose_action():
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    agent = ref_hirl.Agent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 1000, 128, True, "g", data["expert_s"], data["expert_a"], 0.5, True)
    load_nets(agent, params, True)
    rec = Recorder(99)
    install(rec, ref_buf)
    states = data["replay"][:32, 0:13].astype(np.float64)
    a_noise = np.asarray([agent.chooseAction(s) for s in states])
    n1 = np.asarray(rec.normals)
    rec.normals.clear()
    a_small = np.asarray([agent.chooseActionSmallNoise(s) for s in states])
    n2 = np.asarray(rec.normals)
    a_clean = np.asarray([agent.chooseActionNoNoise(s) for s in states])
    np.savez_compressed(os.path.join(HERE, "hirl_choose_action.npz"), states=states.astype(np.float32), noise=n1, small_noise=n2,
                        action=a_noise, action_small=a_small, action_clean=a_clean, param_checksum=D.checksum(params))
    print("chooseAction:", a_clean[0])


if __name__ == "__main__":
    _choice, _normal = np.random.choice, torch.normal
    if sys.argv[1:] == ["noln"]:
        run_hirl("soft_noln", 0, lambda ep: (100, 0.2 - 0.1 * ep), layer_norm=False)
        np.random.choice, torch.normal = _choice, _normal
        sys.exit(0)
    run_hirl("soft_e0", 0, lambda ep: (100, 0.2 - 0.1 * ep))
    run_hirl("soft_e64", 64, lambda ep: (100, 0.0))
    run_hirl("fixed_e32", 32, lambda ep: (0.5, 0.0))
    run_hirl("linear_e0", 0, lambda ep: (max(0.5 - ep / 4.0, 0.0), 0.0))
    run_td3()
    run_bc()
    run_choose_action()
    run_hirl("soft_noln", 0, lambda ep: (100, 0.2 - 0.1 * ep), layer_norm=False)
    np.random.choice, torch.normal = _choice, _normal
