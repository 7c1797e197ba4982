"""GPU tests of the reference-API façades (env classes, Agent classes) and of the vectorised driver: they read like the
reference's own call sites (train_all.py, validate_all.py)."""
import os
import random

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import hirl_oracle as H  # noqa: E402
from tests import _hirl_data as D  # noqa: E402

RTOL, ATOL = 1e-5, 1e-6


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("cls,tag", [("HarfangEnv", "straight_line"), ("HarfangSerpentineEnv", "serpentine"), ("HarfangCircularEnv", "circular")])
def test_env_classes_replay_reference_traces(cls, tag, golden_dir):
    """env = HarfangEnv(); state = env.reset(); n_state, reward, done, info, step_success = env.step(action) — against the
    traces the reference classes produced over the same simulator."""
    import hirl4ucav_amd.environments.HarfangEnv_GYM as G

    g = np.load(os.path.join(golden_dir, f"env_closedloop_{tag}.npz"))
    env = getattr(G, cls)()
    s = env.reset()
    assert isinstance(s, np.ndarray) and s.shape == (13,) and s.dtype == np.float64
    np.testing.assert_allclose(s, g["obs0"], rtol=RTOL, atol=ATOL)
    assert env.action_space.sample().shape == (4,)
    for t in range(0, g["actions"].shape[0]):
        if tag == "circular":  # validate()'s call form, train_all.py:41
            n_state, reward, done, info, iffire, before, after, locked, step_success = env.step_test(g["actions"][t])
            assert isinstance(iffire, bool) and isinstance(before, bool) and isinstance(locked, bool)
        else:
            n_state, reward, done, info, step_success = env.step(g["actions"][t])
        np.testing.assert_allclose(n_state, g["obs"][t], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(reward, g["reward"][t], rtol=RTOL, atol=ATOL)
        assert isinstance(done, bool) and info == {} and isinstance(step_success, int)
        assert (done, step_success) == (bool(g["done"][t]), int(g["success"][t]))
        assert env.episode_success == bool(g["episode_success"][t]) and env.fire_success == bool(g["fire_success"][t])
    assert env.get_pos().shape == (3,) and env.get_oppo_pos().shape == (3,) and env.loc_diff > 0
    np.testing.assert_allclose(env.get_pos() - env.get_oppo_pos(), n_state[0:3] * 10000, rtol=1e-4, atol=0.05)
    # expert-labelling helpers keep the reference's signatures  (train_all.py:295-297)
    r, sc = env.get_reward(g["obs"][0], g["actions"][1], g["obs"][1])
    assert isinstance(sc, int) and env.get_termination(g["obs"][1]) in (True, False) and r < 0


def test_random_reset_and_infinite_rearm():
    import hirl4ucav_amd.environments.HarfangEnv_GYM as G

    random.seed(0)
    env = G.HarfangSerpentineInfiniteEnv()
    a = env.random_reset()
    b = env.random_reset()
    off = env.get_pos() - np.array([0, 3500, -4000.0])
    assert not np.array_equal(a, b) and np.all(np.abs(off) <= 100) and np.all(off == np.round(off))
    fires = 0
    for t in range(1, 200):
        # the rail is re-armed BEFORE steps 60, 120, 180 (HarfangEnv_GYM.py:484-486); the wrapper sees the new missile in
        # the observation of that step, so a launch on the FOLLOWING step is the first one that counts
        out = env.step_test(np.array([0, 0, 0, 1.0 if t % 60 == 1 else -1.0]))
        fires += int(out[8] != 0)
    assert fires == 4 and env.infinite_total_fire == 4 and env.infinite_total_success == 0 and env.infinite_total_step == 199
    # firing on the re-arm step itself spends the missile uncounted (missile1_state is still False, :250-251,121-129)
    env2 = G.HarfangSerpentineInfiniteEnv()
    env2.reset()
    n2 = sum(int(env2.step_test(np.array([0, 0, 0, 1.0]))[8] != 0) for _ in range(199))
    assert n2 == 1


def test_agent_facade_reads_like_the_reference():
    """agent = HIRLAgent(actorLR, criticLR, stateDim, actionDim, h1, h2, tau, gamma, bufferSize, batchSize, useLayerNorm, name,
    expert_states, expert_actions, bc_weight, expert_warm_up)   — train_all.py:226,343-361."""
    from hirl4ucav_amd.agents.HIRL import Agent as HIRLAgent
    from hirl4ucav_amd.utils.seed import set_seed

    params, data = D.make_params(5), D.make_data(6)
    set_seed(7)
    agent = HIRLAgent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 10 ** 5, 128, True, "Harfang_GYM", data["expert_s"], data["expert_a"], 0.5, True)
    for net, p in ((agent.actor, params["actor"]), (agent.targetActor, params["actor"]), (agent.critic, params["critic"]),
                   (agent.targetCritic, params["critic"]), (agent.bc_actor, params["bc_actor"])):
        net.load_state_dict({k: torch.tensor(v) for k, v in p.items()})
    for row in data["replay"][:600]:
        agent.store(row[0:13], row[13:17], row[17:30], row[30], row[31], 0)
    for row in data["expert_rows"]:
        agent.expert_buffer.store(row[0:13], row[13:17], row[17:30], row[30], row[31], 0)
    assert agent.buffer.fullEnough(agent.batchSize) and len(agent.expert_buffer.memory) == D.N_EXPERT_ROWS
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    # acting
    s = data["replay"][0, 0:13].astype(np.float64)
    a = agent.chooseActionNoNoise(s)
    assert isinstance(a, np.ndarray) and a.shape == (4,)
    np.testing.assert_allclose(a, o.choose_action(s.astype(np.float32)), rtol=1e-5, atol=1e-6)
    a2 = agent.chooseAction(s)
    assert np.all(np.abs(a2) <= 1) and not np.allclose(a, a2)
    # learning: replay the host generators to know what the façade will draw, then compare with the oracle
    bc_weight_now, expert_num = 100, 32
    for k in range(4):
        st_r, st_np, st_t = random.getstate(), np.random.get_state(), torch.get_rng_state()
        idx = random.sample(range(len(agent.buffer)), 128 - expert_num) + random.sample(range(len(agent.expert_buffer)), expert_num)
        ibc = np.random.choice(D.N_EXPERT, 128, replace=False)
        noise = torch.normal(mean=torch.zeros(4), std=torch.ones(4) * 0.2).numpy()
        random.setstate(st_r); np.random.set_state(st_np); torch.set_rng_state(st_t)  # noqa: E702
        ret = agent.learn(bc_weight_now, expert_num, 0.0)
        assert len(ret) == 6
        rows = np.concatenate([data["replay"][idx[:128 - expert_num]], data["expert_rows"][idx[128 - expert_num:]]], 0)
        ref = o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (data["expert_s"][ibc], data["expert_a"][ibc]),
                      noise, bc_weight_now, 0.0)
        np.testing.assert_allclose([float(v) for v in ret], ref, rtol=5e-5, atol=5e-6, err_msg=f"call {k}")
        bc_weight_now = ret[5]  # train_all.py:361
    assert agent.actorTrainable and agent.update_count == 2


def test_agent_facade_without_layernorm():
    """HIRLAgent(..., layerNorm=False, ...) — a constructor argument of the reference (HIRL.py:149-152; the driver passes True,
    train_all.py:204): acting and learning follow the `else` branches of HIRL.py:58-80,126-140."""
    from hirl4ucav_amd.agents.HIRL import Agent as HIRLAgent
    from hirl4ucav_amd.utils.seed import set_seed

    params, data = D.make_params(5), D.make_data(6)  # (LayerNorm entries perturbed: a layerNorm=False agent must ignore them)
    set_seed(7)
    agent = HIRLAgent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 10 ** 5, 128, False, "Harfang_GYM", data["expert_s"], data["expert_a"], 0.5, True)
    for net, p in ((agent.actor, params["actor"]), (agent.targetActor, params["actor"]), (agent.critic, params["critic"]),
                   (agent.targetCritic, params["critic"]), (agent.bc_actor, params["bc_actor"])):
        net.load_state_dict({k: torch.tensor(v) for k, v in p.items()})
    plain = D.plain_layernorm(params)
    o = H.HirlOracle(plain["actor"], plain["critic"], plain["bc_actor"], layer_norm=False)
    s = data["replay"][0, 0:13].astype(np.float64)
    np.testing.assert_allclose(agent.chooseActionNoNoise(s), o.choose_action(s.astype(np.float32)), rtol=1e-5, atol=1e-6)
    o_ln = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    assert np.abs(agent.chooseActionNoNoise(s) - o_ln.choose_action(s.astype(np.float32))).max() > 1e-3  # it is NOT the LayerNorm network
    for row in data["replay"][:600]:
        agent.store(row[0:13], row[13:17], row[17:30], row[30], row[31], 0)
    for k in range(2):
        st_r, st_np, st_t = random.getstate(), np.random.get_state(), torch.get_rng_state()
        idx = random.sample(range(len(agent.buffer)), 128)
        ibc = np.random.choice(D.N_EXPERT, 128, replace=False)
        noise = torch.normal(mean=torch.zeros(4), std=torch.ones(4) * 0.2).numpy()
        random.setstate(st_r); np.random.set_state(st_np); torch.set_rng_state(st_t)  # noqa: E702
        ret = agent.learn(0.5, 0, 0.0)
        rows = data["replay"][idx]
        ref = o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (data["expert_s"][ibc], data["expert_a"][ibc]), noise, 0.5, 0.0)
        np.testing.assert_allclose([float(v) for v in ret], ref, rtol=5e-5, atol=5e-6, err_msg=f"call {k}")


def test_checkpoints_round_trip(tmp_path):
    from hirl4ucav_amd.agents.TD3 import Agent as TD3Agent

    a = TD3Agent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 1000, 128, True, "Harfang_GYM")
    b = TD3Agent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 1000, 128, True, "Harfang_GYM")
    a.saveCheckpoints("Agent1_0_-5_", str(tmp_path))
    b.loadCheckpoints("Agent1_0_-5_", str(tmp_path))
    s = np.linspace(-1, 1, 13)
    np.testing.assert_array_equal(a.chooseActionNoNoise(s), b.chooseActionNoNoise(s))
    assert sorted(os.listdir(tmp_path)) == ["Agent1_0_-5_Actor_Harfang_GYM", "Agent1_0_-5_Critic_Harfang_GYM",
                                            "Agent1_0_-5_TargetActor_Harfang_GYM", "Agent1_0_-5_TargetCritic_Harfang_GYM"]
    for row in D.make_data(1)["replay"][:200]:
        a.store(row[0:13], row[13:17], row[17:30], row[30], row[31], 0)
    out = a.learn()
    assert len(out) == 2 and np.isfinite(out[0]) and np.isfinite(out[1])


@pytest.mark.parametrize("dtype", ["f32", "f32x9"])
def test_vectorised_driver_runs_and_learns_something(tmp_path, capsys, dtype):
    """Two short 'episodes' of the vectorised train_all on 256 envs: exploration, expert labelling, act/step/learn,
    schedules, stats — end to end through the HIP path (f32x9: with the acting kernel's product as the exact bf16 split)."""
    from hirl4ucav_amd import train_all as T

    T.MAX_STEP["straight_line"] = 40
    try:
        cfg = T.parser().parse_args(["--agent", "HIRL", "--type", "soft", "--env", "straight_line", "--random", "--seed", "1", "--dtype", dtype,
                                     "--num_envs", "256", "--episodes", "2", "--result_dir", str(tmp_path), "--buffer_size", "65536", "--synthetic_expert"])
        T.main(cfg)
    finally:
        T.MAX_STEP["straight_line"] = 1500
    out = capsys.readouterr().out
    assert "Episode 2:" in out and "env steps/s" in out and "nan" not in out.lower()


def test_bc_agent_facade_learns_the_expert(tmp_path):
    """agent = BCAgent(actorLR, stateDim, actionDim, h1, h2, useLayerNorm, name, batchSize, expert_states, expert_actions);
    bc_loss = agent.train_actor()  (train_all.py:229,244-250) — the loss falls, the checkpoint loads as HIRL's bc_actor."""
    from hirl4ucav_amd.agents.BC import Agent as BCAgent
    from hirl4ucav_amd.agents.HIRL import Agent as HIRLAgent

    rng = np.random.default_rng(0)
    s = rng.uniform(-1, 1, (2000, 13))
    a = np.tanh(s[:, :4] * 1.5)  # a learnable expert
    agent = BCAgent(1e-3, 13, 4, 256, 512, True, "Harfang_GYM", 128, s, a)
    losses = [float(agent.train_actor()) for _ in range(300)]
    assert np.mean(losses[-20:]) < 0.25 * np.mean(losses[:5])
    agent.saveCheckpoints("Agent1_", str(tmp_path))
    h = HIRLAgent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 1000, 128, True, "Harfang_GYM", s, a, 0.5, True)
    h.load_bc_actor("Agent1_", str(tmp_path))  # train_all.py:311-312
    np.testing.assert_array_equal(h.bc_actor.state_dict()["full2.weight"].numpy(), agent.actor.state_dict()["full2.weight"].numpy())
    assert agent.chooseActionNoNoise(s[0]).shape == (4,)


def test_sac_agent_facade_and_driver(tmp_path, capsys):
    """agent = SACAgent(observation_space=..., action_space=..., log_dir=..., batch_size=128, lr=1e-3, hidden_units=[256, 512],
    memory_size=..., gamma=..., tau=...)  (train_sac.py:214-215); memory.append / explore / exploit / learn(False) / save_models;
    then two short episodes of the vectorised driver with --agent SAC on serpentine envs."""
    import types

    from hirl4ucav_amd import train_all as T
    from hirl4ucav_amd.agents.SAC.agent import SacAgent

    box = lambda n: types.SimpleNamespace(shape=(n,), sample=lambda: np.random.uniform(-1, 1, n))  # noqa: E731
    agent = SacAgent(observation_space=box(13), action_space=box(4), log_dir=str(tmp_path), batch_size=128, lr=1e-3, hidden_units=[256, 512],
                     memory_size=2e5, gamma=0.99, tau=0.005)
    data = D.make_data(3)
    for row in data["replay"][:300]:
        agent.memory.append(row[0:13], row[13:17], row[30], row[17:30], bool(row[31]), bool(row[31]))
    assert len(agent.memory) == 300 and len(agent.memory) > agent.batch_size
    s = data["replay"][0, 0:13]
    a, b = agent.explore(s), agent.exploit(s)
    assert a.shape == (4,) and b.shape == (4,) and np.all(np.abs(a) <= 1) and not np.allclose(a, b)
    for _ in range(4):
        agent.learn(False)
    q1, q2, pl, el, ent, alpha = agent.eng.losses_host()
    assert np.isfinite([q1, q2, pl, el, ent]).all() and 0.99 < alpha < 1.0 and agent.learning_steps == 4
    agent.save_models("Agent1_0_-5_")
    assert sorted(os.listdir(agent.model_dir)) == ["critic_Agent1_0_-5_.pth", "critic_target_Agent1_0_-5_.pth", "policy_Agent1_0_-5_.pth"]
    # the files hold GaussianPolicy / TwinnedQNetwork state_dicts with the reference's key names (SAC/model.py:21,35-38,58)
    pol = torch.load(os.path.join(agent.model_dir, "policy_Agent1_0_-5_.pth"))
    cri = torch.load(os.path.join(agent.model_dir, "critic_Agent1_0_-5_.pth"))
    assert sorted(pol) == sorted(f"policy.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")) and pol["policy.4.weight"].shape == (8, 512)
    assert sorted(cri) == sorted(f"{q}.Q.{i}.{w}" for q in ("Q1", "Q2") for i in (0, 2, 4) for w in ("weight", "bias"))
    before = agent.eng.arena.clone()
    agent.eng.policy.zero_(); agent.eng.critic.zero_(); agent.eng.target_critic.zero_()
    agent.eng.load_models(agent.model_dir, "Agent1_0_-5_")
    assert torch.equal(agent.eng.policy, before[:agent.eng.policy.numel()]) and float(agent.eng.critic.abs().sum()) > 0
    # validate_sac.py on those files: python -m hirl4ucav_amd.validate_all --agent SAC (exploit = tanh(mean), SAC/agent.py:191-196)
    from hirl4ucav_amd import validate_all as V

    vr, vs, _ = V.main(V.parser().parse_args(["--agent", "SAC", "--model_dir", agent.model_dir, "--model_name", "Agent1_0_-5_", "--random", "--seed", "5",
                                              "--episodes", "6", "--validation_step", "120"]))
    assert len(vr) == 2 and all(np.isfinite(vr)) and all(0.0 <= x <= 1.0 for x in vs)
    capsys.readouterr()
    T.MAX_STEP["serpentine"] = 30
    try:
        T.main(T.parser().parse_args(["--agent", "SAC", "--env", "serpentine", "--random", "--seed", "2", "--num_envs", "256", "--episodes", "2",
                                      "--result_dir", str(tmp_path), "--buffer_size", "65536"]))
        T.main(T.parser().parse_args(["--agent", "SAC", "--type", "ESAC", "--env", "serpentine", "--seed", "2", "--num_envs", "256", "--episodes", "1",
                                      "--result_dir", str(tmp_path), "--buffer_size", "65536", "--synthetic_expert"]))
    finally:
        T.MAX_STEP["serpentine"] = 1500
    out = capsys.readouterr().out
    assert "Episode 2:" in out and "alpha" in out and "nan" not in out.lower()


@pytest.mark.parametrize("agent_args", [["--agent", "HIRL", "--type", "soft", "--env", "straight_line"],
                                        ["--agent", "SAC", "--type", "ESAC", "--env", "serpentine"]])
def test_resumed_run_continues_bit_identically(tmp_path, capsys, agent_args):
    """True resume: 3 episodes straight == 2 episodes, stop, --resume, 1 more — every network, Adam moment, env
    state word, replay row and counter equal bit for bit; validation + checkpoint files are written on the way."""
    from hirl4ucav_amd import train_all as T

    env_name = agent_args[-1]
    common = agent_args + ["--random", "--seed", "3", "--num_envs", "256", "--buffer_size", "32768", "--checkpoint_rate", "2", "--synthetic_expert",
                           "--separate_launches"]  # one env workgroup: the replay insert order, and with it the run, is reproducible
    T.MAX_STEP[env_name] = 48
    try:
        a = T.main(T.parser().parse_args(common + ["--episodes", "3", "--snapshot_every", "3", "--result_dir", str(tmp_path / "a")]))
        b1 = T.main(T.parser().parse_args(common + ["--episodes", "2", "--snapshot_every", "2", "--result_dir", str(tmp_path / "b")]))
        b2 = T.main(T.parser().parse_args(common + ["--episodes", "3", "--snapshot_every", "3", "--result_dir", str(tmp_path / "c"), "--resume", b1]))
    finally:
        T.MAX_STEP[env_name] = 1900 if env_name == "circular" else 1500
    out = capsys.readouterr().out
    assert out.count("Validation 1:") == 2 and len(os.listdir(os.path.join(b1, "model"))) in (3, 4)
    sa = torch.load(os.path.join(a, "state_rank0.pt"), weights_only=False)
    sb = torch.load(os.path.join(b2, "state_rank0.pt"), weights_only=False)
    assert sa["driver"] == sb["driver"] and sa["driver"]["episode"] == 3
    assert sa["engine"]["counters"] == sb["engine"]["counters"]
    # the 8 logged loss scalars are sums of float atomics (order-dependent in the last bit even between two identical runs):
    # everything that feeds the next step must match exactly, the log slots only closely
    if agent_args[1] == "SAC":
        from hirl4ucav_amd.agents.sac_engine import SacEngine as Eng
    else:
        from hirl4ucav_amd.agents.engine import HirlEngine as Eng
    eng = Eng()
    lo = (eng.losses.data_ptr() - eng.arena.data_ptr()) // 4
    la, lb = sa["engine"]["arena"][lo:lo + 8].clone(), sb["engine"]["arena"][lo:lo + 8].clone()
    np.testing.assert_allclose(la.numpy(), lb.numpy(), rtol=1e-4, atol=1e-6)
    sa["engine"]["arena"][lo:lo + 8] = 0
    sb["engine"]["arena"][lo:lo + 8] = 0
    for part, keys in (("engine", ["arena"]), ("env", ["state", "obs", "episode_ctr", "stats"]), ("replay", ["ring", "success"])):
        for k in keys:
            assert torch.equal(sa[part][k].view(torch.uint8), sb[part][k].view(torch.uint8)), (part, k)
    assert sa["replay"]["total"] == sb["replay"]["total"] > 0


def test_expert_collector_bc_pipeline(tmp_path, capsys):
    """the scripted pilot flies the build's own sim the way ai_data_col.py flies Harfang's IA — per-tick
    (obs, action) until the opponent is destroyed, the episode filter, the two-row CSV — and the file feeds the expert
    labeller and the BC branch of the driver (train_all.py:227-229,244-260,289-306)."""
    from hirl4ucav_amd import train_all as T
    from hirl4ucav_amd.data import ai_data_col as C
    from hirl4ucav_amd.utils.data_processor import read_data

    out = C.main(["--env", "straight_line", "--random", "--episodes", "6", "--seed", "4", "--out", str(tmp_path / "expert.csv")])
    log = capsys.readouterr().out
    assert "invalid data 0" in log and "Finish" in log
    s, a = read_data(out)
    assert s.shape[1] == 13 and a.shape == (s.shape[0], 4) and 6 * 600 < s.shape[0] < 6 * 1400
    ends = np.nonzero(s[:, 12] <= 0)[0]
    assert len(ends) == 6 and ends[-1] == s.shape[0] - 1          # each episode ends on the first destroyed observation
    assert int((a[:, 3] > 0).sum()) == 6                           # one launch per episode: the rail is empty afterwards
    starts = np.concatenate([[0], ends[:-1] + 1])
    for b, e in zip(starts, ends):
        fire = b + int(np.nonzero(a[b:e + 1, 3] > 0)[0][0])
        assert s[fire, 7] > 0 and s[fire, 8] > 0 and (s[b:fire, 7] < 0).all()  # launched on the first locked tick
        assert np.abs(s[b, 0:3] * 1e4 - np.array([0, -700, -4000])).max() <= 100.5  # random_reset offsets, HarfangEnv_GYM.py:74
    # the labeller sees one +600 kill transition per episode and skips the episode joints (train_all.py:289-306)
    rows, succ = T.label_expert(s, a, "cuda")
    r = rows[:, 30].cpu().numpy()
    assert int((r > 500).sum()) == 6 and rows.shape[0] == s.shape[0] - 1 - 6 + 1
    T.MAX_STEP["straight_line"] = 300
    try:
        d = T.main(T.parser().parse_args(["--agent", "BC", "--env", "straight_line", "--random", "--seed", "1", "--episodes", "3", "--expert_csv", out,
                                          "--checkpoint_rate", "3", "--bc_validate_from", "3", "--result_dir", str(tmp_path)]))
    finally:
        T.MAX_STEP["straight_line"] = 1500
    log = capsys.readouterr().out
    losses = [float(l.split("bc_loss")[1]) for l in log.splitlines() if l.startswith("Episode")]
    # an untrained actor's MSE against the pilot's stick levels is ~0.1-0.5; 900 steps bring the minibatch loss below 1e-3
    assert len(losses) == 3 and max(losses) < 1e-3 and "Validation 1:" in log
    files = os.listdir(os.path.join(d, "model"))
    assert len(files) == 1 and files[0].endswith("Actor_Harfang_GYM")
    sd = torch.load(os.path.join(d, "model", files[0]))
    assert sorted(sd) == sorted(H.ACTOR_KEYS)


def test_validation_latches_success_at_the_first_done():
    """validate() keeps stepping envs that are already done (no auto reset), so EPISODE_SUCCESS raised LATER — a missile
    still in flight when the aircraft left the altitude band — must not count: the reference reads env.episode_success at the step
    it first sees done (train_all.py:59-64)."""
    from hirl4ucav_amd import _lib
    from hirl4ucav_amd import train_all as T
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.agents.HIRL import init_actor_state_dict, init_critic_state_dict
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv

    n = 4
    env = BatchedHarfangEnv(n, scenario="straight_line", seed=0, auto_reset=False, random_reset=False, collect_stats=False)
    env.reset()
    st = env.state.clone()
    flags = st[35].view(torch.int32)
    # envs 0, 1: above the altitude band (done at step 1) with a guided missile 600 m behind the opponent: it kills ~40 ticks later
    for i in (0, 1):
        st[1, i] = 10500.0
        st[26:29, i] = st[13:16, i] - torch.tensor([0.0, 0.0, 600.0], device=st.device)
        st[29:32, i] = torch.tensor([0.0, 0.0, 900.0], device=st.device)
        flags[i] = (int(flags[i]) | _lib.F_M_ACTIVE | _lib.F_M_GUIDED) & ~_lib.F_SIM_SLOT
    # env 2: a kill that IS the cause of done (missile about to hit, aircraft inside the band) counts
    st[26:29, 2] = st[13:16, 2] - torch.tensor([0.0, 0.0, 30.0], device=st.device)
    st[29:32, 2] = torch.tensor([0.0, 0.0, 900.0], device=st.device)
    flags[2] = (int(flags[2]) | _lib.F_M_ACTIVE | _lib.F_M_GUIDED | _lib.F_FIRE_SUCCESS) & ~_lib.F_SIM_SLOT
    env.set_state(st, env.obs)
    eng = E.HirlEngine(batch=128)
    eng.load_params(init_actor_state_dict(), init_critic_state_dict())
    calls = eng.act_calls
    mean, std, succ, fire = T.validate(eng, "straight_line", n, 200, False, 0, torch.device("cuda"), env=env)
    f = env.state[35].view(torch.int32)
    assert all(int(f[i]) & _lib.F_EPISODE_SUCCESS for i in (0, 1, 2))  # the late kills DID happen in the simulator ...
    assert succ == 1 and fire == 1                                      # ... but only env 2's counts
    assert eng.act_calls == calls                                       # validation leaves the exploration-noise counter alone


def test_front_trip_falls_back_to_the_reference_order_from_the_last_snapshot(tmp_path, capsys, monkeypatch):
    """The front launch's in-launch waits lean on an undocumented dispatch order (include/hirl4ucav.h hx_hirl_front), so a wait that gives up must be cheap
    to see and survivable: the driver reads the status word every --status_check_every vector steps, and on a trip reloads the newest snapshot and
    continues in the reference's order (hirl/train_all.py:343-361: store, then sample) inside the same process — or exits with code 3 when there is
    nothing to go back to.  The trip is injected at the host's read of the word (the kernels never produced one: tools/soak_front_roles.sh)."""
    from hirl4ucav_amd import train_all as T
    from hirl4ucav_amd.agents.engine import HirlEngine

    common = ["--agent", "HIRL", "--type", "soft", "--env", "straight_line", "--random", "--seed", "3", "--num_envs", "256", "--buffer_size", "32768",
              "--checkpoint_rate", "100", "--synthetic_expert", "--status_check_every", "16"]
    reads = {"n": 0, "trip_at": 10}  # 48 steps per episode, a read every 16 steps + one at each episode's end = 4 per episode: read 10 is the second of episode 3

    def status(self):
        reads["n"] += 1
        return 2 if reads["n"] == reads["trip_at"] else 0

    monkeypatch.setattr(HirlEngine, "front_status", status)
    T.MAX_STEP["straight_line"] = 48
    try:
        d = T.main(T.parser().parse_args(common + ["--episodes", "4", "--snapshot_every", "1", "--result_dir", str(tmp_path / "a")]))
        out = capsys.readouterr().out
        assert "status word 2: launch C waiting for launches A / B" in out and "back to the snapshot of episode 2" in out and "REFERENCE's order" in out
        assert out.count("Episode 3:") == 1 and "Episode 4:" in out  # the tripped episode printed nothing; the run finished
        snap = torch.load(os.path.join(d, "state_rank0.pt"), weights_only=False)
        assert snap["driver"]["episode"] == 4
        reads["n"] = 0  # ... and with nothing to go back to, or when told so: exit code 3
        with pytest.raises(SystemExit) as e:
            T.main(T.parser().parse_args(common + ["--episodes", "4", "--snapshot_every", "0", "--result_dir", str(tmp_path / "b")]))
        assert e.value.code == 3 and "no snapshot to go back to" in capsys.readouterr().out
        reads["n"] = 0
        with pytest.raises(SystemExit) as e:
            T.main(T.parser().parse_args(common + ["--episodes", "4", "--snapshot_every", "1", "--on_front_trip", "exit", "--result_dir", str(tmp_path / "c")]))
        assert e.value.code == 3
    finally:
        T.MAX_STEP["straight_line"] = 1500


def test_step_learn_that_is_refused_leaves_the_engine_as_it_was():
    """step_learn advanced the host's counters before hx_hirl_front could refuse (here: a ring smaller than 2 n) — a caller that catches the
    error was left with a host epoch ahead of the device's hand-off counters.  Counters are put back and the hand-off words start over."""
    from hirl4ucav_amd import _lib
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.agents.HIRL import init_actor_state_dict, init_critic_state_dict
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    n = 512
    eng = E.HirlEngine(batch=128, use_bc=False, slope=0.01)
    eng.load_params(init_actor_state_dict(), init_critic_state_dict())
    small, big = DeviceReplay(768), DeviceReplay(1 << 14)
    env = BatchedHarfangEnv(n, scenario="straight_line", seed=1, replay=big)
    env.reset()
    for _ in range(2):
        env.step(torch.rand((n, 4), device="cuda") * 2 - 1)
    eng.step_learn(env, None, None, act_sigma=0.1, sample_seed=5)
    before = (eng._front_epoch, eng.critic_step, eng.actor_step, eng.update_count, eng.sample_calls, eng.act_calls, env.steps_issued, eng.actor_trainable)
    env2 = BatchedHarfangEnv(n, scenario="straight_line", seed=1, replay=small)
    env2.reset()
    env2.step(torch.rand((n, 4), device="cuda") * 2 - 1)
    issued2 = env2.steps_issued
    with pytest.raises(_lib.HxError):
        eng.step_learn(env2, None, None, act_sigma=0.1, sample_seed=5)
    assert (0, eng.critic_step, eng.actor_step, eng.update_count, eng.sample_calls, eng.act_calls, env.steps_issued, eng.actor_trainable) == (eng._front_epoch,) + before[1:]
    assert env2.steps_issued == issued2 and int(eng._front[0].abs().sum()) == 0 and eng.front_status() == 0
    for _ in range(3):  # ... and the loop goes on (the first call draws with a launch of its own)
        eng.step_learn(env, None, None, act_sigma=0.1, sample_seed=5)
    eng.front_check()
    assert np.isfinite(eng.losses_host()[0])


def test_step_learn_that_fails_behind_the_front_launch_is_not_rolled_back(tmp_path, monkeypatch):
    """the rollback above is right only while nothing has been enqueued.  A failure AFTER hx_hirl_front has returned (here: the rest
    of learn() refuses) finds a device that has stepped the envs and inserted into the ring: putting the host's counters back would replay the same Philox
    noise and sample calls against it.  The counters stay advanced, the engine refuses further steps (`needs_reload`) until a snapshot is loaded, and a
    snapshot load puts host and device back in step."""
    from hirl4ucav_amd import _lib
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.agents.HIRL import init_actor_state_dict, init_critic_state_dict
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils import checkpoint as CK
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    n = 512
    eng = E.HirlEngine(batch=128, use_bc=False, slope=0.01)
    eng.load_params(init_actor_state_dict(), init_critic_state_dict())
    rep = DeviceReplay(1 << 14)
    env = BatchedHarfangEnv(n, scenario="straight_line", seed=1, replay=rep)
    env.reset()
    for _ in range(2):
        env.step(torch.rand((n, 4), device="cuda") * 2 - 1)
    eng.step_learn(env, None, None, act_sigma=0.1, sample_seed=5)
    snap = str(tmp_path / "state.pt")
    CK.save_run(snap, eng, env, rep, {"episode": 0})
    held = (eng.critic_step, eng.sample_calls, eng.act_calls, env.steps_issued, eng._front_epoch)
    real = _lib.call

    def failing(name, *a):
        if name == "hx_hirl_learn_back":
            raise _lib.HxError("hx_hirl_learn_back failed (-2): injected")
        return real(name, *a)

    monkeypatch.setattr(_lib, "call", failing)
    with pytest.raises(_lib.HxError, match="injected"):
        eng.step_learn(env, None, None, act_sigma=0.1, sample_seed=5)
    monkeypatch.setattr(_lib, "call", real)
    # the front launch WAS enqueued: nothing is put back, and the engine says so
    assert eng.needs_reload and (eng.critic_step, eng.sample_calls, eng.act_calls, eng._front_epoch) == (held[0] + 1, held[1] + 1, held[2] + 1, held[4] + 1)
    with pytest.raises(_lib.HxError, match="load a snapshot"):
        eng.step_learn(env, None, None, act_sigma=0.1, sample_seed=5)
    with pytest.raises(_lib.HxError, match="load a snapshot"):
        eng.learn()
    CK.load_run(snap, eng, env, rep)  # back to a state where host and device agree
    assert not eng.needs_reload and (eng.critic_step, eng.sample_calls, eng.act_calls) == held[:3] and eng._front_epoch == 0
    for _ in range(3):
        eng.step_learn(env, None, None, act_sigma=0.1, sample_seed=5)
    eng.front_check()
    assert np.isfinite(eng.losses_host()[0])


def test_snapshot_records_the_acting_format(tmp_path):
    """what decides the acting arithmetic beyond --dtype (the row count from which the exact split applies, the number of partial products,
    the front launch's format) is part of a whole-run snapshot; loading it into an engine that acts otherwise warns instead of continuing silently."""
    import warnings

    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.agents.HIRL import init_actor_state_dict, init_critic_state_dict
    from hirl4ucav_amd.utils import checkpoint as CK

    eng = E.HirlEngine(batch=128, use_bc=False, slope=0.01)
    eng.load_params(init_actor_state_dict(), init_critic_state_dict())
    st = CK.engine_state(eng)
    assert st["acting_format"] == {"act_dtype": "f32", "update_dtype": "f32", "x9_rows": 4096, "front_x9": True, "x9_terms": 6}
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        CK.load_engine_state(eng, st)  # the same format: silent
    other = E.HirlEngine(batch=128, use_bc=False, slope=0.01)
    other.load_params(init_actor_state_dict(), init_critic_state_dict())
    other.x9_rows = 16384  # round 4's rule
    with pytest.warns(UserWarning, match="other acting arithmetic"):
        CK.load_engine_state(other, st)
    old = {k: v for k, v in st.items() if k != "acting_format"}  # a snapshot from before round 6
    with pytest.warns(UserWarning, match="unrecorded"):
        CK.load_engine_state(eng, old)


class _ScriptedFire:
    """an engine whose policy is the loaded actor's, with the launch decision scripted (fire on the steps listed): a random-init actor never locks, and the
    counters of the infinite env only move when missiles leave the rail"""

    def __init__(self, eng, fire_steps):
        self.eng, self.fire_steps, self.t, self.act_calls = eng, set(fire_steps), 0, 0

    def act(self, obs):
        a = self.eng.act(obs).clone()
        a[:, 3] = 1.0 if self.t in self.fire_steps else -1.0
        self.t += 1
        return a


@pytest.mark.parametrize("infinite", [False, True])
def test_validate_all_round_is_the_references_loop_episode_by_episode(infinite):
    """hirl4ucav_amd.validate_all.validate_round (one batch) against validate() of hirl/validate_all.py:25-77 restated over the single-env classes, one fresh env per
    episode: scores, fire successes and — with --infinite — the launch counters of HarfangSerpentineInfiniteEnv (re-arm before every 60th call, :484-486)."""
    import hirl4ucav_amd.environments.HarfangEnv_GYM as G
    from hirl4ucav_amd import validate_all as V
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv

    params = D.make_params(D.PARAM_SEED)
    eng = E.HirlEngine(batch=128)
    eng.load_params(params["actor"], params["critic"], params["bc_actor"])
    episodes, steps, seed, dev = 5, 260, 11, torch.device("cuda")
    fire_steps = (3, 61, 70, 125, 190, 241)  # 3: the first missile; 61 / 125 / 190 / 241: behind a re-arm; 70: nothing on the rail
    benv = BatchedHarfangEnv(episodes, scenario="serpentine", seed=seed, auto_reset=False, random_reset=True, collect_stats=False)
    benv.episode_ctr += 1  # (the single-env classes count the episode before they reset: the same Philox counter on both sides)
    benv.reset()
    got = V.validate_round(_ScriptedFire(eng, fire_steps), episodes, steps, True, seed, dev, infinite=infinite, env=benv)
    # the reference's loop (validate_all.py:25-77), one env per episode so that every episode's call counter starts at 0 (the module's documented deviation)
    scores, fire_success, launches, locked = [], 0, 0, 0
    for e in range(episodes):
        env = (G.HarfangSerpentineInfiniteEnv if infinite else G.HarfangSerpentineEnv)()
        env._env = BatchedHarfangEnv(1, scenario="serpentine", seed=seed, auto_reset=False, random_reset=True, env_id0=e, collect_stats=False)
        agent = _ScriptedFire(eng, fire_steps)
        state = env.random_reset()
        total, done = 0.0, False
        for step in range(steps):
            if not done:
                action = agent.act(torch.from_numpy(np.asarray(state, np.float32)[None]).cuda())[0].cpu().numpy()
                state, reward, done, _, *_rest = env.step_test(action)
                total += reward
                if step == steps - 1:
                    break
            else:
                fire_success += int(env.fire_success)
                break
        scores.append(total)
        if infinite:
            launches += env.infinite_total_fire
            locked += env.infinite_total_success
    assert abs(got[0] - np.mean(scores)) <= 1e-4 * max(1.0, abs(np.mean(scores))), (got[0], np.mean(scores))
    assert got[1] == fire_success / episodes
    if infinite:
        assert launches >= 3 * episodes - 2 and got[2] == (locked / launches if launches else 0.0), (got, launches, locked)  # re-armed missiles DID leave the rail
    else:
        assert got[2] == 0.0


def test_validate_all_driver_loads_a_checkpoint_and_prints_the_references_lines(tmp_path, capsys):
    """python -m hirl4ucav_amd.validate_all: the checkpoint files are the reference's (Agent.saveCheckpoints, HIRL.py:336-342), the output the two lines of
    validate_all.py:202-204 (or the one of :200 with --infinite)."""
    from hirl4ucav_amd import validate_all as V
    from hirl4ucav_amd.agents import engine as E

    params = D.make_params(D.PARAM_SEED)
    torch.save({k: torch.as_tensor(v) for k, v in params["actor"].items()}, os.path.join(tmp_path, "Agent7_100_5_Actor_Harfang_GYM"))
    common = ["--model_dir", str(tmp_path), "--model_name", "Agent7_100_5_", "--random", "--seed", "3", "--episodes", "8", "--validation_step", "200"]
    r, s, f = V.main(V.parser().parse_args(["--agent", "HIRL"] + common))
    out = capsys.readouterr().out.strip().splitlines()
    assert len(r) == 2 and len(s) == 2 and f == [0.0, 0.0]
    assert [float(x) for x in out[-2].split()] == pytest.approx([np.mean(r), np.std(r, ddof=1)]) and [float(x) for x in out[-1].split()] == pytest.approx([np.mean(s), np.std(s, ddof=1)])
    r2, _, _ = V.main(V.parser().parse_args(["--agent", "HIRL"] + common))
    assert r2 == r  # same seed, same rounds
    rt, _, _ = V.main(V.parser().parse_args(["--agent", "TD3"] + common))
    assert rt != r  # the same file read as a LeakyReLU network (TD3.py) is another policy
    _, _, fi = V.main(V.parser().parse_args(["--agent", "HIRL", "--infinite"] + common))
    last = capsys.readouterr().out.strip().splitlines()[-1].split()
    assert len(last) == 2 and float(last[0]) == pytest.approx(np.mean(fi))
    with pytest.raises(FileNotFoundError):
        V.main(V.parser().parse_args(["--agent", "HIRL", "--model_dir", str(tmp_path), "--model_name", "missing_"]))
    del E
