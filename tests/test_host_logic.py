"""CPU tests of the host-side logic that mirrors the reference's drivers and data formats (no GPU, no HIP calls)."""
import os

import numpy as np
import pytest
import torch

from oracle import hirl_oracle as H


def test_flat_layout_follows_reference_state_dict_order():
    from hirl4ucav_amd.agents import engine as E

    assert [k for k, _, _ in E.ACTOR_LAYOUT] == list(H.ACTOR_KEYS)
    assert [k for k, _, _ in E.CRITIC_LAYOUT] == list(H.CRITIC_KEYS)
    assert E.ACTOR_SIZE == 138756 and E.Q_PADDED == 138244 and E.CRITIC_SIZE == 276488  # 138,241 padded to 16 B
    rng = np.random.default_rng(0)
    p = H.init_critic(rng)
    flat = E.pack(p, E.CRITIC_LAYOUT, E.CRITIC_SIZE, "cpu")
    back = E.unpack(flat, E.CRITIC_LAYOUT)
    for k in p:
        np.testing.assert_array_equal(back[k].numpy(), p[k])
    assert float(flat[138241:138244].abs().sum()) == 0.0 and float(flat[-3:].abs().sum()) == 0.0  # pads stay zero


def test_bc_weight_and_expert_num_schedules():
    """train_all.py:328-339 and :356-357."""
    from hirl4ucav_amd.train_all import bc_weight_schedule, checkpoint_tag, expert_num_after

    assert bc_weight_schedule("linear", 0, 0.5) == (0.5, 0.0)
    assert bc_weight_schedule("linear", 1000, 0.5) == (0.3, 0.0)
    assert bc_weight_schedule("linear", 2500, 0.5) == (0.0, 0.0) and bc_weight_schedule("linear", 4000, 0.5) == (0, 0.0)
    assert bc_weight_schedule("fixed", 777, 0.25) == (0.25, 0.0)
    assert bc_weight_schedule("soft", 3, 0.5) == (100, 0.0)
    w, warm = bc_weight_schedule("soft", 100, 0.5, bc_warm_up=True)
    assert w == 100 and abs(warm - 0.2) < 1e-12
    # 128 -> 0, one per 10 steps, including step 0
    e, trace = 128, []
    for step in range(1500):
        e = expert_num_after(e, step)
        trace.append(e)
    assert trace[0] == 127 and trace[9] == 127 and trace[10] == 126 and trace[1269] == 1 and trace[1270] == 0 and trace[-1] == 0
    assert checkpoint_tag(3, 32, 50, -412.6) == "Agent3_64_-413_"


def test_expert_pair_indices_skip_after_done():
    """train_all.py:289-306: after a terminal pair the walk skips one extra row."""
    from hirl4ucav_amd.train_all import expert_pair_indices

    done = np.array([0, 0, 1, 0, 0, 0, 1], bool)  # 8 states, pairs (0,1) .. (6,7)
    np.testing.assert_array_equal(expert_pair_indices(done), [0, 1, 2, 4, 5, 6])
    np.testing.assert_array_equal(expert_pair_indices(np.zeros(4, bool)), [0, 1, 2, 3])


def test_expert_csv_round_trip(tmp_path):
    """The 2-row stringified-array CSV of hirl/utils/data_processor.py:5-18."""
    from hirl4ucav_amd.utils.data_processor import read_data, up_sample, write_data

    rng = np.random.default_rng(1)
    s, a = rng.uniform(-1, 1, (37, 13)), rng.uniform(-1, 1, (37, 4))
    a[:, 3] = np.where(rng.random(37) < 0.2, 1, -1)
    path = os.path.join(tmp_path, "expert_data_ai.csv")
    write_data(s, a, path)
    s2, a2 = read_data(path)
    np.testing.assert_allclose(s2, s, rtol=0, atol=1e-8)  # numpy's str() keeps 8 significant digits
    np.testing.assert_allclose(a2, a, rtol=0, atol=1e-8)
    np.testing.assert_array_equal(up_sample(a2), np.where(a[:, 3] == 1)[0])


def test_uniform_memory_ring_semantics():
    """UniformMemory store/sample/fullEnough/len (buffer.py:11-54) on whatever device torch has here."""
    import random

    from hirl4ucav_amd.utils.buffer import UniformMemory

    m = UniformMemory(5, False)
    for i in range(7):
        m.store(np.full(13, i), np.full(4, i), np.full(13, i + 0.5), float(i), i % 2 == 0, 0)
    assert len(m) == 5 and m.fullEnough(5) and not m.fullEnough(6) and len(m.memory) == 5 and m.position == 2
    assert sorted(m.ring[:, 30].tolist()) == [2.0, 3.0, 4.0, 5.0, 6.0]  # the two oldest were overwritten
    random.seed(3)
    st, ac, ns, rw, dn = m.sample(4)
    assert len(st) == 4 and st[0].shape == (13,) and len(set(rw)) == 4  # without replacement
    for s_, n_, r_ in zip(st, ns, rw):
        assert s_[0] == r_ and n_[0] == r_ + 0.5
    with pytest.raises(NotImplementedError):
        UniformMemory(5, True)


def test_df_shim_and_constants():
    import hirl4ucav_amd.environments.dogfight_client as df
    from hirl4ucav_amd.environments.constants import NormStates

    assert df.connect("127.0.0.1", 50888) is None and df.disable_log() is None
    assert df.set_renderless_mode(True) is None and df.set_client_update_mode(True) is None
    assert NormStates["Plane_position"] == 10000 and abs(NormStates["Plane_Euler_angles"] - np.pi) < 1e-15


def test_checkpoint_view_uses_reference_key_names(tmp_path):
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.agents.HIRL import _NetView, init_actor_state_dict, init_critic_state_dict

    sd = init_critic_state_dict()
    assert list(sd) != [] and set(sd) == set(H.CRITIC_KEYS) and set(init_actor_state_dict()) == set(H.ACTOR_KEYS)
    flat = torch.zeros(E.CRITIC_SIZE)
    v = _NetView(flat, E.CRITIC_LAYOUT, "Critic_Harfang_GYM")
    v.load_state_dict(sd)
    v.saveCheckpoint("Agent1_50_-3_", str(tmp_path))
    flat2 = torch.zeros(E.CRITIC_SIZE)
    v2 = _NetView(flat2, E.CRITIC_LAYOUT, "Critic_Harfang_GYM")
    v2.loadCheckpoint("Agent1_50_-3_", str(tmp_path))
    assert torch.equal(flat, flat2)
    loaded = torch.load(os.path.join(tmp_path, "Agent1_50_-3_Critic_Harfang_GYM"))
    assert list(loaded) == list(H.CRITIC_KEYS)  # a reference Critic.load_state_dict accepts it
    # hidden-weight bounds of kaiming_uniform_(a=0.01, 'relu'): sqrt(6 / fan_in)
    assert sd["full1.weight"].abs().max() <= np.sqrt(6 / 17) and sd["full2.weight"].abs().max() <= np.sqrt(6 / 256)


def test_expert_pilot_steers_to_the_target_and_fires_only_on_lock():
    """hirl4ucav_amd/data/expert_pilot.py: body-frame bearing -> stick levels; launch rule of ai_data_col.py:62-63."""
    from hirl4ucav_amd.data.expert_pilot import ALLY_QUAT, OPPO_POS, pursuit_actions

    n = 4
    state = torch.zeros((37, n))
    state[ALLY_QUAT] = 1.0                       # identity attitude: nose along +Z, Y up, X right
    state[OPPO_POS + 2] = 1000.0                 # all targets 1 km ahead ...
    state[OPPO_POS + 1, 1] = 300.0               # ... env 1: and above
    state[OPPO_POS + 0, 2] = 300.0               # ... env 2: and to the right
    state[OPPO_POS + 0, 3] = -300.0              # ... env 3: and to the left
    obs = torch.zeros((n, 13))
    obs[:, 7], obs[:, 8] = torch.tensor([1.0, 1.0, -1.0, 1.0]), torch.tensor([1.0, -1.0, 1.0, 1.0])
    a = pursuit_actions(state, obs)
    assert a.shape == (n, 4) and float(a.abs().max()) <= 1.0
    assert torch.all(a[0, :3] == 0)                                   # dead ahead: sticks centred
    assert a[1, 0] < 0 and a[1, 1] == 0 and a[1, 2] == 0               # above: pull up (positive pitch level = nose down)
    assert a[2, 2] > 0 and a[2, 1] < 0 and a[3, 2] < 0 and a[3, 1] > 0  # right / left: yaw toward it, bank into the turn
    assert a[:, 3].tolist() == [1.0, -1.0, -1.0, 1.0]                  # locked AND missile on the rail
    g = torch.Generator().manual_seed(0)
    b = pursuit_actions(state, obs, noise_std=0.05, generator=g)
    assert not torch.equal(a[:, :3], b[:, :3]) and torch.equal(a[:, 3], b[:, 3])


def test_collector_episode_filter():
    """ai_data_col.py:86: keep an episode iff destroyed within 2000 steps and 0 <= fire - lock <= 20; rows run to the first
    destroyed observation inclusive; kept episodes are concatenated in order."""
    from hirl4ucav_amd.data.ai_data_col import filter_episodes

    T, E = 60, 5
    S = np.zeros((T, E, 13))
    A = np.zeros((T, E, 4))
    S[:, :, 7], S[:, :, 8], S[:, :, 12], A[:, :, 3] = -1, 1, 0.2, -1
    S[:, :, 0] = np.arange(T)[:, None] + 100 * np.arange(E)[None, :]   # tag rows: step + 100 * episode
    # 0: lock at 10, fire at 10, destroyed at 30 -> kept (31 rows)         1: never destroyed -> dropped
    # 2: lock at 5, fire at 30 (delay 25) -> dropped                       3: lock at 8, no launch recorded (fire = 0) -> dropped (negative delay)
    # 4: lock at 20, fire at 40 (delay 20), destroyed at 50 -> kept (51 rows)
    for e, (lock, fire, dead) in {0: (10, 10, 30), 2: (5, 30, 45), 3: (8, None, 40), 4: (20, 40, 50)}.items():
        S[lock:, e, 7] = 1
        if fire is not None:
            A[fire, e, 3] = 1
            S[fire + 1:, e, 8] = -1
        S[dead:, e, 12] = 0
    S[12:, 1, 7] = 1
    s, a, info = filter_episodes(S, A)
    assert info == {"episodes": 5, "invalid": 3, "lengths": [31, 51], "delt": [0, 20]}
    assert s.shape == (82, 13) and a.shape == (82, 4)
    assert s[:31, 0].tolist() == list(range(31)) and s[31:, 0].tolist() == [400 + t for t in range(51)]
    assert s[30, 12] == 0 and s[29, 12] > 0 and int((a[:, 3] > 0).sum()) == 2


def test_missing_inputs_raise_instead_of_training_on_noise(tmp_path):
    """a mistyped --expert_csv / no expert data at all must raise like the reference's read_data does; the synthetic
    stand-in exists only behind --synthetic_expert."""
    from hirl4ucav_amd import train_all as T

    with pytest.raises(FileNotFoundError):
        T.load_expert(T.parser().parse_args(["--expert_csv", str(tmp_path / "nope.csv")]))
    with pytest.raises(ValueError):
        T.load_expert(T.parser().parse_args([]))
    es, ea = T.load_expert(T.parser().parse_args(["--synthetic_expert"]))
    assert es.shape == (20000, 13) and ea.shape == (20000, 4)


def test_scalar_writer_keeps_the_reference_tags(tmp_path):
    """utils/scalars.py: SummaryWriter when tensorboard is installed, otherwise scalars.jsonl with the same add_scalar calls."""
    import json

    from hirl4ucav_amd.train_all import log_validation
    from hirl4ucav_amd.utils.scalars import JsonlWriter, make_writer

    w = make_writer(str(tmp_path / "summary"))
    log_validation(w, 24, -310.5, 12.0, 0.64, 0.7)  # train_all.py:97-100
    w.close()
    if isinstance(w, JsonlWriter):
        rows = [json.loads(line) for line in open(w.path)]
        assert [r["tag"] for r in rows] == ["Validation/Avg Reward", "Validation/Std Reward", "Validation/Success Rate", "Validation/Fire Success Rate"]
        assert rows[2] == {"tag": "Validation/Success Rate", "value": 0.64, "step": 24}


def test_snapshot_is_replaced_atomically(tmp_path, monkeypatch):
    """utils/checkpoint.save_run writes <file>.tmp and renames: a failed write must leave the previous snapshot untouched."""
    from hirl4ucav_amd.utils import checkpoint as CK

    path = tmp_path / "state_rank0.pt"
    path.write_bytes(b"previous snapshot")
    monkeypatch.setattr(CK.torch.cuda, "synchronize", lambda: None)

    def boom(obj, f):
        open(f, "wb").write(b"partial")
        raise OSError("disk full")

    monkeypatch.setattr(CK.torch, "save", boom)
    monkeypatch.setattr(CK, "engine_state", lambda e: {})
    monkeypatch.setattr(CK, "env_state", lambda e: {})
    monkeypatch.setattr(CK, "replay_state", lambda r: {})
    monkeypatch.setattr(CK.torch.cuda, "get_rng_state", lambda d=None: None)

    class Env:
        device = "cpu"

    with pytest.raises(OSError):
        CK.save_run(str(path), None, Env(), None, {})
    assert path.read_bytes() == b"previous snapshot"


def test_env_snapshot_with_single_way_statistics_still_loads():
    """The env statistics are kept HX_STAT_WAYS times since round 3 (one 128-byte line per way; a statistic is the sum over the ways).  A
    snapshot written before that holds ONE set of 8 or 9 totals: they go into way 0, the sums stay what they were."""
    import types

    import torch

    from hirl4ucav_amd import _lib
    from hirl4ucav_amd.utils import checkpoint as ck

    def fake_env():
        return types.SimpleNamespace(n=4, state=torch.zeros(_lib.ENV_WORDS, 4), obs=torch.zeros(4, 13), episode_ctr=torch.zeros(4, dtype=torch.int32),
                                     stats=torch.ones((_lib.STAT_WAYS, _lib.STAT_PITCH), dtype=torch.int64))
    for words in (8, 9):
        env = fake_env()
        old = {"state": torch.ones(_lib.ENV_WORDS, 4), "obs": torch.ones(4, 13), "episode_ctr": torch.full((4,), 3, dtype=torch.int32),
               "stats": torch.arange(1, words + 1, dtype=torch.int64)}
        ck.load_env_state(env, old)
        assert env.stats.sum(0)[:words].tolist() == list(range(1, words + 1)) and int(env.stats.sum()) == words * (words + 1) // 2
        assert int(env.episode_ctr[0]) == 3 and float(env.state.sum()) == _lib.ENV_WORDS * 4
    env, env2 = fake_env(), fake_env()
    env.stats.random_(0, 1000)
    ck.load_env_state(env2, ck.env_state(env))  # the current format round-trips way by way
    assert torch.equal(env.stats, env2.stats)


def test_front_vs_reference_summaries_follow_from_the_committed_logs(tmp_path):
    """The learning-outcome evidence for the front loop is reproducible from the logs beside it: tools/demo_front_summary.py over
    profiles/r05_demo_front_vs_reference[_final | _seeds3to8]/ gives the committed summary.md, every run is there (3 scenarios x 3 or 6 seeds x {BC, HIRL-soft,
    TD3} x loop), each log says which loop it ran, and front minus reference lies inside the seed-to-seed spread in every row; the nine seeds run on the
    round's final binaries (_final + _seeds3to8), pooled and PAIRED by (scenario, seed), show no difference between the loops."""
    import shutil
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def summary(d):
        return subprocess.run([sys.executable, os.path.join(root, "tools", "demo_front_summary.py"), str(d)], capture_output=True, text=True, check=True).stdout

    def verdicts(out):
        return [ln.split("|")[-2].strip() for ln in out.splitlines() if ln.startswith("| ") and ln.rstrip().endswith(("yes |", "NO |"))]

    for tag, seeds in (("r05_demo_front_vs_reference", 3), ("r05_demo_front_vs_reference_final", 3), ("r05_demo_front_vs_reference_seeds3to8", 6)):
        d = os.path.join(root, "profiles", tag)
        out = summary(d)
        assert out.strip() == open(os.path.join(d, "summary.md")).read().strip(), tag
        rows = [ln for ln in out.splitlines() if ln.startswith("| ") and "| front |" in ln or "| reference |" in ln]
        per_run = [ln for ln in rows if ln.count("|") == 11]
        assert len(per_run) == 3 * seeds * 2 * 2 and "LOG SAYS" not in out, (tag, len(per_run))
        assert len(verdicts(out)) == 6 and all(v == "yes" for v in verdicts(out)), (tag, verdicts(out))
    for env in ("straight_line", "serpentine", "circular"):
        for tag in ("r05_demo_front_vs_reference_final", "r05_demo_front_vs_reference_seeds3to8"):
            for sd in os.listdir(os.path.join(root, "profiles", tag, env)):
                if sd.startswith("seed"):
                    shutil.copytree(os.path.join(root, "profiles", tag, env, sd), tmp_path / env / sd)
    out = summary(tmp_path)
    assert out.strip() == open(os.path.join(root, "profiles", "r05_demo_front_vs_reference_9seeds_summary.md")).read().strip()
    assert all(v == "yes" for v in verdicts(out))
    paired = [ln for ln in out.splitlines() if ln.startswith(("| HIRL-soft | 27 |", "| TD3 | 27 |"))]
    assert len(paired) == 2 and all(ln.rstrip().endswith("| no |") for ln in paired), paired

def test_front_launch_waiters_stay_below_the_chip_in_the_default_shape():
    """Why the front launch's in-launch waits end under any dispatch order in the default shape (include/hirl4ucav.h hx_hirl_front "WHY THE WAITS END"):
    every workgroup is a whole CU, only launch B's target-critic jobs (and launch C's, when it rides) wait, and at B = 128 without launch C they are 128 of
    the 256 CUs' worth; the shapes where they can fill the chip are the opt-in ones."""
    from hirl4ucav_amd.agents.engine import HirlEngine

    w = HirlEngine.front_waiting_workgroups
    assert w(128) == 128 and w(64) == 64 and w(100) == 7 * 16  # two jobs x 8 column workgroups per 16-row tile
    assert w(256) == 256 and w(128, with_c=True) == 128 + 256   # B = 256, or launch C riding: the count reaches the 256 CUs
    assert w(128) < 256


def test_validate_all_flags_are_the_references():
    """hirl4ucav_amd.validate_all accepts every flag of hirl/validate_all.py:214-227 with the reference's defaults (and names the checkpoint, which the
    reference hard-codes, through two required extras)."""
    from hirl4ucav_amd import validate_all as V

    a = V.parser().parse_args(["--model_dir", "d", "--model_name", "Agent1_50_0_"])
    assert (a.agent, a.port, a.type, a.bc_weight, a.load_model, a.render, a.plot, a.seed, a.env, a.random, a.infinite) == \
        ("HIRL", None, "linear", 1, False, False, False, None, "straight_line", False, False)
    assert (a.nums, a.episodes, a.validation_step, a.scenario) == (2, 50, 1200, None)  # validate_all.py:192, :123, :187-190; serpentine whatever --env says
    with pytest.raises(SystemExit):
        V.parser().parse_args([])  # no checkpoint named


def test_sac_entry_points_keep_the_references_defaults():
    """python -m hirl4ucav_amd.train_sac / validate_sac: the command lines of hirl/train_sac.py:441-457 and hirl/validate_sac.py:191-201 on the shared drivers."""
    from hirl4ucav_amd import train_sac, validate_sac

    a = train_sac.parser().parse_args(["--env", "serpentine", "--random"])
    assert (a.agent, a.type, a.env, a.random, a.port, a.render, a.plot, a.seed) == ("SAC", "ESAC", "serpentine", True, None, False, False, None)
    assert train_sac.parser().parse_args(["--type", "SAC"]).type == "SAC"
    v = validate_sac.parser().parse_args(["--model_dir", "d", "--model_name", "t", "--infinite"])
    assert (v.agent, v.type, v.infinite, v.env) == ("SAC", "ESAC", True, "straight_line")
