"""Is a rare miss of the front loop's oracle check (tests/test_front_gpu.py::test_front_loop_against_the_oracles) a matter of the CHECK (ring slots are handed
out by an atomic: another order -> another minibatch -> now and then a parameter whose gradient sits at a ReLU kink) or of the front launch?  Per step, from
shared states: (1) front launch vs separate launches bit for bit, (2) front vs the update oracle with check_params, (3) the separate launches vs the oracle.
A miss of (2) together with a pass of (1) and a miss of (3) is the check; a miss of (1) would be the kernel.
    python3 tools/ubench/front_oracle_stress.py [rounds]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402
from hirl4ucav_amd.agents import engine as E  # noqa: E402
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv  # noqa: E402
from hirl4ucav_amd.utils.buffer import DeviceReplay  # noqa: E402
from oracle import hirl_oracle as H  # noqa: E402
from tests import _hirl_data as D  # noqa: E402
from tests.test_front_gpu import NETS, sync  # noqa: E402
from tests.test_hirl_gpu import check_params, sync_oracle  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = _lib.load()
params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
bc = np.zeros((data["expert_s"].shape[0], 32), np.float32)
bc[:, 0:13], bc[:, 13:17] = data["expert_s"], data["expert_a"]
bc_t = torch.from_numpy(bc).cuda()
exp = DeviceReplay(D.N_EXPERT)
exp.store_rows(torch.from_numpy(data["expert_rows"]))
n = 768
miss_front = miss_ref = bit_diff = steps = 0
for rnd in range(rounds):
    side = []
    for _ in range(2):
        e = E.HirlEngine(batch=128)
        e.load_params(params["actor"], params["critic"], params["bc_actor"])
        rep = DeviceReplay(4096)
        env = BatchedHarfangEnv(n, scenario="straight_line", seed=1 + rnd, auto_reset=False, replay=rep)
        env.reset()
        side.append((e, env, rep))
    (a, env_a, rep_a), (b, env_b, rep_b) = side
    a0 = np.random.default_rng(rnd).uniform(-1, 1, (n, 4)).astype(np.float32)
    env_a.step(torch.from_numpy(a0).cuda())
    oa, ob = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"]), H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    snap = torch.zeros(1, dtype=torch.int64, device="cuda")
    for k in range(8):
        sync(side[1], side[0])
        sync_oracle(oa, a, E)
        sync_oracle(ob, b, E)
        was_actor = a.actor_trainable
        w = 100 if k % 4 == 0 else (None if k % 4 < 3 else 0.3)
        wo = w if w is not None else float(a.wstate.item())
        a.step_learn(env_a, exp, bc_t, n_main=96, act_sigma=0.1, act_seed=3, sample_seed=11, bc_weight_now=w, bc_warm_up_weight=0.05)
        snap.copy_(rep_b.total)
        b.front_x9 = a.front_x9
        if a.front_x9:
            b.set_act_dtype("f32x9")
        b.act_step(env_b, sigma=0.1, seed=3)
        b.sample(rep_b, exp, bc_t, n_main=96, seed=11, defer=True)
        b._pending[0].total, b._pending[0].guard = snap.data_ptr(), n
        L.hx_debug_set_fwd_nt(64, 1, 1)
        b.learn(bc_weight_now=w, bc_warm_up_weight=0.05)
        L.hx_debug_set_fwd_nt(0, 0, 0)
        steps += 1
        same = all(torch.equal(getattr(a, nm), getattr(b, nm)) for nm in NETS) and torch.equal(a.rows, b.rows)
        bit_diff += 0 if same else 1
        res = []
        for eng, o in ((a, oa), (b, ob)):
            rows, bcr = eng.rows.cpu().numpy().reshape(128, 32), eng.bc_rows.cpu().numpy().reshape(128, 32)
            o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (bcr[:, 0:13], bcr[:, 13:17]), eng._noise.cpu().numpy(), wo, 0.05)
            try:
                check_params(eng, o, E, "x", was_actor_call=was_actor)
                res.append("")
            except AssertionError as ex:
                res.append(str(ex)[:120])
        miss_front += bool(res[0])
        miss_ref += bool(res[1])
        if res[0] or res[1] or not same:
            print(f"round {rnd} step {k}: front == separate launches: {same}; front vs oracle: {res[0] or 'ok'}; separate vs oracle: {res[1] or 'ok'}", flush=True)
    a.front_check()
print(f"{steps} steps: front != separate launches {bit_diff}; oracle check missed by the front loop {miss_front}, by the separate launches {miss_ref}")
