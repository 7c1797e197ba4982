"""The exchange step of the PRODUCT engine with two ranks on a GPU box (gloo for the RCCL-shaped path, hipIpc mappings for the one-shot
kernel; both ranks share the one GPU): one message per phase, replicas bit-identical, and the result equal to ONE engine updating the
global batch (256 rows = the two shards' minibatches) — SURVEY.md 8e "grad of global batch = mean of shard grads"."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import _hirl_data as D

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def single_engine_reference():
    """6 learn() calls of ONE engine with batch 256: rows = rank 0's 128 ++ rank 1's 128 of tests/_sharded_check.py"""
    from hirl4ucav_amd.agents import engine as E

    params, data = D.make_params(31), D.make_data(32)
    ring = torch.from_numpy(data["replay"]).cuda().contiguous()
    bc = np.zeros((D.N_EXPERT, 32), np.float32)
    bc[:, 0:13], bc[:, 13:17] = data["expert_s"], data["expert_a"]
    bc = torch.from_numpy(bc).cuda()
    e = E.HirlEngine(batch=256)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    rng = np.random.default_rng(7)
    for k in range(6):
        idx, ibc = rng.integers(0, D.N_REPLAY, (2, 128)).astype(np.int32), rng.integers(0, D.N_EXPERT, (2, 128)).astype(np.int32)
        noise = rng.normal(0, 0.2, 4).astype(np.float32)
        e.assemble(ring, torch.from_numpy(idx.reshape(-1)).cuda(), bc_table=bc, idx_bc=torch.from_numpy(ibc.reshape(-1)).cuda())
        e.learn(noise=torch.from_numpy(noise).cuda(), bc_weight_now=100 if k % 4 == 0 else None, bc_warm_up_weight=0.05)
    torch.cuda.synchronize()
    return {k: getattr(e, k).cpu().numpy() for k in ("actor", "critic", "target_actor", "target_critic")}, np.asarray(e.losses_host())


@pytest.mark.parametrize("exchange", ["rccl", "oneshot"])
def test_two_rank_update_equals_the_global_batch_update(exchange, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = str(tmp_path / "sharded.npz")
    port = str(29800 + os.getpid() % 150 + (50 if exchange == "oneshot" else 0))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(ROOT, "tests", "_sharded_check.py"), exchange], cwd=ROOT,
                       env={**os.environ, "SHARDED_OUT": out, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0 and "SHARDED_OK " + exchange in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    got = np.load(out)
    ref, ref_losses = single_engine_reference()
    # same global batch, different summation tree (two shard sums added vs one 256-row reduction) and, in the sharded run, the two actor
    # losses combined AFTER the exchange: fp32 rounding only.  Adam turns a rounding-level gradient difference on a near-zero entry
    # into up to 2 lr of parameter difference, so: 99.9 % of the entries within 2e-5, none beyond 6 x 2 lr.
    for k in ("actor", "critic", "target_actor", "target_critic"):
        d = np.abs(got[k] - ref[k])
        assert (d <= 2e-5).mean() >= 0.999 and d.max() <= 1.2e-2, (k, (d <= 2e-5).mean(), d.max())
    # (rank 0's logged loss values are its own shard's means; only the weight is a global quantity)
    assert abs(got["losses"][5] - ref_losses[5]) < 1e-6  # the BC weight comes from the GLOBAL soft count: equal to the single engine's
