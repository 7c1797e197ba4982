"""Run under `python -m torch.distributed.run --nproc-per-node W ... tests/_sharded_check.py rccl,oneshot,twostage,twostage-bf16` (gloo, all W ranks on the
one GPU of the box; W = 2, 3, 8): for every exchange of the comma-separated list, on ONE process group (a cold `import torch` per process is what a
fresh box charges for: one launch per world size, not one per case), K sharded learn() calls of the PRODUCT engine — each rank its own minibatch of 128,
ONE exchange per phase — to be compared with a single engine of batch 128 W fed the minibatches concatenated (the same global batch).  Rank 0 writes
$SHARDED_OUT.<exchange>.npz and prints SHARDED_OK <exchange> per case."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirl4ucav_amd.agents import engine as E  # noqa: E402
from tests import _hirl_data as D  # noqa: E402


def main(exchanges):
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo")
    for exchange in exchanges.split(","):
        one(exchange)
        torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def one(exchange):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    params, data = D.make_params(31), D.make_data(32)
    ring = torch.from_numpy(data["replay"]).cuda().contiguous()
    bc = np.zeros((D.N_EXPERT, 32), np.float32)
    bc[:, 0:13], bc[:, 13:17] = data["expert_s"], data["expert_a"]
    bc = torch.from_numpy(bc).cuda()
    e = E.HirlEngine(batch=128)
    assert e.world == world and e.staged
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    if exchange in ("oneshot", "twostage", "twostage-bf16"):
        e.use_oneshot_exchange(timeout_ms=20000, two_stage=exchange != "oneshot", bf16=exchange == "twostage-bf16")
    rng = np.random.default_rng(7)
    K = 6
    draws = [(rng.integers(0, D.N_REPLAY, (world, 128)).astype(np.int32), rng.integers(0, D.N_EXPERT, (world, 128)).astype(np.int32),
              rng.normal(0, 0.2, 4).astype(np.float32)) for _ in range(K)]
    for k, (idx, ibc, noise) in enumerate(draws):
        e.assemble(ring, torch.from_numpy(idx[rank]).cuda(), bc_table=bc, idx_bc=torch.from_numpy(ibc[rank]).cuda())
        e.learn(noise=torch.from_numpy(noise).cuda(), bc_weight_now=100 if k % 4 == 0 else None, bc_warm_up_weight=0.05)
    if e.xchg is not None:
        e.xchg.check()
    torch.cuda.synchronize()
    mine = torch.tensor([e.replica_checksum()], dtype=torch.int64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(every, mine)
    assert all(int(c) == int(mine) for c in every), "replicas diverged"
    out = {k: getattr(e, k).cpu().numpy() for k in ("actor", "critic", "target_actor", "target_critic")}
    out["losses"] = np.asarray(e.losses_host())
    if rank == 0:
        np.savez(os.environ["SHARDED_OUT"] + "." + exchange + ".npz", **out, exchange=np.asarray(e.exchange_name), world=np.asarray(world))
        print("SHARDED_OK", e.exchange_name, flush=True)
    if e.xchg is not None:
        torch.distributed.barrier()
        e.close()  # checks the status word once more, then releases the peer mappings


if __name__ == "__main__":
    main(sys.argv[1])
