#!/usr/bin/env python3
"""Times all-reduces of ONE message through the peer-read kernels (hx_allreduce_twostage / hx_allreduce_oneshot over hipIpc mappings) in a process
group of its own — bench.py starts one of these per rank as a CHILD process (probe_exchanges), so that the first contact of those kernels with real
xGMI peers cannot take the benchmark's own processes down with it: whatever happens here (a refused mapping, a wait that times out, a memory fault
that aborts the process), the parent sees an exit code and a missing result, keeps RCCL and says so in its line.

    RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT from the environment (rendezvous over gloo: the bytes that travel are hipIpc handles)
    python tools/exchange_probe.py --floats 276488 --messages 200 --kind twostage

Rank 0 prints ONE JSON line: {"transport", "ok", "median_us", "p10_us", "p90_us", "per_rank_median_us", "distinct_gpus", ...}.
The reference has no counterpart (single process, hirl/agents/HIRL.py:52)."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--floats", type=int, default=276488)
    p.add_argument("--messages", type=int, default=200)
    p.add_argument("--kind", default="twostage", choices=["oneshot", "twostage", "twostage-bf16"])
    p.add_argument("--timeout-ms", dest="timeout_ms", type=int, default=5000)
    a = p.parse_args(argv)
    import torch

    from hirl4ucav_amd.agents.exchange import OneShotExchange

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ngpu = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(ngpu, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.distributed.init_process_group("gloo")
    x = OneShotExchange({"probe": a.floats}, dev, None, a.timeout_ms, two_stage=a.kind != "oneshot", bf16=a.kind == "twostage-bf16")
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.messages)]

    for _ in range(4):
        x.write_buffer("probe").fill_(1.0)
        x.allreduce("probe")
    torch.cuda.synchronize()
    torch.distributed.barrier()
    last = None
    for s, e in evs:
        x.write_buffer("probe").fill_(1.0)  # (outside the timed pair: the events bracket the exchange kernel alone)
        s.record()
        last = x.allreduce("probe")
        e.record()
    x.check()  # raises if any wait timed out
    ok = bool((last[:a.floats] == float(world)).all())
    us = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    mine = {"median_us": us[len(us) // 2], "p10_us": us[len(us) // 10], "p90_us": us[(9 * len(us)) // 10], "ok": ok,
            "gpu": str(getattr(torch.cuda.get_device_properties(local), "uuid", local))}
    every = [None] * world
    torch.distributed.all_gather_object(every, mine)
    torch.distributed.barrier()
    x.close()
    torch.distributed.destroy_process_group()
    if rank == 0:
        worst = max(every, key=lambda r: r["median_us"])
        print(json.dumps({"transport": a.kind, "ok": all(r["ok"] for r in every), "bytes": 4 * a.floats, "messages": a.messages, "world_size": world,
                          "median_us": round(worst["median_us"], 2), "p10_us": round(worst["p10_us"], 2), "p90_us": round(worst["p90_us"], 2),
                          "per_rank_median_us": [round(r["median_us"], 2) for r in every], "distinct_gpus": len({r["gpu"] for r in every}),
                          "busbw_GBps": round(2 * (world - 1) / world * 4 * a.floats / worst["median_us"] / 1e3, 2),
                          "statistic": "the slowest rank's median (events around the exchange kernel on the stream it is enqueued on)"}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
