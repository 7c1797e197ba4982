#!/usr/bin/env python3
"""Several bench.py runs in ONE set of processes (tests: a cold `import torch` costs ~10 s per process on a fresh box, and the two-rank forms alone
were six launches of two ranks each): reads a JSON list of bench.py argument lists, runs them one after the other through bench.run_rank on one
process group, and rank 0 prints one JSON line per run — {"argv": [...], "record": {...}} or {"argv": [...], "error": "..."}.

    python tools/bench_many.py '[["--steps", "60", ...], ["--agent", "sac", ...]]'
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/bench_many.py '[[...], [...]]'

Not a benchmark entry point: the driver's contract (ONE line from `python bench.py ...`) is bench.py's; this only shares its process start-up."""
import json
import os
import sys
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    import bench

    runs = json.loads(sys.argv[1])
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    backend = os.environ.get("HX_BENCH_BACKEND", "nccl")
    if launched:  # ONE process group for every run (bench.run_rank keeps a group it finds initialised)
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
    rc = 0
    for argv in runs:
        try:
            res = bench.run_rank(bench.parse(argv))
            line = {"argv": argv, "record": res}
        except BaseException as e:  # noqa: BLE001 — report and go on to the next run (SystemExit included: a refused flag combination)
            traceback.print_exc()
            sys.stderr.write(f"bench.py rank {rank}: {type(e).__name__}: {' '.join(str(e).split())[:1500]}\n")
            line, rc = {"argv": argv, "error": f"{type(e).__name__}: {e}"}, 1
            if world > 1:  # the ranks' collective sequences may have parted: do not run the next set on this group
                print(json.dumps(line), flush=True) if rank == 0 else None
                break
        if rank == 0:
            print(json.dumps(line), flush=True)
    if launched:
        torch.distributed.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
