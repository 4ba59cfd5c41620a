"""Stand-in for bench.py's per-rank body, used by the CPU test of `bench.self_launch`: no GPU, no engine — it proves the
launcher plumbing (N children with RANK / WORLD_SIZE / MASTER_* set, a working rendezvous, ONE relayed JSON line, exit codes)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t)
dist.barrier()
print(f"noise from rank {rank}", flush=True)                 # must NOT reach the parent's stdout
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    dist.destroy_process_group()
    sys.exit(3)
if rank == 0:
    print(json.dumps({"metric": "stub", "n_gpus": world, "sum": float(t.item()), "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
