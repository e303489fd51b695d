#!/usr/bin/env python3
"""Does RCCL accept two ranks of one communicator on ONE device?  (Decides whether the two-rank form of the
slab exchange can be rehearsed on a one-GPU box.)  Run as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_two_ranks_one_device.py
Every rank uses cuda:0; the first collective runs under a watchdog, so a hang ends in a message, not a wait."""
import os
import sys
import threading

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
t = threading.Timer(60.0, lambda: (sys.stderr.write(f"rank {rank}: no answer from the first all-reduce in 60 s\n"), os._exit(3)))
t.daemon = True
t.start()
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    x = torch.ones(4, device="cuda:0") * (rank + 1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all-reduce over two ranks on one device gave {x.tolist()}")
except Exception as e:      # noqa: BLE001 - the message is the result
    print(f"rank {rank}: refused: {type(e).__name__}: {str(e)[:400]}")
finally:
    t.cancel()
