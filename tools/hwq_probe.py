#!/usr/bin/env python
"""How many fresh HIP streams run side by side in this process?  (dev tool behind qv_probe_concurrent_streams)

    python tools/hwq_probe.py [early|late|dist]

early: the package is imported (GPU_MAX_HW_QUEUES=8 exported) before HIP initialises; late: torch.cuda is touched
first; dist: early + a one-rank RCCL process group (run under torch.distributed.run).  Prints elapsed / spin time
for 2..12 streams.
"""
import ctypes, json, os, sys

mode = sys.argv[1] if len(sys.argv) > 1 else "early"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
if mode == "late":
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
    import torch
    torch.zeros(8, device="cuda").sum().item()
import offline_tarteel_amd  # noqa: E402,F401
import torch  # noqa: E402
from offline_tarteel_amd.engine import load_library  # noqa: E402

if mode == "dist":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    x = torch.zeros(4, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
torch.zeros(8, device="cuda").sum().item()
lib = load_library()
lib.qv_debug_probe_rounds.restype = ctypes.c_double
lib.qv_debug_probe_rounds.argtypes = [ctypes.c_int32]
print(json.dumps({"mode": mode, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                  "rounds": {n: round(lib.qv_debug_probe_rounds(n), 2) for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12)}}))
