"""which run differs when tests/test_dist_gpu.py::test_reducer_one_rank flakes: the ragged nano step, three trainer steps, run
plain / plain / reducer / reducer in one fresh process: python tools/probe/diag_reducer_flaky.py <port>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[1] if len(sys.argv) > 1 else "29741"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from tests import dist_gpu_worker as W
runs = []
for force in (False, False, True, True):
    s, t, l, en, nb = W._steps("ragged", dev, force=force)
    runs.append((s, l))
    print("force", force, "losses", ["%.9f" % v for v in l])
base = runs[0]
for i, (s, l) in enumerate(runs[1:], 1):
    worst = max(((base[0][k].float() - s[k].float()).abs().max().item(), k) for k in s if s[k].numel())
    print("run", i, "vs run 0: losses equal", l == base[1], "worst param diff", worst)
dist.destroy_process_group()
