"""GPU time of the three parts of the multi-rank optimiser step on one MI355X (hipGraph replays of 64 calls each, so that the host is
out of the measurement): forward/backward + factor packing, the factor all-gather (world 1: a copy), gradient rebuild + clip + Adam."""
import os, socket, sys
import torch
import torch.distributed as dist
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_fullsize_properties import _filled_agent
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
b = _filled_agent(n, 5)
fact, fact_all = b.t["FACTORS"], b.t["FACTORS_ALL"]
b.backward_factors(-1)
parts = {"backward_factors": lambda: b.backward_factors(0), "all_gather": lambda: dist.all_gather_into_tensor(fact_all, fact),
         "apply_factors": lambda: b.apply_factors()}
for name, fn in parts.items():
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(64):
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(8):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%s: %.2f us per call" % (name, e0.elapsed_time(e1) / (8 * 64) * 1e3))
dist.destroy_process_group()
