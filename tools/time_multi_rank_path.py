"""Per-optimiser-step cost of the multi-rank update path on ONE GPU (torch.distributed 'nccl' with world_size 1): the RCCL
calls are degenerate, so this is the floor that kernels + launch + Python overhead put under the N>1 numbers."""
import os, socket, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_fullsize_properties import _filled_agent
from seqdex_amd.a2c_agent import A2CAgent
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
b = _filled_agent(n, 5)
ag = A2CAgent.__new__(A2CAgent)
ag.ppo = b
ag.mini_epochs_num, ag.batch_size, ag.minibatch_size = 1, n * 8, 4
ag.rank, ag.rank_size, ag.multi_gpu = 0, 1, True
ag._update_multi_gpu(); torch.cuda.synchronize()
t = time.time(); ag._update_multi_gpu(); torch.cuda.synchronize(); dt = time.time() - t
steps = n * 8 // 4
print("multi-rank path: %.1f us per optimiser step (%d steps)" % (dt / steps * 1e6, steps))
dist.destroy_process_group()
