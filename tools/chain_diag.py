"""stage 0 of the chain (InsertSim training + T-value fit) and Orient's harvest under the library SDX_LIB_PATH points at: what the fitted
transition value looks like (range over random brick orientations) and how many piles Orient harvests at the chain's gate.
usage: [SDX_LIB_PATH=...] python tools/chain_diag.py [epochs]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.evaluation import main_rlgames  # noqa: E402
from seqdex_amd.tvalue_trainer import state_dict_from_flat  # noqa: E402
from tools.bench_config3 import prepare_tvalue_and_insert_policy  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
tv, _, prep = prepare_tvalue_and_insert_policy(1024, epochs, fit_iters=2000)
print("lib:", os.environ.get("SDX_LIB_PATH", "default"))
print("stage 0:", json.dumps(prep))
if tv is not None:
    sd = state_dict_from_flat(torch.from_numpy(tv))
    g = torch.Generator().manual_seed(0)
    q = torch.randn(20000, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    x = q
    from seqdex_amd.tvalue_trainer import LAYERS
    for name, _, _ in LAYERS:        # 4-256-128-64-2, ELU after every layer (GS:1196-1201)
        x = torch.nn.functional.elu(torch.nn.functional.linear(x, sd[name + ".weight"], sd[name + ".bias"]))
    t = torch.sigmoid(x[:, 1])
    print("T-value over 20000 random orientations: min %.3f mean %.3f max %.3f, fraction > 0.5: %.4f, > 0.28: %.4f" %
          (t.min(), t.mean(), t.max(), (t > 0.5).float().mean(), (t > 0.28).float().mean()))
    orient, st = main_rlgames("BlockAssemblyOrient", 1024, tvalue_state=tv, seed=22, steps=None,
                              until=lambda tk: int(tk.sim.PILE_HARVEST_COUNT.min()) >= 8, max_steps=640,
                              task_kwargs={"tvalue_gate": 0.5, "piles_per_type": 64, "initial_piles": None})
    print("Orient:", json.dumps(st), orient.sim.PILE_HARVEST_COUNT.cpu().tolist())
    orient.sim.close()
