#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): golden vectors of the three fragments of the PPO arithmetic the reference carries IN ITS OWN TREE
(utils/rl_games_custom.py = RC; everything else of R1-R9 lives in the absent rl_games 1.5.2):
  * `_calc_neglogp` (RC:2114-2126, torch.jit.script) - the Gaussian negative log-likelihood of R3,
  * `_calc_ac_loss` (RC:2129-2132) - the composition of actor, critic, entropy and bounds losses of R7,
  * the advantage normalisation of `prepare_dataset` (RC:1621-1683; the lines RC:1639-1651) of R6.
RC cannot be imported here (it imports rl_games and gym, both absent; `collections.Iterable`, RC:15, is gone from Python 3.10; and it names an
undefined `RobotArmPolicy`, RC:39-41).  The three definitions are therefore taken out of the reference's file AS THEY STAND - located with
`ast`, their source text read from /root/reference at generation time - and executed: the two jit functions with their `@torch.jit.script`
decorators, `prepare_dataset` unbound on a stand-in `self` that records what it hands to the dataset (the way oracle/gen_golden.py drives the
task class).  Nothing of the reference's text is stored: the fixture holds inputs and outputs only.

  python oracle/gen_golden_rc.py      # needs /root/reference; writes tests/golden/RC1_ppo_fragments.npz
"""
import ast
import os
import sys
import textwrap
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
RC = "/root/reference/dexteroushandenvs/utils/rl_games_custom.py"
OUT = os.path.join(HERE, "..", "tests", "golden", "RC1_ppo_fragments.npz")


def extract(names):
    src = open(RC).read()
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            lines = src.splitlines()
            first = min([d.lineno for d in node.decorator_list] + [node.lineno])
            found[node.name] = textwrap.dedent("\n".join(lines[first - 1:node.end_lineno]))
    assert set(found) == set(names), set(names) - set(found)
    return found


def main():
    frag = extract(["_calc_neglogp", "_calc_ac_loss", "prepare_dataset"])
    ns = {"torch": torch, "np": np, "LOG2PI": float(np.log(2.0 * np.pi)), "torch_ext": None}
    # torch.jit.script reads a function's source through inspect: the fragments go through a module file in a temporary directory
    import importlib.util
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rc_fragments.py")
        with open(path, "w") as f:
            f.write("import torch\nimport numpy as np\nLOG2PI = np.log(2.0 * np.pi)\ntorch_ext = None\n\n" + frag["_calc_neglogp"] + "\n\n" + frag["_calc_ac_loss"] + "\n\n" + frag["prepare_dataset"] + "\n")
        spec = importlib.util.spec_from_file_location("rc_fragments", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        g = torch.Generator().manual_seed(17)
        n, a = 96, 23
        x = torch.randn(n, a, generator=g)
        mean = torch.randn(n, a, generator=g) * 0.5
        logstd = torch.randn(a, generator=g) * 0.3
        std = torch.exp(logstd).expand(n, a).contiguous()
        nlp = mod._calc_neglogp(x, x, mean, std, logstd.expand(n, a).contiguous(), False)
        a_loss, c_loss, ent, b_loss = (torch.randn(n, generator=g) for _ in range(4))
        tot = mod._calc_ac_loss(a_loss, c_loss.abs(), 1.0, ent, 0.0, b_loss.abs(), 0.001)
        tot2 = mod._calc_ac_loss(a_loss, c_loss.abs(), 4.0, ent, 0.01, b_loss.abs(), 0.05)
        # prepare_dataset on a stand-in self: no value normalisation, no rnn, advantage normalisation on (YG:57-64)
        rec = {}
        me = types.SimpleNamespace(normalize_value=False, normalize_advantage=True, is_rnn=False, has_central_value=False,
                                   dataset=types.SimpleNamespace(update_values_dict=lambda d: rec.update(d)))
        R = 8 * 24
        returns = torch.randn(R, 1, generator=g) * 3 + 1
        values = torch.randn(R, 1, generator=g)
        z = torch.zeros(R, 1)
        batch = dict(obses=z, next_obses=z, control_dicts=None, next_control_dicts=None, control_goals=None, controls=None, returns=returns, dones=z,
                     values=values, actions=z, pre_actions=z, neglogpacs=z, mus=z, sigmas=z)
        mod.prepare_dataset(me, batch)
    np.savez(OUT, x=x.numpy(), mean=mean.numpy(), logstd=logstd.numpy(), neglogp=nlp.numpy(),
             a_loss=a_loss.numpy(), c_loss=c_loss.abs().numpy(), entropy=ent.numpy(), b_loss=b_loss.abs().numpy(),
             ac_loss_coef_1_0_0p001=tot.numpy(), ac_loss_coef_4_0p01_0p05=tot2.numpy(),
             returns=returns.numpy(), values=values.numpy(), advantages=rec["advantages"].numpy(), old_values=rec["old_values"].numpy())
    print("wrote", os.path.normpath(OUT), {k: tuple(v.shape) for k, v in np.load(OUT).items()})


if __name__ == "__main__":
    main()
