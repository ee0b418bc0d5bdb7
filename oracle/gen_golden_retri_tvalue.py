#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): golden vectors of BlockAssemblySearch's transition value network, produced with the REFERENCE's
class (policy_sequencing/terminal_value_function.py::RetriGraspTValue, imported from /root/reference in this container) with the
formula-defined parameters of oracle/task_oracle.py::retri_tvalue_formula_weights.

  python oracle/gen_golden_retri_tvalue.py      # needs /root/reference; writes tests/golden/S7_retri_tvalue.npz (data only)
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (mocks isaacgym / cv2 / h5py and puts the reference on sys.path)
import task_oracle as T  # noqa: E402


def main():
    G.import_reference()
    import importlib
    from unittest import mock
    sys.modules.setdefault("utils.cnn_module", mock.MagicMock())
    tvf = importlib.import_module("policy_sequencing.terminal_value_function")
    net = tvf.RetriGraspTValue(input_dim=65 * 10, output_dim=2)           # SE:397
    sd = T.retri_tvalue_formula_weights()
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    g = torch.Generator().manual_seed(13)
    x = torch.randn(16, 650, generator=g) * 0.7
    with torch.no_grad():
        out = net(x)
        tv = torch.sigmoid(out)[:, 1]                                     # SE:1134
    dst = os.path.join(HERE, "..", "tests", "golden", "S7_retri_tvalue.npz")
    np.savez_compressed(dst, x=x.numpy(), out=out.numpy(), tvalue=tv.numpy())
    print("wrote", dst, os.path.getsize(dst))


if __name__ == "__main__":
    main()
