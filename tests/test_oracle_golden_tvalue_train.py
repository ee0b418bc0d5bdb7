"""CPU: oracle/tvalue_train_oracle.py (restatement of the transition-value trainer's iteration, TT:209-231) against the golden vectors
that oracle/gen_golden_tvalue_train.py captured with the reference's own GraspInsertTValue class (tests/golden/TV1_train.npz)."""
import os

import numpy as np
import torch

from oracle import tvalue_train_oracle as TO

NAMES = ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "linear3.weight", "linear3.bias",
         "output_layer.weight", "output_layer.bias"]


def _sd(g, pre):
    return {n: g[pre + n.replace(".", "_")] for n in NAMES}


def test_tvalue_training_iterations_match_reference_module(golden_dir):
    g = np.load(os.path.join(golden_dir, "TV1_train.npz"))
    sd1, losses, out = TO.train_steps(_sd(g, "w0_"), [g["x0"]])
    np.testing.assert_allclose(losses[0], g["losses"][0], rtol=1e-6)
    for n in NAMES:
        np.testing.assert_allclose(sd1[n].numpy(), g["w1_" + n.replace(".", "_")], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out.numpy(), g["pred0"], rtol=1e-5, atol=1e-6)
    sd4, losses, _ = TO.train_steps(_sd(g, "w0_"), [g["x%d" % i] for i in range(4)])
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for n in NAMES:
        np.testing.assert_allclose(sd4[n].numpy(), g["w4_" + n.replace(".", "_")], rtol=1e-5, atol=1e-6)
    assert g["losses"][3] < g["losses"][0]


def test_batch_construction_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "TV1_train.npz"))
    x = TO.noisy_batch(torch.as_tensor(g["succ"][g["si0"]]), torch.as_tensor(g["fail"][g["fi0"]]), torch.as_tensor(g["noise0"]))
    np.testing.assert_allclose(x.numpy(), g["x0"], rtol=1e-6, atol=1e-7)
    lab = TO.labels(1024).numpy()
    assert lab[:512, 1].all() and lab[512:, 0].all() and lab.sum() == 1024
