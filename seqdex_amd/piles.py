"""Synthetic saved pile states [8, K, 132, 13]: the reference reads them from a pickle produced by
its Search->Orient stages (GS:412-413, 1507-1513); that artefact is not shipped, so the engine settles its own:
72 free bricks dropped from the spawn lattice (GS:737-742) with per-pile jitter, robot parked in the prepare pose,
`steps` simulate() calls on the GPU."""
import pickle

import numpy as np
import torch


def generate_piles(per_type=8, steps=150, device="cuda:0", seed=22):
    from .sim import SdxSim
    n_out = 8 * per_type
    n = ((int(n_out * 1.4) + 7) // 8) * 8          # 40 % spare: a fifth of the dropped piles loses a brick over the bin wall (below)
    sim = SdxSim(n, device=device, seed=seed)
    try:
        sc = sim.scene
        g = torch.Generator().manual_seed(seed)
        root = sim.ROOT.view(n, 142, 13)
        jit = (torch.rand(n, 72, 2, generator=g) * 2 - 1) * 0.012
        root[:, 9:81, 0:2] += jit.to(root.device)
        yaw = (torch.rand(n, 72, generator=g) * 2 - 1) * 3.14159
        root[:, 9:81, 3] = 0.0
        root[:, 9:81, 4] = 0.0
        root[:, 9:81, 5] = torch.sin(yaw / 2).to(root.device)
        root[:, 9:81, 6] = torch.cos(yaw / 2).to(root.device)
        lo, hi = sc.lower, sc.upper
        pose = np.concatenate([np.array(sc.arm_prepare_pose, np.float32),
                               0.5 * (np.array(sc.finger_reset_unscaled, np.float32) + 1) * (hi[7:] - lo[7:]) + lo[7:]])
        dof = torch.zeros(n, 23, 2)
        dof[:, :, 0] = torch.as_tensor(pose)
        sim.DOF.copy_(dof.view(-1, 2).to(sim.DOF.device))
        sim.TARGETS.copy_(torch.as_tensor(np.tile(pose, (n, 1))).to(sim.DOF.device))
        for _ in range(steps):
            sim.simulate()
        torch.cuda.synchronize()
        piles = sim.ROOT.view(n, 142, 13)[:, 9:141].clone()
        piles[:, :, 7:13] = 0.0
        # a pile in five loses a brick over the 10 cm bin wall while the lattice collapses (tools/drop_bricks.py: 133 of 73 728 bricks with the
        # lattice starting above the floor); only complete piles are kept - 40 % more than needed are dropped - and should fewer survive,
        # the rest are copies of complete ones, so that every saved state has its 72 free bricks over the bin (a few still lie on the parked
        # hand: they drop into the pile when a reset moves the hand away)
        fb = piles[:, :72, 0:3]
        inside = ((fb[:, :, 0] - 0.25).abs() < 0.3) & ((fb[:, :, 1] - 0.19).abs() < 0.21) & (fb[:, :, 2] > 0.55)
        good = inside.all(dim=1)
        if not bool(good.any()):
            raise RuntimeError("generate_piles: no pile kept all its bricks inside the bin")
        gi = torch.nonzero(good).flatten()
        pick = gi[torch.arange(n_out, device=gi.device) % len(gi)]          # the first n_out complete piles (wrapping around if fewer)
        piles = piles[pick]
        # env i of the generator uses type group i % 8 only as a label: piles are exchangeable across groups
        out = piles.view(per_type, 8, 132, 13).permute(1, 0, 2, 3).contiguous().cpu().numpy()
        assert np.isfinite(out).all()
        return out
    finally:
        sim.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's on-disk hand-off of pile states (GS:412-413, OR:419-420 read; SE:1349-1350, OR:1505-1510 write):
#   pickle.dump(saved_searching_ternimal_states_list, f) - a python list of 8 torch tensors [slots, 132, 13] (one per brick-type
#   group; slots = 10000 + num_envs in the reference, filled from index 0, unfilled slots all zero).
def save_pile_pickle(path, piles, counts=None):
    """piles: [8, K, 132, 13] (tensor / array) or a list of 8 [K_t, 132, 13]; writes the reference's list-of-8-tensors pickle"""
    if not isinstance(piles, (list, tuple)):
        piles = [torch.as_tensor(np.asarray(piles[t])) for t in range(8)]
    out = []
    for t in range(8):
        x = torch.as_tensor(np.asarray(piles[t].cpu() if torch.is_tensor(piles[t]) else piles[t]), dtype=torch.float32)
        if counts is not None:
            x = x[:int(counts[t])]
        assert x.ndim == 3 and tuple(x.shape[1:]) == (132, 13), tuple(x.shape)
        out.append(x.contiguous())
    with open(path, "wb") as f:
        pickle.dump(out, f)


def load_pile_pickle(path):
    """the reference's pickle (or one written by save_pile_pickle) -> float32 [8, K, 132, 13] for sdx_load_initial_states, K = the
    smallest number of FILLED slots over the 8 groups (a slot is filled when any brick has a non-zero quaternion)"""
    with open(path, "rb") as f:
        lst = pickle.load(f)
    assert isinstance(lst, (list, tuple)) and len(lst) == 8, "expected a list of 8 tensors [slots, 132, 13]"
    arrs, ks = [], []
    for x in lst:
        a = np.asarray(x.detach().cpu().numpy() if torch.is_tensor(x) else x, dtype=np.float32).reshape(-1, 132, 13)
        filled = np.abs(a[:, :, 3:7]).sum(axis=(1, 2)) > 0
        k = int(filled.sum()) if filled.all() or not filled.any() else int(np.argmin(filled))   # filled from index 0: first empty slot
        arrs.append(a); ks.append(k)
    k = min(ks)
    if k == 0:
        raise ValueError("load_pile_pickle: a brick-type group has no saved pile state")
    return np.stack([a[:k] for a in arrs]).astype(np.float32)
