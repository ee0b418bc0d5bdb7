"""worker of tests/test_gpu_two_ranks_one_gpu.py: one of WORLD_SIZE processes that SHARE cuda:0.  RCCL refuses two ranks on one device, so the
process group is gloo and A2CAgent stages its collectives through the host (a2c_agent.py::_collectives); everything else is the multi-rank
path as the driver's N > 1 runs take it: rank-sharded seeds, parameter broadcast, per optimiser step backward -> collective -> apply.
usage (env RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT): python two_ranks_one_gpu.py out_dir num_envs minibatch epochs"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    out_dir, n, minibatch, epochs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    from seqdex_amd.a2c_agent import A2CAgent
    from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
    from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython
    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/allegro_hand_block_assembly_grasp_sim.yaml")))
    train = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/lego/ppo_continuous_grasp.yaml")))
    cfg["env"]["numEnvs"] = n
    pc = train["params"]["config"]
    pc["minibatch_size"] = minibatch
    pc["central_value_config"]["minibatch_size"] = minibatch
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22 + rank, piles_per_type=2)
    env = RLgamesVecTaskPython(task, "cuda:0")
    pc.update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22, multi_gpu=True)
    agent = A2CAgent("run", train["params"])
    assert agent.multi_gpu and agent.rank == rank and agent.rank_size == int(os.environ["WORLD_SIZE"])
    assert agent._collectives().host_staged
    first = {k: agent.ppo.t[k].cpu().numpy().copy() for k in ("AC_PARAMS", "CV_PARAMS")}
    # everything the update phase of the LAST epoch starts from, taken between its rollout and its update: the test replays the two
    # ranks' updates in ONE process (collectives emulated by tensor copies) and expects the same parameters bit for bit
    DATASET = ("AC_PARAMS", "CV_PARAMS", "MB_OBS", "MB_STATES", "MB_ACTIONS", "MB_MUS", "MB_SIGMAS", "MB_NEGLOGP", "MB_VALUES", "RETURNS", "ADVANTAGES",
               "CV_RMS_MEAN", "CV_RMS_VAR", "STATS", "AC_ADAM_M", "AC_ADAM_V", "CV_ADAM_M", "CV_ADAM_V")
    snap = {}
    upd = agent._update_multi_gpu

    def snapshotting_update():
        torch.cuda.synchronize()
        snap.update({"pre_" + k: agent.ppo.t[k].cpu().numpy().copy() for k in DATASET})
        upd()
    agent._update_multi_gpu = snapshotting_update
    import time
    t0 = time.time()
    for _ in range(epochs):
        agent.train_epoch()
    torch.cuda.synchronize()
    print("rank %d: %d epochs in %.1f s" % (rank, epochs, time.time() - t0), flush=True)
    agent.ppo.update_status()
    c = agent.ppo.ctrl()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ac0=first["AC_PARAMS"], cv0=first["CV_PARAMS"],
             ac=agent.ppo.t["AC_PARAMS"].cpu().numpy(), cv=agent.ppo.t["CV_PARAMS"].cpu().numpy(), obs=agent.ppo.t["MB_OBS"].cpu().numpy(),
             lr=np.float64(agent.last_lr), ac_t=np.int64(int(c.ac_t)), factor_path=np.int64("FACTORS" in agent.ppo.t and minibatch <= 8), **snap)
    dist.barrier()
    agent.ppo.close()
    task.sim.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
