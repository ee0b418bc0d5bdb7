#!/usr/bin/env python3
"""bench.py — env-steps/sec of the BlockAssemblyGraspSim rollout + PPO hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one rl_games epoch of the hot path over one batch of synthetic input: horizon_length (8) vectorised env
steps of num_envs (1024) envs per GPU (policy inference + physics + obs/reward) followed by the complete PPO update
with the SHIPPED hyper-parameters (minibatch_size 4, 5 mini-epochs, central value; cfg/lego/ppo_continuous_grasp.yaml).
value = total env-steps of all ranks / max-over-ranks wall time of exactly K steps = rl_games' "fps total" (PS:136-140).
Rank 0 prints ONE JSON line; see DESIGN.md §7 for every field.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_ENV_STEP = 18496     # algorithmic HBM bytes per env-step of the sim+task path
# HBM traffic per launch: PMC counters cannot be read from inside this process, so `traffic` is NOT a measurement of this run.  It is
# QUOTED from the rocprofv3 passes of this same command committed under profiles/ (profiles/pmc_traffic.json: raw FETCH_SIZE /
# WRITE_SIZE per launch in KiB, the files they came from, the kernel build they were taken on) and is null when no entry matches
# the configuration.  Units and corrections as MI355X_MICROARCH.md's HBM section prescribes: on gfx950 FETCH_SIZE tallies each
# 128-B memory-side read request at 64 B, i.e. reports half of the bytes read -> doubled; other access widths and WRITE_SIZE are
# uncalibrated, so the raw counters are kept next to the corrected figure.
def _load_pmc():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


PMC = _load_pmc()


def pmc_traffic(kernel, n):
    e = PMC.get("%s@%d" % (kernel, n))
    if not e:
        return None, None
    f, w = e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"]
    return (2.0 * f + w) * 1024, {"quoted_from": e["source"], "FETCH_SIZE_bytes_as_reported": f * 1024, "WRITE_SIZE_bytes_as_reported": w * 1024,
                                  "correction": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests at 64 B)",
                                  "note": e.get("note", "")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--num-envs", type=int, default=1024, help="envs per GPU (config[1]: 1024)")
    ap.add_argument("--minibatch", type=int, default=None, help="override minibatch_size (labelled variant, not the headline)")
    ap.add_argument("--pretrain-epochs", type=int, default=0, help="untimed training epochs before the warm-up (SURVEY 8(d) config 2: "
                    "second run with a partially trained policy after 200 epochs)")
    ap.add_argument("--piles-per-type", type=int, default=64, help="saved pile states per brick-type group (SURVEY 8(d) config 2: K = 64)")
    ap.add_argument("--mixed-precision", action="store_true", help="bf16 trunk GEMMs (fp32 master weights / accumulation) on the large-minibatch "
                    "path: BASELINE.json configs[4]; only affects minibatch sizes > 8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-minibatch", action="store_true", help="skip the labelled large-minibatch variant")
    ap.add_argument("--cpu-baseline-envs", type=int, default=1024)
    ap.add_argument("--cpu-baseline-steps", type=int, default=16)
    ap.add_argument("--cpu-baseline-ppo-envs", type=int, default=128, help="PPO leg: envs of the sample dataset (x horizon rows; 128 -> 1280 optimiser steps, a few seconds)")
    return ap.parse_args()


def cpu_baseline(args, scene_desc, root0, dof0, targets0, horizon, minibatch, mini_epochs):
    """CPU port of the same hot path, timed on the host cores on a bounded sample (reported beside the GPU number, not a target):
    sim leg  = oracle/physics_oracle.c (plain-C port of the env physics step, OpenMP over envs) on the bench's own settled piles;
    PPO leg  = oracle/ppo_oracle.py (torch-CPU autograd + Adam, same three networks and hyper-parameters) on a small dataset,
               scaled to the optimiser steps of one epoch.  value = env-steps/s of sim leg + PPO leg together ("fps total")."""
    import numpy as np
    import torch
    from oracle import physics_oracle as po
    from oracle.ppo_oracle import PPOOracle, DEFAULT_CFG
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n_full = root0.shape[0]
    n = min(args.cpu_baseline_envs, n_full)
    root, dof, tg = root0[:n].copy(), dof0[:n].copy(), targets0[:n].copy()
    po.simulate(scene_desc, root, dof, tg)            # warm-up (also builds the .so if needed)
    t = time.time()
    for _ in range(args.cpu_baseline_steps):
        po.simulate(scene_desc, root, dof, tg)
    sim_dt = time.time() - t
    sim_rate = n * args.cpu_baseline_steps / sim_dt
    # the same leg with Isaac Gym's default num_threads (4, CF:201) and the shipped YAML's 64 (EG:158): the SAME sample (same envs,
    # same number of steps from the same start state) for every thread count, so that the best-of below compares like with like
    by_threads = {str(cores): sim_rate}
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        for th in (4, 64):
            if th >= cores:
                continue
            gomp.omp_set_num_threads(th)
            r2, d2, t2 = root0[:n].copy(), dof0[:n].copy(), targets0[:n].copy()
            po.simulate(scene_desc, r2, d2, t2)           # the same warm-up step as the all-cores leg
            t = time.time()
            for _ in range(args.cpu_baseline_steps):
                po.simulate(scene_desc, r2, d2, t2)
            by_threads[str(th)] = n * args.cpu_baseline_steps / (time.time() - t)
        gomp.omp_set_num_threads(cores)
    except OSError:
        pass
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # PPO leg (rank-4 minibatches: the torch-CPU step is dispatch/Adam-stream bound, more than ~16 threads only adds fork/join cost)
    ppo_threads = min(16, cores)
    torch.set_num_threads(ppo_threads)
    m = args.cpu_baseline_ppo_envs
    oc = dict(DEFAULT_CFG)
    oc.update(minibatch=minibatch, mini_epochs=mini_epochs)
    orc = PPOOracle(oc, seed=0)
    g = torch.Generator().manual_seed(0)
    R = m * horizon
    ds = dict(obs=torch.randn(R, 396, generator=g), states=torch.randn(R, 564, generator=g), actions=torch.randn(R, 23, generator=g),
              mus=torch.zeros(R, 23), sigmas=torch.ones(R, 23), neglogp=torch.full((R,), 30.0), values=torch.zeros(R),
              returns=torch.randn(R, generator=g))
    t = time.time()
    orc.update(ds)
    ppo_dt = time.time() - t
    ppo_steps = mini_epochs * (R // minibatch)
    us_per_opt_step = ppo_dt / ppo_steps * 1e6
    epoch_opt_steps = mini_epochs * (n_full * horizon // minibatch)
    # the headline CPU figure takes the BEST of the thread counts tried for the sim leg (more threads than ~64 lose to fork/join and
    # memory traffic on this box); `cores` = the threads that figure used
    best_threads = max(by_threads, key=lambda k: by_threads[k])
    sim_best = by_threads[best_threads]
    epoch_s = n_full * horizon / sim_best + epoch_opt_steps * us_per_opt_step * 1e-6
    return {"value": n_full * horizon / epoch_s, "unit": "env-steps/s", "cores": int(best_threads), "host_cores": cores, "kind": "port", "ppo_threads": ppo_threads,
            "cpu_model": cpu_model, "sim_only_env_steps_per_s": sim_best, "sim_only_env_steps_per_s_by_omp_threads": by_threads,
            "ppo_us_per_optimiser_step": us_per_opt_step,
            "sample": "sim: %d envs x %d physics steps of oracle/physics_oracle.c (OpenMP over envs; the same sample for every thread count tried, %d threads: %.1f s); "
                      "PPO: %d optimiser steps (minibatch %d) of oracle/ppo_oracle.py on torch-CPU (%d threads), %.1f s, scaled to the "
                      "%d steps of one epoch; no policy inference / obs kernels in the CPU number"
                      % (n, args.cpu_baseline_steps, cores, sim_dt, ppo_steps, minibatch, torch.get_num_threads(), ppo_dt, epoch_opt_steps)}


FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: v_mfma_f32_32x32x16_bf16, dense (no 2:1 sparsity)


def gemm_roofline(agent, n, horizon, upd_ms_per_epoch):
    """large-minibatch update (sdxp_bigmb.hip): the dominant kernels are the fp32-MFMA GEMMs k_gemm<...>; algorithmic flops of one
    optimiser step = 6 x minibatch x parameters (forward 2, data gradient 2, weight gradient 2 flops per weight and sample)"""
    p = agent.ppo.param_count(0) + agent.ppo.param_count(1)
    nsteps = agent.mini_epochs_num * (n * horizon // agent.minibatch_size)
    flops_step = 6.0 * agent.minibatch_size * p
    ms_step = upd_ms_per_epoch / nsteps
    ach = flops_step / (ms_step * 1e-3) / 1e12
    # what the step executes: no data gradient reaches the network inputs, so layer 0 (2 x obs_dim + state_dim inputs x 1024 units) has a
    # forward and a weight-gradient product only: 6 P - 2 P_layer0 per sample (about 5.2 P)
    c = agent.ppo.cfg
    executed = flops_step - 2.0 * agent.minibatch_size * (2 * c.obs_dim + c.state_dim) * int(c.units[0])
    bf = bool(agent.config.get("mixed_precision", False))
    peak = BF16_MFMA_PEAK_TFLOPS if bf else FP32_MFMA_PEAK_TFLOPS
    return {"kernel": "k_gemm<NT|NN|TN> %s MFMA (forward, data gradient, weight gradient of 3 networks; whole optimiser step incl. "
                      "losses, reductions, clip + Adam in the time)" % ("bf16" if bf else "fp32"), "bound": "mfma", "achieved": ach, "peak": peak,
            "unit": "TFLOP/s", "frac": ach / peak, "avg_launch_ms": ms_step, "us_per_optimiser_step": ms_step * 1e3,
            "algorithmic_flops_per_step": flops_step, "executed_flops_per_step": executed,
            "achieved_on_executed_flops": executed / (ms_step * 1e-3) / 1e12,
            "flops_note": "achieved / frac use SURVEY 8(d)'s accounting (update = 3 x forward = 6 flops per weight and sample, as every round "
                          "before); the step does not compute the layer-0 data gradient, achieved_on_executed_flops is the rate on the flops it runs",
            "optimiser_steps_per_epoch": nsteps, "traffic": None}


def large_minibatch_variant(train, env, n, horizon, args, mbs=None):
    """mbs: minibatch_size of the variant; None = one minibatch of n x horizon samples per mini-epoch (BASELINE.md's labelled row);
    2048 = the schedule DESIGN.md section 17 recommends for training on this engine (the shipped 4 does not learn to lift here)"""
    import copy
    import torch
    from seqdex_amd.a2c_agent import A2CAgent
    tr = copy.deepcopy({k: v for k, v in train["params"].items() if k != "config"})
    pc = {k: v for k, v in train["params"]["config"].items() if k not in ("vec_env", "env_info")}
    pc = copy.deepcopy(pc)
    whole = mbs is None
    mbs = n * horizon if whole else int(mbs)
    pc["minibatch_size"] = mbs
    pc["central_value_config"]["minibatch_size"] = mbs
    pc["mixed_precision"] = bool(args.mixed_precision)
    pc.update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22, multi_gpu=False)
    tr["config"] = pc
    agent = A2CAgent("run_large_minibatch", tr)
    for _ in range(args.warmup):
        agent.train_epoch()
    torch.cuda.synchronize()
    t0 = time.time()
    play_t = upd_t = 0.0
    for _ in range(args.steps):
        r = agent.train_epoch()
        play_t += r[1]; upd_t += r[2]
    torch.cuda.synchronize()
    dt = time.time() - t0
    return {"label": "NOT the shipped schedule: minibatch_size %d instead of 4 (%s)%s"
                     % (mbs, "BASELINE.md protocol row; the insert policy ships 4096" if whole else
                        "the schedule that learns to lift on this engine: DESIGN.md section 17, profiles/r6_grasp_seeds_mb2048.json",
                        "; bf16 trunk GEMMs, fp32 master weights / accumulation (configs[4])" if args.mixed_precision else ""),
            "minibatch_size": mbs, "dtype": "bf16 (trunk GEMM operands) / f32" if args.mixed_precision else "f32", "value": n * horizon * args.steps / dt, "unit": "env-steps/s", "ms_per_step": dt / args.steps * 1e3,
            "update_ms_per_epoch": upd_t / args.steps * 1e3, "rollout_ms_per_epoch": play_t / args.steps * 1e3,
        "update_path": ("multi-rank: hipGraph of [forward/backward -> RCCL all-gather of the rank-MB factors -> rebuild + clip + Adam]" if agent.multi_gpu
                        else agent.ppo.update_impl()),
            "update_impl": agent.ppo.update_impl(), "roofline": gemm_roofline(agent, n, horizon, upd_t / args.steps * 1e3)}


def main():
    args = parse()
    # stdout carries ONE line, the JSON: whatever a library prints through C stdio (RCCL's version banner when a communicator is set up
    # or torn down: it used to land behind the JSON line of a multi-rank run) goes to stderr with everything else
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import yaml
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = None
    force_multi = os.environ.get("SDX_FORCE_MULTI_RANK") == "1"     # multi-rank update path at world size 1 (validation aid)
    if force_multi and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
    if world > 1 or force_multi:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL on ROCm

    from seqdex_amd.a2c_agent import A2CAgent
    from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
    from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython

    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/allegro_hand_block_assembly_grasp_sim.yaml")))
    train = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/lego/ppo_continuous_grasp.yaml")))
    n = args.num_envs
    cfg["env"]["numEnvs"] = n
    if args.minibatch:
        train["params"]["config"]["minibatch_size"] = args.minibatch
        train["params"]["config"]["central_value_config"]["minibatch_size"] = args.minibatch
    train["params"]["config"]["mixed_precision"] = bool(args.mixed_precision)
    seed = 22 + rank
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=local_rank, headless=True, seed=seed, piles_per_type=args.piles_per_type)
    env = RLgamesVecTaskPython(task, "cuda:%d" % local_rank)
    pc = train["params"]["config"]
    pc.update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22, multi_gpu=world > 1 or force_multi)
    agent = A2CAgent("run", train["params"])
    horizon = agent.horizon_length

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.pretrain_epochs):
        agent.train_epoch()
    for _ in range(args.warmup):
        agent.train_epoch()
    barrier()
    t0 = time.time()
    step_t = play_t = upd_t = 0.0
    for _ in range(args.steps):
        r = agent.train_epoch()
        step_t += r[0]; play_t += r[1]; upd_t += r[2]
    barrier()
    dt = time.time() - t0
    tmax = torch.tensor([dt], device="cuda")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    frames = n * horizon * args.steps * world
    value = frames / dt

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    # ---- roofline of the dominant rollout kernel (k_physics): HIP events on the stream the kernel is launched on
    sim = task.sim
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        sim.simulate()
    e0.record()
    for _ in range(reps):
        sim.simulate()
    e1.record()
    torch.cuda.synchronize()
    phys_ms = e0.elapsed_time(e1) / reps
    phys_bytes = BYTES_PER_ENV_STEP * n
    ptraf, pctr = pmc_traffic("k_physics", n)
    cstats = [int(x) for x in sim.CONTACT_STATS.cpu().tolist()]
    nc_mean = float(sim.NCONTACTS.float().mean().item())
    # the warm-started solver keeps (key, 3 impulses) = 16 B per contact in HBM between solves: read + written once per substep
    warm = float(sim._desc.warm_start) > 0
    cache_bytes = int(2 * sim._desc.substeps * 16 * nc_mean * n) if warm else 0
    roof_phys = {"kernel": "k_physics<%d> (%d threads per env, two workgroups per CU)" % (int(sim.lib.sdxk_physics_threads()), int(sim.lib.sdxk_physics_threads())),
                 "bound": "hbm", "achieved": phys_bytes / (phys_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "avg_launch_ms": phys_ms, "algorithmic_bytes_per_launch": phys_bytes,
                 "warm_start_cache_bytes_per_launch": cache_bytes,
                 "bound_actual": "the working set never leaves LDS, so the HBM roofline is nominal.  Round 6 (DESIGN.md section 4a, profiles/r6_kphysics_*): "
                                 "the launch's 512 workgroup slots are about 97 % busy (every env is stamped in the profiling build: sum of env cycles / 512 "
                                 "slots = 1.17 M cycles against a launch of 0.578 ms at about 2.05 GHz), so the average env's 586 k cycles are what counts; "
                                 "inside the solver loop the CU is LDS-throughput-bound (four more row loads per contact: +5.7 % of the kernel), elsewhere it "
                                 "waits: barrier-separated LDS-resident phases, serial FK / factorisation on one wave (the broadphase tests now run beside them)",
                 "contacts_per_env_mean": nc_mean, "contacts_per_env_max": int(sim.NCONTACTS.max().item()),
                 "contact_capacity_per_env": 1536, "contacts_per_env_max_since_create": cstats[0],
                 "env_substeps_over_capacity_since_create": cstats[1], "env_substeps_rebuilt_without_speculative_contacts": cstats[2],
                 "env_substeps_pair_list_overflow": cstats[3],
                 "contact_results_valid": cstats[1] == 0 and cstats[3] == 0,   # a lost contact or an untested pair invalidates the physics of the run
                 "traffic": ptraf, "traffic_counters": pctr}
    roof_phys["frac"] = roof_phys["achieved"] / HBM_PEAK_GBS
    # ---- roofline of the update phase.  Algorithmic bytes: one optimiser step touches 5 x 4 B per parameter
    # (w, g, m, v in, w' out) for all three networks; the persistent kernel runs all optimiser steps of the epoch in ONE launch and
    # keeps w, m, v in VGPRs, so its real HBM traffic is far below that figure
    p_ac, p_cv = agent.ppo.param_count(0), agent.ppo.param_count(1)
    nsteps = agent.mini_epochs_num * (n * horizon // agent.minibatch_size)
    upd_bytes_step = 5 * 4 * (p_ac + p_cv)
    impl = agent.ppo.update_impl() if not agent.multi_gpu else "explicit"
    if impl == "persistent":
        for _ in range(2):      # HIP events on the stream the kernel is launched on (torch's current stream)
            e0.record()
            agent.ppo.update()
            e1.record()
            torch.cuda.synchronize()
        upd_launch_ms = e0.elapsed_time(e1)
        kname, launches = "k_update_persistent (%d optimiser steps, 3 networks, per launch)" % nsteps, 1
    else:
        upd_launch_ms = upd_t / args.steps * 1e3
        kname = ("k_layer/k_head/k_back/k_ctrl hipGraph (per epoch: %d optimiser steps, 3 networks)" % nsteps if impl == "graph" else
                 "k_update_persistent<SINGLE> (forward/backward of one minibatch) + RCCL all-gather of the rank-MB factors + k_grad_all_w/k_sqnorm2/k_adam2 "
                 "(per epoch: %d optimiser steps)" % nsteps)
        launches = nsteps
    upd_ms_step = upd_launch_ms / nsteps
    utraf, uctr = pmc_traffic("k_update_persistent", n) if impl == "persistent" and args.minibatch is None else (None, None)
    if impl == "gemm":
        roof_upd = gemm_roofline(agent, n, horizon, upd_launch_ms)
    else:
      roof_upd = {"kernel": kname, "bound": "hbm", "achieved": upd_bytes_step / (upd_ms_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "avg_launch_ms": upd_launch_ms, "us_per_optimiser_step": upd_ms_step * 1e3,
                "algorithmic_bytes_per_launch": upd_bytes_step * nsteps,
                "traffic": utraf, "traffic_counters": uctr}
      roof_upd["frac"] = roof_upd["achieved"] / HBM_PEAK_GBS
      if impl == "persistent":
          # the bound the kernel actually sits at: a chain of CU-to-CU exchange edges per optimiser step plus the
          # arithmetic that cannot leave the chain; weights and moments live in VGPRs, so HBM carries ~5 % of the algorithmic bytes
          pc_ = PMC.get("k_update_persistent_phase_clock", {})
          fl_ = PMC.get("k_update_persistent_exchange_floor", {})
          edge_floor = sum(fl_.get("edge_floor_us", {}).values()) if fl_ else None
          floor = (edge_floor + pc_["on_chain_us"]) if (edge_floor and pc_.get("on_chain_us")) else None
          roof_upd["bound_actual"] = {"kind": "exchange-latency", "exchange_edges_per_optimiser_step": pc_.get("edges", 5),
                                      "us_wait_on_edges_per_step": pc_.get("edge_wait_us"), "us_on_chain_arithmetic_per_step": pc_.get("on_chain_us"),
                                      "us_shadow_work_per_step": pc_.get("shadow_us"),
                                      "us_per_optimiser_step_measured": upd_ms_step * 1e3, "quoted_from": pc_.get("source"),
                                      # the denominator: the tagged-word all-gather measured alone, per edge, + the on-chain phases
                                      "edge_floor_us": fl_.get("edge_floor_us"), "floor_us_per_step": floor,
                                      "frac_of_floor": (floor / (upd_ms_step * 1e3)) if floor else None, "floor_quoted_from": fl_.get("source"),
                                      "floor_note": fl_.get("note")}
    dominant = roof_upd if upd_t > step_t else roof_phys
    out = {
        "metric": "env-steps/sec BlockAssemblyGraspSim num_envs=%d/GPU" % n, "value": value, "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 trunk GEMMs / f32" if (args.mixed_precision and agent.minibatch_size > 8) else "f32", "data": "synthetic",
        "config": {"workload": "BlockAssemblyGraspSim num_envs=%d per GPU, fp32, PPO MLP policy [1024,512,256], horizon 8, "
                               "minibatch_size %d, mini_epochs 5, central value (configs[1])" % (n, agent.minibatch_size),
                   "step": "one rl_games epoch = 8 env steps x num_envs + full PPO update", "global_envs": n * world,
                   "piles": "synthetic settled piles, %d per brick-type group" % args.piles_per_type,
                   "policy": "random init, seed 22+rank" + (", then %d untimed training epochs" % args.pretrain_epochs if args.pretrain_epochs else "")},
        "fps_step": n * horizon * args.steps / step_t, "fps_step_and_inference": n * horizon * args.steps / play_t,
        "fps_total_rank0": n * horizon * args.steps / (play_t + upd_t),
        "update_ms_per_epoch": upd_t / args.steps * 1e3, "rollout_ms_per_epoch": play_t / args.steps * 1e3,
        "update_path": ("multi-rank: hipGraph of [forward/backward -> RCCL all-gather of the rank-MB factors -> rebuild + clip + Adam]" if agent.multi_gpu
                        else agent.ppo.update_impl()),
        "roofline": {"bound": dominant["bound"], "achieved": dominant["achieved"], "peak": dominant["peak"],
                     "unit": dominant["unit"], "frac": dominant["frac"], "traffic": dominant["traffic"], "kernel": dominant["kernel"],
                     "bound_actual": dominant.get("bound_actual"),
                     "traffic_is": "quoted from the committed rocprofv3 PMC passes (see traffic_counters.quoted_from), not measured in this run"
                                   if dominant["traffic"] is not None else "not available for this configuration"},
        "roofline_physics": roof_phys, "roofline_update": roof_upd,
    }
    if world == 1 and not force_multi and args.minibatch is None and not args.no_large_minibatch:
        try:    # BASELINE.md's labelled variant: the same epoch with ONE minibatch of n x horizon samples per mini-epoch
            out["large_minibatch_variant"] = large_minibatch_variant(train, env, n, horizon, args)
        except Exception as ex:
            out["large_minibatch_variant"] = {"value": None, "error": str(ex)}
        try:    # the same epoch at 2 048-row minibatches (20 optimiser steps per epoch at 1 024 envs): the schedule users are told to train with
            out["large_minibatch_2048"] = large_minibatch_variant(train, env, n, horizon, args, mbs=2048)
        except Exception as ex:
            out["large_minibatch_2048"] = {"value": None, "error": str(ex)}
    if not args.no_cpu_baseline:
        try:
            root = sim.ROOT.view(n, 142, 13).cpu().numpy().copy()
            dof = sim.DOF.view(n, 23, 2).cpu().numpy().copy()
            tg = sim.TARGETS.cpu().numpy().copy()
            out["cpu_baseline"] = cpu_baseline(args, sim._desc, root, dof, tg, horizon, agent.minibatch_size, agent.mini_epochs_num)
        except Exception as ex:   # the checker is optional for the measurement itself
            out["cpu_baseline"] = {"value": None, "error": str(ex)}
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
