"""`A2CAgent` with the surface rl_games' `rl_games.algos_torch.a2c_continuous.A2CAgent` offers to its callers
(train_rlgames.py:88-94 via Runner; policy_sequencing/policy_seq_runner.py:90-129,193-373 touches the attributes):
train_epoch() returning the 11-tuple, play_steps(), env_reset()/env_step(), update_epoch(), set_eval()/set_train(),
save()/restore(), obs/dones/frame/epoch_num/last_lr/batch_size/...  The loops themselves run in libseqdex_hip.so
(seqdex_amd/ppo.py); this class is orchestration + timing + (for world_size > 1) the RCCL exchange."""
import os
import time

import torch

from .ppo import Ctrl, SdxPPO, make_config


class _Meter:
    """AverageMeter stand-in for game_rewards / game_lengths (PS:372-373): epoch means from device-side sums."""

    def __init__(self):
        self.current_size, self.mean = 0, 0.0

    def update_from(self, total, count):
        if count > 0:
            self.mean, self.current_size = total / count, int(count)

    def get_mean(self):
        return [self.mean]


class _Dataset:
    """rl_games' PPODataset surface (`len(dataset)`, `dataset[i]`, `update_mu_sigma`, `update_values_dict`; App. C): minibatch i is
    rows [i*mb, (i+1)*mb) of the env-major experience buffer - contiguous, unshuffled.  The rows are VIEWS of library-owned memory;
    `mb_index` is what calc_gradients hands to sdxp_backward."""

    def __init__(self, agent):
        self.agent = agent

    def __len__(self):
        return self.agent.batch_size // self.agent.minibatch_size

    def __getitem__(self, i):
        a, t = self.agent, self.agent.ppo.t
        if not 0 <= i < len(self):
            raise IndexError(i)
        mb = a.minibatch_size
        sl = slice(i * mb, (i + 1) * mb)
        flat = lambda x: x.reshape(a.batch_size, -1)[sl]
        return {"mb_index": i, "obs": flat(t["MB_OBS"]), "states": flat(t["MB_STATES"]), "actions": flat(t["MB_ACTIONS"]),
                "mu": flat(t["MB_MUS"]), "sigma": flat(t["MB_SIGMAS"]), "old_logp_actions": t["MB_NEGLOGP"].reshape(-1)[sl],
                "old_values": t["MB_VALUES"].reshape(-1)[sl], "returns": t["RETURNS"][sl], "advantages": t["ADVANTAGES"][sl]}

    def update_mu_sigma(self, mu, sigma):
        pass            # the step kernels write the new mu / sigma rows back themselves (RC:1358)

    def update_values_dict(self, values_dict):
        pass            # the dataset IS the library's experience buffer


class A2CAgent:
    def __init__(self, base_name, params):
        self.base_name = base_name
        self.params = params
        self.config = cfg = params["config"]
        self.vec_env = cfg["vec_env"]                                  # injected like TR:84
        self.env_info = cfg.get("env_info") or self.vec_env.get_env_info()
        self.num_actors = cfg["num_actors"]
        self.num_agents = self.env_info.get("agents", 1)
        self.horizon_length = cfg.get("horizon_length", 8)
        self.mini_epochs_num = cfg.get("mini_epochs", 5)
        self.minibatch_size = cfg.get("minibatch_size", 4)
        self.batch_size = self.horizon_length * self.num_actors * self.num_agents
        assert self.batch_size % self.minibatch_size == 0
        self.max_epochs = cfg.get("max_epochs", 100000)
        self.name = cfg.get("name", "allegro")
        self.ppo_device = self.vec_env.rl_device
        self.rank = int(os.environ.get("RANK", "0")) if cfg.get("multi_gpu", False) else 0
        self.rank_size = int(os.environ.get("WORLD_SIZE", "1")) if cfg.get("multi_gpu", False) else 1
        # SDX_FORCE_MULTI_RANK=1: run the multi-rank update path even at world size 1 (degenerate collectives; used to validate and
        # time that path on a single-GPU box)
        self.multi_gpu = self.rank_size > 1 or (cfg.get("multi_gpu", False) and os.environ.get("SDX_FORCE_MULTI_RANK") == "1")
        seed = int(cfg.get("seed", 22)) + self.rank                    # per-rank seed = seed + rank (App. C)
        obs_dim = int(self.env_info["observation_space"].shape[0])        # 396 GraspSim, 186 Orient
        state_dim = int(self.env_info["state_space"].shape[0]) if "state_space" in self.env_info else None
        self.ppo = SdxPPO(self.num_actors, params=params, device=self.ppo_device, seed=seed, world_size=self.rank_size,
                          obs_dim=obs_dim, state_dim=state_dim)
        self.has_central_value = True
        self.is_tensor_obses = True
        self.frame, self.epoch_num = 0, 0
        self.last_lr = float(cfg.get("learning_rate", 3e-4))
        self.entropy_coef = cfg.get("entropy_coef", 0.0)
        self.game_rewards, self.game_lengths = _Meter(), _Meter()
        self.last_mean_rewards = -100500
        self.save_freq = int(cfg.get("save_frequency", 0) or 0)               # App. C: A2CBase reads these from params.config
        self.save_best_after = int(cfg.get("save_best_after", 100))
        self.train_dir = cfg.get("train_dir", "runs")
        self.experiment_name = cfg.get("full_experiment_name", self.name)
        self.nn_dir = os.path.join(self.train_dir, self.experiment_name, "nn")
        self.schedule_type = cfg.get("schedule_type", "legacy")
        self.bounds_loss_coef = cfg.get("bounds_loss_coef", None)
        self.normalize_input = False                                          # YG: only the central value normalises its input
        self.dataset = _Dataset(self)
        self.update_list = ["actions", "neglogpacs", "values", "mus", "sigmas"]
        self.tensor_list = self.update_list + ["obses", "states", "dones"]
        self._stats_off = {k: getattr(Ctrl, k).offset // 4 for k in ("acc", "last_kl", "ac_lr")}
        self.obs, self.dones = None, None
        self.curr_frames = self.batch_size
        self.experience_buffer = type("ExperienceBuffer", (), {})()
        t = self.ppo.t
        self.experience_buffer.tensor_dict = {   # env-major [N,H,...] views (== swap_and_flatten01 layout)
            "obses": t["MB_OBS"], "states": t["MB_STATES"], "actions": t["MB_ACTIONS"], "mus": t["MB_MUS"],
            "sigmas": t["MB_SIGMAS"], "neglogpacs": t["MB_NEGLOGP"], "values": t["MB_VALUES"],
            "rewards": t["MB_REWARDS"], "dones": t["MB_DONES"]}
        self._ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    for _ in range(self.horizon_length)]
        if self.multi_gpu:
            self._broadcast_parameters()

    # ------------------------------------------------------------------ multi-GPU (RCCL over xGMI)
    def _collectives(self):
        """the three collectives of the multi-rank path.  Backend "nccl" (RCCL) takes the library's device tensors as they are, on the current
        stream.  Any other backend (gloo: the world-size-2 CPU tests, and two ranks SHARING one GPU - RCCL refuses two ranks on one device -
        tests/test_gpu_two_ranks_one_gpu.py) gets them staged through the host: device -> host copy, collective, copy back (synchronous;
        a validation path, never the measured one)."""
        import torch.distributed as dist
        if getattr(self, "_coll", None) is None:
            staged = dist.get_backend() != "nccl"

            def all_reduce(t):
                if staged and t.is_cuda:
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
                else:
                    dist.all_reduce(t)

            def all_gather(out, inp):
                if staged and inp.is_cuda:
                    h = torch.empty(out.shape, dtype=out.dtype)
                    dist.all_gather_into_tensor(h, inp.cpu())
                    out.copy_(h)
                else:
                    dist.all_gather_into_tensor(out, inp)

            def broadcast(t, src):
                if staged and t.is_cuda:
                    h = t.cpu()
                    dist.broadcast(h, src)
                    t.copy_(h)
                else:
                    dist.broadcast(t, src)

            self._coll = type("Collectives", (), {"all_reduce": staticmethod(all_reduce), "all_gather": staticmethod(all_gather),
                                                  "broadcast": staticmethod(broadcast), "host_staged": staged})
        return self._coll

    def _broadcast_parameters(self):
        for k in ("AC_PARAMS", "CV_PARAMS"):
            self._collectives().broadcast(self.ppo.t[k], 0)

    # ------------------------------------------------------------------ rl_games-shaped API
    def set_eval(self):
        pass

    def set_train(self):
        pass

    def init_tensors(self):
        pass

    def update_epoch(self):
        self.epoch_num += 1
        return self.epoch_num

    def env_reset(self):
        return self.vec_env.reset()

    def env_step(self, actions):
        obs, rew, dones, infos = self.vec_env.step(actions)
        return obs, rew.unsqueeze(1), dones, infos

    def get_action_values(self, obs, t=0):
        a = self.ppo.act(t, obs["obs"], obs["states"], self.dones)
        return {"actions": a}

    def get_values(self, obs):
        """RC:1725-1745: the central value of obs["states"], shape [N, 1]"""
        return self.ppo.get_values(obs["states"]).unsqueeze(1)

    def discount_values(self, fdones=None, last_extrinsic_values=None, mb_fdones=None, mb_extrinsic_values=None, mb_rewards=None):
        """PS:331-336 on the library's own experience buffer (the mb_* arguments of rl_games are those buffers; passing different
        tensors is not supported).  Returns the raw advantages in rl_games' [H, N, 1] orientation (a view of SDXP_T_ADVANTAGES)."""
        t = self.ppo.t
        for given, own in ((mb_fdones, t["MB_DONES"]), (mb_extrinsic_values, t["MB_VALUES"]), (mb_rewards, t["MB_REWARDS"])):
            if given is not None and given.data_ptr() != own.data_ptr():
                raise ValueError("discount_values works on the agent's own experience buffer")
        lv = last_extrinsic_values.reshape(-1).contiguous()
        ld = None if fdones is None else (fdones.reshape(-1) != 0).to(torch.int64)
        self.ppo.discount_values(lv, ld)
        return t["ADVANTAGES"].view(self.num_actors, self.horizon_length).t().unsqueeze(2)

    def prepare_dataset(self, batch_dict=None):
        """RC:1621-1683: advantage normalisation; the returns / values / actions of `batch_dict` are the library's buffers already.
        Also opens the update phase (control block reset, central-value input normalisation for the epoch)."""
        self.ppo.prepare_dataset()
        self.ppo.backward(0, -1)

    def train_central_value(self):
        """fused into the actor-critic minibatch loop: the two optimisers never exchange data (RC:1323-1324), so running the
        central-value step of minibatch i in the same launches as the actor-critic step of minibatch i changes no result"""
        return 0.0

    def train_actor_critic(self, input_dict):
        return self.calc_gradients(input_dict)

    def calc_gradients(self, input_dict):
        """RC:1767-1877 for minibatch input_dict["mb_index"] (dataset[i]; minibatches must come in order): forward, losses,
        backward, [all-reduce], clip, Adam, LR rule - for all three networks.  Returns rl_games' train_result tuple
        (a_loss, c_loss, entropy, kl, last_lr, lr_mul, mu, sigma, b_loss) with device scalars (no host sync)."""
        ppo, mb = self.ppo, int(input_dict["mb_index"])
        ppo.backward(0, mb)
        st = ppo.t["STATS"]
        acc = st[self._stats_off["acc"]:self._stats_off["acc"] + 8].clone()    # per-minibatch sums of the HEAD kernel: [1]a [2]c [3]b [4]kl [6]entropy
        if self.multi_gpu and self.rank_size > 1:
            self._collectives().all_reduce(ppo.t["ALL_GRADS"])
            ppo.apply(0, float("-inf"))
        else:
            ppo.apply(0)
        ppo.apply(1)
        n = float(self.minibatch_size)
        lr = st[self._stats_off["ac_lr"]].clone()
        return (acc[1] / n, acc[2] / n, acc[6] / n, acc[4] / n, lr, 1.0, input_dict["mu"], input_dict["sigma"], acc[3] / n)

    def update_lr(self, lr):
        self.ppo.set_state(ac_lr=float(lr))
        self.last_lr = float(lr)

    def play_steps(self, deterministic=False):
        """PS:220-275 / RC:1394-1483: horizon loop; every call below is asynchronous on the current stream.  deterministic: zero noise
        into the Gaussian head, i.e. the mean action (rl_games' player with `deterministic: True`, YG:69)."""
        if self.obs is None:
            self.obs = self.env_reset()
            self.dones = self.vec_env.task.reset_buf
        task = self.vec_env.task
        eps = None
        if deterministic:
            if getattr(self, "_zero_eps", None) is None:
                self._zero_eps = torch.zeros(self.num_actors, self.ppo.cfg.act_dim, device=self.ppo.device)
            eps = self._zero_eps
        for n in range(self.horizon_length):
            a = self.ppo.act(n, self.obs["obs"], self.obs["states"], self.dones, eps)
            self._ev[n][0].record()
            self.obs, rew, self.dones, infos = self.vec_env.step(a)
            self._ev[n][1].record()
            self.ppo.store_rewards(n, task.rew_buf, self.dones)
        self.ppo.finish_rollout(self.obs["states"], self.dones)
        return {"played_frames": self.batch_size}

    def train_epoch(self):
        """rl_games train_epoch (mirrored at PS:193-218, RC:1306-1392).  Returns
        (step_time, play_time, update_time, total_time, a_losses, c_losses, b_losses, entropies, kls, last_lr, lr_mul)."""
        torch.cuda.synchronize()
        play_time_start = time.time()
        self.play_steps()
        torch.cuda.synchronize()
        play_time_end = time.time()
        step_time = sum(a.elapsed_time(b) for a, b in self._ev) * 1e-3
        update_time_start = time.time()
        if self.multi_gpu:
            self._update_multi_gpu()
        else:
            self.ppo.update_checked()      # waits; falls back to the hipGraph path if the persistent kernel could not run
        torch.cuda.synchronize()
        update_time_end = time.time()
        c = self.ppo.ctrl()
        n = max(c.n_mb, 1)
        self.last_lr = c.ac_lr
        self.game_rewards.update_from(c.games_sum_rew, c.games_cnt)
        self.game_lengths.update_from(c.games_sum_len, c.games_cnt)
        tt = lambda v: [torch.tensor(v)]
        return (step_time, play_time_end - play_time_start, update_time_end - update_time_start,
                update_time_end - play_time_start, tt(c.sum_a_loss / n), tt(c.sum_c_loss / n), tt(c.sum_b_loss / n),
                tt(c.sum_entropy / n), tt(c.sum_kl / n), self.last_lr, 1.0)

    def _update_multi_gpu(self):
        """world_size > 1: per optimiser step the flat gradients (8.5 MB actor-critic + 4.9 MB central value, fp32) and
        the scalar KL are summed over ranks with RCCL (torch.distributed backend "nccl") on the current stream; the
        division by world_size, clip_grad_norm_, Adam and the LR rule run in sdxp_apply (rl_games multi-GPU semantics,
        SURVEY.md App. C; PS:308-310)."""
        dist = self._collectives()
        ppo = self.ppo
        nmb = self.batch_size // self.minibatch_size
        if "FACTORS" in ppo.t and hasattr(ppo, "backward_factors") and self.minibatch_size <= 8:
            # preferred: all-gather the rank-MB factors (194 KB per rank) and rebuild the summed gradient locally
            # flat views: the concatenating form of all_gather_into_tensor is the one every backend accepts (gloo rejects [W, F] <- [F])
            fact, fact_all = ppo.t["FACTORS"].view(-1), ppo.t["FACTORS_ALL"].view(-1)

            def steps(k):
                for _ in range(k):
                    ppo.backward_factors(0)          # the minibatch is the one under the DEVICE cursor; the argument is only range-checked
                    dist.all_gather(fact_all, fact)
                    ppo.apply_factors()

            ppo.backward_factors(-1)
            total = self.mini_epochs_num * nmb
            done = 0
            # the optimiser step (one persistent forward/backward launch, the RCCL all-gather, three apply launches) carries no
            # host-side argument that changes from step to step - cursor, exchange tags, Adam counters and the LR all live in the
            # device control block - so a chunk of steps is captured ONCE into a hipGraph (RCCL collectives are capturable) and
            # replayed: per step the host then issues 1/64 of a graph launch instead of four launches and a collective call
            chunk = next((c for c in (64, 32, 16, 8) if total % c == 0 and total >= 2 * c), 0)
            # default: on at world size 1 (where it is measured: 77.5 -> 70.4 us per step); at world size > 1 the capture contains a real
            # RCCL collective, which this build could never run (gpurun boxes have one GPU), so it is opt-in there: SDX_MULTI_RANK_GRAPH=1
            want = os.environ.get("SDX_MULTI_RANK_GRAPH", "1" if self.rank_size == 1 else "0") == "1"
            if chunk and want and not dist.host_staged and getattr(self, "_mr_graph", None) is not False:
                if getattr(self, "_mr_graph", None) is None:
                    steps(chunk)                     # eagerly once: communicator set-up, lazy module loads, function attributes
                    done = chunk
                    try:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            steps(chunk)
                        self._mr_graph, self._mr_chunk = g, chunk
                    except Exception as ex:          # capture refused (old RCCL / torch): stay on eager launches for good
                        print("seqdex_amd: multi-rank step not captured (%s); eager launches" % (str(ex).splitlines()[0] if str(ex) else type(ex).__name__))
                        self._mr_graph = False
                        torch.cuda.synchronize()
                if self._mr_graph:
                    while done + self._mr_chunk <= total:
                        self._mr_graph.replay()
                        done += self._mr_chunk
            steps(total - done)
            try:
                ppo.update_status()
            except Exception:
                # an exchange-word timeout makes the library leave the persistent forward/backward launch for the multi-kernel one;
                # the captured graph still holds the persistent launch, so it must not be replayed again: drop it (None = capture
                # afresh, on the path the handle has switched to, the next time this method runs) and let the caller see the error
                self._mr_graph = None
                raise
            return
        ppo.backward(0, -1)
        if "ALL_GRADS" in ppo.t:
            # one collective per optimiser step: both flat gradients and the KL word share one library-owned buffer
            # (this is also the large-minibatch path: with thousands of samples per step the gradient IS the small object)
            g_all = ppo.t["ALL_GRADS"]
            for _ in range(self.mini_epochs_num):
                for mb in range(nmb):
                    ppo.backward(0, mb)
                    dist.all_reduce(g_all)
                    ppo.apply(0, float("-inf"))
                    ppo.apply(1)
            return
        g_ac, g_cv, kl = ppo.t["AC_GRADS"], ppo.t["CV_GRADS"], ppo.kl_view()
        for _ in range(self.mini_epochs_num):
            for mb in range(nmb):
                ppo.backward(0, mb)
                dist.all_reduce(g_ac)
                dist.all_reduce(g_cv)
                dist.all_reduce(kl)
                ppo.apply(0)
                ppo.apply(1)

    def train(self):
        """rl_games A2CBase.train(): epoch loop + the fps line that IS the metric (PS:136-140)."""
        total_time = 0.0
        while True:
            epoch_num = self.update_epoch()
            step_time, play_time, update_time, sum_time, a_l, c_l, b_l, ent, kls, last_lr, lr_mul = self.train_epoch()
            total_time += sum_time
            curr_frames = self.curr_frames * self.rank_size
            self.frame += curr_frames
            if self.rank == 0:
                fps_step = curr_frames / max(step_time, 1e-6)
                fps_step_inference = curr_frames / play_time
                fps_total = curr_frames / sum_time
                print(f"fps step: {fps_step:.0f} fps step and policy inference: {fps_step_inference:.0f} "
                      f"fps total: {fps_total:.0f} epoch: {epoch_num}/{self.max_epochs}")
            if self.rank == 0 and self.game_rewards.current_size > 0:      # A2CBase.train: periodic + best checkpoints (App. C)
                mean_rewards = self.game_rewards.get_mean()[0]
                name = "%s_ep_%d_rew_%s" % (self.name, epoch_num, mean_rewards)
                if self.save_freq > 0 and epoch_num % self.save_freq == 0:
                    os.makedirs(self.nn_dir, exist_ok=True)
                    self.save(os.path.join(self.nn_dir, "last_" + name))
                if mean_rewards > self.last_mean_rewards and epoch_num >= self.save_best_after:
                    self.last_mean_rewards = mean_rewards
                    os.makedirs(self.nn_dir, exist_ok=True)
                    self.save(os.path.join(self.nn_dir, self.name))
            if epoch_num >= self.max_epochs:
                return self.game_rewards.get_mean()[0], epoch_num

    def play(self, games_num=1, max_steps=150, deterministic=None):
        """--play/--test: rollouts only, no update.  The action is the policy mean when the YAML's player block says
        `deterministic: True` (as shipped, YG:69), else sampled with the learned sigma."""
        if deterministic is None:
            deterministic = bool(self.config.get("player", {}).get("deterministic", True)) if hasattr(self, "config") else True
        n, rew, length = 0, 0.0, 0.0
        while n < max(1, games_num):
            self.play_steps(deterministic)
            c = self.ppo.ctrl()
            n += int(c.games_cnt)
            rew += c.games_sum_rew
            length += c.games_sum_len
            self.game_rewards.update_from(rew, n)            # means over every game finished during this play() call
            self.game_lengths.update_from(length, n)
        torch.cuda.synchronize()
        print("mean episode reward: %.3f  mean episode length: %.1f  (%d games)" % (self.game_rewards.get_mean()[0],
                                                                                   self.game_lengths.get_mean()[0], n))

    # ------------------------------------------------------------------ checkpoints (rl_games .pth-shaped dict)
    def get_full_state_weights(self):
        """rl_games' checkpoint dictionary (A2CBase.get_full_state_weights, 1.5.2 layout as recalled in SURVEY.md App. C): `model` and
        `assymetric_vf_nets` are named state_dicts (seqdex_amd/rlgames_checkpoint.py); the Adam moments are kept flat under `optimizer`
        (rl_games stores torch.optim state there, which only another rl_games process could consume)."""
        from .rlgames_checkpoint import rlgames_from_flat
        t, st = self.ppo.t, self.ppo.get_state()
        cfg = self.ppo.cfg
        obs_cols, state_cols = self._checkpoint_widths()
        model, vf = rlgames_from_flat(t["AC_PARAMS"], t["CV_PARAMS"], cfg.obs_dim, cfg.state_dim, cfg.act_dim, tuple(cfg.units),
                                      t["CV_RMS_MEAN"], t["CV_RMS_VAR"], st["rms_count"], obs_cols=obs_cols, state_cols=state_cols)
        # `optimizer` is torch.optim.Adam's state_dict over model.parameters() (what rl_games' set_full_state_weights hands to
        # optimizer.load_state_dict); the central-value optimiser's moments, which rl_games does not checkpoint, ride along as extra keys
        from .rlgames_checkpoint import torch_adam_from_flat
        opt = torch_adam_from_flat(t["AC_ADAM_M"], t["AC_ADAM_V"], st["ac_t"], st["ac_lr"], cfg.obs_dim, cfg.act_dim, tuple(cfg.units), obs_cols=obs_cols)
        opt.update({"cv_m": t["CV_ADAM_M"].cpu(), "cv_v": t["CV_ADAM_V"].cpu(), "ac_t": st["ac_t"], "cv_t": st["cv_t"], "cv_lr": st["cv_lr"]})
        return {"model": model, "assymetric_vf_nets": vf, "optimizer": opt,
                "epoch": self.epoch_num, "frame": self.frame, "last_mean_rewards": self.last_mean_rewards,
                "last_lr": st["ac_lr"], "env_state": None}

    def _checkpoint_widths(self):
        """real input widths of the two networks: the library pads observation rows to a multiple of 4 (Orient / Search 186 -> 188)
        and keeps 564-wide state rows for every task (InsertSim's one frame is 188 wide); rl_games' first layers have the real
        widths, so checkpoints are written / read with them and the padded columns carry zero weights."""
        cfg = self.ppo.cfg
        obs_cols = cfg.obs_cols if cfg.obs_cols > 0 else cfg.obs_dim
        task = getattr(getattr(self, "vec_env", None), "task", None)
        state_cols = cfg.state_dim
        if task is not None and getattr(task, "stack_obs", 3) == 1:
            state_cols = int(getattr(task, "one_frame_num_states", cfg.state_dim))
        return obs_cols, state_cols

    def save(self, fn):
        torch.save(self.get_full_state_weights(), fn + ".pth")

    def restore(self, fn):
        """loads a checkpoint written by save() or by rl_games itself (`model` / `assymetric_vf_nets` state_dicts; keys matched by
        suffix, shapes checked; first-layer columns the library pads are zero-filled).  Adam moments are restored when they are ours."""
        from .rlgames_checkpoint import flat_from_rlgames
        ck = torch.load(fn, map_location="cpu", weights_only=False)
        t = self.ppo.t
        cfg = self.ppo.cfg
        if "ac_flat" in ck.get("model", {}):                         # round-1 layout of this package
            ac, cv = ck["model"]["ac_flat"], ck["model"]["cv_flat"]
            rms = (ck["running_mean_std"]["mean"], ck["running_mean_std"]["var"], None)
        else:
            obs_cols, state_cols = self._checkpoint_widths()
            try:      # real widths first (what rl_games and save() write), then the library's padded widths (round-1 files)
                ac, cv, rms = flat_from_rlgames(ck["model"], ck["assymetric_vf_nets"], cfg.obs_dim, cfg.state_dim, cfg.act_dim,
                                                tuple(cfg.units), obs_cols=obs_cols, state_cols=state_cols)
            except ValueError:
                ac, cv, rms = flat_from_rlgames(ck["model"], ck["assymetric_vf_nets"], cfg.obs_dim, cfg.state_dim, cfg.act_dim,
                                                tuple(cfg.units))
        t["AC_PARAMS"].copy_(ac); t["CV_PARAMS"].copy_(cv)
        if rms is not None:
            t["CV_RMS_MEAN"].copy_(rms[0]); t["CV_RMS_VAR"].copy_(rms[1])
        opt = ck.get("optimizer", {})
        torch_adam_step = None
        if isinstance(opt, dict) and "ac_m" in opt:                  # flat moments (files of rounds 2-3)
            t["AC_ADAM_M"].copy_(opt["ac_m"]); t["AC_ADAM_V"].copy_(opt["ac_v"])
        elif isinstance(opt, dict) and "state" in opt:               # torch.optim.Adam.state_dict() (rl_games, and save() since round 4)
            from .rlgames_checkpoint import flat_from_torch_adam
            obs_cols, _ = self._checkpoint_widths()
            try:
                mv = flat_from_torch_adam(opt, cfg.obs_dim, cfg.act_dim, tuple(cfg.units), obs_cols=obs_cols)
            except ValueError as ex:             # an optimiser state of another layout: say so, start from fresh moments AND a fresh step counter
                print("restore: optimizer state NOT restored (%s); Adam moments and step counter start from zero" % ex)
                mv = None
                t["AC_ADAM_M"].zero_(); t["AC_ADAM_V"].zero_()
                torch_adam_step = 0
            if mv is not None:
                t["AC_ADAM_M"].copy_(mv[0]); t["AC_ADAM_V"].copy_(mv[1])
                torch_adam_step = mv[2]
        if isinstance(opt, dict) and "cv_m" in opt:
            t["CV_ADAM_M"].copy_(opt["cv_m"]); t["CV_ADAM_V"].copy_(opt["cv_v"])
        # the rest of the optimiser state (rl_games restores all of it): running_mean_std.count, Adam step counters (bias-correction
        # powers follow them), the adaptive learning rate
        state = {}
        if rms is not None and rms[2] is not None and float(rms[2]) > 0:
            state["rms_count"] = float(rms[2])
        if torch_adam_step is not None:
            state["ac_t"] = torch_adam_step
        if isinstance(opt, dict):
            for k in ("ac_t", "cv_t", "cv_lr"):
                if k in opt:
                    state[k] = opt[k]
        if "last_lr" in ck:
            state["ac_lr"] = float(ck["last_lr"])
            self.last_lr = float(ck["last_lr"])
        if state:
            self.ppo.set_state(**state)
        self.epoch_num, self.frame = ck.get("epoch", 0), ck.get("frame", 0)
        self.last_mean_rewards = ck.get("last_mean_rewards", getattr(self, "last_mean_rewards", -100500))
