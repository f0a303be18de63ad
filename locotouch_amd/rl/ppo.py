"""PPO with clipped surrogate / clipped value loss and the adaptive-KL learning-rate schedule.

Behavioural twin of the reference's `PPO` (loco_rl/loco_rl/algorithms/ppo.py:19-385) for the feed-forward
ActorCritic path the LocoTouch agents use (agents/rsl_rl_ppo_cfg.py:11-30: clip 0.2, gamma 0.99, lam 0.95, lr 1e-3
adaptive, desired KL 0.01, entropy 0.01, value coef 1.0, grad-norm 1.0, 5 epochs x 4 minibatches); the RND and
symmetry branches of the reference are never enabled by those configs and are out of scope.

Data parallelism (new): gradients live in one flat fp32 bucket that is all-reduced (mean) between
`backward()` and the gradient clip, and the KL estimate is all-reduced before the learning-rate decision.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .dist import Dist
from .modules import ActorCritic
from .storage import RolloutStorage


class PPO:
    def __init__(self, actor_critic: ActorCritic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998,
                 lam=0.95, value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01, device="cpu",
                 normalize_advantage_per_mini_batch=False, dist: Dist | None = None, **unused):
        self.device = device
        self.dist = dist or Dist()
        self.actor_critic = actor_critic.to(device)
        on_gpu = torch.device(device).type == "cuda"
        self.fused_loss = on_gpu and bool(unused.get("fused_loss", True))  # csrc/lt_ppo.hip: the loss chain and its backward in one launch
        self.packed_forward = on_gpu and bool(unused.get("packed_forward", True))  # csrc/lt_mlp.hip: both networks' training forward in one launch
        self.direct_update = on_gpu and bool(unused.get("direct_update", True))  # no autograd graph, no host read between minibatch steps (_direct_update)
        self._pair = None
        if on_gpu and bool(unused.get("tuned_gemms", True)):
            from . import tuned_gemms

            tuned_gemms.enable()  # library-GEMM algorithm selection recorded for this stack (rl/tuned_gemms.py); no run-time tuning
        self.optimizer = torch.optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.storage: RolloutStorage | None = None
        self.learning_rate = learning_rate
        self.schedule, self.desired_kl = schedule, desired_kl
        self.clip_param, self.gamma, self.lam = clip_param, gamma, lam
        self.num_learning_epochs, self.num_mini_batches = num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.max_grad_norm = value_loss_coef, entropy_coef, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.normalize_advantage_per_mini_batch = normalize_advantage_per_mini_batch
        self._t = {}
        self._flat_grad: torch.Tensor | None = None
        # clip_grad_norm_ + Adam.step() in two launches on flat buffers (rl/flat_adam.py); the tensors the module and the optimizer
        # hold become views of those buffers, checkpoints keep the reference's layout
        self._flat_adam = None
        use_flat_adam = on_gpu and bool(unused.get("fused_adam", True))
        if self.dist.world_size > 1:
            if not use_flat_adam:
                self._make_flat_grad_bucket()
            self.broadcast_parameters()
        if use_flat_adam:
            from .flat_adam import FlatAdam

            self._flat_adam = FlatAdam(self.optimizer)
            self._flat_grad = self._flat_adam.flat_g  # also the all-reduce bucket of a multi-rank job

    # ---- data-parallel plumbing ------------------------------------------------------------------
    def _make_flat_grad_bucket(self) -> None:
        params = [p for p in self.actor_critic.parameters() if p.requires_grad]
        total = sum(p.numel() for p in params)
        self._flat_grad = torch.zeros(total, device=params[0].device, dtype=params[0].dtype)
        off = 0
        for p in params:
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)  # gradients accumulate straight into the bucket
            off += p.numel()

    def broadcast_parameters(self) -> None:
        for t in list(self.actor_critic.parameters()) + list(self.actor_critic.buffers()):
            self.dist.broadcast_(t.data, src=0)

    # ---- rollout -----------------------------------------------------------------------------------
    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape,
                     obs_dtype: torch.dtype = torch.float32) -> None:
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape[0], critic_obs_shape[0],
                                      action_shape[0], self.device, obs_dtype=obs_dtype)

    def test_mode(self) -> None:
        self.actor_critic.eval()

    def train_mode(self) -> None:
        self.actor_critic.train()

    def act(self, obs: torch.Tensor, critic_obs: torch.Tensor) -> torch.Tensor:
        ac, t = self.actor_critic, self._t
        if getattr(ac, "is_recurrent", False):  # the state the memories hold BEFORE this step (ppo.py:130-131)
            t["hidden"] = tuple(None if h is None else (tuple(x.clone() for x in h) if isinstance(h, tuple) else h.clone())
                                for h in ac.get_hidden_states())
        t["actions"] = ac.act(obs).detach()
        t["values"] = ac.evaluate(critic_obs).detach()
        t["log_prob"] = ac.get_actions_log_prob(t["actions"]).detach()
        t["mu"] = ac.action_mean.detach()
        t["sigma"] = ac.action_std.detach()
        t["obs"], t["critic_obs"] = obs, critic_obs  # recorded before env.step() overwrites the env's buffers
        return t["actions"]

    def process_env_step(self, rewards: torch.Tensor, dones: torch.Tensor, infos: dict) -> None:
        t = self._t
        rew = rewards.clone()
        if "time_outs" in infos:  # bootstrap the value through time-limit terminations (ppo.py:162-165)
            rew += self.gamma * torch.squeeze(t["values"] * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add(t["obs"], t["critic_obs"], t["actions"], rew, dones, t["values"], t["log_prob"], t["mu"], t["sigma"],
                         hidden_states=t.get("hidden"))
        self._t = {}
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs: torch.Tensor) -> None:
        last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam,
                                     normalize_advantage=not self.normalize_advantage_per_mini_batch, dist=self.dist)

    # ---- update ------------------------------------------------------------------------------------
    def _fused_loss_ok(self, b) -> bool:
        ac = self.actor_critic
        st = self.storage
        return (self.fused_loss and st.observations.is_cuda and type(ac) is ActorCritic and getattr(ac, "noise_std_type", "scalar") == "scalar"
                and st.actions.shape[-1] <= 16)

    def _packed_pair(self):
        """PackedPair of the policy's two stacks (None: shapes the MLP kernel does not cover, or `packed_forward=False`)."""
        if not self.packed_forward:
            return None
        if self._pair is None:
            from .mlp import PackedPair, describe

            ac = self.actor_critic
            self._pair = False
            if describe(ac.actor) is not None and describe(ac.critic) is not None:
                try:
                    self._pair = PackedPair(ac.actor, ac.critic)
                except ValueError:
                    self._pair = False
        return self._pair or None

    def _fused_update(self):
        """The update of the plain ActorCritic on the GPU: per minibatch step two row gathers (obs, critic obs), the two MLPs,
        ONE loss launch that reads the seven small per-row tensors of the rollout storage through the minibatch index
        (csrc/lt_ppo.hip), backward, two launches of clip + Adam; one host read per step (the KL of the learning-rate rule,
        ppo.py:273-281), the statistics once at the end.  Same arithmetic as the op chain below (tests/test_hip_ppo_graph.py)."""
        from .fused_loss import fused_ppo_loss

        ac, st = self.actor_critic, self.storage
        f = lambda t: t.flatten(0, 1)  # noqa: E731
        obs, cobs = f(st.observations), f(st.privileged_observations)
        small = [f(t).contiguous() for t in (st.actions, st.actions_log_prob, st.advantages, st.returns, st.values, st.mu, st.sigma)]
        stats = torch.zeros(3, device=obs.device)
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        pair = self._packed_pair()
        if pair is not None and self._flat_adam is not None and self.direct_update and ac.std.requires_grad:
            return self._direct_update(pair)
        if pair is not None:
            pair.arm_domain_check()  # the first minibatch of the update reports the largest layer input (LT_MLP_INPUT_CLAMP)
        for idx in st.mini_batch_indices(self.num_mini_batches, self.num_learning_epochs):
            # both networks' forward in one launch of the MFMA MLP kernel where their shape allows (rl/mlp.py PackedPair)
            o, co = obs[idx], cobs[idx]
            if o.dtype != torch.float32:  # bf16 observation storage (BASELINE config 5): the update computes in f32
                o, co = o.float(), co.float()
            mu, value = pair(o, co) if pair is not None else (ac.actor(o), ac.critic(co))
            loss, surrogate_loss, value_loss, ent, kl_mean = fused_ppo_loss(
                mu, ac.std, value, *small, self.clip_param, self.value_loss_coef, self.entropy_coef,
                self.use_clipped_value_loss, idx=idx)
            # the 1-float KL all-reduce starts here and its host read waits until the backward pass is enqueued: the collective's
            # latency (and the host round trip of the learning-rate rule, ppo.py:273-281) hides under the backward GEMMs; the
            # decision still precedes this step's optimizer update, as in the reference
            kl_handle = self.dist.all_reduce_mean_begin(kl_mean.detach().clone()) if adaptive else None
            self._zero_grad()
            loss.backward()
            if adaptive:
                self._apply_kl(self.dist.all_reduce_mean_end(kl_handle), reduced=True)
            self._optim_step()
            stats = stats + torch.stack((value_loss, surrogate_loss, ent.detach()))
        n = self.num_learning_epochs * self.num_mini_batches
        sv, ss, se = (stats / n).tolist()
        if pair is not None and pair.domain_violated():
            self._warn_mlp_domain()
        st.clear()
        return sv, ss, se, None, None

    def _direct_update(self, pair, split_buckets: bool = True):
        """The update of the plain ActorCritic on the GPU with NO host read between its minibatch steps and no autograd graph: per step
        two row gathers, the packed forward (rl/mlp.py), ONE loss launch that also writes d loss / d (mu, value, std) (csrc/lt_ppo.hip),
        the adaptive-KL learning-rate rule as a one-lane launch on a device scalar (`lt_ppo_lr_rule`, ppo.py:273-281), the two
        stacks' backward chains writing their gradients straight into the flat bucket (rl/mlp.py `backward_chain`), [gradient
        all-reduce], clip + Adam with the learning rate read from the device (`lt_adam_clip_step_dev`).  Statistics and the learning
        rate come back in ONE read at the end.  Arithmetic equal to `_fused_update` (tests/test_hip_ppo_graph.py); the host used to
        wait for the KL of every step - 20 stalls per iteration, ~350 us of idle GPU each."""
        import ctypes

        from .. import _abi

        lib = _abi.load()
        vp = ctypes.c_void_p
        ac, st, fa = self.actor_critic, self.storage, self._flat_adam
        f = lambda t: t.flatten(0, 1)  # noqa: E731
        obs, cobs = f(st.observations), f(st.privileged_observations)
        small = [f(t).contiguous() for t in (st.actions, st.actions_log_prob, st.advantages, st.returns, st.values, st.mu, st.sigma)]
        small = [t.view(-1) if t.shape[-1] == 1 else t for t in small]
        dev = obs.device
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        a_dim = small[0].shape[1]
        state = torch.zeros(4, device=dev)  # [0] learning rate, [1..3] sums of (value loss, surrogate, entropy)
        state[0] = self.learning_rate
        lr_dev, stats = state[:1], state[1:]
        grad_of = fa.grad_views()
        std_c = ac.std.detach()
        pair.arm_domain_check()
        n_steps = 0
        # The reference draws ONE permutation per update and reuses it in every epoch (rollout_storage.py:189): the two observation
        # matrices are gathered through it once - minibatch i of every epoch is rows [i mb, (i + 1) mb) of the permuted copies -
        # instead of 2 x 16 us of row gathers in each of the 20 steps.  (The small per-row tensors are read through the index
        # by the loss kernel itself.)
        critic_at = fa.offset_of(next(iter(ac.critic.parameters())))
        two_buckets = split_buckets and 0 < critic_at < self._flat_grad.numel()
        perm, m = st.mini_batch_permutation(self.num_mini_batches)
        perm_o, perm_co = obs[perm], cobs[perm]
        if perm_o.dtype != torch.float32:  # bf16 observation storage (BASELINE config 5): the update computes in f32
            perm_o, perm_co = perm_o.float(), perm_co.float()
        # the same rows in the MLP kernels' split format (f16 hi | f16 lo): what the first layer's weight gradient multiplies, 20 times
        split_o = (pair.split_rows(perm_o), pair.split_rows(perm_co)) if pair._fused_backward_possible(perm_o[:m], perm_co[:m]) else None
        for step_i in range(self.num_learning_epochs * self.num_mini_batches):
            i = step_i % self.num_mini_batches
            idx = perm[i * m:(i + 1) * m]
            o, co = perm_o[i * m:(i + 1) * m], perm_co[i * m:(i + 1) * m]
            xs = (split_o[0][i * m:(i + 1) * m], split_o[1][i * m:(i + 1) * m]) if split_o is not None else None
            (mu, value), acts = pair.forward_raw(o, co)
            dmu = torch.empty_like(mu)
            dvalue = torch.empty(m, 1, device=dev, dtype=torch.float32)
            acc = torch.empty(24, device=dev, dtype=torch.float32)
            out = torch.empty(24, device=dev, dtype=torch.float32)
            stream = vp(torch.cuda.current_stream(dev).cuda_stream)
            _abi.check(lib.lt_ppo_loss(vp(mu.data_ptr()), vp(std_c.data_ptr()), vp(value.data_ptr()), *[vp(t.data_ptr()) for t in small],
                                       vp(idx.data_ptr()), m, a_dim, float(self.clip_param), float(self.value_loss_coef), float(self.entropy_coef),
                                       int(bool(self.use_clipped_value_loss)), vp(dmu.data_ptr()), vp(dvalue.data_ptr()), vp(acc.data_ptr()),
                                       vp(out.data_ptr()), stream), "lt_ppo_loss")
            kl = None
            if adaptive:
                kl = out[4:5]
                if self.dist.world_size > 1:  # every rank must take the same decision; the collective is stream-ordered
                    kl = self.dist.all_reduce_mean_(kl.clone())
            _abi.check(lib.lt_ppo_lr_rule(vp(kl.data_ptr()) if kl is not None else vp(None), float(self.desired_kl or 0.0), 1e-5, 1e-2, 1.5,
                                          vp(lr_dev.data_ptr()), vp(stats.data_ptr()), vp(out.data_ptr()), vp(grad_of[ac.std].data_ptr()), a_dim, stream),
                       "lt_ppo_lr_rule")
            if self.dist.world_size > 1 and two_buckets:
                # the bucket in two halves (std + actor | critic): the actor's all-reduce runs on RCCL's stream under the critic's
                # three weight-gradient launches (~105 us at 24 576 rows), only the critic's half stays exposed
                handles = []
                pair.backward_raw(o, co, acts, dmu, dvalue, grad_of, xs, after_first=lambda: handles.append(self.dist.all_reduce_mean_begin(self._flat_grad[:critic_at])),
                                  dy_amax=(acc[20:21], acc[21:22]))
                # (a backward pass on the library path never calls back: the whole bucket then)
                handles.append(self.dist.all_reduce_mean_begin(self._flat_grad[critic_at:] if handles else self._flat_grad))
                for h in handles:
                    self.dist.all_reduce_mean_end(h)
            else:
                pair.backward_raw(o, co, acts, dmu, dvalue, grad_of, xs, dy_amax=(acc[20:21], acc[21:22]))
                if self.dist.world_size > 1:
                    self.dist.all_reduce_mean_(self._flat_grad)  # RCCL all-reduce of the policy gradients over xGMI
            fa.step_dev(self.max_grad_norm, lr_dev)
            n_steps += 1
        sat = pair.saturated()
        lr, sv, ss, se, nsat = torch.cat((state, sat if sat is not None else state[:0].new_zeros(1))).tolist()  # the update's only host read
        if nsat > 0:
            import warnings

            sat.zero_()
            warnings.warn(f"lt_mlp_backward_pair: {int(nsat)} workgroup-layers of the gradient chain reached the f16 image's bound "
                          "(a gradient grew by more than 500x through the layers): those rows were saturated; set LT_FUSED_BACKWARD=0 "
                          "to run the chain as f32 library GEMMs")
        self.learning_rate = lr
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        if pair.domain_violated():
            self._warn_mlp_domain()
        st.clear()
        return sv / n_steps, ss / n_steps, se / n_steps, None, None

    def _warn_mlp_domain(self) -> None:
        import warnings

        self.mlp_domain_violations = getattr(self, "mlp_domain_violations", 0) + 1
        warnings.warn("an observation or hidden activation reached the MLP kernel's saturation bound (|x| >= LT_MLP_INPUT_CLAMP, "
                      "include/lt_env.h): the fused forward computes MLP(clamp(x)) there, unlike the fp32 reference; "
                      "construct PPO(..., packed_forward=False) and FusedRollout(..., use_packed_mlp=False) to run the fp32 GEMM path")

    def _adapt_learning_rate(self, mu, sigma, old_mu, old_sigma) -> None:
        with torch.inference_mode():
            kl = torch.sum(torch.log(sigma / old_sigma + 1.0e-5)
                           + (torch.square(old_sigma) + torch.square(old_mu - mu)) / (2.0 * torch.square(sigma)) - 0.5, dim=-1)
            self._apply_kl(torch.mean(kl))

    def _apply_kl(self, kl_mean, reduced: bool = False) -> None:
        with torch.inference_mode():
            if self.dist.world_size > 1 and not reduced:
                kl_mean = self.dist.all_reduce_mean_(kl_mean.clone())
            kl_mean = float(kl_mean)  # host decision, as in the reference (ppo.py:273-281)
        if kl_mean > self.desired_kl * 2.0:
            self.learning_rate = max(1e-5, self.learning_rate / 1.5)
        elif self.desired_kl / 2.0 > kl_mean > 0.0:
            self.learning_rate = min(1e-2, self.learning_rate * 1.5)
        for group in self.optimizer.param_groups:
            group["lr"] = self.learning_rate

    def update(self):
        return self._eager_update()

    def _eager_update(self):
        ac = self.actor_critic
        sum_value = sum_surr = sum_ent = 0.0
        stats = None
        recurrent = getattr(ac, "is_recurrent", False)
        if not recurrent and not self.normalize_advantage_per_mini_batch and self._fused_loss_ok(None):
            return self._fused_update()
        batches = (self.storage.recurrent_mini_batches(self.num_mini_batches, self.num_learning_epochs) if recurrent
                   else self.storage.mini_batches(self.num_mini_batches, self.num_learning_epochs))
        for raw in batches:
            if recurrent:  # padded whole trajectories + the hidden states saved at their first steps (ppo.py:195-196,251-256)
                from .storage import Batch

                b, (hid_a, hid_c), masks = Batch(*raw[:9]), raw[9], raw[10]
            else:
                b, hid_a, hid_c, masks = raw, None, None, None
            adv = b.advantages
            if self.normalize_advantage_per_mini_batch:
                with torch.no_grad():
                    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
            if self._fused_loss_ok(b):
                # GPU: log-prob, KL, surrogate, value loss, entropy and their gradients in one launch (csrc/lt_ppo.hip)
                from .fused_loss import fused_ppo_loss

                loss, surrogate_loss, value_loss, ent, kl_mean = fused_ppo_loss(
                    ac.actor(b.obs), ac.std, ac.critic(b.critic_obs), b.actions, b.log_prob, adv, b.returns, b.values, b.mu, b.sigma,
                    self.clip_param, self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss)
                if self.desired_kl is not None and self.schedule == "adaptive":
                    self._apply_kl(kl_mean)
                if stats is None:
                    stats = torch.zeros(3, device=loss.device)
                self._zero_grad()
                loss.backward()
                self._optim_step()
                stats = stats + torch.stack((value_loss, surrogate_loss, ent.detach()))  # read once, after the last step
                continue
            if recurrent:
                ac.act(b.obs, masks=masks, hidden_states=hid_a)
                log_prob = ac.get_actions_log_prob(b.actions)
                value = ac.evaluate(b.critic_obs, masks=masks, hidden_states=hid_c)
            else:
                ac.act(b.obs)
                log_prob = ac.get_actions_log_prob(b.actions)
                value = ac.evaluate(b.critic_obs)
            mu, sigma, entropy = ac.action_mean, ac.action_std, ac.entropy
            if self.desired_kl is not None and self.schedule == "adaptive":
                self._adapt_learning_rate(mu, sigma, b.mu, b.sigma)
            ratio = torch.exp(log_prob - torch.squeeze(b.log_prob))
            a = torch.squeeze(adv)
            surrogate_loss = torch.max(-a * ratio, -a * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
            if self.use_clipped_value_loss:
                clipped = b.values + (value - b.values).clamp(-self.clip_param, self.clip_param)
                value_loss = torch.max((value - b.returns).pow(2), (clipped - b.returns).pow(2)).mean()
            else:
                value_loss = (b.returns - value).pow(2).mean()
            loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
            self._zero_grad()
            loss.backward()
            self._optim_step()
            sum_value += value_loss.item()
            sum_surr += surrogate_loss.item()
            sum_ent += entropy.mean().item()
        n = self.num_learning_epochs * self.num_mini_batches
        if stats is not None:
            sv, ss, se = stats.tolist()
            sum_value, sum_surr, sum_ent = sum_value + sv, sum_surr + ss, sum_ent + se
        self.storage.clear()
        return sum_value / n, sum_surr / n, sum_ent / n, None, None

    def _zero_grad(self) -> None:
        if self._flat_adam is not None:
            self._flat_adam.zero_grad()
        elif self._flat_grad is not None:
            self._flat_grad.zero_()
        else:
            self.optimizer.zero_grad()

    def _optim_step(self) -> None:
        """All-reduce of the gradients (multi-rank), clip_grad_norm_, Adam (ppo.py:318-319)."""
        if self._flat_adam is not None:
            self._flat_adam.gather_grads()               # one multi-tensor copy into the flat bucket
        if self.dist.world_size > 1:
            self.dist.all_reduce_mean_(self._flat_grad)  # RCCL all-reduce of the policy gradients over xGMI
        if self._flat_adam is not None:
            self._flat_adam.step(self.max_grad_norm, gathered=True)  # clip + Adam in two launches (csrc/lt_ppo.hip)
        else:
            nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
            self.optimizer.step()
