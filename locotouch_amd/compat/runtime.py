"""Import surface that lets the reference's launch scripts run UNMODIFIED on top of this package.

`install()` registers stand-ins for every third-party module `locotouch/scripts/{train,play}.py` import that is not in the
image - `isaaclab*` (config classes, managers API: locotouch_amd/compat/isaaclab_shim.py), `isaaclab.app.AppLauncher`,
`isaaclab_tasks.utils{,.hydra,.parse_cfg}`, `isaaclab_rl.rsl_rl`, `gymnasium` (registry + make) and `git` - and aliases
`loco_rl` to the PyTorch-ROCm trainer (locotouch_amd/compat/loco_rl).  `gym.make(task, cfg=env_cfg)` then builds the HIP env
(`LocoTouchVecEnv`) for the registered task ids this build implements; the reference's own config classes are imported and
instantiated as they are (they only need the config-class machinery), and the values the scripts override on them
(`scene.num_envs`, `seed`, `sim.device`) are honoured.

    python -m locotouch_amd.compat.run_reference <path-to>/locotouch/scripts/train.py --task Isaac-...-v1 --num_envs 4096 --headless
"""
from __future__ import annotations

import argparse
import importlib
import os
import re
import sys
import types

import torch

from . import isaaclab_shim
from .configclass import MISSING, configclass

SUPPORTED_TASKS = ("Isaac-Locomotion-LocoTouch-v1", "Isaac-RandCylinderTransportTeacher-LocoTouch-v1")
_env_factory = None
_INSTALLED = False


def set_env_factory(fn) -> None:
    """Override how `gym.make` builds the env: fn(task_id, env_cfg) -> VecEnv-protocol object (tests inject a CPU stand-in)."""
    global _env_factory
    _env_factory = fn


def _module(name: str, **attrs) -> types.ModuleType:
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_module(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


# ---------------------------------------------------------------------------------------------------------
# gymnasium: registry + make
# ---------------------------------------------------------------------------------------------------------
class EnvSpec:
    def __init__(self, id, entry_point, kwargs):
        self.id, self.entry_point, self.kwargs = id, entry_point, dict(kwargs or {})


def _install_gym():
    try:
        import gymnasium  # noqa: F401  (a real gymnasium wins if the image has one)
        return
    except ImportError:
        pass
    registry: dict[str, EnvSpec] = {}

    def register(id, entry_point=None, disable_env_checker=False, kwargs=None, **_unused):
        registry[id] = EnvSpec(id, entry_point, kwargs)

    def spec(id):
        return registry[id]

    def make(id, cfg=None, render_mode=None, **kwargs):
        if id not in registry:
            raise KeyError(f"gym.make: no registered env with id {id!r}")
        return make_env(id, cfg)

    class RecordVideo:  # there is no renderer: the wrapper is transparent
        def __new__(cls, env, **kwargs):
            return env

    _module("gymnasium", register=register, make=make, spec=spec, registry=registry, Env=object)
    _module("gymnasium.wrappers", RecordVideo=RecordVideo)


class ManagedEnv:
    """What `gym.make` returns: the HIP env behind the attribute surface the scripts and `RslRlVecEnvWrapper` touch
    (`unwrapped`, `cfg`, `num_envs`, `device`, `step_dt`, `max_episode_length`, `episode_length_buf`, `close`)."""

    def __init__(self, task_id: str, cfg, vec_env, extra_rewards=(), extra_terminations=(), extra_observations=()):
        self.task_id, self.cfg, self.vec = task_id, cfg, vec_env
        # reward terms the fused kernels do not know: evaluated in torch on IsaacLab-layout views after every step and added to
        # the kernel's reward (compat/scene_views.py; the slow path of SURVEY.md §8(b) B3)
        self.extra = None
        if extra_rewards:
            from .scene_views import ExtraTerms

            self.extra = ExtraTerms(vec_env)
            for name, func, weight, params in extra_rewards:
                self.extra.add_reward(name, func, weight, params)
        for name, func, params, *flag in extra_terminations:
            self.add_termination_term(name, func, params, time_out=bool(flag and flag[0]))
        for o in extra_observations:
            self.add_observation_term(**o)

    def add_reward_term(self, name: str, func, weight: float, params: dict | None = None) -> None:
        """Attach a user reward term `func(env, **params) -> (N,)` (reference term signature, mdp/rewards.py:15-20)."""
        if self.extra is None:
            from .scene_views import ExtraTerms

            self.extra = ExtraTerms(self.vec)
        self.extra.add_reward(name, func, weight, params)

    def add_termination_term(self, name: str, func, params: dict | None = None, time_out: bool = False) -> None:
        """Attach a user termination term `func(env, **params) -> bool (N,)` (mdp/terminations.py:10-23); it takes effect one env
        step after it fires (include/lt_env.h, LT_T_USER; `time_out`: LT_T_USER_TIME_OUT - the env ends by time-out)."""
        if self.extra is None:
            from .scene_views import ExtraTerms

            self.extra = ExtraTerms(self.vec)
        self.extra.add_termination(name, func, params, time_out=time_out)

    def add_observation_term(self, group: str, name: str, func, params: dict | None = None, history_length: int = 0, scale=None, clip=None,
                             noise=None) -> None:
        """Attach a user observation term `func(env, **params) -> (N, d)` (IsaacLab ObservationTermCfg semantics: noise if the group
        corrupts, clip, scale, history flattened oldest -> newest) behind the fused terms of `group` ("policy" / "critic"): the rows
        the trainer sees grow by d x history columns; the fused rollout (which reads the kernel's own rows) is bypassed."""
        if self.extra is None:
            from .scene_views import ExtraTerms

            self.extra = ExtraTerms(self.vec)
        self.extra.add_observation(group, name, func, params, history_length, scale, clip, noise)
        self._obs_dims = None

    def _extra_dims(self) -> dict:
        if getattr(self, "_obs_dims", None) is None:
            self._obs_dims = self.extra.observation_dims() if (self.extra is not None and self.extra.observations) else {}
        return self._obs_dims

    @property
    def num_obs(self) -> int:
        return int(self.vec.num_obs) + int(self._extra_dims().get("policy", 0))

    @property
    def num_privileged_obs(self) -> int:
        return int(getattr(self.vec, "num_privileged_obs", self.vec.num_obs)) + int(self._extra_dims().get("critic", 0))

    def _with_user_observations(self, obs, extras, dones):
        if self.extra is None or not self.extra.observations:
            return obs, extras
        more = self.extra.observe(dones)
        groups = dict(extras.get("observations", {}))
        if "policy" in more:
            obs = torch.cat((obs, more["policy"]), dim=1)
            groups["policy"] = obs
        if "critic" in more and "critic" in groups:
            groups["critic"] = torch.cat((groups["critic"], more["critic"]), dim=1)
        extras = dict(extras, observations=groups)
        return obs, extras

    def get_observations(self):
        obs, extras = self.vec.get_observations()
        if self.extra is not None and self.extra.observations:
            # the rows of the state the env is in: re-emit the user terms' current history (no new frame is pushed)
            more = {}
            for o in self.extra.observations:
                if o["buf"] is None:
                    return self._with_user_observations(obs, extras, None)
                more.setdefault(o["group"], []).append(o["buf"].reshape(self.vec.num_envs, -1))
            groups = dict(extras.get("observations", {}))
            if "policy" in more:
                obs = torch.cat([obs] + more["policy"], dim=1)
                groups["policy"] = obs
            if "critic" in more and "critic" in groups:
                groups["critic"] = torch.cat([groups["critic"]] + more["critic"], dim=1)
            extras = dict(extras, observations=groups)
        return obs, extras

    def reset(self):
        obs, extras = self.vec.reset()
        return self._with_user_observations(obs, extras, None)

    @property
    def scene(self):
        if self.extra is None:
            from .scene_views import ExtraTerms

            self.extra = ExtraTerms(self.vec)
        return self.extra.env.scene

    def step(self, actions):
        if self.extra:
            self.extra.pre_step()
        obs, rew, dones, extras = self.vec.step(actions)
        if self.extra:
            if self.extra.terms:
                rew = self.extra.apply(rew, dones)
            if self.extra.terminations:
                self.extra.request_terminations(dones)
            self.extra.post_step()
            obs, extras = self._with_user_observations(obs, extras, dones)
        return obs, rew, dones, extras

    @property
    def unwrapped(self):
        return self

    def __getattr__(self, name):  # everything else is the VecEnv's
        return getattr(self.vec, name)

    @property
    def episode_length_buf(self):
        return self.vec.episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        self.vec.episode_length_buf = value

    def close(self):
        pass


def translate_env_cfg(task_id: str, cfg):
    """(lt_cfg, object_sizes | None) for the env-cfg tree the launch script built (and possibly edited): every manager term
    is routed into lt_cfg by cfg_translate.translate, which raises on anything the fused kernels cannot honour.  Per-env
    cylinders of a MultiAssetSpawnerCfg travel as an explicit size table.  `cfg is None`: the registration's preset."""
    from .. import _abi
    from . import cfg_translate

    if cfg is None:
        return _abi.preset_cfg(task_id), None
    lt = cfg_translate.translate(cfg, collect_unknown_rewards=True)
    sizes = None
    spawn = getattr(getattr(cfg.scene, "object", None), "spawn", None)
    if spawn is not None and type(spawn).__name__ == "MultiAssetSpawnerCfg":
        import torch

        # MultiAssetSpawnerCfg(random_choice=False) deals the asset list to the envs in order, cyclically [DEP]; the -Play-
        # registration builds 20 cylinders and then raises num_envs to 50 (rand_cylinder..._PLAY + smaller_scene_for_playing)
        assets = spawn.assets_cfg
        if getattr(spawn, "random_choice", False):
            raise cfg_translate.UnsupportedCfg("MultiAssetSpawnerCfg(random_choice=True) is not implemented")
        sizes = torch.tensor([[float(assets[i % len(assets)].radius), float(assets[i % len(assets)].height)] for i in range(lt.num_envs)],
                             dtype=torch.float32)
        lt.obj_size_explicit = 1
    return lt, sizes


def make_env(task_id: str, cfg):
    if _env_factory is not None:
        vec = _env_factory(task_id, cfg)
        return ManagedEnv(task_id, cfg, vec, extra_rewards=getattr(getattr(vec, "cfg", None), "extra_reward_terms", ()),
                          extra_terminations=getattr(getattr(vec, "cfg", None), "extra_termination_terms", ()),
                          extra_observations=getattr(getattr(vec, "cfg", None), "extra_observation_terms", ()))
    from ..env import LocoTouchVecEnv

    lt, sizes = translate_env_cfg(task_id, cfg)
    device = getattr(getattr(cfg, "sim", None), "device", None) or "cuda:0"
    return ManagedEnv(task_id, cfg, LocoTouchVecEnv(task_id, device=device, cfg=lt, object_sizes=sizes),
                      extra_rewards=getattr(lt, "extra_reward_terms", ()), extra_terminations=getattr(lt, "extra_termination_terms", ()),
                      extra_observations=getattr(lt, "extra_observation_terms", ()))


# ---------------------------------------------------------------------------------------------------------
# isaaclab.app, isaaclab_tasks.utils, isaaclab_rl.rsl_rl
# ---------------------------------------------------------------------------------------------------------
class _App:
    """`simulation_app`: there is no window to close, so the play loops (`while simulation_app.is_running()`, play.py:139,
    distillation.py:181) end after LT_APP_MAX_STEPS polls when that variable is set (unset = run until interrupted)."""

    def __init__(self):
        self._polls = 0
        self._limit = int(os.environ["LT_APP_MAX_STEPS"]) if os.environ.get("LT_APP_MAX_STEPS") else None

    def is_running(self) -> bool:
        self._polls += 1
        return self._limit is None or self._polls <= self._limit

    def close(self) -> None:
        pass


class AppLauncher:
    """No simulator application to launch; keeps the CLI surface (`--headless --device --enable_cameras ...`)."""

    def __init__(self, launcher_args=None, **kwargs):
        self.app = _App()

    @staticmethod
    def add_app_launcher_args(parser: argparse.ArgumentParser) -> None:
        g = parser.add_argument_group("app_launcher", description="Arguments of the (absent) simulator application.")
        g.add_argument("--headless", action="store_true", default=False)
        g.add_argument("--livestream", type=int, default=-1)
        g.add_argument("--enable_cameras", action="store_true", default=False)
        g.add_argument("--device", type=str, default=None)
        g.add_argument("--verbose", action="store_true", default=False)
        g.add_argument("--experience", type=str, default="")
        g.add_argument("--kit_args", type=str, default="")


def load_cfg_from_registry(task_name: str, entry_point_key: str):
    import gymnasium as gym

    entry = gym.spec(task_name).kwargs.get(entry_point_key)
    if entry is None:
        raise ValueError(f"no entry point {entry_point_key!r} registered for {task_name!r}")
    if isinstance(entry, str):
        if entry.endswith(".yaml"):
            import yaml

            with open(entry) as f:
                return yaml.safe_load(f)
        mod, attr = entry.split(":")
        entry = getattr(importlib.import_module(mod), attr)
    cfg = entry() if callable(entry) else entry
    # LT_CFG_OVERRIDES='{"distillation_cfg_entry_point": {"bc_data_steps": 2000, "num_iterations": 2}}': dotted-path edits per
    # entry point, for the cfgs the reference scripts offer no command-line override for (distill.py has no Hydra hook)
    ov = os.environ.get("LT_CFG_OVERRIDES")
    if ov:
        import json

        for dotted, value in json.loads(ov).get(entry_point_key, {}).items():
            obj, parts = cfg, dotted.split(".")
            for p_ in parts[:-1]:
                obj = getattr(obj, p_)
            if not hasattr(obj, parts[-1]):
                raise AttributeError(f"LT_CFG_OVERRIDES: {entry_point_key} has no field {dotted!r}")
            setattr(obj, parts[-1], value)
    return cfg


def parse_env_cfg(task_name: str, device: str = "cuda:0", num_envs: int | None = None, use_fabric: bool | None = None):
    cfg = load_cfg_from_registry(task_name, "env_cfg_entry_point")
    cfg.sim.device = device
    if num_envs is not None:
        cfg.scene.num_envs = num_envs
    return cfg


def _apply_override(cfg, dotted: str, value: str) -> None:
    import ast

    obj = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        obj = getattr(obj, p)
    try:
        val = ast.literal_eval(value)
    except Exception:
        val = value
    setattr(obj, parts[-1], val)


def hydra_task_config(task_name: str, agent_cfg_entry_point: str):
    """Decorator of the scripts' main(env_cfg, agent_cfg): loads both configs from the registry and applies
    `env.<path>=<value>` / `agent.<path>=<value>` overrides left in sys.argv (the part of Hydra the scripts rely on)."""

    def decorator(func):
        def wrapper(*args, **kwargs):
            env_cfg = load_cfg_from_registry(task_name, "env_cfg_entry_point")
            agent_cfg = load_cfg_from_registry(task_name, agent_cfg_entry_point)
            for tok in sys.argv[1:]:
                m = re.match(r"^(env|agent)\.([\w.]+)=(.*)$", tok)
                if m:
                    _apply_override(env_cfg if m.group(1) == "env" else agent_cfg, m.group(2), m.group(3))
            return func(env_cfg, agent_cfg, *args, **kwargs)

        return wrapper

    return decorator


def get_checkpoint_path(log_path: str, run_dir: str = ".*", checkpoint: str = ".*", other_dirs=None, sort_alpha: bool = True) -> str:
    """Latest run directory matching `run_dir` (regex) under `log_path`, latest file matching `checkpoint` inside it."""
    try:
        runs = [e.path for e in os.scandir(log_path) if e.is_dir() and re.match(run_dir, e.name)]
    except FileNotFoundError:
        runs = []
    if not runs:
        raise ValueError(f"no runs in {log_path!r} match {run_dir!r}")
    runs.sort() if sort_alpha else runs.sort(key=os.path.getmtime)
    run_path = os.path.join(runs[-1], *(other_dirs or []))
    files = [f for f in os.listdir(run_path) if re.match(checkpoint, f)]
    if not files:
        raise ValueError(f"no checkpoints in {run_path!r} match {checkpoint!r}")
    files.sort(key=lambda m: f"{m:0>15}")
    return os.path.join(run_path, files[-1])


def _rsl_rl_cfgs():
    @configclass
    class RslRlPpoActorCriticCfg:
        class_name: str = "ActorCritic"
        init_noise_std: float = MISSING
        actor_hidden_dims: list = MISSING
        critic_hidden_dims: list = MISSING
        activation: str = MISSING

    @configclass
    class RslRlPpoAlgorithmCfg:
        class_name: str = "PPO"
        value_loss_coef: float = MISSING
        use_clipped_value_loss: bool = MISSING
        clip_param: float = MISSING
        entropy_coef: float = MISSING
        num_learning_epochs: int = MISSING
        num_mini_batches: int = MISSING
        learning_rate: float = MISSING
        schedule: str = MISSING
        gamma: float = MISSING
        lam: float = MISSING
        desired_kl: float = MISSING
        max_grad_norm: float = MISSING

    @configclass
    class RslRlOnPolicyRunnerCfg:
        seed: int = 42
        device: str = "cuda:0"
        num_steps_per_env: int = MISSING
        max_iterations: int = MISSING
        empirical_normalization: bool = MISSING
        policy: object = MISSING
        algorithm: object = MISSING
        save_interval: int = MISSING
        experiment_name: str = MISSING
        run_name: str = ""
        logger: str = "tensorboard"
        neptune_project: str = "isaaclab"
        wandb_project: str = "isaaclab"
        resume: bool = False
        load_run: str = ".*"
        load_checkpoint: str = "model_.*.pt"

    return RslRlPpoActorCriticCfg, RslRlPpoAlgorithmCfg, RslRlOnPolicyRunnerCfg


class ObsGroups(tuple):
    """What `get_observations()` returns.  The reference is written against TWO generations of the IsaacLab wrapper (quirk Q1):
    its runner / play.py / ReplayBuffer.evaluate unpack `obs, extras = env.get_observations()` (on_policy_runner.py:158,
    play.py:122, replay_buffer.py:137) while its distillation code indexes a group mapping - `env_obs["policy"]`,
    `env_obs.items()` (distillation.py:51-54, replay_buffer.py:37-39).  This object is the 2-tuple AND the mapping."""

    def __new__(cls, obs, extras):
        return super().__new__(cls, (obs, extras))

    @property
    def groups(self) -> dict:
        return tuple.__getitem__(self, 1)["observations"]

    def __getitem__(self, k):
        return self.groups[k] if isinstance(k, str) else tuple.__getitem__(self, k)

    def __contains__(self, k):
        return k in self.groups if isinstance(k, str) else tuple.__contains__(self, k)

    def keys(self):
        return self.groups.keys()

    def values(self):
        return self.groups.values()

    def items(self):
        return self.groups.items()


class ObsTensor(torch.Tensor):
    """First element of `step()`: the policy rows (what the runner / play.py feed to the policy) that can also be indexed by
    group name and iterated with `.items()` (what the distillation code does with `next_obs`, replay_buffer.py:52-54).
    Shares the policy rows' storage; every torch operation on it returns a plain Tensor."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def wrap(obs: torch.Tensor, groups: dict) -> "ObsTensor":
        t = torch.Tensor._make_subclass(ObsTensor, obs)
        t._groups = groups
        return t

    def __getitem__(self, k):
        return self._groups[k] if isinstance(k, str) else torch.Tensor.__getitem__(self.as_subclass(torch.Tensor), k)

    def keys(self):
        return self._groups.keys()

    def items(self):
        return self._groups.items()


class RslRlVecEnvWrapper:
    """`RslRlVecEnvWrapper(env)`: the VecEnv protocol of loco_rl/loco_rl/env/vec_env.py:12-101 over what gym.make returned.
    The HIP env already speaks that protocol, so this only forwards (and performs the reset the IsaacLab wrapper does)."""

    def __init__(self, env, clip_actions=None):
        self.env = env
        self.clip_actions = clip_actions
        vec = env.unwrapped
        self.num_envs, self.num_actions, self.device = vec.num_envs, vec.num_actions, vec.device
        self.max_episode_length = vec.max_episode_length
        self.num_obs = vec.num_obs  # (ManagedEnv: the kernel's rows + the user observation terms' columns)
        self.num_privileged_obs = getattr(vec, "num_privileged_obs", vec.num_obs)

    @property
    def cfg(self):
        return self.env.cfg

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def episode_length_buf(self):
        return self.env.unwrapped.episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        self.env.unwrapped.episode_length_buf = value

    def get_observations(self):
        obs, extras = self.env.unwrapped.get_observations()
        return ObsGroups(obs, extras)

    def reset(self):
        obs, extras = self.env.unwrapped.reset()
        return ObsGroups(obs, extras)

    def step(self, actions: torch.Tensor):
        if self.clip_actions is not None:
            actions = torch.clamp(actions, -self.clip_actions, self.clip_actions)
        obs, rew, dones, extras = self.env.unwrapped.step(actions)
        return ObsTensor.wrap(obs, extras["observations"]), rew, dones, extras

    def fused_target(self):
        """The HIP env itself when nothing between the trainer and it changes a step - no action clipping here, no user terms on
        the ManagedEnv - so that `OnPolicyRunner` can run its fused rollout (hipGraph, two launches per step) through the import
        surface of the unmodified launch scripts as well; None otherwise (the trainer then steps through `step`)."""
        if self.clip_actions is not None:
            return None
        managed = self.env.unwrapped
        if getattr(managed, "extra", None):
            return None
        return getattr(managed, "vec", None)

    def seed(self, seed: int = -1) -> int:
        return seed

    def close(self):
        return self.env.close()

    def __getattr__(self, name):
        return getattr(self.env.unwrapped, name)


class _PolicyExport(torch.nn.Module):
    """Deployment module of `export_policy_as_jit` / `_onnx` [DEP isaaclab_rl.rsl_rl.exporter]: normaliser -> actor."""

    def __init__(self, actor_critic, normalizer=None):
        super().__init__()
        import copy

        if getattr(actor_critic, "is_recurrent", False):
            raise NotImplementedError("recurrent actor export is not implemented (no registered LocoTouch task trains one)")
        layers = []  # plain nn.Sequential of plain nn.Linear (rl/linear.py's training-time nodes are not scriptable)
        for m in copy.deepcopy(actor_critic.actor).cpu():
            if isinstance(m, torch.nn.Linear) and type(m) is not torch.nn.Linear:
                lin = torch.nn.Linear(m.in_features, m.out_features, bias=m.bias is not None)
                lin.load_state_dict(m.state_dict())
                m = lin
            layers.append(m)
        self.actor = torch.nn.Sequential(*layers)
        self.normalizer = copy.deepcopy(normalizer).cpu() if normalizer is not None else torch.nn.Identity()

    def forward(self, x):
        return self.actor(self.normalizer(x))


def export_policy_as_jit(actor_critic, normalizer=None, path: str = ".", filename: str = "policy.pt") -> str:
    """TorchScript file `path/filename` of obs -> mean action (what the robot-side runtime loads)."""
    os.makedirs(path, exist_ok=True)
    mod = _PolicyExport(actor_critic, normalizer).eval()
    out = os.path.join(path, filename)
    torch.jit.script(mod).save(out)
    return out


def export_policy_as_onnx(actor_critic, path: str = ".", normalizer=None, filename: str = "policy.onnx", verbose: bool = False) -> str:
    """ONNX file (inputs `obs`, outputs `actions`, opset 11, as the IsaacLab exporter writes it).  torch's exporter needs the
    `onnx` package, which this image does not ship: the ImportError says so instead of writing nothing."""
    try:
        import onnx  # noqa: F401
    except ImportError as e:
        raise ImportError("export_policy_as_onnx needs the `onnx` package (torch.onnx.export serialises through it); "
                          "export_policy_as_jit writes the same network as TorchScript") from e
    os.makedirs(path, exist_ok=True)
    mod = _PolicyExport(actor_critic, normalizer).eval()
    out = os.path.join(path, filename)
    n_in = mod.actor[0].in_features
    torch.onnx.export(mod, torch.zeros(1, n_in), out, export_params=True, opset_version=11, verbose=verbose, input_names=["obs"],
                      output_names=["actions"], dynamic_axes={})
    return out


_STOCK_MDP = ["generated_commands", "base_ang_vel", "base_lin_vel", "projected_gravity", "joint_pos_rel", "joint_vel_rel", "last_action",
              "is_alive", "time_out", "bad_orientation", "root_height_below_minimum", "illegal_contact", "randomize_rigid_body_mass",
              "randomize_rigid_body_material", "reset_root_state_uniform", "reset_joints_by_offset", "reset_joints_by_scale",
              "push_by_setting_velocity", "height_scan", "joint_pos_out_of_limit", "is_terminated", "action_rate_l2", "joint_acc_l2",
              "joint_torques_l2", "flat_orientation_l2", "lin_vel_z_l2", "ang_vel_xy_l2", "undesired_contacts", "reset_scene_to_default"]


def _stock_mdp() -> dict:
    """Names the reference configs reference as `mdp.<name>` (locomotion_base_env_cfg.py).  In this build the terms are part of
    the fused step kernel; the functions exist so that the configs import, and say so when called."""

    def make(name):
        def term(env, *args, **kwargs):
            raise RuntimeError(f"isaaclab.envs.mdp.{name} is evaluated inside lt_step_kernel; the Python term is a config-time placeholder")

        term.__name__ = name
        return term

    return {n: make(n) for n in _STOCK_MDP}


def _permissive(modname: str) -> None:
    """Unknown attribute of a stand-in module: `...Cfg` -> a permissive config class, anything else -> a placeholder term.
    (The reference registers Go2W / other-robot tasks from the same package tree; importing their configs must not fail just
    because this build does not implement those tasks.)"""
    mod = _module(modname)

    def getattr_(name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.endswith("Cfg"):
            value = isaaclab_shim._anycfg(name)
        elif name[:1].isupper():
            value = type(name, (isaaclab_shim._Placeholder,), {})
        else:
            def value(*args, _n=name, **kwargs):
                raise RuntimeError(f"{modname}.{_n} is a config-time placeholder in this build")
            value.__name__ = name
        setattr(mod, name, value)
        return value

    mod.__getattr__ = getattr_


def install(env_factory=None) -> None:
    global _INSTALLED
    if env_factory is not None:
        set_env_factory(env_factory)
    if _INSTALLED:
        return
    _INSTALLED = True
    _install_gym()
    isaaclab_shim.install(extra_mdp=_stock_mdp(), import_subpackages=True)
    for m in ("isaaclab.envs.mdp", "isaaclab.envs.mdp.actions", "isaaclab.envs.mdp.commands", "isaaclab.sim", "isaaclab.sensors",
              "isaaclab.assets", "isaaclab.actuators", "isaaclab.terrains", "isaaclab.utils.noise", "isaaclab.managers", "isaaclab.envs",
              "isaaclab.scene", "isaaclab.sim.spawners", "isaaclab.sim.schemas", "isaaclab.utils.assets", "isaaclab.markers",
              "isaaclab.markers.config", "isaaclab.terrains.config.rough"):
        _permissive(m)
    _module("isaaclab.app", AppLauncher=AppLauncher)
    tu = _module("isaaclab_tasks.utils")
    tu.get_checkpoint_path, tu.parse_env_cfg, tu.load_cfg_from_registry = get_checkpoint_path, parse_env_cfg, load_cfg_from_registry
    _module("isaaclab_tasks.utils.hydra", hydra_task_config=hydra_task_config)
    _module("isaaclab_tasks.utils.parse_cfg", load_cfg_from_registry=load_cfg_from_registry, parse_env_cfg=parse_env_cfg,
            get_checkpoint_path=get_checkpoint_path)
    ac, alg, runner = _rsl_rl_cfgs()
    _module("isaaclab_rl")
    _module("isaaclab_rl.rsl_rl", RslRlVecEnvWrapper=RslRlVecEnvWrapper, RslRlOnPolicyRunnerCfg=runner, RslRlPpoActorCriticCfg=ac,
            RslRlPpoAlgorithmCfg=alg, export_policy_as_jit=export_policy_as_jit, export_policy_as_onnx=export_policy_as_onnx)
    try:  # `from torch.utils.tensorboard import SummaryWriter` (distillation.py:85): the real one needs the tensorboard package
        importlib.import_module("torch.utils.tensorboard")
    except ImportError:
        from ..rl import tb_writer

        _module("torch.utils.tensorboard", SummaryWriter=tb_writer.SummaryWriter)
    # loco_rl -> this package's trainer
    from . import loco_rl as alias

    sys.modules["loco_rl"] = alias
    for sub in ("runners", "models", "algorithms", "modules", "storage", "env", "utils"):
        sys.modules[f"loco_rl.{sub}"] = importlib.import_module(f"{alias.__name__}.{sub}")
