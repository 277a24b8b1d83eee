"""MI355X drop-in for the reference's ``TorchVectorMnkEnv``.

Same constructor, attributes, methods and return conventions as
``/root/reference/src/env/torch_vector_mnk_env.py:7-119``, so
``TorchSelfPlayWrapper``, ``PPOAgent.learn``, ``validate_gpu``, ``MatchRunner`` and
``play.py`` of the reference run on it unchanged.  Underneath, the boards live
bit-packed on the GPU (``planes u64[2][W][N]`` + ``meta u32[N]``, see
``include/mnk_hip.h``) and every operation is one launch of a hand-written HIP
kernel on torch's current stream -- no ATen op chain, no host synchronisation.

``boards`` / ``current_player`` / ``move_counts`` are *views* of the packed state
that behave like the reference's dense tensors (readable, index-assignable, usable
in torch expressions); they are meant for tests, ``play.py`` and debugging, the hot
path never touches them.

There is no CPU mode: ``device`` must be a HIP device and ``libmnk_hip.so`` must be
built, otherwise construction raises.
"""
from typing import Dict, Optional, Tuple

import torch

import mnk_hip
from .constants import PLAYER_BLACK, PLAYER_WHITE  # noqa: F401  (re-exported like the reference module)


class _DenseView:
    """A dense-tensor face on packed device state.

    Reads materialise the dense tensor (one unpack kernel); writes go
    read-modify-write through ``_commit`` (one pack kernel).  Supports what the
    reference's callers do with these attributes: ``view[idx]``, ``view[idx] = v``,
    ``view[idx] ^= 1``, comparisons, arithmetic, ``.sum()``, ``.cpu()``, ``.clone()``,
    in-place methods such as ``.zero_()``, and use as an argument of ``torch.*``.
    """

    def __init__(self, env):
        object.__setattr__(self, "_env", env)

    # subclasses implement these two
    def _dense(self) -> torch.Tensor:
        raise NotImplementedError

    def _commit(self, dense: torch.Tensor) -> None:
        raise NotImplementedError

    def __getitem__(self, key):
        return self._dense()[_unwrap(key)]

    def __setitem__(self, key, value):
        d = self._dense()
        d[_unwrap(key)] = _unwrap(value)
        self._commit(d)

    def __getattr__(self, name):
        d = self._dense()
        attr = getattr(d, name)
        if callable(attr) and name.endswith("_") and not name.startswith("_"):
            def inplace(*args, **kwargs):
                attr(*[_unwrap(a) for a in args], **{k: _unwrap(v) for k, v in kwargs.items()})
                self._commit(d)
                return self
            return inplace
        return attr

    def __setattr__(self, name, value):
        raise AttributeError("views of the packed env state have no settable attributes")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        args = [_unwrap(a) for a in args]
        kwargs = {k: _unwrap(v) for k, v in (kwargs or {}).items()}
        return func(*args, **kwargs)

    def __array__(self, dtype=None):
        a = self._dense().cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def __len__(self):
        return self._env.num_envs

    def __iter__(self):
        return iter(self._dense())

    def __repr__(self):
        return f"{type(self).__name__}({self._dense()!r})"

    def __bool__(self):
        return bool(self._dense())


def _unwrap(x):
    if isinstance(x, _DenseView):
        return x._dense()
    if isinstance(x, tuple):
        return tuple(_unwrap(v) for v in x)
    return x


def _binary(name):
    def op(self, other):
        return getattr(self._dense(), name)(_unwrap(other))
    return op


def _inplace(name):
    def op(self, other):
        d = self._dense()
        getattr(d, name)(_unwrap(other))
        self._commit(d)
        return self
    return op


for _n in ("eq", "ne", "lt", "le", "gt", "ge", "add", "sub", "mul", "and", "or", "xor", "radd", "rsub", "rmul",
           "rand", "ror", "rxor", "floordiv", "truediv", "mod"):
    setattr(_DenseView, f"__{_n}__", _binary(f"__{_n}__"))
for _n in ("iadd", "isub", "imul", "iand", "ior", "ixor"):
    setattr(_DenseView, f"__{_n}__", _inplace(f"__{_n}__"))
_DenseView.__invert__ = lambda self: ~self._dense()
_DenseView.__neg__ = lambda self: -self._dense()
_DenseView.__hash__ = None


class _BoardsView(_DenseView):
    """``env.boards``: f32 (N, 2, m, n), plane 0 black / 1 white (reference env:17)."""

    def _dense(self):
        return self._env._unpack_boards()

    def _commit(self, dense):
        self._env._pack_boards(dense)


class _SideView(_DenseView):
    """``env.current_player``: i64 (N,) (reference env:18)."""

    def _dense(self):
        self._env.check_errors()
        return (self._env._meta & 1).to(torch.int64)

    def _commit(self, dense):
        env = self._env
        env._meta.copy_((env._meta & ~1) | (dense.to(env._meta.device, torch.int32) & 1))


class _CountView(_DenseView):
    """``env.move_counts``: i64 (N,) (reference env:19)."""

    def _dense(self):
        self._env.check_errors()
        return (self._env._meta >> 1).to(torch.int64)

    def _commit(self, dense):
        env = self._env
        env._meta.copy_((env._meta & 1) | (dense.to(env._meta.device, torch.int32) << 1))


class TorchVectorMnkEnv:
    """Batched m,n,k-game on bit-packed boards; surface of reference env:7-119."""

    def __init__(self, m: int, n: int, k: int, num_envs: int, device: str = "cuda", strict: bool = False,
                 obs_dtype: torch.dtype = torch.float32):
        assert m >= k and n >= k, f"Board ({m}x{n}) is too small for k={k}"  # reference env:9
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(
                f"TorchVectorMnkEnv(device={device!r}): this is the MI355X HIP implementation and needs a GPU "
                "device; it has no CPU mode (the CPU restatement under oracle/ is test infrastructure)."
            )
        mnk_hip.load()
        if not mnk_hip.geometry_supported(m, n, k):
            raise ValueError(f"board {m}x{n} (k={k}) is outside the packed layout's range "
                             "(2 <= n <= 61, m*(n+1) <= 1024 bits)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        elif dev.index != torch.cuda.current_device():
            # kernels are launched on torch's current stream of this device; HIP wants that device current
            raise ValueError(f"TorchVectorMnkEnv(device={device!r}): make it the current device first "
                             f"(torch.cuda.set_device({dev.index})); one process per GPU is the intended layout")
        self.m, self.n, self.k = int(m), int(n), int(k)
        self.num_envs = int(num_envs)
        self.device = device
        self._dev = dev
        self.max_moves = self.m * self.n
        self.strict = bool(strict)
        # dtype of the observations this env (and a wrapper around it) hands out: float32 is the reference's; bfloat16 /
        # uint8 are opt-in narrow forms of the same 0 / 1 cells (include/mnk_hip.h MNK_OBS_*), half / a quarter of the bytes
        mnk_hip.obs_dtype_code(obs_dtype)
        self.obs_dtype = obs_dtype
        self.words = mnk_hip.state_words(self.m, self.n)
        # packed state; int64 / int32 tensors carry the u64 / u32 bit patterns
        self._planes = torch.zeros((2, self.words, self.num_envs), dtype=torch.int64, device=dev)
        self._meta = torch.zeros(self.num_envs, dtype=torch.int32, device=dev)
        self._err = torch.zeros(2, dtype=torch.int32, device=dev)
        self.env_indices = torch.arange(self.num_envs, device=dev)  # reference env:23

    # ------------------------------------------------------------------ dense views
    @property
    def boards(self):
        return _BoardsView(self)

    @boards.setter
    def boards(self, value):
        self._pack_boards(_unwrap(value))

    @property
    def current_player(self):
        return _SideView(self)

    @current_player.setter
    def current_player(self, value):
        _SideView(self)._commit(torch.as_tensor(_unwrap(value), device=self._dev))

    @property
    def move_counts(self):
        return _CountView(self)

    @move_counts.setter
    def move_counts(self, value):
        _CountView(self)._commit(torch.as_tensor(_unwrap(value), device=self._dev))

    def _stream(self):
        return mnk_hip.stream_ptr(self._dev)

    def _unpack_boards(self) -> torch.Tensor:
        self.check_errors()
        out = torch.empty((self.num_envs, 2, self.m, self.n), dtype=torch.float32, device=self._dev)
        if self.num_envs:
            mnk_hip.call("mnk_unpack_boards", mnk_hip.ptr(self._planes), mnk_hip.ptr(out), self.num_envs, self.m,
                         self.n, self._stream())
        return out

    def _pack_boards(self, dense: torch.Tensor) -> None:
        dense = torch.as_tensor(dense, dtype=torch.float32, device=self._dev).contiguous()
        assert dense.shape == (self.num_envs, 2, self.m, self.n), "boards must be (num_envs, 2, m, n)"
        if self.num_envs:
            mnk_hip.call("mnk_pack_boards", mnk_hip.ptr(dense), mnk_hip.ptr(self._planes), self.num_envs, self.m,
                         self.n, self._stream())

    # ------------------------------------------------------------------ errors
    def check_errors(self) -> None:
        """Surfaces a device-side error recorded by an earlier kernel (synchronises)."""
        code, where = self._err.tolist()
        if code == mnk_hip.ERR_NONE:
            return
        self._err.zero_()
        if code == mnk_hip.ERR_ILLEGAL_MOVE:
            # message of reference env:102-104
            raise ValueError(f"Illegal Move: Env {where} tried to play in occupied cell.")
        raise IndexError(f"index {where} is out of bounds (actions must lie in [-{self.max_moves}, {self.max_moves}), "
                         f"env indices in [-{self.num_envs}, {self.num_envs}))")

    # ------------------------------------------------------------------ reference surface
    def reset(self, env_indices: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """reference env:34-44"""
        if self.num_envs == 0:  # nothing to reset; zero-element tensors have no storage to point at
            return self.observe()
        if env_indices is None:
            mnk_hip.call("mnk_reset_all", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs,
                         self.words, self._stream())
        else:
            idx = torch.as_tensor(_unwrap(env_indices), device=self._dev)
            if idx.dtype == torch.bool:
                mask = idx.to(torch.uint8).contiguous()
                assert mask.shape == (self.num_envs,)
                mnk_hip.call("mnk_reset_mask", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs,
                             self.words, mnk_hip.ptr(mask), self._stream())
            else:
                idx = idx.to(torch.int64).reshape(-1).contiguous()
                mnk_hip.call("mnk_reset_idx", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs,
                             self.words, mnk_hip.ptr(idx), idx.numel(), mnk_hip.ptr(self._err), self._stream())
        return self.observe()

    def observe(self) -> Dict[str, torch.Tensor]:
        """reference env:46-53: fresh tensors, absolute planes, row-major legal mask"""
        obs = torch.empty((self.num_envs, 2, self.m, self.n), dtype=self.obs_dtype, device=self._dev)
        mask = torch.empty((self.num_envs, self.max_moves), dtype=torch.bool, device=self._dev)
        self.observe_into(obs, mask)
        return {"observation": obs, "action_mask": mask}

    def legal_mask(self) -> torch.Tensor:
        """bool (N, m*n): the ``action_mask`` of ``observe()`` without the observation."""
        mask = torch.empty((self.num_envs, self.max_moves), dtype=torch.bool, device=self._dev)
        self.observe_into(None, mask)
        return mask

    def step(self, actions: torch.Tensor) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor]:
        """reference env:55-58"""
        actions = self._as_actions(actions, self.num_envs)
        obs = torch.empty((self.num_envs, 2, self.m, self.n), dtype=self.obs_dtype, device=self._dev)
        mask = torch.empty((self.num_envs, self.max_moves), dtype=torch.bool, device=self._dev)
        rewards = torch.empty(self.num_envs, dtype=torch.float32, device=self._dev)
        dones = torch.empty(self.num_envs, dtype=torch.bool, device=self._dev)
        self.step_into(actions, rewards, dones, mask, obs)
        return {"observation": obs, "action_mask": mask}, rewards, dones

    def step_subset(self, actions: torch.Tensor, active_indices: torch.Tensor):
        """reference env:60-84: full-size rewards / dones, full observation"""
        idx = torch.as_tensor(_unwrap(active_indices), device=self._dev).to(torch.int64).reshape(-1).contiguous()
        actions = self._as_actions(actions, idx.numel())
        obs = torch.empty((self.num_envs, 2, self.m, self.n), dtype=self.obs_dtype, device=self._dev)
        mask = torch.empty((self.num_envs, self.max_moves), dtype=torch.bool, device=self._dev)
        rewards = torch.empty(self.num_envs, dtype=torch.float32, device=self._dev)
        dones = torch.empty(self.num_envs, dtype=torch.bool, device=self._dev)
        if self.num_envs:
            mnk_hip.call("mnk_step", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs, self.m,
                         self.n, self.k, mnk_hip.ptr(actions), mnk_hip.ptr(idx), idx.numel(), mnk_hip.ptr(rewards),
                         mnk_hip.ptr(dones), mnk_hip.ptr(mask), mnk_hip.ptr(obs), mnk_hip.obs_code(obs),
                         mnk_hip.ptr(self._err), self._flags(), self._stream())
        if self.strict:
            self.check_errors()
        return {"observation": obs, "action_mask": mask}, rewards, dones

    # ------------------------------------------------------------------ buffer-reusing forms (graph-capturable)
    def step_into(self, actions, rewards, dones, mask=None, obs=None, autoreset: bool = False) -> None:
        """``step`` into caller-owned buffers; ``mask`` / ``obs`` may be None to skip them (``obs``: float32,
        bfloat16 or uint8 -- the kernel writes the dtype it is given).
        One kernel launch, nothing allocated, nothing synchronised (unless ``strict``).
        ``autoreset``: finished games restart in the same launch and ``mask`` / ``obs`` show the fresh boards --
        ``step(a); reset(nonzero(done)); observe()`` of the raw loop as one kernel."""
        if self.num_envs:
            mnk_hip.call("mnk_step", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs, self.m,
                         self.n, self.k, mnk_hip.ptr(actions), None, self.num_envs, mnk_hip.ptr(rewards),
                         mnk_hip.ptr(dones), mnk_hip.ptr(mask), mnk_hip.ptr(obs), mnk_hip.obs_code(obs),
                         mnk_hip.ptr(self._err), self._flags() | (mnk_hip.STEP_AUTORESET if autoreset else 0),
                         self._stream())
        if self.strict:
            self.check_errors()

    def step_random_into(self, rewards, dones, mask=None, obs=None, actions=None, *, seed: int, step: int, env_id0: int = 0,
                         stream_id: int = mnk_hip.STREAM_MOVE, step_dev=None, autoreset: bool = True) -> None:
        """One ply of uniform random play in ONE launch (``mnk_step_random``): every env draws its own legal move
        from Philox(seed, env_id0 + i, step) -- the draw of ``sample_legal_into`` -- plays it, restarts if the game
        ended (``autoreset``), and the legal mask / observation of the position that follows are written:
        ``RandomPolicy.act -> env.step -> env.reset(nonzero(done)) -> env.observe()`` of the raw loop (SURVEY.md
        Appendix A; policy.py:18-29, env:34-84).  ``actions`` (optional, int64 (N,)) receives the moves played."""
        if self.num_envs:
            mnk_hip.call("mnk_step_random", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs, self.m,
                         self.n, self.k, seed, step, mnk_hip.ptr(step_dev), env_id0, stream_id, mnk_hip.ptr(actions),
                         mnk_hip.ptr(rewards), mnk_hip.ptr(dones), mnk_hip.ptr(mask), mnk_hip.ptr(obs),
                         mnk_hip.obs_code(obs), mnk_hip.STEP_AUTORESET if autoreset else 0, self._stream())

    def observe_into(self, obs=None, mask=None, flip_side=None, fix_empty_mask=False, packed=None) -> None:
        """``obs``: float32 / bfloat16 / uint8 (N, 2, m, n); ``mask``: bool (N, C); ``packed``: int64 (2, W, N), the same
        view as packed planes (channel 0 first).  Any of them may be None."""
        if self.num_envs and (obs is not None or mask is not None or packed is not None):
            mnk_hip.call("mnk_observe", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs, self.m,
                         self.n, mnk_hip.ptr(flip_side), mnk_hip.ptr(obs), mnk_hip.obs_code(obs), mnk_hip.ptr(mask),
                         1 if fix_empty_mask else 0, mnk_hip.ptr(packed), self._stream())

    def sample_legal_into(self, actions, seed: int, step: int, env_id0: int = 0,
                          stream_id: int = mnk_hip.STREAM_MOVE, step_dev=None) -> None:
        """Uniform legal action per env (the reference's ``RandomPolicy``, policy.py:13-29) from Philox.
        ``step_dev``: optional device int64[1] added to ``step`` (a captured graph's advancing counter)."""
        if self.num_envs:
            mnk_hip.call("mnk_sample_legal", mnk_hip.ptr(self._planes), self.num_envs, self.m, self.n, seed, step,
                         mnk_hip.ptr(step_dev), env_id0, stream_id, mnk_hip.ptr(actions), self._stream())

    def specialise_kernels(self, kinds=None) -> int:
        """On a board without a built-in kernel variant (anything but 3x3x3, 9x9x5, 13x13x5, 15x15x5, 19x19x5) the
        library compiles the board's own variant of an API kernel with hiprtc once that kernel is hot (1 024 launches),
        never while a stream is being captured.  This compiles and loads NOW the variants of the kernels launched on
        this board so far (``kinds``: an iterable of ``mnk_hip.JIT_API_*`` instead) -- call it after a warm-up run and
        before capturing a hipGraph of your own (``selfplay.graphed`` does).  Returns how many variants are ready (0 on
        the built-in boards)."""
        return mnk_hip.jit_prepare(self.m, self.n, self.k, kinds)

    def reset_mask_(self, mask_u8) -> None:
        """Fixed-shape reset: envs with a non-zero byte in ``mask_u8`` (bool / uint8, (N,)) start over."""
        if self.num_envs:
            mnk_hip.call("mnk_reset_mask", mnk_hip.ptr(self._planes), mnk_hip.ptr(self._meta), self.num_envs,
                         self.words, mnk_hip.ptr(mask_u8), self._stream())

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The whole env state (packed planes + meta words, 36 B per env at 9x9) as CPU tensors; with the Philox
        step counters of the wrapper / rollout driver a run resumes bit-exactly."""
        self.check_errors()
        return {"geometry": torch.tensor([self.m, self.n, self.k, self.num_envs]),
                "planes": self._planes.cpu(), "meta": self._meta.cpu()}

    def load_state_dict(self, state: Dict[str, torch.Tensor]) -> None:
        if state["geometry"].tolist() != [self.m, self.n, self.k, self.num_envs]:
            raise ValueError(f"state is for {state['geometry'].tolist()}, this env is "
                             f"{[self.m, self.n, self.k, self.num_envs]}")
        self._planes.copy_(state["planes"])
        self._meta.copy_(state["meta"])
        self._err.zero_()

    # ------------------------------------------------------------------ helpers
    def _flags(self) -> int:
        return mnk_hip.STEP_STRICT if self.strict else 0

    def _as_actions(self, actions, count: int) -> torch.Tensor:
        a = torch.as_tensor(_unwrap(actions), device=self._dev).to(torch.int64).reshape(-1).contiguous()
        if a.numel() != count:
            raise IndexError(f"shape mismatch: {a.numel()} actions for {count} envs")
        return a
