"""BatchedTruss: B independent truss designs resident in device memory, stepped by one HIP launch.

This is the product-side host mirror of the reference's `Game_research04` transition
(`_game_modify`, truss2D_ENV.py:370-525) and reset analysis (`_game_get_1_state`, :336-351) for a
whole batch of environments.  PyTorch is used for device memory, streams and (in bench/training)
torch.distributed only; all arithmetic happens in the native library behind the C ABI
(include/truss_mi355.h), reached through the PyTorch custom operators `torch.ops.truss_mi355.*`
(truss_mi355/ops.py, csrc/truss_torch_ops.cpp).

Tensors live on the device that matches the loaded library's backend: "hip" -> a cuda device.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch

from . import _lib, ops
from .topology import TrussTopology

_NULL_CTX = contextlib.nullcontext()


class BatchedTruss:
    """Device-resident state of B envs sharing one topology.

    Per-env constants (set once per episode): x[B,N], target[B,N], env_params[B,8].
    Design state (double buffered):           y[B,N] float32, sec[B,E] int32.
    Results of the last step/analysis:        disp, q0, sr, comp, point, obj, max_up, max_down, status.
    """

    def __init__(self, topo: TrussTopology, n_envs: int, device=None, lib: "_lib.TrussLib | None" = None,
                 debug_f64: bool = False):
        self.lib = lib or _lib.load()
        if device is None:
            device = "cuda" if self.lib.backend == "hip" else "cpu"
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.lib.backend == "hip" and self.device.type != "cuda":
            raise _lib.TrussError("the HIP library needs tensors on a cuda (ROCm) device")
        self.topo = topo
        self._ops = ops.namespace()
        self._lib_id = ops.bind(self.lib)
        # the native topology owns device tables: one handle per (library, device), created on THIS device
        with self._on_device():
            self.h = topo.native(self.lib, self.device.index if self.device.type == "cuda" else None)
        self.B, self.N, self.E = int(n_envs), topo.N, topo.E
        _, _, self.ndof = topo.dofs(self.lib)
        B, N, E = self.B, self.N, self.E
        dev = self.device

        def z(shape, dt):
            return torch.zeros(shape, dtype=dt, device=dev)

        f32, f64, i32, u8 = torch.float32, torch.float64, torch.int32, torch.uint8
        self.x = z((B, N), f32)
        self.target = z((B, N), f32)
        self.env_params = z((B, _lib.NPARAM), f64)
        self.ybuf = [z((B, N), f32), z((B, N), f32)]
        self.secbuf = [z((B, E), i32), z((B, E), i32)]
        self.cur = 0
        self.max_up = z((B, N), f32)
        self.max_down = z((B, N), f32)
        self.disp = z((B, N, 2), f32)
        self.q0 = z((B, E), f32)
        self.sr = z((B, E), f32)
        self.comp = z((B, E), u8)
        self.point = z((B, 4), f32)
        self.obj = z((B, 2), f32)
        self.status = z((B,), i32)
        self.energy = z((B,), f64)
        self.reactions = z((B, max(2 * N - self.ndof, 1)), f64)
        self.disp_f64 = z((B, N, 2), f64) if debug_f64 else None
        self.q0_f64 = z((B, E), f64) if debug_f64 else None
        self._coin0 = z((B,), u8)

    # current design
    @property
    def y(self):
        return self.ybuf[self.cur]

    @property
    def sec(self):
        return self.secbuf[self.cur]

    def _on_device(self):
        """Native launches go to the CURRENT HIP device: make it this env's device for the call (a no-op
        context on the one-rank-per-GPU path, where it already is)."""
        if self.device.type == "cuda" and torch.cuda.current_device() != self.device.index:
            return torch.cuda.device(self.device)
        return _NULL_CTX

    OBS_KEYS = ("x_n", "A_s", "A_n_ts", "A_n_cs", "nN_x_n", "nN_x_e")

    def obs_buffers(self):
        """The env's own observation tensors (allocated on first use)."""
        if not hasattr(self, "_obs"):
            B, N, E = self.B, self.N, self.E
            mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=self.device)
            self._obs = dict(x_n=mk(B, N, 13), A_s=mk(B, N, N), A_n_ts=mk(B, N, N), A_n_cs=mk(B, N, N),
                             nN_x_n=mk(B, N, 12), nN_x_e=mk(B, E, 21))
        return self._obs

    def _obs_out(self, out, nact):
        """Validated observation outputs: None -> the env's own buffers; a dict may leave tensors out."""
        if out is None or out is True:
            return self.obs_buffers()
        N, E = self.N, self.E
        shapes = dict(x_n=(N, 13), A_s=(N, N), A_n_ts=(N, N), A_n_cs=(N, N), nN_x_n=(N, 12), nN_x_e=(E, 21))
        for k in self.OBS_KEYS:
            t = out.get(k)
            if t is None:
                continue
            if (t.dim() != 3 or t.shape[0] < nact or tuple(t.shape[1:]) != shapes[k] or t.dtype != torch.float32
                    or not t.is_contiguous() or t.device != self.device):
                raise ValueError(f"obs[{k}]: expected contiguous float32 [>={nact}, {shapes[k][0]}, {shapes[k][1]}] on "
                                 f"{self.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
        return out

    @property
    def persistent_rollout(self):
        """True when rollout() runs its chained steps as ONE launch for this topology (state resident in LDS)."""
        return bool(self.lib.dll.truss_topo_persistent_rollout(self.h))

    @property
    def fused_obs(self):
        """True when step(obs=...) writes the observations from the step's own launch for this topology."""
        return bool(self.lib.dll.truss_topo_fused_obs(self.h))

    def _step_tensors(self, a_geo, a_topo, coin, mu_in, md_in, y_in, sec_in, y_out, sec_out, want_energy, obs):
        o = obs or {}
        return (self.x, y_in, sec_in, mu_in, md_in, a_geo, a_topo, coin, self.target, self.env_params, y_out, sec_out,
                self.max_up, self.max_down, self.disp, self.q0, self.sr, self.comp, self.point, self.obj, self.disp_f64,
                self.q0_f64, self.energy if want_energy else None, self.reactions if want_energy else None, self.status,
                o.get("x_n"), o.get("A_s"), o.get("A_n_ts"), o.get("A_n_cs"), o.get("nN_x_n"), o.get("nN_x_e"))

    def _step_op(self, flags, n_envs, a_geo, a_topo, coin, mu_in, md_in, y_in, sec_in, y_out, sec_out, want_energy=True,
                 obs=None):
        """torch.ops.truss_mi355.step: one truss_step launch (+ the observation tensors with `obs`)."""
        if obs is not None:
            flags |= _lib.F_EMIT_OBS
        with self._on_device():
            ops.call(self._ops.step, self._lib_id, self.h.value, ops.stream_of(self.device), flags, n_envs, self.N, self.E,
                           *self._step_tensors(a_geo, a_topo, coin, mu_in, md_in, y_in, sec_in, y_out, sec_out, want_energy, obs))

    def _chk(self, t, shape, dtype, name):
        if t is None:
            return
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous() or t.device != self.device:
            raise ValueError(f"{name}: expected contiguous {dtype} {tuple(shape)} on {self.device}, got "
                             f"{t.dtype} {tuple(t.shape)} on {t.device}")

    # ---- episode set-up -------------------------------------------------------------------
    def set_constants(self, x, target, y_max, d_min, max_def, load_x, load_y, is_roof):
        """Per-env constants; array-likes broadcast over B."""
        dev = self.device
        self.x.copy_(torch.as_tensor(np.broadcast_to(np.asarray(x, np.float32), (self.B, self.N)).copy(), device=dev))
        self.target.copy_(torch.as_tensor(np.broadcast_to(np.asarray(target, np.float32), (self.B, self.N)).copy(),
                                          device=dev))
        P = np.zeros((self.B, _lib.NPARAM), np.float64)
        P[:, _lib.P_YMAX], P[:, _lib.P_DMIN], P[:, _lib.P_MAXDEF] = y_max, d_min, max_def
        P[:, _lib.P_LOADX], P[:, _lib.P_LOADY], P[:, _lib.P_ISROOF] = load_x, load_y, is_roof
        P[:, _lib.P_INTOBJ1] = 1.0
        P[:, _lib.P_INTOBJ2] = 1.0
        self.env_params.copy_(torch.as_tensor(P, device=dev))

    def set_design(self, y, sec):
        self.y.copy_(torch.as_tensor(np.broadcast_to(np.asarray(y, np.float32), (self.B, self.N)).copy(),
                                     device=self.device))
        self.sec.copy_(torch.as_tensor(np.broadcast_to(np.asarray(sec, np.int32), (self.B, self.E)).copy(),
                                       device=self.device))

    def _n(self, n_active):
        if n_active is None:
            return self.B
        n = int(n_active)
        if not 0 <= n <= self.B:
            raise ValueError(f"n_active {n} outside 0..{self.B}")
        return n

    def analyze(self, set_normalisers: bool = False, n_active=None, obs=None):
        """Model.restore(); Model.gen_all() on the current design (reset path).  With
        set_normalisers the raw objectives become int_obj1/int_obj2 (Game_research04.__init__,
        truss2D_ENV.py:264-274).  n_active: only the first n envs of the resident buffers (callers that
        compact their live envs to the front, truss_mi355/marl.py).  obs: True / dict -> also write the
        observation tensors (`_game_get_1_state`, truss2D_ENV.py:336-351) in the same native call."""
        n = self._n(n_active)
        if n == 0:
            return None
        out = None if obs is None or obs is False else self._obs_out(obs, n)
        self._step_op(_lib.F_NO_DECODE, n, None, None, None, None, None, self.y, self.sec, self.y, self.sec, obs=out)
        if set_normalisers:
            self.env_params[:, _lib.P_INTOBJ1] = self.obj[:, 0].double()
            self.env_params[:, _lib.P_INTOBJ2] = self.obj[:, 1].double()
            self.point[:, 0] = 1.0
            self.point[:, 1] = 1.0
        return out

    # ---- the transition ---------------------------------------------------------------------
    def step(self, a_geo, a_topo, coin=None, max_up_in=None, max_down_in=None, clamp_inplace=False, n_active=None,
             obs=None):
        """One `_game_modify` per env from the current design; the new design becomes current.
        a_geo [B,N,2], a_topo [B,N,3] float32 on the env's device ([n_active, ...] with n_active).
        obs: True (the env's own buffers) or a dict of output tensors -> the observation tensors of the new
        design (state_data + state_data_not_norm, truss2D_ENV.py:497-500) are written by the same native call
        -- by the same kernel launch where `fused_obs` -- and returned."""
        B, N = self._n(n_active), self.N
        if B == 0:
            return None
        out = None if obs is None or obs is False else self._obs_out(obs, B)
        if coin is None:
            coin = self._coin0
        nxt = self.cur ^ 1
        flags = _lib.F_CLAMP_INPLACE if clamp_inplace else 0
        if tuple(a_geo.shape) != (B, N, 2) or tuple(a_topo.shape) != (B, N, 3):
            raise ValueError(f"actions: expected [{B}, {N}, 2] and [{B}, {N}, 3], got {tuple(a_geo.shape)} and {tuple(a_topo.shape)}")
        for t, nm in ((max_up_in, "max_up_in"), (max_down_in, "max_down_in")):
            if t is not None and tuple(t.shape) != (B, N):
                raise ValueError(f"{nm}: expected [{B}, {N}], got {tuple(t.shape)}")
        if coin is not self._coin0 and tuple(coin.shape) != (B,):
            raise ValueError(f"coin: expected [{B}], got {tuple(coin.shape)}")
        # dtype / device / contiguity / size of every tensor are checked by the operator itself
        self._step_op(flags, B, a_geo, a_topo, coin, max_up_in, max_down_in, self.ybuf[self.cur], self.secbuf[self.cur],
                      self.ybuf[nxt], self.secbuf[nxt], obs=out)
        self.cur = nxt
        return out

    def rollout(self, a_geo_sets, a_topo_sets, n_steps, coin=None):
        """n_steps chained transitions in one native call; action set s % S is used at step s.
        a_geo_sets [S,B,N,2], a_topo_sets [S,B,N,3]."""
        S = a_geo_sets.shape[0]
        self._chk(a_geo_sets, (S, self.B, self.N, 2), torch.float32, "a_geo_sets")
        self._chk(a_topo_sets, (S, self.B, self.N, 3), torch.float32, "a_topo_sets")
        if coin is None:
            coin = self._coin0
        nxt = self.cur ^ 1
        with self._on_device():
            ops.call(self._ops.rollout, self._lib_id, self.h.value, ops.stream_of(self.device), 0, self.B, self.N, self.E, int(n_steps), int(S),
                              *self._step_tensors(a_geo_sets, a_topo_sets, coin, None, None, self.ybuf[self.cur],
                                                  self.secbuf[self.cur], self.ybuf[nxt], self.secbuf[nxt], False, None))
        if n_steps & 1:
            self.cur = nxt

    def observe(self, out=None, n_active=None):
        """state_data + state_data_not_norm (truss2D_ENV.py:40-193) for the current design and the last
        analysis, for every env: x_n[B,N,13], A_s/A_n_ts/A_n_cs[B,N,N], nN_x_n[B,N,12], nN_x_e[B,E,21]
        (device tensors).  A_n, mask and nC_e are topology-static: TrussTopology.normalized_adjacency()
        / .incidence()."""
        nact = self._n(n_active)
        out = self._obs_out(out, nact)
        if nact == 0:
            return out
        with self._on_device():
            ops.call(self._ops.obs, self._lib_id, self.h.value, ops.stream_of(self.device), nact, self.N, self.E, self.x, self.y, self.sec,
                          self.max_up, self.max_down, self.target, self.disp, self.q0, self.sr, self.comp, self.env_params,
                          out.get("x_n"), out.get("A_s"), out.get("A_n_ts"), out.get("A_n_cs"), out.get("nN_x_n"), out.get("nN_x_e"))
        return out

    def check(self):
        """Synchronising health check of the last native call (status[B], include/truss_mi355.h): raises TrussError when an env's
        stiffness matrix was not positive definite (bit 0: the reference would raise LinAlgError, FEM_2Dtruss.py:337) or when the
        fused observation writer gave up waiting for a step's results (bit 1: the observation tensors of that call are incomplete).
        Long-running loops call it at episode boundaries; a step itself never synchronises."""
        st = self.status
        bad = int((st & _lib.STATUS_NOT_SPD).ne(0).sum().item())
        late = int((st & _lib.STATUS_OBS_TIMEOUT).ne(0).sum().item())
        if late:
            raise _lib.TrussError(f"{late} env(s): the observation stream of the fused step timed out (TRUSS_STATUS_OBS_TIMEOUT); "
                                  "the observation tensors of the last step are incomplete")
        if bad:
            raise _lib.TrussError(f"{bad} env(s): non-positive pivot (K not SPD)")

    def results(self):
        """Host copies (numpy) of the last step's outputs."""
        g = lambda t: t.detach().cpu().numpy()
        out = dict(y=g(self.y), sec=g(self.sec), max_up=g(self.max_up), max_down=g(self.max_down), disp=g(self.disp),
                   q0=g(self.q0), sr=g(self.sr), comp=g(self.comp), point=g(self.point), obj=g(self.obj),
                   status=g(self.status), energy=g(self.energy), reactions=g(self.reactions))
        if self.disp_f64 is not None:
            out["disp_f64"] = g(self.disp_f64)
            out["q0_f64"] = g(self.q0_f64)
        return out

