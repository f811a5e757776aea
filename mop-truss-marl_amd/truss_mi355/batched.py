"""BatchedTruss: B independent truss designs resident in device memory, stepped by one HIP launch.

This is the product-side host mirror of the reference's `Game_research04` transition
(`_game_modify`, truss2D_ENV.py:370-525) and reset analysis (`_game_get_1_state`, :336-351) for a
whole batch of environments.  PyTorch is used for device memory, streams and (in bench/training)
torch.distributed only; all arithmetic happens in the native library behind the C ABI
(include/truss_mi355.h).

Tensors live on the device that matches the loaded library's backend: "hip" -> a cuda device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .topology import TrussTopology


import contextlib

_NULL_CTX = contextlib.nullcontext()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


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
        self._step_cache = {}
        self._obs_cache = {}

    # current design
    @property
    def y(self):
        return self.ybuf[self.cur]

    @property
    def sec(self):
        return self.secbuf[self.cur]

    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

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
    def fused_obs(self):
        """True when step(obs=...) writes the observations from the step's own launch for this topology."""
        return bool(self.lib.dll.truss_topo_fused_obs(self.h))

    def _args(self, flags, a_geo, a_topo, coin, mu_in, md_in, y_in, sec_in, y_out, sec_out, want_energy=True, n_envs=None,
              obs=None):
        a = _lib.StepArgs()
        a.struct_size = C.sizeof(_lib.StepArgs)
        a.n_envs, a.flags = (self.B if n_envs is None else int(n_envs)), flags
        a.x, a.y_in, a.sec_in = _ptr(self.x), _ptr(y_in), _ptr(sec_in)
        a.max_up_in, a.max_down_in = _ptr(mu_in), _ptr(md_in)
        a.a_geo, a.a_topo, a.coin = _ptr(a_geo), _ptr(a_topo), _ptr(coin)
        a.target, a.env_params = _ptr(self.target), _ptr(self.env_params)
        a.y_out, a.sec_out = _ptr(y_out), _ptr(sec_out)
        a.max_up_out, a.max_down_out = _ptr(self.max_up), _ptr(self.max_down)
        a.disp, a.q0, a.sr, a.comp = _ptr(self.disp), _ptr(self.q0), _ptr(self.sr), _ptr(self.comp)
        a.point, a.obj = _ptr(self.point), _ptr(self.obj)
        a.disp_f64, a.q0_f64 = _ptr(self.disp_f64), _ptr(self.q0_f64)
        a.energy = _ptr(self.energy) if want_energy else None
        a.reactions = _ptr(self.reactions) if want_energy else None
        a.status = _ptr(self.status)
        if obs is not None:
            a.flags |= _lib.F_EMIT_OBS
            a.x_n, a.A_s, a.A_n_ts, a.A_n_cs = (_ptr(obs.get(k)) for k in ("x_n", "A_s", "A_n_ts", "A_n_cs"))
            a.nN_x_n, a.nN_x_e = _ptr(obs.get("nN_x_n")), _ptr(obs.get("nN_x_e"))
        return a

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
        a = self._args(_lib.F_NO_DECODE, None, None, None, None, None, self.y, self.sec, self.y, self.sec,
                       n_envs=n, obs=out)
        with self._on_device():
            self.lib.check(self.lib.dll.truss_step(self.h, C.byref(a), self._stream()), "truss_step(analyze)")
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
        # an RL loop passes the same buffers every step: validated argument blocks are kept per buffer set
        key = (self.cur, flags, a_geo.data_ptr(), a_topo.data_ptr(), coin.data_ptr(),
               0 if max_up_in is None else max_up_in.data_ptr(), 0 if max_down_in is None else max_down_in.data_ptr(),
               a_geo.shape, a_topo.shape, a_geo.dtype, a_topo.dtype, a_geo.is_contiguous(), a_topo.is_contiguous(), B,
               None if out is None else tuple(0 if out.get(k) is None else out[k].data_ptr() for k in self.OBS_KEYS))
        a = self._step_cache.get(key)
        if a is None:
            self._chk(a_geo, (B, N, 2), torch.float32, "a_geo")
            self._chk(a_topo, (B, N, 3), torch.float32, "a_topo")
            if coin is not self._coin0:
                self._chk(coin, (B,), torch.uint8, "coin")
            self._chk(max_up_in, (B, N), torch.float32, "max_up_in")
            self._chk(max_down_in, (B, N), torch.float32, "max_down_in")
            a = self._args(flags, a_geo, a_topo, coin, max_up_in, max_down_in, self.ybuf[self.cur], self.secbuf[self.cur],
                           self.ybuf[nxt], self.secbuf[nxt], n_envs=B, obs=out)
            if len(self._step_cache) > 64:
                self._step_cache.clear()
            self._step_cache[key] = a
        with self._on_device():
            rc = self.lib.dll.truss_step(self.h, C.byref(a), self._stream())
        if rc:
            self.lib.check(rc, "truss_step")
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
        a = self._args(0, a_geo_sets, a_topo_sets, coin, None, None, self.ybuf[self.cur], self.secbuf[self.cur],
                       self.ybuf[nxt], self.secbuf[nxt], want_energy=False)
        with self._on_device():
            self.lib.check(self.lib.dll.truss_rollout(self.h, C.byref(a), int(n_steps), int(S), self._stream()),
                           "truss_rollout")
        if n_steps & 1:
            self.cur = nxt

    def observe(self, out=None, n_active=None):
        """state_data + state_data_not_norm (truss2D_ENV.py:40-193) for the current design and the last
        analysis, for every env: x_n[B,N,13], A_s/A_n_ts/A_n_cs[B,N,N], nN_x_n[B,N,12], nN_x_e[B,E,21]
        (device tensors).  A_n, mask and nC_e are topology-static: TrussTopology.normalized_adjacency()
        / .incidence()."""
        nact = self._n(n_active)
        key0 = None if out is None else tuple(0 if out.get(k) is None else out[k].data_ptr() for k in self.OBS_KEYS)
        if out is None or (self.cur, id(out), nact) + key0 not in self._obs_cache:
            out = self._obs_out(out, nact)
        if nact == 0:
            return out
        key = (self.cur, id(out), nact) + tuple(0 if out.get(k) is None else out[k].data_ptr()
                                                for k in ("x_n", "A_s", "A_n_ts", "A_n_cs", "nN_x_n", "nN_x_e"))
        a = self._obs_cache.get(key)
        if a is not None:
            with self._on_device():
                rc = self.lib.dll.truss_obs(self.h, C.byref(a), self._stream())
            if rc:
                self.lib.check(rc, "truss_obs")
            return out
        a = _lib.ObsArgs()
        a.struct_size = C.sizeof(_lib.ObsArgs)
        a.n_envs, a.flags = nact, 0
        a.x, a.y, a.sec = _ptr(self.x), _ptr(self.y), _ptr(self.sec)
        a.max_up, a.max_down, a.target = _ptr(self.max_up), _ptr(self.max_down), _ptr(self.target)
        a.disp, a.q0, a.sr, a.comp = _ptr(self.disp), _ptr(self.q0), _ptr(self.sr), _ptr(self.comp)
        a.env_params = _ptr(self.env_params)
        a.x_n, a.A_s, a.A_n_ts, a.A_n_cs = _ptr(out.get("x_n")), _ptr(out.get("A_s")), _ptr(out.get("A_n_ts")), _ptr(out.get("A_n_cs"))
        a.nN_x_n, a.nN_x_e = _ptr(out.get("nN_x_n")), _ptr(out.get("nN_x_e"))
        if len(self._obs_cache) > 16:
            self._obs_cache.clear()
        self._obs_cache[key] = a
        with self._on_device():
            self.lib.check(self.lib.dll.truss_obs(self.h, C.byref(a), self._stream()), "truss_obs")
        return out

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

