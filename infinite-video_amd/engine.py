"""Thin torch-facing wrapper of one ``infv_ltm_handle``: device pointers in, device tensors out.

PyTorch is plumbing here (device memory, the current HIP stream); every number is produced by
the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .basis_maps import NB_BINS, NB_SAMPLES, Plan, build_gaussian_plan, build_plan, padded_N

ProjTensors = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]   # (wk, bk, wv, bv)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _np_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _np_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


TOKEN_DTYPES = {torch.float32: 0, torch.bfloat16: 1}      # infv_token_dtype


def _check_dev(t: torch.Tensor, device: torch.device, name: str, dtype=torch.float32):
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, engine is on {device}")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


class LTMEngine:
    """``n_layers`` LTM instances (one per cross-attention layer) stepped together on one GPU."""

    def __init__(self, num_basis: int, n_heads: int, head_size: int, d_in: int, tokens_per_frame: int,
                 tau: float, sticky: bool, n_layers: int = 1, max_q: int = 32,
                 device: Optional[torch.device] = None, nb_samples: int = NB_SAMPLES,
                 max_batch_chunks: int = 32, gaussian_sigmas: Optional[Sequence[float]] = None):
        """``gaussian_sigmas``: use the reference's Gaussian basis family (``add_gaussian_basis_functions``,
        long_term_attention_gibbs.py:167-174: centres ``linspace(0, 1, num_basis // len(sigmas))`` x these widths) instead
        of the rectangular one the active reference module builds; every step then takes the dense per-call path."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("LTMEngine needs a HIP device (no CPU fallback exists)")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.N, self.H, self.dh, self.d, self.P = num_basis, n_heads, head_size, d_in, tokens_per_frame
        self.dm = n_heads * head_size
        self.tau, self.sticky, self.L, self.S = float(tau), bool(sticky), n_layers, nb_samples
        self.max_q = max_q
        self.gaussian_sigmas = tuple(float(x) for x in gaussian_sigmas) if gaussian_sigmas else None
        # the reference takes any --num_basis (run_inference_inf_video_llama_nextqa.py:61); the kernels tile the basis dimension
        # by 16, so the device holds Np = the next multiple of 16 -- the extra basis functions are inert (basis_maps.padded_N)
        self.Np = padded_N(num_basis)
        if self.Np != num_basis and self.gaussian_sigmas:
            raise ValueError("the Gaussian family needs num_basis to be a multiple of 16")
        cfg = _lib.Config(self.Np, n_heads, head_size, d_in, tokens_per_frame, n_layers, nb_samples,
                          int(self.sticky), max_q, max_batch_chunks)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_create(C.byref(cfg), C.byref(handle)))
        self._h = handle
        self._plans = {}
        self._token_dtype = torch.float32

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.lib.infv_ltm_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ plans
    def ensure_plan(self, T: int) -> Plan:
        if T in self._plans:
            return self._plans[T]
        p = (build_gaussian_plan(int(T), self.N, self.tau, self.gaussian_sigmas, self.S) if self.gaussian_sigmas
             else build_plan(int(T), self.N, self.tau, self.S))
        arrs = dict(
            first_row_box=_np_i32(p.first_row_box), first_row_begin=_np_i32(p.first_row_begin),
            first_row_end=_np_i32(p.first_row_end), first_box_val=_np_f32(p.first_box_val),
            inf_row_box=_np_i32(p.inf_row_box), inf_row_begin=_np_i32(p.inf_row_begin),
            inf_row_end=_np_i32(p.inf_row_end), inf_box_val=_np_f32(p.inf_box_val),
            inf_old_ptr=_np_i32(p.inf_old_ptr), inf_old_slot=_np_i32(p.inf_old_slot),
            readout_w=_np_f32(p.readout_w), edge_box=_np_i32(p.edge_box), edge_dx=_np_f32(p.edge_dx),
            bin_box=_np_i32(p.bin_box), uniform_idx=_np_i32(p.uniform_idx))
        ip = lambda k: arrs[k].ctypes.data_as(_lib.i32p)
        fp = lambda k: arrs[k].ctypes.data_as(_lib.f32p)
        s = _lib.PlanStruct(
            T=int(T), first_rows=len(p.first_row_box), first_row_box=ip("first_row_box"),
            first_row_begin=ip("first_row_begin"), first_row_end=ip("first_row_end"),
            first_box_val=fp("first_box_val"), inf_rows=len(p.inf_row_box), inf_row_box=ip("inf_row_box"),
            inf_row_begin=ip("inf_row_begin"), inf_row_end=ip("inf_row_end"), inf_box_val=fp("inf_box_val"),
            inf_old_ptr=ip("inf_old_ptr"), inf_old_slot=ip("inf_old_slot"), readout_w=fp("readout_w"),
            readout_w_out=p.readout_w_out, n_bins=NB_BINS, edge_box=ip("edge_box"), edge_dx=fp("edge_dx"),
            bin_box=ip("bin_box"), uniform_idx=ip("uniform_idx"))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_set_plan(self._h, C.byref(s)))
            if p.dense:
                # num_basis whose fp32 boxes overlap where the step looks: dense operators (x . G on fp32 MFMA) and
                # two-box tables; every entry point then takes the per-call dense step
                dn = dict(first_GT=_np_f32(p.first_GT), inf_GT=_np_f32(p.inf_GT), bin_box2=_np_i32(p.bin_box2),
                          edge_box2=_np_i32(p.edge_box2), uniform_box2=_np_i32(p.uniform_box2))
                ds = _lib.DensePlanStruct(
                    T=int(T), first_K=int(p.first_GT.shape[1]), first_GT=dn["first_GT"].ctypes.data_as(_lib.f32p),
                    inf_K=int(p.inf_GT.shape[1]), inf_GT=dn["inf_GT"].ctypes.data_as(_lib.f32p),
                    bin_box2=dn["bin_box2"].ctypes.data_as(_lib.i32p), edge_box2=dn["edge_box2"].ctypes.data_as(_lib.i32p),
                    uniform_box2=dn["uniform_box2"].ctypes.data_as(_lib.i32p))
                _lib.check(self.lib.infv_ltm_set_dense_plan(self._h, C.byref(ds)))
            if p.psi:
                # a basis family whose psi(t) is a dense row (the reference's Gaussian family): psi wherever the step evaluates it
                pn = {k: _np_f32(getattr(p, k)) for k in ("psi_edge", "psi_bin", "psi_uniform", "psi_grid", "grid_w")}
                ps = _lib.PsiPlanStruct(T=int(T), n_grid=int(p.psi_grid.shape[0]),
                                        **{k: v.ctypes.data_as(_lib.f32p) for k, v in pn.items()})
                _lib.check(self.lib.infv_ltm_set_psi_plan(self._h, C.byref(ps)))
        self._plans[T] = p
        return p

    def set_dense_operators(self, T: int, first_GT: np.ndarray, inf_GT: Optional[np.ndarray] = None):
        """Replace the operators of chunk length ``T`` by caller-supplied DENSE ones (``first_GT`` [N, T], ``inf_GT`` [N, S + T],
        transposed ``G`` of long_term_attention_gibbs.py:68-84): every step of that length then runs the dense kernels
        (``B = x . G`` on fp32 MFMA).  The resampling / histogram tables stay those of the rectangular basis -- this is the
        operator-level hook for another basis family (e.g. ``basis_maps.gaussian_first_operator_T``), not a second model."""
        from .basis_maps import NB_BINS as _NB, boxes2_of
        self.ensure_plan(T)
        first = _np_f32(first_GT)
        inf = _np_f32(inf_GT) if inf_GT is not None else np.zeros((self.N, self.S + T), np.float32)
        if first.shape != (self.N, T) or inf.shape != (self.N, self.S + T):
            raise ValueError(f"operators must be [{self.N}, {T}] and [{self.N}, {self.S + T}]")
        if self.Np != self.N:
            first = _np_f32(np.concatenate([first, np.zeros((self.Np - self.N, T), np.float32)]))
            inf = _np_f32(np.concatenate([inf, np.zeros((self.Np - self.N, self.S + T), np.float32)]))
        bins = torch.linspace(0, 1, _NB + 1)
        mod = bins.clone()
        mod[0] = -.000001
        mod[-1] = 1.000001
        t_uni = (torch.arange(1, self.S + 1).float() * self.tau / self.S) / self.tau
        tabs = [_np_i32(boxes2_of(x, self.N)) for x in (bins[:-1], mod, t_uni)]
        ds = _lib.DensePlanStruct(
            T=int(T), first_K=int(T), first_GT=first.ctypes.data_as(_lib.f32p), inf_K=int(self.S + T),
            inf_GT=inf.ctypes.data_as(_lib.f32p), bin_box2=tabs[0].ctypes.data_as(_lib.i32p),
            edge_box2=tabs[1].ctypes.data_as(_lib.i32p), uniform_box2=tabs[2].ctypes.data_as(_lib.i32p))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_set_dense_plan(self._h, C.byref(ds)))

    # ------------------------------------------------------------------ helpers
    def _proj_array(self, projs: Sequence[ProjTensors]):
        if len(projs) != self.L:
            raise ValueError(f"expected projections for {self.L} layers, got {len(projs)}")
        # the same weight tensors call after call (the per-chunk loop): validated once, the ctypes array is reused
        key = tuple((t.data_ptr(), tuple(t.shape), t.stride(), t.dtype, t.device) for p in projs for t in p)
        cached = getattr(self, "_proj_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        arr = (_lib.Proj * self.L)()
        for l, (wk, bk, wv, bv) in enumerate(projs):
            for name, t, shape in (("wk", wk, (self.dm, self.d)), ("bk", bk, (self.dm,)),
                                   ("wv", wv, (self.dm, self.d)), ("bv", bv, (self.dm,))):
                _check_dev(t, self.device, f"{name}[{l}]")
                if tuple(t.shape) != shape:
                    raise ValueError(f"{name}[{l}] has shape {tuple(t.shape)}, expected {shape}")
            arr[l] = _lib.Proj(wk.data_ptr(), bk.data_ptr(), wv.data_ptr(), bv.data_ptr())
        self._proj_cache = (key, arr, [tuple(p) for p in projs])      # (the tensors are kept alive with their addresses)
        return arr

    def _check_q(self, q: torch.Tensor) -> int:
        _check_dev(q, self.device, "q")
        if q.dim() != 3 or q.shape[0] != self.L or q.shape[2] != self.dm:
            raise ValueError(f"q must be [{self.L}, Q, {self.dm}], got {tuple(q.shape)}")
        if q.shape[1] > self.max_q:
            raise ValueError(f"Q={q.shape[1]} exceeds max_q={self.max_q}")
        return int(q.shape[1])

    def _check_u(self, u: Optional[torch.Tensor], lead: Tuple[int, ...]):
        if u is None:
            return
        _check_dev(u, self.device, "u", torch.float64)
        if tuple(u.shape) != lead + (self.L, self.S):
            raise ValueError(f"u must be {lead + (self.L, self.S)}, got {tuple(u.shape)}")

    @property
    def has_memory(self) -> bool:
        return bool(_lib.check(self.lib.infv_ltm_has_memory(self._h)))

    def reset(self):
        _lib.check(self.lib.infv_ltm_reset(self._h))

    def _tokens(self, k: torch.Tensor):
        """Frame tokens may be fp32 (the reference's layout) or bf16 (a producer that halves the HBM stream): tell the
        handle which one this call passes."""
        if k.dtype not in TOKEN_DTYPES:
            raise TypeError(f"frame tokens must be float32 or bfloat16, got {k.dtype}")
        _check_dev(k, self.device, "k", k.dtype)
        # always: the dtype is sticky state of the handle and other users of the handle (the video Q-former's C path) set it
        # too, so a Python-side cache of it could go stale
        _lib.check(self.lib.infv_ltm_set_token_dtype(self._h, TOKEN_DTYPES[k.dtype]))
        self._token_dtype = k.dtype
        self._token_code = TOKEN_DTYPES[k.dtype]

    # ------------------------------------------------------------------ operators
    def pool(self, k: torch.Tensor) -> torch.Tensor:
        """k [..., T*P, d] (fp32 or bf16) -> frame means [..., T, d] fp32   (reference :304)."""
        self._tokens(k)
        if k.shape[-1] != self.d or k.shape[-2] % self.P:
            raise ValueError(f"k must be [..., T*{self.P}, {self.d}], got {tuple(k.shape)}")
        n_frames = k.numel() // (self.P * self.d)
        out = torch.empty(k.shape[:-2] + (k.shape[-2] // self.P, self.d), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_pool(self._h, _ptr(k), n_frames, _ptr(out), _stream(self.device)))
        return out

    def pool_rows(self, k: torch.Tensor) -> torch.Tensor:
        """k [C, T*P, d] (fp32 or bf16) -> the memory's new rows R [C, rows, d] fp32: frame means (reference :304) summed
        per box row with the operator's weights (reference :216), one pass over the tokens.  Sparse plans only."""
        self._tokens(k)
        if k.dim() != 3 or k.shape[-1] != self.d or k.shape[-2] % self.P:
            raise ValueError(f"k must be [C, T*{self.P}, {self.d}], got {tuple(k.shape)}")
        n_chunks, T = int(k.shape[0]), int(k.shape[1]) // self.P
        self.ensure_plan(T)
        rows = _lib.check(self.lib.infv_ltm_new_rows(self._h, T))
        out = torch.empty(n_chunks, rows, self.d, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_pool_rows(self._h, _ptr(k), n_chunks, T, _ptr(out), _stream(self.device)))
        return out

    def step(self, kbar: torch.Tensor, q: torch.Tensor, projs: Sequence[ProjTensors],
             u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One chunk from pooled frames kbar [T, d]; q [L, Q, dm]; u [L, S] f64 -> ctx [L, Q, dm]."""
        _check_dev(kbar, self.device, "kbar")
        T = int(kbar.shape[0])
        Q = self._check_q(q)
        self._check_u(u, ())
        self.ensure_plan(T)
        ctx = torch.empty(self.L, Q, self.dm, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_step(self._h, _ptr(kbar), T, _ptr(q), Q, self._proj_array(projs),
                                               _ptr(u), _ptr(ctx), _stream(self.device)))
        return ctx

    # -- lean entry points of the drop-in module's per-call path: the caller has validated shapes / dtypes / devices once for
    #    this call signature; nothing is re-checked here, no context manager is entered (the caller's device is current)
    def pool_into(self, k: torch.Tensor, kbar: torch.Tensor, n_frames: int, token_code: int, stream: C.c_void_p):
        # the token dtype is sticky state of the C handle and other callers of the same handle change it (the video
        # Q-former's C path forces fp32): always set it, one cheap host call, never trust a Python-side cache
        _lib.check(self.lib.infv_ltm_set_token_dtype(self._h, token_code))
        rc = self.lib.infv_ltm_pool(self._h, C.c_void_p(k.data_ptr()), n_frames, C.c_void_p(kbar.data_ptr()), stream)
        if rc < 0:
            _lib.check(rc)

    def step_raw(self, kbar_ptr: int, T: int, q_ptr: int, Q: int, proj_arr, u_ptr: int, ctx_ptr: int, stream: C.c_void_p):
        """infv_ltm_step on raw addresses; ``u_ptr`` may be pinned host memory (the draw reads its 4 KB through the mapping)."""
        rc = self.lib.infv_ltm_step(self._h, C.c_void_p(kbar_ptr), T, C.c_void_p(q_ptr), Q, proj_arr,
                                    C.c_void_p(u_ptr) if u_ptr else None, C.c_void_p(ctx_ptr), stream)
        if rc < 0:
            _lib.check(rc)

    def forward_into_raw(self, k_ptr: int, token_code: int, T: int, kbar_ptr: int, q_ptr: int, Q: int, proj_arr, u_ptr: int,
                         ctx_ptr: int, stream: C.c_void_p):
        """infv_ltm_forward_into on raw addresses: set the token dtype, pool k into the caller's kbar, step from it -- ONE C call
        (the drop-in module's steady-state forward; ``pool_into`` + ``step_raw`` were three)."""
        rc = self.lib.infv_ltm_forward_into(self._h, C.c_void_p(k_ptr), token_code, T, C.c_void_p(kbar_ptr), C.c_void_p(q_ptr), Q,
                                            proj_arr, C.c_void_p(u_ptr) if u_ptr else None, C.c_void_p(ctx_ptr), stream)
        if rc < 0:
            _lib.check(rc)

    def forward(self, k: torch.Tensor, q: torch.Tensor, projs: Sequence[ProjTensors],
                u: Optional[torch.Tensor] = None, new_doc: bool = False) -> torch.Tensor:
        """LongTermAttention.forward for all layers: k [T*P, d], q [L, Q, dm] -> ctx [L, Q, dm]."""
        self._tokens(k)
        if k.dim() != 2 or k.shape[1] != self.d or k.shape[0] % self.P:
            raise ValueError(f"k must be [T*{self.P}, {self.d}], got {tuple(k.shape)}")
        T = k.shape[0] // self.P
        Q = self._check_q(q)
        self._check_u(u, ())
        self.ensure_plan(T)
        ctx = torch.empty(self.L, Q, self.dm, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_forward(self._h, _ptr(k), T, _ptr(q), Q, self._proj_array(projs),
                                                  _ptr(u), int(new_doc), _ptr(ctx), _stream(self.device)))
        return ctx

    def consolidate(self, k: torch.Tensor, q: torch.Tensor, projs: Sequence[ProjTensors],
                    u: Optional[torch.Tensor] = None, new_doc: bool = True,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Whole-video loop: k [C, T*P, d] (fp32 or bf16), q [L, Q, dm], u [C, L, S] -> ctx [C, L, Q, dm]."""
        self._tokens(k)
        if k.dim() != 3 or k.shape[2] != self.d or k.shape[1] % self.P:
            raise ValueError(f"k must be [C, T*{self.P}, {self.d}], got {tuple(k.shape)}")
        Cn, T = int(k.shape[0]), k.shape[1] // self.P
        Q = self._check_q(q)
        self._check_u(u, (Cn,))
        self.ensure_plan(T)
        if out is None:
            out = torch.empty(Cn, self.L, Q, self.dm, device=self.device, dtype=torch.float32)
        else:
            _check_dev(out, self.device, "out")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_consolidate(self._h, _ptr(k), Cn, T, _ptr(q), Q, self._proj_array(projs),
                                                      _ptr(u), int(new_doc), _ptr(out), _stream(self.device)))
        return out

    def consolidate_pooled(self, kbar: torch.Tensor, q: torch.Tensor, projs: Sequence[ProjTensors],
                           u: Optional[torch.Tensor] = None, new_doc: bool = True) -> torch.Tensor:
        """consolidate() from frame means the caller already holds: kbar [C, T, d] fp32 (= pool() of the tokens).
        Bit-identical to consolidate() on those tokens; the pooling stage drops out."""
        _check_dev(kbar, self.device, "kbar")
        if kbar.dim() != 3 or kbar.shape[2] != self.d or kbar.dtype != torch.float32:
            raise ValueError(f"kbar must be fp32 [C, T, {self.d}], got {kbar.dtype} {tuple(kbar.shape)}")
        Cn, T = int(kbar.shape[0]), int(kbar.shape[1])
        Q = self._check_q(q)
        self._check_u(u, (Cn,))
        self.ensure_plan(T)
        out = torch.empty(Cn, self.L, Q, self.dm, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_consolidate_pooled(self._h, _ptr(kbar), Cn, T, _ptr(q), Q, self._proj_array(projs),
                                                             _ptr(u), int(new_doc), _ptr(out), _stream(self.device)))
        return out

    def consolidate_q(self, k: torch.Tensor, q: torch.Tensor, projs: Sequence[ProjTensors],
                      u: Optional[torch.Tensor] = None, new_doc: bool = True) -> torch.Tensor:
        """Whole-video loop with a DIFFERENT query per chunk (a cross-attention layer after the first, Qformer.py:211):
        k [C, T*P, d], q [C, L, Q, dm], u [C, L, S] -> ctx [C, L, Q, dm].  Pooling and new-row projections are batched,
        the memory chain runs chunk by chunk; equals C calls of forward()."""
        self._tokens(k)
        if k.dim() != 3 or k.shape[2] != self.d or k.shape[1] % self.P:
            raise ValueError(f"k must be [C, T*{self.P}, {self.d}], got {tuple(k.shape)}")
        Cn, T = int(k.shape[0]), k.shape[1] // self.P
        _check_dev(q, self.device, "q")
        if q.dim() != 4 or q.shape[0] != Cn or q.shape[1] != self.L or q.shape[3] != self.dm or q.shape[2] > self.max_q:
            raise ValueError(f"q must be [{Cn}, {self.L}, Q <= {self.max_q}, {self.dm}], got {tuple(q.shape)}")
        Q = int(q.shape[2])
        self._check_u(u, (Cn,))
        self.ensure_plan(T)
        out = torch.empty(Cn, self.L, Q, self.dm, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_consolidate_q(self._h, _ptr(k), Cn, T, _ptr(q), Q, self._proj_array(projs),
                                                        _ptr(u), int(new_doc), _ptr(out), _stream(self.device)))
        return out

    # ------------------------------------------------------------------ state
    def export_state(self, layer: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(B_past [N, d], unnormalised sticky bin masses [127]) of one layer."""
        B = torch.empty(self.Np, self.d, device=self.device, dtype=torch.float32)
        mass = torch.empty(NB_BINS, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_export_state(self._h, layer, _ptr(B), _ptr(mass), _stream(self.device)))
        return B[:self.N], mass[:NB_BINS - 1]

    def import_state(self, layer: int, B: torch.Tensor, bin_mass: Optional[torch.Tensor], proj: ProjTensors):
        _check_dev(B, self.device, "B")
        if tuple(B.shape) != (self.N, self.d):
            raise ValueError(f"B must be [{self.N}, {self.d}]")
        if self.Np != self.N:                                   # inert rows of the padding basis functions
            B = torch.cat([B, torch.zeros(self.Np - self.N, self.d, device=B.device, dtype=B.dtype)])
        if bin_mass is not None:
            _check_dev(bin_mass, self.device, "bin_mass")
            if bin_mass.numel() != NB_BINS - 1:
                raise ValueError(f"bin_mass must have {NB_BINS - 1} entries")
        p = _lib.Proj(*(t.data_ptr() for t in proj))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_import_state(self._h, layer, _ptr(B), _ptr(bin_mass), C.byref(p),
                                                       _stream(self.device)))

    def export_chain_state(self, Q: int) -> torch.Tensor:
        """Everything ``consolidate`` carries from chunk to chunk as one fp32 device tensor (infv_ltm_export_chain_state):
        importing it into another engine and calling ``consolidate(..., new_doc=False)`` with the same query and weights
        continues the chain bit for bit."""
        n = int(self.lib.infv_ltm_chain_state_bytes(self._h, Q))
        if n <= 0:
            raise ValueError(f"bad query length {Q}")
        blob = torch.empty(n // 4, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_export_chain_state(self._h, Q, _ptr(blob), _stream(self.device)))
        return blob

    def import_chain_state(self, Q: int, blob: torch.Tensor):
        _check_dev(blob, self.device, "blob")
        if blob.numel() * 4 != int(self.lib.infv_ltm_chain_state_bytes(self._h, Q)):
            raise ValueError("chain-state blob does not match this engine's shape")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_import_chain_state(self._h, Q, _ptr(blob), _stream(self.device)))

    def chain_state_numel(self, Q: int) -> int:
        return int(self.lib.infv_ltm_chain_state_bytes(self._h, Q)) // 4

    def last_scores_device(self, Q: int) -> torch.Tensor:
        """Bias-free scores of the last step under the last call's query, [L, H, Q, N] on the device: the scores slice of the
        chain-state blob (layout: include/infv_ltm.h, infv_ltm_export_chain_state).  What SURVEY.md section 8e lists beside
        ``B_past`` in the all-gather payload: with them a peer can re-derive the sticky density on any grid."""
        blob = self.export_chain_state(Q)
        Np = self.Np                                                            # (the handle's basis count: padded to a multiple of 16)
        off = 16 + self.L * Np * self.d + self.L * Np * 2 * self.dm             # header (16 int32) | B | projected memory | scores | masses
        return blob[off:off + self.L * self.H * Q * Np].reshape(self.L, self.H, Q, Np)[..., :self.N]

    def reproject(self, projs: Sequence[ProjTensors]):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_reproject(self._h, self._proj_array(projs), _stream(self.device)))

    def last_draw(self, layer: int, want_scores: bool = False):
        """(bins [S], idx [S], probs [127][, scores [H,Q,N]]) of the last sticky step (host numpy)."""
        bins = np.empty(self.S, np.int32)
        idx = np.empty(self.S, np.int32)
        probs = np.empty(NB_BINS - 1, np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_get_draw(
                self._h, layer, bins.ctypes.data_as(_lib.i32p), idx.ctypes.data_as(_lib.i32p),
                probs.ctypes.data_as(_lib.f32p), None, _stream(self.device)))
        return bins, idx, probs

    def last_scores(self, layer: int, Q: int) -> np.ndarray:
        sc = np.empty((self.H, Q, self.Np), np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_get_draw(self._h, layer, None, None, None,
                                                  sc.ctypes.data_as(_lib.f32p), _stream(self.device)))
        return sc[:, :, :self.N]

    def set_probs(self, layer: int, probs: np.ndarray):
        p = _np_f32(probs)
        if p.size != NB_BINS - 1:
            raise ValueError(f"probs must have {NB_BINS - 1} entries")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_set_probs(self._h, layer, p.ctypes.data_as(_lib.f32p)))

    def set_bins(self, layer: int, bins: np.ndarray):
        """Forced draw of the next per-call step of ``layer`` (see infv_ltm_set_bins)."""
        b = _np_i32(bins)
        if b.size != self.S:
            raise ValueError(f"bins must have {self.S} entries")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_set_bins(self._h, layer, b.ctypes.data_as(_lib.i32p)))

    def set_trace(self, n_chunks: int):
        """Record the draw of every chunk of the following consolidate() calls: returns device tensors
        (bins [n_chunks, L, S] int32 initialised to -1, probs [n_chunks, L, 128] fp32); ``n_chunks = 0`` stops tracing."""
        if n_chunks <= 0:
            self._trace = None
            _lib.check(self.lib.infv_ltm_set_trace(self._h, None, None, 0))
            return None
        bins = torch.full((n_chunks, self.L, self.S), -1, device=self.device, dtype=torch.int32)
        probs = torch.zeros(n_chunks, self.L, NB_BINS, device=self.device, dtype=torch.float32)
        self._trace = (bins, probs)                       # keep the buffers alive while the library writes them
        _lib.check(self.lib.infv_ltm_set_trace(self._h, _ptr(bins), _ptr(probs), n_chunks))
        return bins, probs

    def sync(self):
        """Wait for the engine's work on the current stream and raise if a device-side failure was latched
        (the persistent chain kernel of consolidate() timed out: its outputs are invalid)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.infv_ltm_sync(self._h, _stream(self.device)))

    # ------------------------------------------------------------------ measurement
    def profile(self, on: bool):
        """Bracket every kernel launch of this engine with HIP events (bench.py's roofline leg)."""
        _lib.check(self.lib.infv_ltm_profile_enable(self._h, int(on)))

    def profile_read(self):
        """{kernel: (launches, total device ms)} since profiling was enabled; synchronises."""
        out = {}
        for i, name in enumerate(_lib.KERNELS):
            n, ms = C.c_int64(), C.c_double()
            _lib.check(self.lib.infv_ltm_profile_read(self._h, i, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out
