"""Drop-in ``LongTermAttention`` for infinity-Video on MI355X.

Mirrors the operator surface of the reference module

    infty-Video-LLaMA/InfVideoLLaMA/models/long_term_attention_gibbs.py:25-346   (Video-LLaMA)
    infty-VideoChat2/models/blip2/long_term_attention_gibbs.py                   (VideoChat2)

same constructor kwargs (as the Q-former passes them, Qformer.py:135-158), same
``forward(k, q, new_doc, layer_n)``, same mutable attributes (``length``, ``target_len``,
``B_past``, ``x_past``).  The arithmetic runs in the HIP kernels of ``libinfv_ltm.so``; there is
no PyTorch fallback -- constructing the module's engine raises if the library or a GPU is missing.

Differences from the reference, all deliberate:

* ``get_basis`` is not rebuilt on every call: the operator tables depend only on
  ``(T, num_basis, tau)`` and are cached (``basis_maps.build_plan``).
* The density-pickle side effect of the Video-LLaMA variant (reference :320-345, writes
  ``./alphas_uniform`` on every call, result unused by the model) is not reproduced.
* The Gibbs draw consumes torch's global **CPU** generator exactly like the reference's CPU path:
  512 float64 uniforms for the bin draw, then 512 more for the degenerate in-bin draw (:204-206).
  Seed it with ``torch.manual_seed`` to reproduce the reference CPU run draw for draw.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
import torch.nn as nn

from .basis_maps import NB_SAMPLES
import ctypes as C

from .engine import TOKEN_DTYPES, LTMEngine

# raw current-stream handle / current device index: torch's C bindings when present (torch >= 1.10), the public API otherwise
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device

# one-entry cache of pooled frames: the Q-former calls every layer's LTM with the same
# encoder_hidden_states tensor OBJECT, so the frame tokens (25 MB at the headline shape) are read once per
# chunk.  Keyed by a weak reference to that object (an address could be recycled by the allocator) + its version.
_pool_cache = {"ref": None, "version": -1, "kbar": None}


class LongTermAttention(nn.Module):
    tokens_per_frame = 32      # reference :291,304 hard-codes k.reshape(B, klen, 32, 768)
    encoder_width = 768

    def __init__(self, head_size: int, length: int, target_len: int, attn_func: str, attn_num_basis: int,
                 continuous: bool, attn_drop: float, infinite_memory: bool, n_layers: int,
                 n_heads: int, affines: bool, mask: bool, mask_type: str, kl_regularizer: bool, proj_key, proj_value,
                 sigma_0, mu_0, sticky_memories, sigmas, tau, **kwargs):
        super().__init__()
        # the attribute set of the reference constructor (:32-65)
        self.device = 'cuda'
        self.length = length
        self.target_len = target_len
        self.head_size = head_size
        self.attn_num_basis = attn_num_basis
        self.continuous = continuous
        self.attn_func = attn_func
        self.n_head = n_heads
        self.sigmas = sigmas
        self.kl_regularizer = kl_regularizer
        self.sigma_0 = sigma_0
        self.mu_0 = mu_0
        # borrowed references to the cross-attention's own key/value Linear (Qformer.py:156-157);
        # kept out of this module's parameter list, exactly like a plain attribute would not be... they
        # are nn.Modules, so they do register -- as in the reference.
        self.proj_key = proj_key
        self.proj_value = proj_value
        self.affines = affines
        self.sticky_memories = sticky_memories
        self.mem_threshold = 2048
        self.infinite_memory = infinite_memory
        self.nb_samples = NB_SAMPLES
        self.tau = tau
        self.count = 0
        self.x_past = None            # never read by the reference either (:221)
        self.ridge_penalty = 0.5
        self.padding = True
        self.spacing = 'linear'
        self.d_model = kwargs.get("d_model", n_heads * head_size)
        if not continuous:
            raise NotImplementedError("only the continuous-attention path exists in the reference (forward :290)")
        if attn_func != "softmax":
            raise NotImplementedError("the active reference path is the softmax density (:248)")
        if tau is None or attn_num_basis is None:
            # the image Q-former builds LTM objects with num_basis/tau = None and never calls them
            # (blip2.py:48-65, infinityqa.py:265); allow construction, refuse forward.
            self._callable = False
        else:
            self._callable = True
        self._engine: Optional[LTMEngine] = None

    # ------------------------------------------------------------------ engine
    def _get_engine(self, device: torch.device, Q: int) -> LTMEngine:
        eng = self._engine
        if eng is None or eng.device != device or eng.max_q < Q:
            carried = None
            if eng is not None and eng.has_memory:
                if eng.device != device:
                    raise RuntimeError("LongTermAttention moved device mid-document")
                carried = eng.export_state(0)             # a longer query mid-document: carry the memory into a larger engine
            self._engine = eng = LTMEngine(
                self.attn_num_basis, self.n_head, self.head_size, self.encoder_width, self.tokens_per_frame,
                tau=self.tau, sticky=bool(self.sticky_memories), n_layers=1, max_q=max(Q, 32), device=device,
                nb_samples=self.nb_samples)
            if carried is not None:
                eng.import_state(0, carried[0].contiguous(), carried[1].contiguous(), self._proj(device))
        return eng

    def _proj(self, device):
        # the borrowed key/value Linear layers rarely change between calls: reuse the fp32 views while the parameters are the
        # same tensors at the same in-place version
        pk, pv = self.proj_key, self.proj_value
        stamp = (pk.weight.data_ptr(), pk.weight._version, pv.weight.data_ptr(), pv.weight._version,
                 None if pk.bias is None else (pk.bias.data_ptr(), pk.bias._version),
                 None if pv.bias is None else (pv.bias.data_ptr(), pv.bias._version), device)
        cached = getattr(self, "_proj_views", None)
        if cached is not None and cached[0] == stamp:
            return cached[1]
        views = self._proj_build(device)
        # cache only views that ALIAS the parameters (fp32, contiguous, on the device): a converted copy would go stale when
        # the parameter is rewritten through ``.data`` (which does not bump ``_version``)
        params = [t for t in (pk.weight, pk.bias, pv.weight, pv.bias) if t is not None]
        aliased = all(any(v.data_ptr() == t.data_ptr() for t in params) for v in views) and len(params) == 4
        self._proj_views = (stamp, views) if aliased else None
        return views

    def _proj_build(self, device):
        def f32(t):
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
                t = t.to(device=device, dtype=torch.float32).contiguous()
            return t
        dm, d = self.n_head * self.head_size, self.encoder_width
        bk = self.proj_key.bias if self.proj_key.bias is not None else torch.zeros(dm, device=device)
        bv = self.proj_value.bias if self.proj_value.bias is not None else torch.zeros(dm, device=device)
        wk, wv = f32(self.proj_key.weight), f32(self.proj_value.weight)
        if tuple(wk.shape) != (dm, d) or tuple(wv.shape) != (dm, d):
            raise ValueError(f"proj_key/proj_value must map {d} -> {dm}")
        return (wk, f32(bk), wv, f32(bv))

    # ------------------------------------------------------------------ reference-visible state
    @property
    def B_past(self) -> Optional[torch.Tensor]:
        """Coefficient matrix [1, N, d] after the last call, ``None`` before the first / after new_doc."""
        if self._engine is None or not self._engine.has_memory:
            return None
        return self._engine.export_state(0)[0].unsqueeze(0)

    @B_past.setter
    def B_past(self, value):
        if value is not None:
            raise AttributeError("B_past can only be cleared (set to None); use engine.import_state to load a memory")
        if self._engine is not None:
            self._engine.reset()

    # ------------------------------------------------------------------ persistence of the consolidated memory
    def memory_state(self) -> Optional[dict]:
        """Everything needed to continue the document later or elsewhere: the coefficient matrix ``B_past``
        and the sticky bin masses derived from the last scores (the reference never serialises its LTM state;
        SURVEY.md section 5).  ``None`` if the memory is empty.  Tensors are on the CPU."""
        if self._engine is None or not self._engine.has_memory:
            return None
        B, mass = self._engine.export_state(0)
        return {"B_past": B.cpu(), "bin_mass": mass.cpu(), "num_basis": self.attn_num_basis, "tau": self.tau,
                "sticky": bool(self.sticky_memories), "version": 1}

    def load_memory_state(self, state: Optional[dict], device, max_q: Optional[int] = None) -> None:
        """Inverse of :meth:`memory_state` (``None`` clears the memory).  The projected memory is rebuilt
        from ``B_past`` with the current key/value weights.  A memory only continues correctly under the knobs it was
        consolidated with: ``num_basis``, ``tau`` and ``sticky_memories`` must match this module's."""
        device = torch.device(device)
        if state is None:
            if self._engine is not None:
                self._engine.reset()
            return
        if state.get("version") != 1 or state["num_basis"] != self.attn_num_basis:
            raise ValueError("memory state does not match this LongTermAttention (version / num_basis)")
        if abs(float(state["tau"]) - float(self.tau)) > 1e-12 or bool(state["sticky"]) != bool(self.sticky_memories):
            raise ValueError(f"memory was consolidated with tau={state['tau']}, sticky={bool(state['sticky'])}; this module has "
                             f"tau={self.tau}, sticky={bool(self.sticky_memories)}")
        if tuple(state["B_past"].shape) != (self.attn_num_basis, self.encoder_width) or state["bin_mass"].numel() != 127:
            raise ValueError("memory state tensors have the wrong shape")
        want_q = max_q if max_q is not None else (self._engine.max_q if self._engine is not None else 32)
        eng = self._get_engine(device, want_q)
        eng.reset()
        eng.import_state(0, state["B_past"].to(device=device, dtype=torch.float32).contiguous(),
                         state["bin_mass"].to(device=device, dtype=torch.float32).contiguous(), self._proj(device))

    _U_RING = 64            # pinned slots of Gibbs uniforms; an event every _U_GROUP calls guards their reuse
    _U_GROUP = 16
    _U_VIA_COPY = False     # True: H2D copy of the 512 uniforms per call instead of letting the draw read the pinned slot

    def _draw_uniforms(self, device) -> int:
        """512 float64 uniforms for the bin draw + 512 for the degenerate in-bin draw (:204-206) from the global CPU generator
        (one ``torch.rand(1024)``: the same stream of draws as the reference's two calls), written into a slot of a pinned
        ring.  Returns the HOST address of the first 512: the draw kernel reads them through the pinned mapping, so there is
        no copy to issue; a slot is only rewritten after the event of the group that last used it has completed."""
        ring = getattr(self, "_u_ring", None)
        if ring is None or ring["device"] != device:
            S = self.nb_samples
            pin = torch.empty(self._U_RING, 2 * S, dtype=torch.float64).pin_memory()
            ring = {"device": device, "i": 0, "pin": pin, "rows": [pin[j] for j in range(self._U_RING)],
                    "addr": [pin[j].data_ptr() for j in range(self._U_RING)],
                    "ev": [None] * (self._U_RING // self._U_GROUP)}
            self._u_ring = ring
        i = ring["i"]
        ring["i"] = (i + 1) % self._U_RING
        g = i // self._U_GROUP
        if i % self._U_GROUP == 0 and ring["ev"][g] is not None:
            ring["ev"][g].synchronize()                   # the steps that read this group's slots have finished
        torch.rand(2 * self.nb_samples, dtype=torch.float64, out=ring["rows"][i])
        if self._U_VIA_COPY:
            # staged into a device slot of the same ring: the draw then reads HBM instead of crossing PCIe inside the kernel
            dev = ring.get("dev")
            if dev is None:
                dev = ring["dev"] = torch.empty(self._U_RING, self.nb_samples, dtype=torch.float64, device=device)
                ring["dev_rows"] = [dev[j] for j in range(self._U_RING)]
                ring["dev_addr"] = [dev[j].data_ptr() for j in range(self._U_RING)]
                ring["pin_first"] = [ring["rows"][j][:self.nb_samples] for j in range(self._U_RING)]
            ring["dev_rows"][i].copy_(ring["pin_first"][i], non_blocking=True)
            return ring["dev_addr"][i]
        return ring["addr"][i]

    def _uniforms_used(self, device):
        """Call after the step that reads the slot handed out last: every _U_GROUP-th slot records the group's event."""
        ring = self._u_ring
        last = (ring["i"] - 1) % self._U_RING
        if last % self._U_GROUP == self._U_GROUP - 1:
            g = last // self._U_GROUP
            ev = ring["ev"][g] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            ring["ev"][g] = ev

    # ------------------------------------------------------------------ forward
    def forward(self, k, q, new_doc, layer_n):
        """k [1, T*P, d] frame tokens, q [1, Q, H*dh] -> [1, Q, H*dh]   (reference :288-346)."""
        # validation runs once per call signature; the steady-state call (same shapes / dtypes / device) skips it
        sig = (k.shape, k.dtype, k.device, q.shape, q.dtype, q.device)
        if sig != getattr(self, "_sig", None):
            if not self._callable:
                raise RuntimeError("this LongTermAttention was built with num_basis/tau = None (image Q-former) "
                                   "and must not be called")
            if not k.is_cuda:
                raise RuntimeError("LongTermAttention runs on the HIP device only (no CPU fallback)")
            if k.size(0) != 1 or q.size(0) != 1:
                raise ValueError("batch size must be 1 (reference :208,346)")
            P, d = self.tokens_per_frame, self.encoder_width
            if k.size(2) != d or k.size(1) % P:
                raise ValueError(f"k must be [1, T*{P}, {d}]")
            if q.device != k.device or q.size(2) != self.n_head * self.head_size:
                raise ValueError(f"q must be [1, Q, {self.n_head * self.head_size}] on {k.device}")
            self._get_engine(k.device, q.size(1)).ensure_plan(k.size(1) // P)
            self._sig = sig
        self.device = k.device
        P = self.tokens_per_frame
        klen = k.size(1) // P
        self.length = klen
        qlen = q.size(1)
        eng = self._get_engine(k.device, qlen)
        if new_doc or not self.infinite_memory:
            eng.reset()                                   # :300-302 (and the non-infinite branch :310)
        # (raw handle of the device's current stream and the current-device check through torch._C: the public wrappers cost 2 us
        #  and 1 us per use on a path whose whole host side is ~35 us)
        dev_index = k.device.index if k.device.index is not None else torch.cuda.current_device()
        stream = C.c_void_p(_raw_stream(dev_index))
        other_device = _cur_device() != dev_index
        # torch.multinomial on the CPU path draws its uniforms from the global CPU generator
        sticky_step = eng.has_memory and self.sticky_memories
        u_addr = self._draw_uniforms(k.device) if sticky_step else 0
        qf = q.detach()
        if qf.dtype != torch.float32 or not qf.is_contiguous():
            qf = qf.float().contiguous()
        ctx = torch.empty(1, qlen, self.n_head * self.head_size, device=k.device, dtype=torch.float32)
        proj_arr = eng._proj_array([self._proj(k.device)])
        ref = _pool_cache["ref"]
        if ref is not None and ref() is k and _pool_cache["version"] == k._version:
            # the frames of this k are pooled already (the chunk's other cross-attention layer did it): step from them
            kbar = _pool_cache["kbar"]
            if other_device:
                with torch.cuda.device(k.device):
                    eng.step_raw(kbar.data_ptr(), klen, qf.data_ptr(), qlen, proj_arr, u_addr, ctx.data_ptr(), stream)
            else:
                eng.step_raw(kbar.data_ptr(), klen, qf.data_ptr(), qlen, proj_arr, u_addr, ctx.data_ptr(), stream)
        else:
            kf = k
            if kf.dtype not in TOKEN_DTYPES:                      # bf16 tokens are pooled as they are
                kf = kf.float()
            if not kf.is_contiguous():
                kf = kf.contiguous()
            kbar = torch.empty(klen, self.encoder_width, device=k.device, dtype=torch.float32)
            # pool (:304) + step (:306-346) behind ONE C call
            if other_device:
                with torch.cuda.device(k.device):
                    eng.forward_into_raw(kf.data_ptr(), TOKEN_DTYPES[kf.dtype], klen, kbar.data_ptr(), qf.data_ptr(), qlen, proj_arr,
                                         u_addr, ctx.data_ptr(), stream)
            else:
                eng.forward_into_raw(kf.data_ptr(), TOKEN_DTYPES[kf.dtype], klen, kbar.data_ptr(), qf.data_ptr(), qlen, proj_arr,
                                     u_addr, ctx.data_ptr(), stream)
            _pool_cache["ref"], _pool_cache["version"], _pool_cache["kbar"] = weakref.ref(k), k._version, kbar
        if sticky_step:
            self._uniforms_used(k.device)
        self.count += 1
        return ctx if q.dtype == torch.float32 else ctx.to(q.dtype)


class LongTermAttentionVC(LongTermAttention):
    """VideoChat2 variant: frames are 14x14 UMT-L patches of width 1024 (reference
    infty-VideoChat2/models/blip2/long_term_attention_gibbs.py:291,304)."""
    tokens_per_frame = 14 * 14
    encoder_width = 1024
