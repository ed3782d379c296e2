"""Video Q-former + ``encode_video`` + per-chunk loop on MI355X (rows a11-a13, next-row f1).

Mirrors the part of the reference model that surrounds the LTM:

* ``init_video_Qformer``                 infty-Video-LLaMA/InfVideoLLaMA/models/infinityqa.py:36-55
* the slimming done after construction   infinityqa.py:202-209  (no word/position embeddings, no text FFN)
* ``encode_video``                       infinityqa.py:280-344
* the eval scripts' chunk loop           eval_code/eval/run_inference_inf_video_llama_nextqa.py:179-196,228

The parameter tree uses the reference's own module names (``bert.embeddings.LayerNorm``,
``bert.encoder.layer.N.{attention,crossattention}.{self.{query,key,value},output.{dense,LayerNorm}}``,
``intermediate_query.dense``, ``output_query.{dense,LayerNorm}``), so a reference checkpoint's ``video_Qformer.*`` /
``video_query_tokens`` / ``llama_proj.*`` / ``video_frame_position_embedding.*`` entries load with
``load_state_dict`` unchanged.  The modules are parameter holders only: the arithmetic of a chunk runs as one
C-ABI call (``infv_vqf_encode_chunk``, include/infv_vqf.h) into the HIP kernels; there is no PyTorch fallback.

Only the configuration the video Q-former is actually run in is supported: query tokens only (no text), eval
mode, all-ones masks, batch 1, 64-wide heads.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from .basis_maps import NB_SAMPLES
from .qformer_hook import build_long_term_attention


class _Cfg:
    """The BertConfig attributes init_video_Qformer sets (infinityqa.py:37-48) plus bert-base defaults."""

    def __init__(self, **kw):
        self.hidden_size = 768
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.layer_norm_eps = 1e-12
        self.add_cross_attention = True
        self.cross_attention_freq = 1
        self.__dict__.update(kw)


class _SelfOutput(nn.Module):                        # Qformer.py:315-326
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _SelfAttention(nn.Module):                     # Qformer.py:115-175
    def __init__(self, cfg, is_cross_attention: bool):
        super().__init__()
        self.is_cross_attention = is_cross_attention
        self.alpha = cfg.alpha
        self.num_attention_heads = cfg.num_attention_heads
        self.attention_head_size = cfg.hidden_size // cfg.num_attention_heads
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        kv_in = cfg.encoder_width if is_cross_attention else cfg.hidden_size
        self.key = nn.Linear(kv_in, cfg.hidden_size)
        self.value = nn.Linear(kv_in, cfg.hidden_size)
        if is_cross_attention:
            self.long_term_attention = build_long_term_attention(cfg, self.key, self.value, self.num_attention_heads,
                                                                 self.attention_head_size)


class _Attention(nn.Module):                         # Qformer.py:329-334
    def __init__(self, cfg, is_cross_attention=False):
        super().__init__()
        self.self = _SelfAttention(cfg, is_cross_attention)
        self.output = _SelfOutput(cfg)


class _Intermediate(nn.Module):                      # Qformer.py:389-403
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)


class _Output(nn.Module):                            # Qformer.py:406-418
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.intermediate_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _Layer(nn.Module):                             # Qformer.py:419-441 after infinityqa.py:206-208
    def __init__(self, cfg, layer_num):
        super().__init__()
        self.layer_num = layer_num
        self.attention = _Attention(cfg)
        self.crossattention = _Attention(cfg, is_cross_attention=True)
        self.has_cross_attention = True
        self.intermediate_query = _Intermediate(cfg)
        self.output_query = _Output(cfg)


class _Embeddings(nn.Module):                        # Qformer.py:55-83 after infinityqa.py:203-205
    def __init__(self, cfg):
        super().__init__()
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg, i) for i in range(cfg.num_hidden_layers)])


class _Bert(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)


class VideoQformer(nn.Module):
    """``video_Qformer`` (the reference's BertLMHeadModel with ``cls = None``): only ``.bert`` remains."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.bert = _Bert(cfg)

    @property
    def ltm_modules(self):
        return [layer.crossattention.self.long_term_attention for layer in self.bert.encoder.layer]


def init_video_Qformer(num_query_token: int, vision_width: int, num_hidden_layers: int = 2, sticky: bool = True,
                       num_basis: int = 256, sigmas=(0.005, 0.01), tau: float = 0.75, alpha: float = 0.9
                       ) -> Tuple[VideoQformer, nn.Parameter]:
    """Same signature and return value as the reference classmethod (infinityqa.py:36-55)."""
    cfg = _Cfg(num_hidden_layers=num_hidden_layers, encoder_width=vision_width, query_length=num_query_token,
               sticky=sticky, num_basis=num_basis, sigmas=list(sigmas), tau=tau, alpha=alpha, initializer_range=0.02)
    qformer = VideoQformer(cfg)
    query_tokens = nn.Parameter(torch.zeros(1, num_query_token, cfg.hidden_size))
    query_tokens.data.normal_(mean=0.0, std=cfg.initializer_range)
    return qformer, query_tokens


class ShortMemoryBuffer:
    """Producer side of ``encode_video``: the frame tokens of the current video fragment, kept in the layout the
    frame mean-pool reads -- one contiguous ``[T, P, d]`` block -- instead of the reference's Python list of per-frame
    tensors that ``encode_video`` unsqueezes, concatenates and rearranges on every call (infinityqa.py:251-278 fills the
    list, :285,306-323 rebuilds the tensor).

    * ``replace(q_hidden_state, n_frame)`` is ``encode_short_memory_frame``'s epilogue (:270-278): the buffer is
      emptied and the frames ``cur_frame <= n_frame`` of the image Q-former's output ``[F, P, d]`` are kept
      (so at most ``n_frame + 1`` of them, as in the reference).
    * ``frames()`` applies ``encode_video``'s cap (:286-288,306-307): with ``n_position = min(32, ceil(sqrt(T)))``
      only the newest ``n_position**2`` frames survive; it returns them as ``[1, T*P, d]`` without copying.

    ``dtype=torch.bfloat16`` stores the tokens at half the bytes: ``infv_ltm_pool`` / ``infv_ltm_consolidate`` read them
    directly (``infv_ltm_set_token_dtype``); the result then differs from the fp32-token run by the rounding of the
    tokens (off by default, never used for the headline number)."""

    def __init__(self, tokens_per_frame: int, width: int, capacity_frames: int = 2049, dtype=torch.float32,
                 device=None):
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("ShortMemoryBuffer holds float32 or bfloat16 tokens")
        self.P, self.d = int(tokens_per_frame), int(width)
        self.store = torch.empty(int(capacity_frames), self.P, self.d, dtype=dtype, device=device)
        self.n = 0

    def __len__(self) -> int:
        return self.n

    def clear(self):
        self.n = 0

    def replace(self, q_hidden_state: torch.Tensor, n_frame: int = 2048):
        if q_hidden_state.dim() != 3 or tuple(q_hidden_state.shape[1:]) != (self.P, self.d):
            raise ValueError(f"q_hidden_state must be [F, {self.P}, {self.d}]")
        keep = min(q_hidden_state.size(0), int(n_frame) + 1)            # frames with cur_frame <= n_frame (:275)
        if keep > self.store.size(0):
            raise ValueError(f"{keep} frames exceed the buffer's capacity of {self.store.size(0)}")
        self.store[:keep].copy_(q_hidden_state[:keep])                   # cast + layout in one copy, no list, no cat
        self.n = keep
        return self

    @staticmethod
    def frame_cap(n_frames: int):
        """(n_position, frames kept) of encode_video for a buffer of ``n_frames`` frames (:286-288,306-307)."""
        n_position = min(32, math.ceil(math.sqrt(n_frames)))
        return n_position, min(n_frames, n_position * n_position)

    def frames(self) -> torch.Tensor:
        if self.n == 0:
            raise RuntimeError("short_memory_buffer is empty")
        _, kept = self.frame_cap(self.n)
        return self.store[self.n - kept:self.n].reshape(1, kept * self.P, self.d)     # the oldest frames are dropped


def _dev_f32(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
        t = t.to(device=device, dtype=torch.float32).contiguous()
    return t


class InfVideoEncoder(nn.Module):
    """The in-scope slice of the reference model ``InfinityQA``: the members ``encode_video`` touches
    (infinityqa.py:195-209,217-236) and ``encode_video`` itself.  The ViT / image Q-former producer
    (``encode_short_memory_frame``) and the LLM are out of scope: fill ``short_memory_buffer`` with per-frame
    token blocks ``[num_query_token, hidden]`` and consume ``inputs_llama``."""

    def __init__(self, num_video_query_token: int = 32, hidden_size: int = 768, llama_hidden: int = 4096,
                 max_frame_pos: int = 32, sticky: bool = True, num_basis: int = 256, sigmas=(0.005, 0.01),
                 tau: float = 0.75, alpha: float = 0.9, num_hidden_layers: int = 2, tokens_per_frame: int = 32):
        super().__init__()
        self.video_frame_position_embedding = nn.Embedding(max_frame_pos, hidden_size)   # used as a flag only (Qformer.py:216)
        self.num_video_query_token = num_video_query_token
        self.video_Qformer, self.video_query_tokens = init_video_Qformer(
            num_video_query_token, hidden_size, num_hidden_layers, sticky, num_basis, sigmas, tau, alpha)
        self.llama_proj = nn.Linear(hidden_size, llama_hidden)
        self.short_memory_buffer: List[torch.Tensor] = []
        self.sticky = sticky
        self.n_position = 8
        self.tokens_per_frame = tokens_per_frame
        self.eval()
        for p in self.parameters():
            p.requires_grad = False
        self._vqf = None
        self._vqf_dev = None
        self.exact_fp32 = False        # True: exact-fp32 MFMA for the short-term attention instead of split-bf16
        self.last_hidden: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ weights
    def load_reference_state_dict(self, sd: dict, strict: bool = True):
        """Load tensors named as in the reference model's checkpoint.  Accepts both the model-level names
        (``video_Qformer.bert...``) and the bare ``bert...`` names of ``synth.video_qformer_weights``."""
        own = {}
        for k, v in sd.items():
            v = torch.as_tensor(v)
            if k.startswith("bert."):
                k = "video_Qformer." + k
            own[k] = v
        res = self.load_state_dict(own, strict=False)
        # long_term_attention.proj_{key,value} alias the layer's key/value Linear (Qformer.py:156-157)
        missing = [k for k in res.missing_keys if ".long_term_attention.proj_" not in k
                   and not k.startswith("video_frame_position_embedding")]
        if strict and (missing or res.unexpected_keys):
            raise KeyError(f"missing {missing}, unexpected {res.unexpected_keys}")
        return res

    def _weight_sources(self):
        """The parameters in the order :meth:`_weights` visits them."""
        out = [self.video_query_tokens]
        bert = self.video_Qformer.bert

        def lin(m):
            out.extend((m.weight, m.bias))
        lin(bert.embeddings.LayerNorm)
        for layer in bert.encoder.layer:
            a, x = layer.attention, layer.crossattention
            for m in (a.self.query, a.self.key, a.self.value, a.output.dense, a.output.LayerNorm,
                      x.self.query, x.self.key, x.self.value, x.output.dense, x.output.LayerNorm,
                      layer.intermediate_query.dense, layer.output_query.dense, layer.output_query.LayerNorm):
                lin(m)
        lin(self.llama_proj)
        return out

    def _weights(self, device: torch.device):
        """ctypes view of the parameters.  Rebuilt only when a parameter moved or was updated in place (the signature
        is every parameter's storage address and version), so a steady-state call costs one pass over ~55 tensors."""
        src = self._weight_sources()
        sig = (str(device),) + tuple((p.data_ptr(), p._version, p.dtype) for p in src)
        cached = getattr(self, "_w_cache", None)
        if cached is not None and cached[0] == sig:
            return cached[1], cached[2]
        w, keep = self._build_weights(device)
        self._w_cache = (sig, w, keep)
        return w, keep

    def _build_weights(self, device: torch.device):
        keep = []                                             # keeps converted copies alive while the struct is cached

        def t(x):
            y = _dev_f32(x, device)
            keep.append(y)
            return y.data_ptr()

        def lin(m: nn.Linear):
            return _lib.Linear(t(m.weight), t(m.bias))

        def ln(m: nn.LayerNorm):
            return _lib.LayerNorm(t(m.weight), t(m.bias))

        w = _lib.VqfWeights()
        w.query_tokens = t(self.video_query_tokens)
        bert = self.video_Qformer.bert
        w.emb_ln = ln(bert.embeddings.LayerNorm)
        for l, layer in enumerate(bert.encoder.layer):
            a, x = layer.attention, layer.crossattention
            w.layer[l] = _lib.VqfLayer(
                lin(a.self.query), lin(a.self.key), lin(a.self.value), lin(a.output.dense), ln(a.output.LayerNorm),
                lin(x.self.query), lin(x.self.key), lin(x.self.value), lin(x.output.dense), ln(x.output.LayerNorm),
                lin(layer.intermediate_query.dense), lin(layer.output_query.dense), ln(layer.output_query.LayerNorm))
        w.llama_proj = lin(self.llama_proj)
        return w, keep

    def _prefix_epoch(self) -> int:
        """Signature of the weights that determine layer 0's chunk-independent prefix (query tokens, embedding
        LayerNorm, self-attention block, cross query and key weights): storage address and in-place version of each.
        Non-zero; a new value tells the library to recompute its cached prefix."""
        layer = self.video_Qformer.bert.encoder.layer[0]
        a, x = layer.attention, layer.crossattention
        mods = (self.video_Qformer.bert.embeddings.LayerNorm, a.self.query, a.self.key, a.self.value, a.output.dense,
                a.output.LayerNorm, x.self.query)
        sig = [(self.video_query_tokens.data_ptr(), self.video_query_tokens._version),
               (x.self.key.weight.data_ptr(), x.self.key.weight._version)]
        for m in mods:
            sig.append((m.weight.data_ptr(), m.weight._version))
            sig.append((m.bias.data_ptr(), m.bias._version))
        return (hash(tuple(sig)) & 0x7FFFFFFFFFFFFFFF) | 1

    def _handle(self, device: torch.device):
        h = self._handle_raw(device)
        if getattr(self, "_vqf_exact", None) != bool(self.exact_fp32):
            _lib.check(_lib.load().infv_vqf_set_precision(h, int(bool(self.exact_fp32))))
            self._vqf_exact = bool(self.exact_fp32)
        return h

    def _handle_raw(self, device: torch.device):
        if self._vqf is not None and self._vqf_dev == device:
            return self._vqf
        cfg = self.video_Qformer.config
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("the video Q-former path needs a HIP device (no CPU fallback exists)")
        c = _lib.VqfConfig(cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_size, cfg.intermediate_size,
                           cfg.encoder_width, self.tokens_per_frame, self.num_video_query_token,
                           self.llama_proj.out_features, NB_SAMPLES, float(cfg.alpha), float(cfg.layer_norm_eps))
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.infv_vqf_create(C.byref(c), C.byref(h)))
        self._release()
        self._vqf, self._vqf_dev = h, device
        self._vqf_exact = None
        return h

    def _release(self):
        h, self._vqf = getattr(self, "_vqf", None), None
        if h:
            try:
                _lib.load().infv_vqf_destroy(h)
            except Exception:
                pass

    def __del__(self):
        self._release()

    # ------------------------------------------------------------------ one chunk
    def encode_frames(self, frame_hidden_state: torch.Tensor, new_video: bool,
                      u: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """frame_hidden_state [1, T*P, d] (device) -> (last_hidden_state [1, Q, hidden], inputs_llama [1, Q, llama]).
        ``u`` [n_layers, 512] float64: the Gibbs uniforms; by default drawn from torch's global CPU generator in
        the order the reference's CPU path consumes them (512 + 512 discarded per LTM call that resamples)."""
        if not frame_hidden_state.is_cuda:
            raise RuntimeError("the video Q-former path runs on the HIP device only (no CPU fallback)")
        if frame_hidden_state.dim() != 3 or frame_hidden_state.size(0) != 1:
            raise ValueError("frame_hidden_state must be [1, T*P, d] (batch 1, infinityqa.py:283)")
        device = frame_hidden_state.device
        cfg = self.video_Qformer.config
        P, Q = self.tokens_per_frame, self.num_video_query_token
        if frame_hidden_state.size(2) != cfg.encoder_width or frame_hidden_state.size(1) % P:
            raise ValueError(f"frame_hidden_state must be [1, T*{P}, {cfg.encoder_width}]")
        T = frame_hidden_state.size(1) // P
        k = _dev_f32(frame_hidden_state[0], device)
        h = self._handle(device)
        lib = _lib.load()
        use_ltm = cfg.alpha != 1.0
        handles = (C.c_void_p * cfg.num_hidden_layers)()
        if use_ltm:
            need_u = False
            for l, m in enumerate(self.video_Qformer.ltm_modules):
                m.length = m.target_len = frame_hidden_state.size(1)             # Qformer.py:218-219
                eng = m._get_engine(device, Q)
                eng.ensure_plan(T)
                handles[l] = eng._h
                need_u = need_u or (eng.has_memory and not new_video and bool(m.sticky_memories))
                m.count += 1
            if u is None and need_u:
                draws = []
                for _ in range(cfg.num_hidden_layers):
                    draws.append(torch.rand(NB_SAMPLES, dtype=torch.float64))
                    torch.rand(NB_SAMPLES, dtype=torch.float64)                  # the in-bin draw of LTM.py:206
                u = torch.stack(draws)
            if u is not None:
                u = u.to(device=device, dtype=torch.float64).contiguous()
                if tuple(u.shape) != (cfg.num_hidden_layers, NB_SAMPLES):
                    raise ValueError(f"u must be [{cfg.num_hidden_layers}, {NB_SAMPLES}]")
        w, keep = self._weights(device)
        # weights that had to be converted for this call (dtype / device / layout) live in temporaries: no reuse then
        stable = all(k.data_ptr() == p.data_ptr() for k, p in zip(keep, self._weight_sources()))
        _lib.check(lib.infv_vqf_set_weights_epoch(h, self._prefix_epoch() if stable else 0))
        hidden = torch.empty(1, Q, cfg.hidden_size, device=device, dtype=torch.float32)
        llama = torch.empty(1, Q, self.llama_proj.out_features, device=device, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        with torch.cuda.device(device):
            _lib.check(lib.infv_vqf_encode_chunk(h, handles if use_ltm else None, C.c_void_p(k.data_ptr()), T,
                                                 C.byref(w), C.c_void_p(0 if u is None else u.data_ptr()),
                                                 int(bool(new_video)), C.c_void_p(hidden.data_ptr()),
                                                 C.c_void_p(llama.data_ptr()), stream))
        del keep
        self.last_hidden = hidden
        return hidden, llama

    # ------------------------------------------------------------------ a whole video, layer-major
    def encode_frames_batch(self, frames: torch.Tensor, new_video: bool, u: Optional[torch.Tensor] = None,
                            want_hidden: bool = False):
        """frames [C, T*P, d] (C chunks of equal length, device) -> (inputs_llama per chunk [C, Q, llama],
        their mean over chunks [1, Q, llama], last_hidden_state per chunk [C, Q, hidden] or None).
        Equals C calls of :meth:`encode_frames` (``new_video`` on the first only) but runs layer-major in one C call
        (``infv_vqf_encode_video``): layer 0's LTM takes the whole-video fast path, every query-token block is batched
        over chunks.  ``u`` [C, n_layers, 512] float64; by default drawn from torch's global CPU generator in the
        order a per-chunk run of the reference would consume it."""
        if not frames.is_cuda:
            raise RuntimeError("the video Q-former path runs on the HIP device only (no CPU fallback)")
        cfg = self.video_Qformer.config
        P, Q, Ln = self.tokens_per_frame, self.num_video_query_token, cfg.num_hidden_layers
        if frames.dim() != 3 or frames.size(2) != cfg.encoder_width or frames.size(1) % P:
            raise ValueError(f"frames must be [C, T*{P}, {cfg.encoder_width}]")
        device = frames.device
        Cn, T = frames.size(0), frames.size(1) // P
        k = _dev_f32(frames, device)
        h = self._handle(device)
        lib = _lib.load()
        use_ltm = cfg.alpha != 1.0
        handles = (C.c_void_p * Ln)()
        if use_ltm:
            had_memory = False
            for l, m in enumerate(self.video_Qformer.ltm_modules):
                m.length = m.target_len = frames.size(1)
                eng = m._get_engine(device, Q)
                eng.ensure_plan(T)
                handles[l] = eng._h
                had_memory = had_memory or eng.has_memory
                m.count += Cn
            sticky = bool(self.video_Qformer.ltm_modules[0].sticky_memories)
            if u is None and sticky:
                first_draws = 0 if (had_memory and not new_video) else 1      # chunk 0 of a new video resamples nothing
                u = torch.zeros(Cn, Ln, NB_SAMPLES, dtype=torch.float64)
                for c in range(first_draws, Cn):
                    for l in range(Ln):
                        u[c, l] = torch.rand(NB_SAMPLES, dtype=torch.float64)
                        torch.rand(NB_SAMPLES, dtype=torch.float64)              # the in-bin draw of LTM.py:206
            if u is not None:
                u = u.to(device=device, dtype=torch.float64).contiguous()
                if tuple(u.shape) != (Cn, Ln, NB_SAMPLES):
                    raise ValueError(f"u must be [{Cn}, {Ln}, {NB_SAMPLES}]")
        w, keep = self._weights(device)
        hidden = torch.empty(Cn, Q, cfg.hidden_size, device=device, dtype=torch.float32) if want_hidden else None
        llama = torch.empty(Cn, Q, self.llama_proj.out_features, device=device, dtype=torch.float32)
        mean = torch.empty(1, Q, self.llama_proj.out_features, device=device, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        with torch.cuda.device(device):
            _lib.check(lib.infv_vqf_encode_video(h, handles if use_ltm else None, C.c_void_p(k.data_ptr()), Cn, T,
                                                 C.byref(w), C.c_void_p(0 if u is None else u.data_ptr()),
                                                 int(bool(new_video)),
                                                 C.c_void_p(0 if hidden is None else hidden.data_ptr()),
                                                 C.c_void_p(llama.data_ptr()), C.c_void_p(mean.data_ptr()), stream))
        del keep
        return llama, mean, hidden

    # ------------------------------------------------------------------ encode_video (infinityqa.py:280-344)
    def encode_video(self, new_video: bool = True):
        if isinstance(self.short_memory_buffer, ShortMemoryBuffer):
            # tokens already in the pooling layout: no per-frame unsqueeze / cat / rearrange (:285,317-323)
            frame_hidden_state = self.short_memory_buffer.frames()
            self.n_position, _ = ShortMemoryBuffer.frame_cap(len(self.short_memory_buffer))
            if not frame_hidden_state.is_cuda:
                raise RuntimeError("the video Q-former path runs on the HIP device only (no CPU fallback)")
            _, inputs_llama = self.encode_frames(frame_hidden_state, new_video)
            atts_llama = torch.ones(inputs_llama.size()[:-1], dtype=torch.long, device=frame_hidden_state.device)
            return inputs_llama, atts_llama
        if not self.short_memory_buffer:
            raise RuntimeError("short_memory_buffer is empty")
        buf = [f if f.dim() == 3 else f.unsqueeze(0) for f in self.short_memory_buffer]   # :285
        self.n_position = min(32, math.ceil(math.sqrt(len(buf))))                            # :286-288
        cap = self.n_position * self.n_position
        while len(buf) > cap:                                                                # :306-307
            buf.pop(0)
        self.short_memory_buffer = buf
        device = buf[0].device
        if not buf[0].is_cuda:
            raise RuntimeError("the video Q-former path runs on the HIP device only (no CPU fallback)")
        frame_hidden_state = torch.cat(buf, dim=0).reshape(1, -1, buf[0].size(-1))          # :317-323  b (t q) h
        _, inputs_llama = self.encode_frames(frame_hidden_state, new_video)
        atts_llama = torch.ones(inputs_llama.size()[:-1], dtype=torch.long, device=device)   # :343
        return inputs_llama, atts_llama


def encode_long_video(model: InfVideoEncoder, frame_tokens: torch.Tensor, max_int: int,
                      u_of_chunk=None, batched: bool = False) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """The eval scripts' loop (run_inference_inf_video_llama_nextqa.py:179-196,228): split the video's per-frame
    token blocks [F, P, d] into chunks of ``max_int`` frames (ragged tail kept), ``new_video=(i == 0)``, mean of the
    per-chunk LLM-side embeddings.  Returns (mean [1, Q, llama], per-chunk list).
    ``batched=True`` pushes the full-length chunks through the layer-major whole-video path
    (:meth:`InfVideoEncoder.encode_frames_batch`) and only a ragged tail chunk through the per-chunk path."""
    blocks = list(torch.split(frame_tokens, max_int, dim=0))
    embs: List[torch.Tensor] = []
    start = 0
    if batched:
        n_full = sum(1 for b in blocks if b.size(0) == max_int)
        if n_full > 0:
            full = frame_tokens[:n_full * max_int].reshape(n_full, -1, frame_tokens.size(-1))
            u = None
            if u_of_chunk is not None:
                u = torch.stack([torch.as_tensor(u_of_chunk(i)) for i in range(n_full)])
            llama, _, _ = model.encode_frames_batch(full, new_video=True, u=u)
            embs.extend(llama[i:i + 1] for i in range(n_full))
            start = n_full
    for i in range(start, len(blocks)):
        blk = blocks[i]
        if u_of_chunk is None:
            model.short_memory_buffer = list(blk)
            emb, _ = model.encode_video(new_video=(i == 0))
        else:
            _, emb = model.encode_frames(blk.reshape(1, -1, blk.size(-1)), new_video=(i == 0), u=u_of_chunk(i))
        embs.append(emb)
    stacked = torch.stack(embs).contiguous()
    out = torch.empty_like(embs[0])
    lib = _lib.load()
    dev = stacked.device
    with torch.cuda.device(dev):
        _lib.check(lib.infv_vqf_mean(C.c_void_p(stacked.data_ptr()), len(embs), embs[0].numel(),
                                     C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out, embs


# ---------------------------------------------------------------------------------------------- multi-GPU (chunk blocks)
def gather_video_embeddings(emb_sum: torch.Tensor, n_chunks: int, memories: Sequence[torch.Tensor], group=None):
    """One all-gather (RCCL over xGMI under backend ``nccl``) of what the LLM side needs from every rank's block of
    chunks: the SUM of its per-chunk embeddings with the chunk count (the eval loop's reduction is a plain mean over
    chunks, run_inference_inf_video_llama_nextqa.py:194, so block sums + counts reproduce it exactly) and the rank's
    consolidated memories (flat tensors, e.g. ``B_past`` / bin masses per layer).
    Returns (global mean embedding, list over ranks of the memory tensors, per-rank chunk counts).
    Works without an initialised process group (world = 1, no collective)."""
    import torch.distributed as dist
    shapes = [tuple(m.shape) for m in memories]
    flat = [emb_sum.reshape(-1).float(), torch.tensor([float(n_chunks)], device=emb_sum.device)] + \
           [m.reshape(-1).float() for m in memories]
    payload = torch.cat(flat).contiguous()
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1:
        gathered = torch.empty(world * payload.numel(), device=payload.device, dtype=payload.dtype)
        dist.all_gather_into_tensor(gathered, payload, group=group)
    else:
        gathered = payload
    gathered = gathered.reshape(world, -1)
    n_emb = emb_sum.numel()
    counts = gathered[:, n_emb]
    mean = (gathered[:, :n_emb].sum(0) / counts.sum()).reshape(emb_sum.shape)
    mems, off = [], n_emb + 1
    per_rank = []
    for r in range(world):
        o, items = off, []
        for shp in shapes:
            n = int(torch.tensor(shp).prod()) if len(shp) else 1
            items.append(gathered[r, o:o + n].reshape(shp))
            o += n
        per_rank.append(items)
    return mean, per_rank, counts


def encode_long_video_sharded(model: InfVideoEncoder, frames_local: torch.Tensor, u_local: Optional[torch.Tensor] = None,
                              group=None):
    """This rank's contiguous block of chunks [C_local, T*P, d] through the layer-major path as its own document
    (``new_video=True``, the reference's semantics for a new video), then one all-gather.
    Returns (mean LLM-side embedding over ALL ranks' chunks [1, Q, llama], per-rank memories, per-chunk embeddings of
    this rank)."""
    llama, _, _ = model.encode_frames_batch(frames_local, new_video=True, u=u_local)
    mems = []
    if model.video_Qformer.config.alpha != 1.0:
        for m in model.video_Qformer.ltm_modules:
            st = m.memory_state()
            mems.extend((st["B_past"].to(llama.device), st["bin_mass"].to(llama.device)))
    mean, per_rank, _ = gather_video_embeddings(llama.sum(0, keepdim=True), llama.size(0), mems, group)
    return mean, per_rank, llama
