"""VideoChat2 binding of the LTM path (BASELINE configs[4]).

The VideoChat2 Q-former (infty-VideoChat2/models/blip2/Qformer.py) is a 12-layer bert-base encoder with a
cross-attention in every second layer (``cross_attention_freq = 2``, blip2.py:62): six ``LongTermAttention``
instances, the hook fires on every cross-attention (no ``position_embedding_ext`` test, Qformer.py:215-222) with
``sigmas = 1``; frames are 14x14 UMT-L patches of width 1024 (blip2/long_term_attention_gibbs.py:291,304); the
cross-attention query is the 96 query tokens (32 + 64 extra, configs/config_mistral.json), the instruction's text
tokens only ride along in the self-attention and have their own FFN (Qformer.py:473-505).

What runs where (the hook, the merge and the op are the path; "rest of BERT stays
stock PyTorch-ROCm"):

* HIP, through the C ABI -- per cross-attention layer: the LTM operator (``LongTermAttentionVC`` ->
  ``infv_ltm_forward``; the six layers share ONE pooled copy of the chunk's frames), the short-term cross-attention
  over the chunk's ``T*196`` tokens re-associated around the frame tokens, and the merge
  ``alpha * short + (1 - alpha) * long`` fused into its epilogue (``infv_vqf_short_attention``);
* stock PyTorch on the same device -- the BERT scaffolding around it: self-attention over the 96 + text tokens, the
  attention output blocks, the two FFNs.

``VideoChat2Encoder.encode_tokens`` is the counterpart of ``encode_img`` after the vision encoder
(videochat2_it_mistral.py:196-252), ``encode_long_video_vc`` that of the eval loop ``infer_*_inf``
(eval_code/run_nextqa_mistral.py:141-152).  Parameters live under the reference's state-dict names.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .basis_maps import NB_SAMPLES
from .qformer_hook import LongTermMemoryHook


class _Cfg:
    """The BertConfig attributes Blip2Base.init_Qformer sets (blip2.py:54-66) plus bert-base defaults."""

    def __init__(self, **kw):
        self.hidden_size = 768
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.layer_norm_eps = 1e-12
        self.num_hidden_layers = 12
        self.add_cross_attention = True
        self.cross_attention_freq = 2
        self.sigmas = 1
        self.__dict__.update(kw)


class _SelfOutput(nn.Module):                         # Qformer.py:313-324
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dense(hidden_states) + input_tensor)


class _SelfAttention(nn.Module):                      # Qformer.py:115-175
    def __init__(self, cfg, is_cross_attention: bool):
        super().__init__()
        self.is_cross_attention = is_cross_attention
        self.H = cfg.num_attention_heads
        self.dh = cfg.hidden_size // cfg.num_attention_heads
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        kv_in = cfg.encoder_width if is_cross_attention else cfg.hidden_size
        self.key = nn.Linear(kv_in, cfg.hidden_size)
        self.value = nn.Linear(kv_in, cfg.hidden_size)
        if is_cross_attention:
            # construction, call and merge rules of Qformer.py:135-159,215-222,302-303 (VideoChat2 variant)
            hook = LongTermMemoryHook(cfg, self.key, self.value, self.H, self.dh, variant="VC")
            object.__setattr__(self, "_hook", hook)           # keep the state-dict names of the reference:
            self.long_term_attention = hook.long_term_attention   # ...crossattention.self.long_term_attention

    def self_attention(self, hidden_states):              # Qformer.py:232-300 with all-ones masks, eval mode
        n = hidden_states.size(1)
        split = lambda x: x.view(1, n, self.H, self.dh).permute(0, 2, 1, 3)
        q, k, v = split(self.query(hidden_states)), split(self.key(hidden_states)), split(self.value(hidden_states))
        probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.dh), dim=-1)
        return torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(1, n, self.H * self.dh)


class _Attention(nn.Module):                          # Qformer.py:327-332
    def __init__(self, cfg, is_cross_attention=False):
        super().__init__()
        self.self = _SelfAttention(cfg, is_cross_attention)
        self.output = _SelfOutput(cfg)


class _Intermediate(nn.Module):                       # Qformer.py:387-401
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)

    def forward(self, x):
        return F.gelu(self.dense(x))


class _Output(nn.Module):                             # Qformer.py:404-416
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.intermediate_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dense(hidden_states) + input_tensor)


class _Layer(nn.Module):                              # Qformer.py:419-441
    def __init__(self, cfg, layer_num):
        super().__init__()
        self.layer_num = layer_num
        self.attention = _Attention(cfg)
        self.has_cross_attention = cfg.add_cross_attention and layer_num % cfg.cross_attention_freq == 0
        if self.has_cross_attention:
            self.crossattention = _Attention(cfg, is_cross_attention=True)
        self.intermediate = _Intermediate(cfg)
        self.output = _Output(cfg)
        self.intermediate_query = _Intermediate(cfg)
        self.output_query = _Output(cfg)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg, i) for i in range(cfg.num_hidden_layers)])


class _Bert(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.encoder = _Encoder(cfg)


class VideoChat2Qformer(nn.Module):
    """``self.qformer`` of VideoChat2_it_mistral with ``cls = None`` (videochat2_it_mistral.py:64-80): ``.bert.encoder``."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.bert = _Bert(cfg)

    @property
    def cross_layers(self) -> List[_Layer]:
        return [l for l in self.bert.encoder.layer if l.has_cross_attention]

    @property
    def ltm_modules(self):
        return [l.crossattention.self.long_term_attention for l in self.cross_layers]


def init_Qformer(num_query_token: int, vision_width: int, tau: float, alpha: float, sticky: bool, num_basis: int,
                 num_hidden_layers: int = 12, cross_attention_freq: int = 2):
    """Counterpart of ``Blip2Base.init_Qformer`` (blip2.py:47-77): (Q-former, query tokens)."""
    cfg = _Cfg(encoder_width=vision_width, sticky=sticky, num_basis=num_basis, tau=tau, alpha=alpha,
               cross_attention_freq=cross_attention_freq, query_length=num_query_token,
               num_hidden_layers=num_hidden_layers)
    qformer = VideoChat2Qformer(cfg)
    query_tokens = nn.Parameter(torch.zeros(1, num_query_token, cfg.hidden_size))
    nn.init.normal_(query_tokens, mean=0.0, std=0.02)
    return qformer, query_tokens


class VideoChat2Encoder(nn.Module):
    """The part of ``VideoChat2_it_mistral`` between the vision encoder and the LLM: Q-former (with the LTM in every
    cross-attention) + ``mistral_proj`` on the query part (videochat2_it_mistral.py:64-69,199-252)."""

    tokens_per_frame = 14 * 14

    def __init__(self, num_query_token: int = 32, extra_num_query_token: int = 64, vision_width: int = 1024,
                 llm_hidden: int = 4096, num_basis: int = 256, sticky: bool = True, tau: float = 0.75,
                 alpha: float = 0.75, num_hidden_layers: int = 12, cross_attention_freq: int = 2):
        super().__init__()
        self.n_query = num_query_token + extra_num_query_token
        self.qformer, self.query_tokens = init_Qformer(self.n_query, vision_width, tau, alpha, sticky, num_basis,
                                                       num_hidden_layers, cross_attention_freq)
        self.mistral_proj = nn.Linear(self.qformer.config.hidden_size, llm_hidden)
        self._vqf = None
        self._vqf_dev = None

    # ------------------------------------------------------------------ weights
    def load_reference_state_dict(self, sd: dict, strict: bool = True):
        """``sd`` under the reference's names: ``bert.encoder.layer.*``, ``query_tokens`` (query + extra query tokens
        concatenated, videochat2_it_mistral.py:199-203), ``mistral_proj.*``."""
        t = lambda v: torch.as_tensor(v)
        own = {"qformer." + k: t(v) for k, v in sd.items() if k.startswith("bert.")}
        own["query_tokens"] = t(sd["query_tokens"])
        for k in ("mistral_proj.weight", "mistral_proj.bias"):
            own[k] = t(sd[k])
        res = self.load_state_dict(own, strict=False)
        if strict:
            bad = [k for k in res.missing_keys if ".long_term_attention.proj_" not in k]   # aliases of key / value (Qformer.py:156-157)
            if bad or res.unexpected_keys:
                raise KeyError(f"state dict mismatch: missing {bad}, unexpected {res.unexpected_keys}")
        return res

    # ------------------------------------------------------------------ HIP short-term cross-attention
    def _short_handle(self, device):
        if self._vqf is not None and self._vqf_dev == device:
            return self._vqf
        self._release()
        cfg = self.qformer.config
        c = _lib.VqfConfig(1, cfg.num_attention_heads, cfg.hidden_size, cfg.intermediate_size, cfg.encoder_width,
                           self.tokens_per_frame, self.n_query, 0, NB_SAMPLES, float(cfg.alpha), float(cfg.layer_norm_eps))
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().infv_vqf_create(C.byref(c), C.byref(h)))
        self._vqf, self._vqf_dev = h, device
        return h

    def _release(self):
        h, self._vqf = self._vqf, None
        if h:
            try:
                _lib.load().infv_vqf_destroy(h)
            except Exception:
                pass

    def __del__(self):
        self._release()

    def _cross_attention(self, layer: _Layer, k: torch.Tensor, xq: torch.Tensor, a_long) -> torch.Tensor:
        """merged = alpha * softmax(xq K^T / sqrt(dh)) V + (1 - alpha) * a_long   (Qformer.py:223-303), on the HIP path."""
        att = layer.crossattention.self
        dev = k.device
        if k.size(0) % 32:
            # frame counts that are not a multiple of 8 (196 * T tokens, T % 8 != 0): the HIP contraction tiles the token
            # axis in 32s; the eval loop never produces such chunks (num_segments = max_int * num_samples,
            # run_nextqa_mistral.py:544-547), so they take the stock-PyTorch cross-attention on the same device
            H, dh = att.H, att.dh
            split = lambda x: x.view(1, -1, H, dh).permute(0, 2, 1, 3)
            kk, vv = split(att.key(k.unsqueeze(0))), split(att.value(k.unsqueeze(0)))
            probs = torch.softmax(torch.matmul(split(xq), kk.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
            short = torch.matmul(probs, vv).permute(0, 2, 1, 3).reshape(1, -1, H * dh)
            alpha = self.qformer.config.alpha
            return short if isinstance(a_long, int) else alpha * short + (1 - alpha) * a_long
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        wk, bk, wv, bv = f32(att.key.weight), f32(att.key.bias), f32(att.value.weight), f32(att.value.bias)
        key, val = _lib.Linear(wk.data_ptr(), bk.data_ptr()), _lib.Linear(wv.data_ptr(), bv.data_ptr())
        q = f32(xq[0])
        along = None if isinstance(a_long, int) else f32(a_long[0])
        out = torch.empty(self.n_query, self.qformer.config.hidden_size, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().infv_vqf_short_attention(
                self._short_handle(dev), C.c_void_p(k.data_ptr()), k.size(0), C.c_void_p(q.data_ptr()), C.byref(key),
                C.byref(val), C.c_void_p(0 if along is None else along.data_ptr()), C.c_void_p(out.data_ptr()),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out.unsqueeze(0)

    # ------------------------------------------------------------------ encode_img after the vision encoder
    def encode_tokens(self, image_embeds: torch.Tensor, text_embeds: Optional[torch.Tensor] = None,
                      new_video: bool = False, hidden_in: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """image_embeds [1, T*196, 1024] (layer-normed vision tokens, videochat2_it_mistral.py:194-196);
        text_embeds [1, n_text, hidden] = the embedding layer's rows for the instruction (or None);
        ``hidden_in`` overrides the whole encoder input [1, n_query + n_text, hidden] (tests).
        Returns (inputs_mistral [1, n_query, llm_hidden], last_hidden_state)."""
        if not image_embeds.is_cuda:
            raise RuntimeError("the VideoChat2 binding runs on the HIP device only (no CPU fallback)")
        if image_embeds.dim() != 3 or image_embeds.size(0) != 1:
            raise ValueError("image_embeds must be [1, T*196, width] (batch 1)")
        cfg = self.qformer.config
        P = self.tokens_per_frame
        if image_embeds.size(2) != cfg.encoder_width or image_embeds.size(1) % P:
            raise ValueError(f"image_embeds must be [1, T*{P}, {cfg.encoder_width}]")
        k = image_embeds[0].detach().to(torch.float32).contiguous()
        if hidden_in is not None:
            hidden = hidden_in.to(torch.float32)
        else:
            hidden = self.query_tokens.to(torch.float32)
            if text_embeds is not None:
                hidden = torch.cat([hidden, text_embeds.to(torch.float32)], dim=1)      # Qformer.py:100-106 (embeddings output)
        nq = self.n_query
        use_ltm = cfg.alpha != 1.0
        with torch.no_grad():
            for layer in self.qformer.bert.encoder.layer:
                att = layer.attention
                attention_output = att.output(att.self.self_attention(hidden), hidden)   # Qformer.py:445-458
                query_out = attention_output[:, :nq, :]
                if layer.has_cross_attention:                                            # :463-481
                    x = layer.crossattention
                    hook = x.self._hook
                    xq = x.self.query(query_out)                                         # mixed_query_layer (:209)
                    a_long = hook.long_term(image_embeds, xq, None, layer.layer_num, new_video) if use_ltm else 0
                    merged = self._cross_attention(layer, k, xq, a_long)
                    query_out = x.output(merged, query_out)
                out = layer.output_query(layer.intermediate_query(query_out), query_out)  # :483-488
                if attention_output.size(1) > nq:                                        # text tokens: their own FFN (:489-496)
                    txt = attention_output[:, nq:, :]
                    out = torch.cat([out, layer.output(layer.intermediate(txt), txt)], dim=1)
                hidden = out
            inputs_mistral = self.mistral_proj(hidden[:, :nq, :])                        # videochat2_it_mistral.py:252
        return inputs_mistral, hidden


def encode_long_video_vc(model: VideoChat2Encoder, frame_tokens: torch.Tensor, num_samples: int,
                         text_embeds: Optional[torch.Tensor] = None, hidden_in: Optional[torch.Tensor] = None):
    """The VideoChat2 eval loop (eval_code/run_nextqa_mistral.py:141-152): ``torch.chunk(video, num_samples, dim=1)``
    over frames, ``new_video`` true on the first chunk only, mean of the stacked per-chunk embeddings.
    frame_tokens [F, 196, width] -> (mean [1, n_query, llm_hidden], per-chunk list)."""
    embs = []
    new_video = True
    for blk in torch.chunk(frame_tokens, num_samples, dim=0):
        emb, _ = model.encode_tokens(blk.reshape(1, -1, blk.size(-1)), text_embeds, new_video, hidden_in)
        embs.append(emb)
        new_video = False
    return torch.mean(torch.stack(embs), dim=0, keepdim=True).squeeze(0), embs
