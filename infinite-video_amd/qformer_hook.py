"""The Q-former side of the LTM path: construction, call and memory-merge step.

Mirrors the ~40 lines of ``BertSelfAttention`` that touch the long-term memory in the reference
(infty-Video-LLaMA/InfVideoLLaMA/models/Qformer.py):

* construction with the layer's own ``key`` / ``value`` Linear as projections   :135-159
* the call, guarded by ``position_embedding_ext is not None`` and ``alpha != 1.0``  :216-223
* the merge  ``alpha * short + (1 - alpha) * long``                              :303-304

The VideoChat2 Q-former (infty-VideoChat2/models/blip2/Qformer.py:215-222,302-303) fires the hook
on every cross-attention (no ``position_embedding_ext`` test) and uses ``sigmas=1``.

A maintainer swaps the reference op for this one either by replacing the import at Qformer.py:50
 or by building the cross-attention with :class:`LongTermMemoryHook`.
"""
from __future__ import annotations

from typing import Optional, Union

import torch
import torch.nn as nn

from .long_term_attention_gibbs import LongTermAttention, LongTermAttentionVC


def build_long_term_attention(config, key: nn.Linear, value: nn.Linear, num_attention_heads: int,
                              attention_head_size: int, variant: str = "VL") -> LongTermAttention:
    """The ``partial(LongTermAttention, ...)()`` of Qformer.py:135-159 with the same kwargs."""
    cls = LongTermAttention if variant == "VL" else LongTermAttentionVC
    return cls(
        attn_num_basis=config.num_basis,
        head_size=attention_head_size,
        length=config.encoder_width,
        target_len=config.encoder_width,
        attn_func="softmax",
        infinite_memory=True,
        n_layers=2,
        attn_drop=0.1,
        n_heads=num_attention_heads,
        d_model=num_attention_heads * attention_head_size,
        affines=True,
        mask=True,
        mask_type="cnn",
        kl_regularizer=False,
        sigma_0=None,
        mu_0=None,
        sticky_memories=config.sticky,
        continuous=True,
        sigmas=config.sigmas if variant == "VL" else 1,
        tau=config.tau,
        proj_key=key,
        proj_value=value,
    )


class LongTermMemoryHook(nn.Module):
    """Holds one cross-attention layer's LTM and applies the reference's call/merge rules."""

    def __init__(self, config, key: nn.Linear, value: nn.Linear, num_attention_heads: int,
                 attention_head_size: int, variant: str = "VL"):
        super().__init__()
        self.alpha = config.alpha
        self.variant = variant
        self.long_term_attention = build_long_term_attention(config, key, value, num_attention_heads,
                                                             attention_head_size, variant)

    def active(self, position_embedding_ext) -> bool:
        # Video-LLaMA: only the video Q-former passes a (non-None) position embedding, which acts as
        # a flag (Qformer.py:216,303); VideoChat2: every cross-attention.
        return self.variant != "VL" or position_embedding_ext is not None

    def long_term(self, encoder_hidden_states: torch.Tensor, mixed_query_layer: torch.Tensor,
                  position_embedding_ext, layer: int, new_video: bool) -> Union[torch.Tensor, int]:
        """Qformer.py:216-223.  Returns 0 when alpha == 1.0 (the reference skips the op entirely)."""
        if not self.active(position_embedding_ext):
            return 0
        _, p, _ = encoder_hidden_states.shape
        self.long_term_attention.length = p
        self.long_term_attention.target_len = p
        if self.alpha != 1.0:
            return self.long_term_attention(encoder_hidden_states, mixed_query_layer,
                                            new_doc=new_video, layer_n=layer).detach()
        return 0

    def merge(self, context_layer: torch.Tensor, a_long_term, position_embedding_ext) -> torch.Tensor:
        """Qformer.py:303-304."""
        if not self.active(position_embedding_ext):
            return context_layer
        return self.alpha * context_layer + (1 - self.alpha) * a_long_term


def install(variant: str = "VL") -> None:
    """Make the reference's import site resolve to this implementation: after ``install()``,
    ``from .long_term_attention_gibbs import LongTermAttention`` inside the reference package
    (Qformer.py:50 / blip2/Qformer.py:49) yields the MI355X operator.  Call before importing the
    reference's model modules."""
    import sys
    import types
    name = ("InfVideoLLaMA.models.long_term_attention_gibbs" if variant == "VL"
            else "models.blip2.long_term_attention_gibbs")
    mod = types.ModuleType(name)
    mod.LongTermAttention = LongTermAttention if variant == "VL" else LongTermAttentionVC
    mod.__doc__ = "MI355X-native replacement installed by infinite_video_amd.qformer_hook.install()"
    sys.modules[name] = mod
