"""CPU restatement of the video Q-former path around the LTM (TEST INFRASTRUCTURE -- not a product path).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
It restates, for the query-token-only case the video Q-former is run in (no text tokens, all-ones masks,
eval mode), what the reference computes in

    infty-Video-LLaMA/InfVideoLLaMA/models/Qformer.py
        BertEmbeddings.forward          :85-112   LayerNorm of the learned query tokens
        BertSelfAttention.forward       :197-312  self-attention; cross-attention over the chunk's frame tokens,
                                                  LTM call (:216-223) and merge alpha*short + (1-alpha)*long (:303-304)
        BertSelfOutput.forward          :322-326  dense + residual + LayerNorm
        BertLayer.forward               :442-522  self-attn -> cross-attn -> query FFN (intermediate_query/output_query)
        BertEncoder.forward             :544-640  the layer loop
    infty-Video-LLaMA/InfVideoLLaMA/models/infinityqa.py
        encode_video                    :280-344  frame cap n_position^2, token concat, video Q-former, llama_proj
    eval loop  run_inference_inf_video_llama_nextqa.py:179-196   new_video=(i==0), mean over chunk embeddings

Parity is PINNED: ``tests/golden/qf_*.npz`` hold the outputs of the real reference ``BertEncoder`` (with the real
``LongTermAttention`` inside) on the same inputs (``tests/golden/make_qformer_goldens.py``);
``tests/test_qformer_oracle.py`` checks this restatement against them.  ``encode_video`` itself cannot be imported
here (needs cv2/omegaconf), so its frame-cap rule is restated from the source lines cited above and covered by a
host-logic test only.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .ltm_oracle import ClosedFormOracle

LN_EPS = 1e-12          # BertConfig.layer_norm_eps


def frame_cap(n_frames: int) -> int:
    """Frames kept by encode_video: n_position = min(32, ceil(sqrt(n))), oldest dropped beyond n_position^2
    (infinityqa.py:285-288,306-307)."""
    n_position = min(32, math.ceil(math.sqrt(n_frames)))
    return min(n_frames, n_position * n_position)


class VideoQformerOracle:
    def __init__(self, weights: Dict[str, np.ndarray], num_basis: int, tau: float, alpha: float, sticky: bool,
                 n_layers: int = 2, n_heads: int = 12, tokens_per_frame: int = 32):
        self.w = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in weights.items()}
        self.L, self.H, self.alpha, self.P = n_layers, n_heads, alpha, tokens_per_frame
        hidden = self.w["video_query_tokens"].shape[-1]
        self.dh = hidden // n_heads
        self.ltm: List[ClosedFormOracle] = []
        for l in range(n_layers):
            p = f"bert.encoder.layer.{l}.crossattention.self."
            self.ltm.append(ClosedFormOracle(num_basis, n_heads, self.dh, tau, sticky,
                                             weights[p + "key.weight"], weights[p + "key.bias"],
                                             weights[p + "value.weight"], weights[p + "value.bias"],
                                             tokens_per_frame=tokens_per_frame))
        self.taps: Dict[str, np.ndarray] = {}

    # ---- building blocks
    def _lin(self, x, name):
        return F.linear(x, self.w[name + ".weight"], self.w[name + ".bias"])

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w[name + ".bias"], LN_EPS)

    def _heads(self, x):                        # [n, H*dh] -> [H, n, dh]
        return x.reshape(x.shape[0], self.H, self.dh).permute(1, 0, 2)

    def _attend(self, q, k, v):                 # Qformer.py:244,278,284,298-301 with zero masks
        s = torch.matmul(self._heads(q), self._heads(k).transpose(-1, -2)) / math.sqrt(self.dh)
        p = torch.softmax(s, dim=-1)
        return torch.matmul(p, self._heads(v)).permute(1, 0, 2).reshape(q.shape[0], -1)

    def embed(self):
        return self._ln(self.w["video_query_tokens"][0], "bert.embeddings.LayerNorm")

    # ---- one chunk through the encoder
    def encode_chunk(self, frames: np.ndarray, new_video: bool, u: Optional[np.ndarray] = None):
        """frames [T*P, d]; u [L, 512] float64 (needed when the memory is non-empty and sticky).
        Returns (hidden [Q, hidden], llama [Q, proj_out])."""
        k = torch.from_numpy(np.asarray(frames, np.float32))
        h = self.embed()
        for l in range(self.L):
            p = f"bert.encoder.layer.{l}."
            # self-attention over the query tokens
            a = p + "attention."
            ctx = self._attend(self._lin(h, a + "self.query"), self._lin(h, a + "self.key"), self._lin(h, a + "self.value"))
            h1 = self._ln(self._lin(ctx, a + "output.dense") + h, a + "output.LayerNorm")
            # cross-attention over this chunk's frame tokens, with the long-term memory
            x = p + "crossattention."
            xq = self._lin(h1, x + "self.query")
            self.taps[f"l{l}_xq"] = xq.numpy()
            if self.alpha != 1.0:
                ul = None if u is None else u[l]
                along = torch.from_numpy(self.ltm[l].step(k.numpy(), xq.numpy(), new_video, ul))
                self.taps[f"l{l}_along"] = along.numpy()
            else:
                along = 0
            short = self._attend(xq, self._lin(k, x + "self.key"), self._lin(k, x + "self.value"))
            merged = self.alpha * short + (1 - self.alpha) * along
            self.taps[f"l{l}_xctx"] = merged.numpy()
            h2 = self._ln(self._lin(merged, x + "output.dense") + h1, x + "output.LayerNorm")
            # query FFN
            inter = F.gelu(self._lin(h2, p + "intermediate_query.dense"))
            h = self._ln(self._lin(inter, p + "output_query.dense") + h2, p + "output_query.LayerNorm")
        return h.numpy(), self._lin(h, "llama_proj").numpy()

    # ---- encode_video / eval-loop counterparts
    def encode_video(self, frame_list: List[np.ndarray], new_video: bool, u=None):
        """frame_list: per-frame token blocks [P, d] (the short-memory buffer)."""
        keep = frame_cap(len(frame_list))
        frames = np.concatenate(frame_list[len(frame_list) - keep:], 0)
        return self.encode_chunk(frames, new_video, u)

    def encode_long_video(self, frames: np.ndarray, max_int: int, u_of_chunk):
        """frames [F, P, d] -> (mean over chunks of llama embeddings, per-chunk embeddings)."""
        embs = []
        for i, start in enumerate(range(0, frames.shape[0], max_int)):
            blk = frames[start:start + max_int]
            embs.append(self.encode_video(list(blk), new_video=(i == 0), u=u_of_chunk(i))[1])
        return np.mean(np.stack(embs), 0), embs
