"""CPU restatement of the VideoChat2 Q-former path around the LTM (TEST INFRASTRUCTURE -- not a product path).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
It restates, for all-ones masks in eval mode, what the reference computes in

    infty-VideoChat2/models/blip2/Qformer.py
        BertSelfAttention.forward   :195-310  self-attention over query + text tokens; in cross layers the LTM call on the
                                              query part (:215-222, every cross-attention, no position-embedding test) and
                                              the merge alpha*short + (1-alpha)*long (:302-303)
        BertSelfOutput.forward      :320-324  dense + residual + LayerNorm
        BertLayer.forward           :443-505  self-attn -> (layer % 2 == 0: cross-attn on the query part) -> query FFN for the
                                              query part, text FFN for the text part
        BertEncoder.forward         :544-640  the 12-layer loop
    infty-VideoChat2/models/videochat_mistra/videochat2_it_mistral.py:252   mistral_proj on the query part
    infty-VideoChat2/eval_code/run_nextqa_mistral.py:141-152                  torch.chunk over frames, new_video on the first
                                                                             chunk only, mean of the per-chunk embeddings
with ``oracle.ltm_oracle.ClosedFormOracle`` (196 tokens per frame, width 1024) as the LTM of each cross layer.

Parity is PINNED: ``tests/golden/vc_mistral.npz`` holds the outputs of the real reference ``BertEncoder`` (with the real
VideoChat2 ``LongTermAttention`` inside) on the same inputs (``tests/golden/make_vc_goldens.py``);
``tests/test_vc_oracle.py`` checks this restatement against it.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .ltm_oracle import ClosedFormOracle

LN_EPS = 1e-12


class VideoChat2Oracle:
    def __init__(self, weights: Dict[str, np.ndarray], num_basis: int, tau: float, alpha: float, sticky: bool,
                 n_layers: int = 12, cross_freq: int = 2, n_heads: int = 12, n_query: int = 96, tokens_per_frame: int = 196):
        self.w = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in weights.items()}
        self.L, self.freq, self.H, self.alpha, self.nq, self.P = n_layers, cross_freq, n_heads, alpha, n_query, tokens_per_frame
        self.dh = self.w["query_tokens"].shape[-1] // n_heads
        self.ltm: Dict[int, ClosedFormOracle] = {}
        for l in range(0, n_layers, cross_freq):
            p = f"bert.encoder.layer.{l}.crossattention.self."
            self.ltm[l] = ClosedFormOracle(num_basis, n_heads, self.dh, tau, sticky, weights[p + "key.weight"],
                                           weights[p + "key.bias"], weights[p + "value.weight"], weights[p + "value.bias"],
                                           tokens_per_frame=tokens_per_frame)
        self.taps: Dict[str, np.ndarray] = {}

    def _lin(self, x, name):
        return F.linear(x, self.w[name + ".weight"], self.w[name + ".bias"])

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w[name + ".bias"], LN_EPS)

    def _heads(self, x):
        return x.reshape(x.shape[0], self.H, self.dh).permute(1, 0, 2)

    def _attend(self, q, k, v):
        s = torch.matmul(self._heads(q), self._heads(k).transpose(-1, -2)) / math.sqrt(self.dh)
        return torch.matmul(torch.softmax(s, dim=-1), self._heads(v)).permute(1, 0, 2).reshape(q.shape[0], -1)

    def encode_chunk(self, frames: np.ndarray, h0: np.ndarray, new_video: bool, u: Optional[np.ndarray] = None):
        """frames [T*P, width]; h0 [n_query + n_text, hidden] encoder input; u [n_ltm, 512] float64.
        Returns (last hidden [n_query + n_text, hidden], mistral [n_query, proj_out])."""
        k = torch.from_numpy(np.asarray(frames, np.float32))
        h = torch.from_numpy(np.asarray(h0, np.float32))
        j = 0
        for l in range(self.L):
            p = f"bert.encoder.layer.{l}."
            a = p + "attention."
            ctx = self._attend(self._lin(h, a + "self.query"), self._lin(h, a + "self.key"), self._lin(h, a + "self.value"))
            att = self._ln(self._lin(ctx, a + "output.dense") + h, a + "output.LayerNorm")
            qo = att[:self.nq]
            if l % self.freq == 0:
                x = p + "crossattention."
                xq = self._lin(qo, x + "self.query")
                self.taps[f"l{l}_xq"] = xq.numpy()
                if self.alpha != 1.0:
                    along = torch.from_numpy(self.ltm[l].step(k.numpy(), xq.numpy(), new_video, None if u is None else u[j]))
                    self.taps[f"l{l}_along"] = along.numpy()
                else:
                    along = 0
                j += 1
                short = self._attend(xq, self._lin(k, x + "self.key"), self._lin(k, x + "self.value"))
                merged = self.alpha * short + (1 - self.alpha) * along
                self.taps[f"l{l}_xctx"] = merged.numpy()
                qo = self._ln(self._lin(merged, x + "output.dense") + qo, x + "output.LayerNorm")
            out = self._ln(self._lin(F.gelu(self._lin(qo, p + "intermediate_query.dense")), p + "output_query.dense") + qo,
                           p + "output_query.LayerNorm")
            if att.shape[0] > self.nq:
                txt = att[self.nq:]
                out_t = self._ln(self._lin(F.gelu(self._lin(txt, p + "intermediate.dense")), p + "output.dense") + txt,
                                 p + "output.LayerNorm")
                out = torch.cat([out, out_t], 0)
            h = out
        return h.numpy(), self._lin(h[:self.nq], "mistral_proj").numpy()

    def encode_long_video(self, frames: np.ndarray, h0: np.ndarray, num_samples: int, u_of_chunk):
        """frames [F, P, width] -> (mean over chunks of the mistral embeddings, per-chunk list)."""
        embs = []
        for c, blk in enumerate(torch.chunk(torch.from_numpy(frames), num_samples, dim=0)):
            embs.append(self.encode_chunk(blk.reshape(-1, blk.shape[-1]).numpy(), h0, c == 0, u_of_chunk(c))[1])
        return np.mean(np.stack(embs), 0), embs
