"""On-disk format of a consolidated long-term memory.

The reference keeps its LTM state (``B_past`` per cross-attention layer, plus the keys/queries the next sticky step
derives its density from) only in module attributes and never serialises it.  One file holds
the memories of all LTM layers of a video Q-former:

    container   safetensors (little-endian, memory-mappable, no pickle)
    tensors     ``layer.{l}.B_past``   float32 [num_basis, d]     coefficient matrix (LTM.py:220)
                ``layer.{l}.bin_mass`` float32 [127]              unnormalised sticky bin masses of the last scores
                                                                   (LTM.py:200-202: the histogram has 128 bins, 127 are drawn from)
    metadata    ``format`` = "infv-ltm-memory", ``version`` = "1", ``n_layers``, ``num_basis``, ``layer.{l}.tau`` and
                ``layer.{l}.sticky`` for every layer (plus ``tau`` / ``sticky`` = layer 0's, kept for older readers),
                and free-form ``user.*`` entries (e.g. the video id, frames consumed)

A memory only continues correctly under the knobs it was consolidated with, so ``load_memory`` refuses a file whose
``num_basis``, ``tau`` or ``sticky`` differ from the receiving module's.

The projected memory (K', V') is not stored: it is rebuilt from ``B_past`` with the key/value weights current at
load time (``infv_ltm_import_state``), so a file stays valid across weight casts and devices.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
from safetensors import safe_open
from safetensors.torch import save_file

FORMAT, VERSION = "infv-ltm-memory", "1"


def save_memory(path: str, ltm_modules: Sequence, user: Optional[Dict[str, str]] = None) -> None:
    """Write the memories of ``ltm_modules`` (``LongTermAttention`` instances, layer order).  Raises if a
    memory is empty (nothing consolidated yet)."""
    tensors, first = {}, None
    for l, m in enumerate(ltm_modules):
        st = m.memory_state()
        if st is None:
            raise RuntimeError(f"LTM layer {l} holds no memory (B_past is None)")
        first = first or st
        tensors[f"layer.{l}.B_past"] = st["B_past"].contiguous()
        tensors[f"layer.{l}.bin_mass"] = st["bin_mass"].contiguous()
    meta = {"format": FORMAT, "version": VERSION, "n_layers": str(len(ltm_modules)),
            "num_basis": str(first["num_basis"]), "tau": repr(float(first["tau"])), "sticky": str(int(first["sticky"]))}
    for l, m in enumerate(ltm_modules):
        st = m.memory_state()
        if st["num_basis"] != first["num_basis"]:
            raise ValueError("all LTM layers of one file must share num_basis")
        meta[f"layer.{l}.tau"] = repr(float(st["tau"]))
        meta[f"layer.{l}.sticky"] = str(int(st["sticky"]))
    for k, v in (user or {}).items():
        meta["user." + k] = str(v)
    save_file(tensors, path, metadata=meta)


def load_memory(path: str, ltm_modules: Sequence, device, max_q: Optional[int] = None) -> Dict[str, str]:
    """Load a file written by :func:`save_memory` into ``ltm_modules``; returns the ``user.*`` metadata.
    ``max_q``: largest query length the modules will be called with afterwards (default: what each module already
    allocated, else 32; a longer query later re-imports the memory into a larger engine by itself)."""
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT or meta.get("version") != VERSION:
            raise ValueError(f"{path} is not an {FORMAT} v{VERSION} file")
        if int(meta["n_layers"]) != len(ltm_modules):
            raise ValueError(f"{path} holds {meta['n_layers']} layers, {len(ltm_modules)} expected")
        for l, m in enumerate(ltm_modules):
            if int(meta["num_basis"]) != m.attn_num_basis:
                raise ValueError(f"num_basis mismatch: file {meta['num_basis']}, module {m.attn_num_basis}")
            tau = float(meta.get(f"layer.{l}.tau", meta["tau"]))
            sticky = bool(int(meta.get(f"layer.{l}.sticky", meta["sticky"])))
            m.load_memory_state({"B_past": f.get_tensor(f"layer.{l}.B_past"), "bin_mass": f.get_tensor(f"layer.{l}.bin_mass"),
                                 "num_basis": int(meta["num_basis"]), "tau": tau, "sticky": sticky, "version": 1},
                                device, max_q=max_q)
    return {k[5:]: v for k, v in meta.items() if k.startswith("user.")}
