"""Host-side construction of the LTM "plan" for one chunk length.

The reference rebuilds its basis machinery on every forward (``get_basis``,
reference long_term_attention_gibbs.py:67-165: sample positions, rectangular basis
evaluation, an N x N inverse on the CPU, Python loops).  The result depends only on
``(T, N, tau)``, and because the active basis family is rectangular
(basis_functions.py:214-266) ``F F^T`` is diagonal: every row of the ridge operator ``G`` has
at most one non-zero, ``1/(count_of_its_box + ridge)``.  This module computes that sparse form
once per ``(T, N, tau)`` and hands it to the C ABI (``infv_ltm_set_plan``).

Positions and box bounds are evaluated with the same fp32 torch CPU expressions the
reference uses (``torch.linspace`` etc.), so box membership -- including the half-open
``lo <= t < hi`` edge cases such as ``t = 1.0`` falling in no box -- is decided on bit-identical
numbers.  torch is used here only as the fp32 calculator for a few hundred scalars.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache
from typing import Tuple

import numpy as np
import torch

NB_SAMPLES = 512      # reference long_term_attention_gibbs.py:55
RIDGE_PENALTY = 0.5   # :62
NB_BINS = 128         # :163
GRID_POINTS = 1000    # :251


class UnsupportedBasis(ValueError):
    """The (T, N, tau) combination does not give a one-non-zero-per-row operator."""


def _box_bounds(N: int):
    # mu/width as built at long_term_attention_gibbs.py:177-181, bounds as basis_functions.py:248-249
    width = torch.ones(N) / N
    edges = torch.linspace(0, 1, N + 1)
    mu = (edges[:-1] + edges[1:]) / 2
    return mu - width / 2, mu + width / 2


class OverlappingBoxes(UnsupportedBasis):
    """A point the step evaluates psi at lies in two fp32 boxes: the sparse (one non-zero per row) form does not apply;
    ``build_plan`` then builds the dense form."""


def box_of(t: torch.Tensor, N: int) -> np.ndarray:
    """Index of the box containing each fp32 point, -1 for none."""
    lo, hi = _box_bounds(N)
    t = t.to(torch.float32).reshape(-1, 1)
    inside = (t >= lo) & (t < hi)
    n_hit = inside.sum(1)
    if int(n_hit.max()) > 1:
        raise OverlappingBoxes(f"a sample position lies in two boxes for num_basis={N}")
    idx = inside.float().argmax(1).to(torch.int32)
    idx[n_hit == 0] = -1
    return idx.numpy()


def boxes2_of(t: torch.Tensor, N: int) -> np.ndarray:
    """[M, 2] int32: the (at most two, adjacent) boxes containing each fp32 point, -1 padded, ascending."""
    lo, hi = _box_bounds(N)
    t = t.to(torch.float32).reshape(-1, 1)
    inside = ((t >= lo) & (t < hi)).numpy()
    if int(inside.sum(1).max(initial=0)) > 2:
        raise UnsupportedBasis(f"a point lies in more than two boxes for num_basis={N}")
    out = np.full((inside.shape[0], 2), -1, dtype=np.int32)
    for i, row in enumerate(inside):
        hit = np.flatnonzero(row)
        out[i, :len(hit)] = hit
    return out


def membership(t: torch.Tensor, N: int) -> torch.Tensor:
    """psi(t) as the reference evaluates it (basis_functions.py:227-250): [M, N] fp32 zeros and ones."""
    lo, hi = _box_bounds(N)
    t = t.to(torch.float32).reshape(-1, 1)
    return ((t >= lo) & (t < hi)).float()


def _dense_operator_T(l: int, positions: torch.Tensor, N: int) -> np.ndarray:
    """G^T [N, l] of compute_G (long_term_attention_gibbs.py:68-84) with the reference's own fp32 ATen sequence:
    F = psi(positions)^T, G = F^T (F F^T + ridge I)^-1 (``Tensor.inverse`` on the CPU), padding rows trimmed."""
    F = torch.zeros(N, positions.size(0))
    F[:, :] = membership(positions, N).t()
    G = F.t().matmul((F.matmul(F.t()) + RIDGE_PENALTY * torch.eye(N)).inverse())
    b, e = _trimmed(G.size(0), l)
    G = G[b:e]
    if G.size(0) != l:
        raise UnsupportedBasis("padding trim does not leave one row per sample")
    return np.ascontiguousarray(G.t().numpy(), dtype=np.float32)


def _positions_first(T: int) -> torch.Tensor:          # :104-110
    if T % 2:
        shift = 1 / float(T)
        return torch.linspace(-.5 + shift, 1.5 - shift, 2 * T - 1)
    shift = 1 / float(2 * T)
    return torch.linspace(-.5 + shift, 1.5 - shift, 2 * T)


def _positions_inf(T: int, tau: float, S: int) -> torch.Tensor:     # :135-150
    tm_tau = torch.arange(1, S + 1).float() * tau / S
    tm_l = tau + (1 - tau) * (torch.arange(S + 1, T + S + 1).float() - S) / T
    if T % 2:
        shift = 1 / float(T + S)
        pad = torch.linspace(-.5 + shift, 1.5 - shift, 2 * (T + S) - 1)
    else:
        shift = 1 / float(2 * T + S)
        pad = torch.linspace(-.5 + shift, 1.5 - shift, 2 * (T + S))
    return torch.cat([pad[pad < 0], tm_tau, tm_l, pad[pad > 1]])


def _trimmed(total: int, l: int) -> Tuple[int, int]:
    """[begin, end) of the rows compute_G keeps (:78-82)."""
    if l % 2:
        return (l - 1) // 2, total + (-(l - 1) // 2)
    return l // 2, total - l // 2


def _operator(l: int, positions: torch.Tensor, N: int):
    box = box_of(positions, N)
    counts = np.bincount(box[box >= 0], minlength=N).astype(np.float32)
    box_val = (np.float32(1) / (counts + np.float32(RIDGE_PENALTY))).astype(np.float32)
    b, e = _trimmed(len(box), l)
    rows = box[b:e]
    if len(rows) != l:
        raise UnsupportedBasis("padding trim does not leave one row per sample")
    return rows.astype(np.int32), box_val


def _frame_ranges(frame_box: np.ndarray):
    """Group frames by box; frames of a box must be contiguous (positions are monotone)."""
    row_box, row_begin, row_end = [], [], []
    i, T = 0, len(frame_box)
    while i < T:
        b = int(frame_box[i])
        j = i
        while j < T and frame_box[j] == b:
            j += 1
        if b >= 0:
            if b in row_box:
                raise UnsupportedBasis("frames of one box are not contiguous")
            row_box.append(b); row_begin.append(i); row_end.append(j)
        i = j
    as32 = lambda x: np.asarray(x, dtype=np.int32)
    return as32(row_box), as32(row_begin), as32(row_end)


@dataclass(frozen=True)
class Plan:
    T: int
    N: int
    tau: float
    S: int
    first_row_box: np.ndarray
    first_row_begin: np.ndarray
    first_row_end: np.ndarray
    first_box_val: np.ndarray
    inf_row_box: np.ndarray
    inf_row_begin: np.ndarray
    inf_row_end: np.ndarray
    inf_box_val: np.ndarray
    inf_old_ptr: np.ndarray
    inf_old_slot: np.ndarray
    readout_w: np.ndarray
    readout_w_out: float
    edge_box: np.ndarray
    edge_dx: np.ndarray
    bin_box: np.ndarray
    uniform_idx: np.ndarray
    # dense form (num_basis whose fp32 boxes overlap at a point the step evaluates): infv_ltm_set_dense_plan
    dense: bool = False
    first_GT: np.ndarray = None       # [N, T]      G_first transposed
    inf_GT: np.ndarray = None         # [N, S + T]  G_inf transposed
    bin_box2: np.ndarray = None       # [NB_BINS, 2]
    edge_box2: np.ndarray = None      # [NB_BINS + 1, 2]
    uniform_box2: np.ndarray = None   # [S, 2]
    # general-psi form (a basis family whose psi(t) is a dense row: the Gaussian family): infv_ltm_set_psi_plan
    psi: bool = False
    psi_edge: np.ndarray = None       # [NB_BINS + 1, N]  psi at the modified histogram edges
    psi_bin: np.ndarray = None        # [NB_BINS, N]      psi at the unmodified left edge of every bin
    psi_uniform: np.ndarray = None    # [S, N]            psi at the non-sticky resample positions
    psi_grid: np.ndarray = None       # [GRID_POINTS, N]  psi on the read-out grid
    grid_w: np.ndarray = None         # [GRID_POINTS]     trapezoid weights of that grid
    # num_basis that is not a multiple of 16 (the kernels' tile): the device sees N_pad = the next multiple, the extra basis
    # functions have zero operator rows and zero read-out weight (padded_N(N)); 0 = no padding
    N_pad: int = 0


def gaussian_psi(t: torch.Tensor, N: int, sigmas) -> torch.Tensor:
    """psi(t) [M, N] of the reference's Gaussian family: centres x widths as ``add_gaussian_basis_functions`` builds them
    (long_term_attention_gibbs.py:167-174), values as ``GaussianBasisFunctions.evaluate`` / ``batch_evaluate`` compute them
    (basis_functions.py:155-164: the same fp32 elementwise sequence (t - mu) / sigma -> phi -> / sigma in both)."""
    import math
    mu, sigma = torch.meshgrid(torch.linspace(0, 1, N // len(sigmas)), torch.Tensor(list(sigmas)), indexing="ij")
    mu, sigma = mu.flatten().unsqueeze(0), sigma.flatten().unsqueeze(0)
    if mu.size(1) != N:
        raise UnsupportedBasis("num_basis must be a multiple of len(sigmas)")
    z = (t.to(torch.float32).reshape(-1, 1) - mu) / sigma
    return (1. / math.sqrt(2 * math.pi) * torch.exp(-.5 * z ** 2)) / sigma


@lru_cache(maxsize=64)
def build_gaussian_plan(T: int, N: int, tau: float, sigmas: Tuple[float, ...], S: int = NB_SAMPLES) -> Plan:
    """Plan of the reference's GAUSSIAN basis family for chunks of T frames: dense ridge operators (compute_G,
    long_term_attention_gibbs.py:68-84, with the reference's own ATen sequence) and psi itself at every point the step
    evaluates it -- histogram edges (:197-200), the bins' left edges (:207-208), the uniform resampling positions
    (:153-157) and the 1000-point read-out grid with its trapezoid weights (:251-286)."""
    if T < 2:
        raise UnsupportedBasis("chunks of a single frame are empty in the reference (G[0:-0])")
    sigmas = tuple(float(x) for x in sigmas)
    bins = torch.linspace(0, 1, NB_BINS + 1)
    mod = bins.clone()
    mod[0] = -.000001
    mod[-1] = 1.000001
    edge_dx = (mod[1:] - mod[:-1]).numpy().astype(np.float32)
    t_uni = (torch.arange(1, S + 1).float() * tau / S) / tau
    t = torch.linspace(0, 1, GRID_POINTS)
    dx = (t[1:] - t[:-1]).double().numpy()
    wt = np.zeros(GRID_POINTS)
    wt[:-1] += dx / 2
    wt[1:] += dx / 2
    f32 = lambda x: np.ascontiguousarray(x.numpy(), dtype=np.float32)
    none_i = np.zeros(0, dtype=np.int32)
    zeros_n = np.zeros(N, dtype=np.float32)
    no_box = lambda n: np.full((n, 2), -1, dtype=np.int32)
    return Plan(
        T=T, N=N, tau=tau, S=S,
        # neither the sparse operator tables nor the box tables apply to a dense psi: empty but valid
        first_row_box=none_i, first_row_begin=none_i, first_row_end=none_i, first_box_val=zeros_n,
        inf_row_box=none_i, inf_row_begin=none_i, inf_row_end=none_i, inf_box_val=zeros_n,
        inf_old_ptr=np.zeros(N + 1, dtype=np.int32), inf_old_slot=none_i,
        readout_w=zeros_n, readout_w_out=1.0,
        edge_box=np.full(NB_BINS + 1, -1, np.int32), edge_dx=edge_dx, bin_box=np.zeros(NB_BINS, np.int32),
        uniform_idx=np.full(S, -1, np.int32),
        dense=True, first_GT=gaussian_operator_T(T, _positions_first(T), N, sigmas),
        inf_GT=gaussian_operator_T(S + T, _positions_inf(T, tau, S), N, sigmas),
        bin_box2=no_box(NB_BINS), edge_box2=no_box(NB_BINS + 1), uniform_box2=no_box(S),
        psi=True, psi_edge=f32(gaussian_psi(mod, N, sigmas)), psi_bin=f32(gaussian_psi(bins[:-1], N, sigmas)),
        psi_uniform=f32(gaussian_psi(t_uni, N, sigmas)), psi_grid=f32(gaussian_psi(t, N, sigmas)),
        grid_w=wt.astype(np.float32))


def gaussian_operator_T(l: int, positions: torch.Tensor, N: int, sigmas) -> np.ndarray:
    """G^T [N, l] of compute_G (long_term_attention_gibbs.py:68-84) for the GAUSSIAN basis family of the reference
    (``add_gaussian_basis_functions``, :167-174: centres ``linspace(0, 1, N // len(sigmas))`` x widths ``sigmas``;
    ``GaussianBasisFunctions.evaluate``, basis_functions.py:135-160), with the reference's own fp32 ATen sequence.  No caller of
    the active module reaches this family; it exists here so that the dense ``x . G`` kernel is pinned on a second, fully
    dense operator (operator-level goldens: tests/golden/make_gaussian_goldens.py)."""
    import math
    mu, sigma = torch.meshgrid(torch.linspace(0, 1, N // len(sigmas)), torch.Tensor(list(sigmas)), indexing="ij")
    mu, sigma = mu.flatten().unsqueeze(0), sigma.flatten().unsqueeze(0)
    if mu.size(1) != N:
        raise UnsupportedBasis("num_basis must be a multiple of len(sigmas)")
    t = positions.unsqueeze(1)
    z = (t - mu) / sigma
    psi = (1. / math.sqrt(2 * math.pi) * torch.exp(-.5 * z ** 2)) / sigma         # evaluate(): [l', N]
    F = torch.zeros(N, positions.size(0))
    F[:, :] = psi.t()
    G = F.t().matmul((F.matmul(F.t()) + RIDGE_PENALTY * torch.eye(N)).inverse())
    b, e = _trimmed(G.size(0), l)
    return np.ascontiguousarray(G[b:e].t().numpy(), dtype=np.float32)


def gaussian_first_operator_T(T: int, N: int, sigmas) -> np.ndarray:
    return gaussian_operator_T(T, _positions_first(T), N, sigmas)


def _first_box(pairs: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(pairs[:, 0], dtype=np.int32)


def padded_N(N: int) -> int:
    """What the device allocates for ``num_basis`` = N: the kernels tile the basis dimension by 16."""
    return (N + 15) // 16 * 16


def _build_dense_plan(T: int, N: int, tau: float, S: int) -> Plan:
    """Plan for a num_basis whose boxes overlap where the step looks -- or that is no multiple of 16: dense operators from
    the reference's sequence and two-box tables for the histogram edges and the resampling points.  The 1000-point grid of
    expected_value() (long_term_attention_gibbs.py:251-286) must still see at most one box per point (true for every
    multiple of 16 up to 512 except 432): the read-out keeps its closed form.
    A num_basis that is no multiple of 16 is padded to the next one: the operators get zero rows for the extra basis
    functions (their coefficients stay 0), the read-out weights zero entries (they never receive attention mass), and no
    edge / bin / resampling table refers to them -- the N real basis functions compute exactly what the reference does."""
    t = torch.linspace(0, 1, GRID_POINTS)
    gbox2 = boxes2_of(t, N)
    # a point of the read-out grid in two boxes (num_basis 37, 432, ...): no count-weighted closed form -- the step then takes
    # the general-psi form with the rectangular psi itself as dense 0/1 rows (infv_ltm_set_psi_plan), like the Gaussian family
    grid_overlap = bool(int((gbox2[:, 1] >= 0).sum()))
    gbox = gbox2[:, 0]
    dx = (t[1:] - t[:-1]).double().numpy()
    wt = np.zeros(GRID_POINTS)
    wt[:-1] += dx / 2
    wt[1:] += dx / 2
    w = np.zeros(N)
    np.add.at(w, gbox[gbox >= 0], wt[gbox >= 0])
    bins = torch.linspace(0, 1, NB_BINS + 1)
    mod = bins.clone()
    mod[0] = -.000001
    mod[-1] = 1.000001
    edge_dx = (mod[1:] - mod[:-1]).numpy().astype(np.float32)
    t_uni = (torch.arange(1, S + 1).float() * tau / S) / tau
    bin_box2, edge_box2, uniform_box2 = boxes2_of(bins[:-1], N), boxes2_of(mod, N), boxes2_of(t_uni, N)
    none_i = np.zeros(0, dtype=np.int32)
    Np = padded_N(N)
    zeros_n = np.zeros(Np, dtype=np.float32)
    pad_rows = lambda GT: np.ascontiguousarray(np.concatenate([GT, np.zeros((Np - N, GT.shape[1]), np.float32)], 0))
    psi_kw = {}
    if grid_overlap:
        pad_cols = lambda m: np.ascontiguousarray(np.concatenate([m.numpy().astype(np.float32), np.zeros((m.shape[0], Np - N), np.float32)], 1))
        psi_kw = dict(psi=True, psi_edge=pad_cols(membership(mod, N)), psi_bin=pad_cols(membership(bins[:-1], N)),
                      psi_uniform=pad_cols(membership(t_uni, N)), psi_grid=pad_cols(membership(t, N)), grid_w=wt.astype(np.float32))
    return Plan(
        T=T, N=N, tau=tau, S=S,
        # the sparse operator tables are unused with a dense plan: empty but valid
        first_row_box=none_i, first_row_begin=none_i, first_row_end=none_i, first_box_val=zeros_n,
        inf_row_box=none_i, inf_row_begin=none_i, inf_row_end=none_i, inf_box_val=zeros_n,
        inf_old_ptr=np.zeros(Np + 1, dtype=np.int32), inf_old_slot=none_i,
        readout_w=np.concatenate([w, np.zeros(Np - N)]).astype(np.float32), readout_w_out=float(wt[gbox < 0].sum()),
        edge_box=_first_box(edge_box2), edge_dx=edge_dx, bin_box=_first_box(bin_box2), uniform_idx=_first_box(uniform_box2),
        dense=True, first_GT=pad_rows(_dense_operator_T(T, _positions_first(T), N)),
        inf_GT=pad_rows(_dense_operator_T(S + T, _positions_inf(T, tau, S), N)),
        bin_box2=np.ascontiguousarray(bin_box2), edge_box2=np.ascontiguousarray(edge_box2),
        uniform_box2=np.ascontiguousarray(uniform_box2), N_pad=Np if Np != N else 0, **psi_kw)


@lru_cache(maxsize=64)
def build_plan(T: int, N: int, tau: float, S: int = NB_SAMPLES) -> Plan:
    if T < 2:
        raise UnsupportedBasis("chunks of a single frame are empty in the reference (G[0:-0])")
    if N % 16:
        return _build_dense_plan(T, N, tau, S)            # (the reference takes any --num_basis: padded dense form)
    try:
        return _build_sparse_plan(T, N, tau, S)
    except OverlappingBoxes:
        return _build_dense_plan(T, N, tau, S)


def _build_sparse_plan(T: int, N: int, tau: float, S: int) -> Plan:
    # first-chunk operator
    frame_box, first_val = _operator(T, _positions_first(T), N)
    f_box, f_beg, f_end = _frame_ranges(frame_box)
    # infinite-memory operator: rows = [S resampled old rows ; T new frames]
    rows, inf_val = _operator(S + T, _positions_inf(T, tau, S), N)
    old_box, new_box = rows[:S], rows[S:]
    i_box, i_beg, i_end = _frame_ranges(new_box)
    order = np.argsort(np.where(old_box >= 0, old_box, N), kind="stable")
    kept = order[old_box[order] >= 0]
    old_ptr = np.zeros(N + 1, dtype=np.int32)
    np.add.at(old_ptr, old_box[kept] + 1, 1)
    old_ptr = np.cumsum(old_ptr).astype(np.int32)
    # read-out weights: trapezoid weights of linspace(0,1,1000) summed per box (:264-282)
    t = torch.linspace(0, 1, GRID_POINTS)
    dx = (t[1:] - t[:-1]).double().numpy()
    wt = np.zeros(GRID_POINTS)
    wt[:-1] += dx / 2
    wt[1:] += dx / 2
    gbox = box_of(t, N)
    w = np.zeros(N)
    np.add.at(w, gbox[gbox >= 0], wt[gbox >= 0])
    # sticky histogram (:163,197-199,207-208)
    bins = torch.linspace(0, 1, NB_BINS + 1)
    mod = bins.clone()
    mod[0] = -.000001
    mod[-1] = 1.000001
    edge_dx = (mod[1:] - mod[:-1]).numpy().astype(np.float32)
    # non-sticky resample positions (:153-157): psi(t / tau) for t in tau * s / S
    t_uni = (torch.arange(1, S + 1).float() * tau / S) / tau
    return Plan(
        T=T, N=N, tau=tau, S=S,
        first_row_box=f_box, first_row_begin=f_beg, first_row_end=f_end, first_box_val=first_val,
        inf_row_box=i_box, inf_row_begin=i_beg, inf_row_end=i_end, inf_box_val=inf_val,
        inf_old_ptr=old_ptr, inf_old_slot=kept.astype(np.int32),
        readout_w=w.astype(np.float32), readout_w_out=float(wt[gbox < 0].sum()),
        edge_box=box_of(mod, N).astype(np.int32), edge_dx=edge_dx,
        bin_box=box_of(bins[:-1], N).astype(np.int32), uniform_idx=box_of(t_uni, N).astype(np.int32),
    )
