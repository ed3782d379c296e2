"""CPU oracle for the infinity-Video LTM consolidation path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  The product path (``infinite-video_amd/``) never does: it fails loudly
when the HIP library is missing.

It restates, on the CPU, the algorithm of the reference operator

    /root/reference/infty-Video-LLaMA/InfVideoLLaMA/models/long_term_attention_gibbs.py
        (``LTM.py`` below; the VideoChat2 twin only differs in the pooling shape)
    /root/reference/infty-Video-LLaMA/InfVideoLLaMA/models/basis_functions.py:214-266
        (``BASIS.py`` below, rectangular family only)

in two independent forms:

* :class:`DenseOracle`  -- "reference-shaped": the same sequence of ATen calls the reference
  issues (per-call ridge operator with an N x N inverse, dense ``x @ G``, 129-edge density,
  ``Categorical.sample`` / ``torch.multinomial``, 1000-point integrand + ``trapz``).  It is the
  ``cpu_baseline`` ("port") that bench.py times, and it consumes torch's global CPU generator
  exactly as the reference does.
* :class:`ClosedFormOracle` -- the box-basis closed form (row gather + segmented ridge mean +
  count-weighted softmax) with the Gibbs uniforms ``u`` passed in explicitly.  This is the
  specification the HIP kernels are compared with on the GPU box.

PARITY PIN: the reference holds no tests or golden vectors for this path.
Both forms are therefore pinned against outputs of the reference itself, imported in the build
container by ``tests/golden/make_goldens.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` replays them).  The arithmetic of ``torch.multinomial``,
``torch.trapz``, ``torch.cumulative_trapezoid`` and ``Tensor.inverse`` lives in PyTorch, which
the reference does not pin; the goldens pin torch 2.10.0 CPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

NB_SAMPLES = 512       # LTM.py:55
RIDGE = 0.5            # LTM.py:62
NB_BINS = 128          # LTM.py:163  (129 edges)
GRID_POINTS = 1000     # LTM.py:251  (expected_value num_points)


# --------------------------------------------------------------------------------------
# rectangular basis (BASIS.py:214-266, built by LTM.py:176-182)
# --------------------------------------------------------------------------------------
def box_bounds(num_basis: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 lower / upper bounds of the N boxes, computed with the reference's expressions
    (LTM.py:177-181 for mu/width, BASIS.py:248-249 for the bounds)."""
    width = torch.ones(num_basis) / num_basis
    edges = torch.linspace(0, 1, num_basis + 1)
    mu = (edges[:-1] + edges[1:]) / 2
    return mu - width / 2, mu + width / 2


def box_membership(t: torch.Tensor, num_basis: int) -> torch.Tensor:
    """[M] fp32 points -> [M, N] float one-hot-or-zero rows (BASIS.py:238 / :250 / :266)."""
    lo, hi = box_bounds(num_basis)
    t = t.reshape(-1, 1).to(torch.float32)
    return ((t >= lo.unsqueeze(0)) & (t < hi.unsqueeze(0))).float()


def box_index(t: torch.Tensor, num_basis: int) -> np.ndarray:
    """[M] points -> int32 box index, -1 where the point lies in no box.
    Raises if a point lies in two boxes (cannot happen for power-of-two N)."""
    m = box_membership(t, num_basis)
    hits = m.sum(1)
    if bool((hits > 1).any()):
        raise ValueError("point inside two boxes: closed form not applicable for this N")
    idx = m.argmax(1).to(torch.int32)
    idx[hits == 0] = -1
    return idx.numpy()


# --------------------------------------------------------------------------------------
# positions (LTM.py:104-110 first chunk, LTM.py:135-150 infinite-memory update)
# --------------------------------------------------------------------------------------
def first_positions(length: int) -> torch.Tensor:
    """Padded sample positions of a first chunk of ``length`` frames (LTM.py:104-110)."""
    if length % 2:
        shift = 1 / float(length)
        return torch.linspace(-.5 + shift, 1.5 - shift, 2 * length - 1)
    shift = 1 / float(2 * length)
    return torch.linspace(-.5 + shift, 1.5 - shift, 2 * length)


def inf_positions(length: int, tau: float, nb_samples: int = NB_SAMPLES) -> torch.Tensor:
    """Padded positions of [old samples ; new frames] (LTM.py:135-150)."""
    old = torch.arange(1, nb_samples + 1).float()
    new = torch.arange(nb_samples + 1, length + nb_samples + 1).float()
    old = old * tau / nb_samples
    new = tau + (1 - tau) * (new - nb_samples) / length
    core = torch.cat([old, new], 0)
    if length % 2:
        shift = 1 / float(length + nb_samples)
        pad = torch.linspace(-.5 + shift, 1.5 - shift, 2 * (length + nb_samples) - 1)
    else:
        shift = 1 / float(2 * length + nb_samples)      # precedence as written at LTM.py:146
        pad = torch.linspace(-.5 + shift, 1.5 - shift, 2 * (length + nb_samples))
    return torch.cat([pad[pad < 0], core, pad[pad > 1]], 0)


def _trim(rows: int, l: int) -> slice:
    """Row range kept by the padding trim of compute_G (LTM.py:78-82)."""
    if l % 2:
        return slice((l - 1) // 2, rows + (-(l - 1) // 2))
    return slice(l // 2, rows - (l // 2))


def gaussian_membership(t: torch.Tensor, num_basis: int, sigmas) -> torch.Tensor:
    """[M] points -> [M, N] values of the reference's GAUSSIAN family: centres x widths of add_gaussian_basis_functions
    (LTM.py:167-174), psi(t) = phi((t - mu) / sigma) / sigma as GaussianBasisFunctions.evaluate / batch_evaluate compute it
    (BASIS.py:155-164; both are the same fp32 elementwise sequence)."""
    mu, sigma = torch.meshgrid(torch.linspace(0, 1, num_basis // len(sigmas)), torch.Tensor(list(sigmas)), indexing="ij")
    mu, sigma = mu.flatten().unsqueeze(0), sigma.flatten().unsqueeze(0)
    z = (t.reshape(-1, 1).to(torch.float32) - mu) / sigma
    return (1. / math.sqrt(2 * math.pi) * torch.exp(-.5 * z ** 2)) / sigma


def dense_ridge_operator(l: int, positions: torch.Tensor, num_basis: int, psi=None) -> torch.Tensor:
    """G = F^T (F F^T + ridge I)^-1, trimmed (LTM.py:68-84), as dense fp32 ATen calls; ``psi(t) -> [M, N]`` selects the
    basis family (default: the rectangular one)."""
    F = torch.zeros(num_basis, positions.size(0))
    F[:, :] = (psi(positions) if psi is not None else box_membership(positions, num_basis)).t()
    eye = torch.eye(num_basis)
    G = F.t().matmul((F.matmul(F.t()) + RIDGE * eye).inverse())
    return G[_trim(G.size(0), l), :]


# --------------------------------------------------------------------------------------
# closed-form maps  
# --------------------------------------------------------------------------------------
@dataclass
class BoxMaps:
    """Sparse form of the ridge operators for one (T, N, tau)."""
    first_col: np.ndarray   # [T]   int32 box of frame i on a first chunk (-1: dropped)
    first_val: np.ndarray   # [T]   fp32 1/(count+ridge) of that box
    inf_col: np.ndarray     # [S+T] int32 box of row r of x=[old samples ; new frames]
    inf_val: np.ndarray     # [S+T] fp32
    w: np.ndarray           # [N]   fp32 read-out weight of box n (trapz weights of the 1000-pt grid)
    w_out: float            # weight of the grid points in no box
    uniform_idx: np.ndarray  # [S]  int32 non-sticky resample rows (LTM.py:153-157), -1 = zero row


def _sparse_operator(l: int, positions: torch.Tensor, num_basis: int):
    idx = box_index(positions, num_basis)
    counts = np.bincount(idx[idx >= 0], minlength=num_basis).astype(np.float32)
    inv = (np.float32(1.0) / (counts + np.float32(RIDGE))).astype(np.float32)
    keep = _trim(len(idx), l)
    col = idx[keep].astype(np.int32)
    val = np.where(col >= 0, inv[np.maximum(col, 0)], np.float32(0)).astype(np.float32)
    return col, val


def readout_weights(num_basis: int, points: int = GRID_POINTS):
    """w_n = sum of trapezoid weights of linspace(0,1,points) inside box n (LTM.py:264-282)."""
    t = torch.linspace(0, 1, points)
    dx = (t[1:] - t[:-1]).double()
    wt = torch.zeros(points, dtype=torch.float64)
    wt[:-1] += dx / 2
    wt[1:] += dx / 2
    idx = box_index(t, num_basis)
    w = np.zeros(num_basis, dtype=np.float64)
    inside = idx >= 0
    np.add.at(w, idx[inside], wt.numpy()[inside])
    return w.astype(np.float32), float(wt.numpy()[~inside].sum())


def build_maps(length: int, num_basis: int, tau: float, nb_samples: int = NB_SAMPLES) -> BoxMaps:
    fc, fv = _sparse_operator(length, first_positions(length), num_basis)
    ic, iv = _sparse_operator(nb_samples + length, inf_positions(length, tau, nb_samples), num_basis)
    if len(fc) != length or len(ic) != nb_samples + length:
        raise ValueError("trim does not align with the sample rows for this length")
    w, w_out = readout_weights(num_basis)
    t_uni = torch.arange(1, nb_samples + 1).float() * tau / nb_samples / tau   # LTM.py:137,155
    return BoxMaps(fc, fv, ic, iv, w, w_out, box_index(t_uni, num_basis))


def sticky_bin_rows(num_basis: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Edges of the sticky histogram: modified edges (LTM.py:197-199), the box each edge
    evaluates (-1 = outside every box -> score 0), and the box of each *unmodified* left
    edge used to resample (LTM.py:207-208)."""
    bins = torch.linspace(0, 1, NB_BINS + 1)
    mod = bins.clone()
    mod[0] = -.000001
    mod[-1] = 1.000001
    return mod.numpy(), box_index(mod, num_basis), box_index(bins, num_basis)


# --------------------------------------------------------------------------------------
# Gibbs draw (LTM.py:203-206 -> torch.distributions.Categorical -> torch.multinomial, CPU)
# --------------------------------------------------------------------------------------
def categorical_probs(p_raw: torch.Tensor) -> torch.Tensor:
    """p as it reaches torch.multinomial: normalised at LTM.py:203 and again by Categorical."""
    p = p_raw / p_raw.sum(-1, keepdim=True)
    return p / p.sum(-1, keepdim=True)


def inverse_cdf_draw(probs: np.ndarray, u: np.ndarray) -> np.ndarray:
    """torch.multinomial(probs, len(u), True) on CPU for fp32 ``probs`` given its uniforms.

    torch 2.10 CPU: fp32 *sequential* running sum, divided by the fp32 total, last bucket
    forced to 1, then a lower-bound binary search of each float64 uniform (pinned against
    torch.multinomial itself in tests/test_oracle_golden.py)."""
    probs = np.asarray(probs, dtype=np.float32)
    run = np.float32(0)
    cdf = np.empty(len(probs), dtype=np.float32)
    for j in range(len(probs)):
        run = np.float32(run + probs[j])
        cdf[j] = run
    cdf = (cdf / run).astype(np.float32)
    cdf[-1] = np.float32(1)
    return np.searchsorted(cdf.astype(np.float64), np.asarray(u, dtype=np.float64), side="left").astype(np.int64)


def draw_uniforms(nb_samples: int = NB_SAMPLES, generator: Optional[torch.Generator] = None) -> np.ndarray:
    """The float64 uniforms one sticky step consumes from torch's CPU generator: ``nb_samples``
    for the bin draw, then as many again for the (degenerate, always 0) in-bin draw of
    LTM.py:206, which are discarded."""
    u = torch.rand(nb_samples, dtype=torch.float64, generator=generator)
    torch.rand(nb_samples, dtype=torch.float64, generator=generator)
    return u.numpy()


# --------------------------------------------------------------------------------------
# closed-form oracle
# --------------------------------------------------------------------------------------
class ClosedFormOracle:
    """One LTM instance (one cross-attention layer).  State: B_past [N,d], S_prev [H,Q,N]."""

    def __init__(self, num_basis: int, n_heads: int, head_size: int, tau: float, sticky: bool,
                 wk: np.ndarray, bk: np.ndarray, wv: np.ndarray, bv: np.ndarray,
                 tokens_per_frame: int = 32, nb_samples: int = NB_SAMPLES):
        self.N, self.H, self.dh = num_basis, n_heads, head_size
        self.tau, self.sticky, self.P, self.S = tau, sticky, tokens_per_frame, nb_samples
        self.wk, self.bk = np.asarray(wk, np.float32), np.asarray(bk, np.float32)
        self.wv, self.bv = np.asarray(wv, np.float32), np.asarray(bv, np.float32)
        self.B_past: Optional[np.ndarray] = None
        self.S_prev: Optional[np.ndarray] = None
        self._maps = {}
        self.edges, self.edge_box, self.bin_box = sticky_bin_rows(num_basis)
        # diagnostics of the last step
        self.last_p_raw = self.last_probs = self.last_bins = self.last_idx = None

    def maps(self, T: int) -> BoxMaps:
        if T not in self._maps:
            self._maps[T] = build_maps(T, self.N, self.tau, self.S)
        return self._maps[T]

    @staticmethod
    def pool(k: np.ndarray, P: int) -> np.ndarray:
        """k [T*P, d] -> frame means [T, d] (LTM.py:304)."""
        k = np.asarray(k, np.float32)
        T = k.shape[0] // P
        return torch.from_numpy(k).reshape(T, P, -1).mean(dim=1).numpy()

    def sticky_p_raw(self, S_prev: np.ndarray) -> np.ndarray:
        """Unnormalised bin masses p[127] from the previous scores (LTM.py:200-202)."""
        S_prev = torch.from_numpy(np.asarray(S_prev, np.float32))
        eb = torch.from_numpy(self.edge_box.astype(np.int64))
        sc = torch.where(eb >= 0, S_prev[..., eb.clamp(min=0)], torch.zeros(()))   # [H,Q,129]
        be = torch.from_numpy(self.edges)
        dens = torch.exp(sc)
        dens = dens / torch.trapz(dens, be, dim=-1).unsqueeze(-1)
        cum = torch.cumulative_trapezoid(dens, be, dim=-1)
        return (cum[..., 1:] - cum[..., :-1]).sum(dim=(0, 1)).numpy()

    def step(self, k: np.ndarray, q: np.ndarray, new_doc: bool, u: Optional[np.ndarray] = None,
             probs_override: Optional[np.ndarray] = None, bins_override: Optional[np.ndarray] = None) -> np.ndarray:
        """k [T*P, d], q [Q, H*dh] -> ctx [Q, H*dh].  ``u``: S float64 uniforms (sticky, not first).

        ``bins_override`` (test infrastructure): resample THESE bins (LTM.py:207-208 onwards) while the oracle's own
        probabilities and its own draw from ``u`` are still computed and kept in ``last_probs`` / ``last_bins`` --
        a long chain can then be compared draw by draw with a device path whose fp32 reduction order differs, without
        the two runs parting at the first uniform that falls within rounding of a cdf edge."""
        if new_doc:
            self.B_past = None                                         # LTM.py:300-302
        kbar = self.pool(k, self.P)
        T = kbar.shape[0]
        mp = self.maps(T)
        N = self.N
        if self.B_past is None:
            col, val, x = mp.first_col, mp.first_val, kbar             # LTM.py:218
            self.last_idx = None
        else:
            if self.sticky:
                if probs_override is None:
                    self.last_p_raw = self.sticky_p_raw(self.S_prev)
                    probs = categorical_probs(torch.from_numpy(self.last_p_raw)).numpy()
                else:
                    probs = np.asarray(probs_override, np.float32)
                self.last_probs = probs
                b = inverse_cdf_draw(probs, u)                         # LTM.py:204-205
                self.last_bins = b
                if bins_override is not None:
                    b = np.asarray(bins_override, np.int64)
                idx = self.bin_box[b]                                  # LTM.py:207-208
            else:
                idx = mp.uniform_idx                                   # LTM.py:212
            self.last_idx = idx
            rows = np.where(idx[:, None] >= 0, self.B_past[np.maximum(idx, 0)], np.float32(0))
            x = np.concatenate([rows.astype(np.float32), kbar], 0)     # LTM.py:210,215
            col, val = mp.inf_col, mp.inf_val
        B = np.zeros((N, x.shape[1]), np.float32)
        keep = col >= 0
        np.add.at(B, col[keep], val[keep, None] * x[keep])             # LTM.py:216 (x @ G)^T
        self.B_past = B
        Bt = torch.from_numpy(B)
        K = torch.nn.functional.linear(Bt, torch.from_numpy(self.wk), torch.from_numpy(self.bk))
        V = torch.nn.functional.linear(Bt, torch.from_numpy(self.wv), torch.from_numpy(self.bv))
        H, dh = self.H, self.dh
        Q = q.shape[0]
        qh = torch.from_numpy(np.asarray(q, np.float32)).view(Q, H, dh).transpose(0, 1) / (dh ** 0.5)
        Kh = K.view(N, H, dh).transpose(0, 1)
        Vh = V.view(N, H, dh).transpose(0, 1)
        S_cur = qh @ Kh.transpose(-1, -2)                              # [H,Q,N]  LTM.py:226-229
        w = torch.from_numpy(mp.w)
        e = w * torch.exp(S_cur)
        alpha = e / (e.sum(-1, keepdim=True) + mp.w_out)               # LTM.py:247-248,269-282
        ctx = alpha @ Vh                                               # [H,Q,dh]  LTM.py:284
        self.S_prev = S_cur.numpy()
        return ctx.transpose(0, 1).reshape(Q, H * dh).numpy()          # LTM.py:346


# --------------------------------------------------------------------------------------
# reference-shaped dense oracle (the cpu_baseline "port")
# --------------------------------------------------------------------------------------
class DenseOracle:
    """Same ATen op sequence as LTM.py:288-346 (minus the density-pickle side effect of
    :320-345, whose result the model never reads; ``density_side_effect=True`` re-adds its
    four ``compute_probability`` calls, without the disk write, for timing studies).

    Consumes torch's *global* CPU generator exactly like the reference: call
    ``torch.manual_seed`` before :meth:`forward` to fix the Gibbs draw."""

    def __init__(self, num_basis: int, n_heads: int, head_size: int, tau: float, sticky: bool,
                 proj_key: torch.nn.Linear, proj_value: torch.nn.Linear,
                 pool_shape: Sequence[int] = (32,), nb_samples: int = NB_SAMPLES,
                 density_side_effect: bool = False, gaussian_sigmas: Sequence[float] = None):
        """``gaussian_sigmas``: the reference module with its builder hook pointed at its own Gaussian builder
        (add_gaussian_basis_functions, LTM.py:167-174) -- every psi evaluation below then uses that family."""
        self.N, self.H, self.dh = num_basis, n_heads, head_size
        self._psi = (lambda t: gaussian_membership(t, num_basis, gaussian_sigmas)) if gaussian_sigmas else \
                    (lambda t: box_membership(t, num_basis))
        self.tau, self.sticky, self.S = tau, sticky, nb_samples
        self.proj_key, self.proj_value = proj_key, proj_value
        self.pool_shape = tuple(pool_shape)
        self.P = int(np.prod(self.pool_shape))
        self.density_side_effect = density_side_effect
        self.B_past = None
        self.queries = self.keys = self.values = None

    # -- per-call basis construction (LTM.py:67-165; rebuilt on every forward, :298) --
    def _build(self, L: int):
        N = self.N
        self.G_first = dense_ridge_operator(L, first_positions(L), N, self._psi)
        self.G_inf = dense_ridge_operator(self.S + L, inf_positions(L, self.tau, self.S), N, self._psi)
        old = torch.arange(1, self.S + 1).float() * self.tau / self.S
        rows = None
        for t in old:                                   # LTM.py:153-157 (S sequential cats)
            r = self._psi((t / self.tau).reshape(1))
            rows = r if rows is None else torch.cat([rows, r], 0)
        self.uniform_samples = rows
        self.bins = torch.linspace(0, 1, NB_BINS + 1)

    def _scores_at(self, t: torch.Tensor) -> torch.Tensor:
        psis = self._psi(t)                                          # [M,N]  LTM.py:225
        query = self.queries / (self.dh ** 0.5)
        keys = torch.matmul(self.keys.transpose(-1, -2), psis.T)     # [1,H,dh,M]
        return torch.matmul(query, keys)                             # [1,H,Q,M]

    def _density(self, t: torch.Tensor) -> torch.Tensor:
        sc = self._scores_at(t)                                      # LTM.py:247-248
        return torch.exp(sc) / torch.trapz(torch.exp(sc), t, dim=-1).unsqueeze(-1)

    def _update(self, x: torch.Tensor) -> torch.Tensor:
        if self.B_past is not None:
            if self.sticky:
                bins = self.bins.clone()
                bins[0] = -.000001
                bins[-1] = 1.000001
                dens = self._density(bins)
                cum = torch.cumulative_trapezoid(dens, bins, dim=-1)
                p = (cum[..., 1:] - cum[..., :-1]).sum(dim=(1, 2))
                p = p / p.sum(-1, keepdim=True)
                cat = torch.distributions.Categorical(p)          # normalises p a second time
                self.last_probs = cat.probs
                b = cat.sample((self.S,))                         # S float64 uniforms
                t = torch.distributions.Categorical(torch.ones(1)).sample((self.S, 1))
                ts = (t * (self.bins[b + 1] - self.bins[b]) / 1 + self.bins[b]).transpose(1, 0)
                self.last_bins = b.reshape(-1)
                samples = self._psi(ts[0]).contiguous()
            else:
                samples = self.uniform_samples
            old = self.B_past.transpose(-1, -2).matmul(samples.transpose(-1, -2))   # [1,d,S]
            x = torch.cat([old, x], dim=2)
            B = torch.matmul(x, self.G_inf).permute(0, 2, 1)
        else:
            B = torch.matmul(x, self.G_first).permute(0, 2, 1)
        self.B_past = B.detach()
        return B

    def forward(self, k: torch.Tensor, q: torch.Tensor, new_doc: bool) -> torch.Tensor:
        """k [1, T*P, d], q [1, Q, H*dh] -> [1, Q, H*dh]."""
        L = k.size(1) // self.P
        Q = q.size(1)
        self._build(L)
        if new_doc:
            self.B_past = None
        kb = k.reshape(1, L, *self.pool_shape, k.size(-1)).mean(dim=tuple(range(2, 2 + len(self.pool_shape))))
        B = self._update(kb.transpose(1, 2))
        keys, values = self.proj_key(B), self.proj_value(B)
        self.queries = q.view(1, Q, self.H, self.dh).transpose(1, 2)
        self.keys = keys.view(1, self.N, self.H, self.dh).transpose(1, 2)
        self.values = values.view(1, self.N, self.H, self.dh).transpose(1, 2)
        # expected value on the 1000-point grid (LTM.py:251-286)
        t = torch.linspace(0, 1, GRID_POINTS)
        psi = self._psi(t)                                            # [M,N]
        prob = self._density(t)                                       # [1,H,Q,M]
        # p(t) psi_n(t) as the reference materialises it (LTM.py:276-282): a batched
        # [.,1,1] x [.,1,N] outer product per grid point -> [1,H,Q,N,M] (393 MB at the
        # headline shape), then a trapezoid rule along the grid axis.
        M = GRID_POINTS
        col = prob.movedim(-1, 0).reshape(M, 1, self.H, Q, 1, 1)
        row = psi.reshape(M, 1, 1, 1, 1, self.N).expand(M, 1, self.H, Q, 1, self.N)
        integrand = torch.matmul(col, row).squeeze(-2).movedim(0, -1)
        integral = torch.trapz(integrand, t, dim=-1)                  # [1,H,Q,N]
        ctx = torch.matmul(integral, self.values)                     # [1,H,Q,dh]
        if self.density_side_effect:                                   # LTM.py:328-339
            for lo, hi in ((0, .25), (.25, .5), (.5, 1)):
                self._density(torch.linspace(lo, hi, 256))
            self._density(torch.linspace(0, 1, 2048))
        return ctx.contiguous().transpose(1, 2).reshape(1, Q, -1)

    def scores(self) -> torch.Tensor:
        """S[h,q,n] of the last forward, for seeding a ClosedFormOracle / HIP state."""
        return (self.queries / (self.dh ** 0.5) @ self.keys.transpose(-1, -2))[0]
