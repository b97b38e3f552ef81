"""Synthetic GROUP BY workloads of the benchmark configs, following the shape of the
reference's examples/performance_10k_groups_R/generate_test_data.sql:37-51,61-79,144-165:

    per group   beta0 ~ U(-10, 10),  beta_j ~ U(-5, 5)
    per row     x_ij ~ U(-10, 10) iid,   y = beta0 + sum_j beta_j x_ij + 2.0 * N(0, 1)   (Box-Muller)
    WLS weight  U(0, 1) + 0.5            (examples/performance_1m_groups/benchmark_wls.sql:14)

Counter-based (SplitMix64 of seed / group / row / column), so any shard of the groups can be generated
independently, on any device, and reproduces the same values.  Written with torch integer ops so the same
code runs on the host and on the GPU; data are laid out as "grouped columns" (one array per feature).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

_M64 = (1 << 64)


def _s64(v: int) -> int:
    v &= _M64 - 1
    return v - _M64 if v >= (1 << 63) else v


_GAMMA = _s64(0x9E3779B97F4A7C15)
_C1 = _s64(0xBF58476D1CE4E5B9)
_C2 = _s64(0x94D049BB133111EB)


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix64(x: torch.Tensor) -> torch.Tensor:
    z = x + _GAMMA
    z = (z ^ _lsr(z, 30)) * _C1
    z = (z ^ _lsr(z, 27)) * _C2
    return z ^ _lsr(z, 31)


def _uniform01(bits: torch.Tensor) -> torch.Tensor:
    """53 random bits -> (0, 1) open interval."""
    return (_lsr(bits, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _stream(seed: int, g: torch.Tensor, r: torch.Tensor, col: int) -> torch.Tensor:
    h = _mix64(g * _s64(0xD1B54A32D192ED03) + seed)
    h = _mix64(h ^ (r * _s64(0xAEF17502108EF2D9) + col))
    return h


def make_grouped(n_groups: int, n_per_group: int, p: int, *, seed: int = 42, group_start: int = 0,
                 weights: bool = False, device="cpu", chunk_groups: int = 8192
                 ) -> Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor], Optional[torch.Tensor]]:
    """Returns (row_offsets int64[G+1], y[N], x_cols: p x [N], w[N] or None), N = G * n_per_group."""
    dev = torch.device(device)
    N = n_groups * n_per_group
    y = torch.empty(N, dtype=torch.float64, device=dev)
    x_cols = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(p)]
    w = torch.empty(N, dtype=torch.float64, device=dev) if weights else None
    offsets = torch.arange(0, n_groups + 1, dtype=torch.int64, device=dev) * n_per_group
    r = torch.arange(n_per_group, dtype=torch.int64, device=dev)
    for c0 in range(0, n_groups, chunk_groups):
        c1 = min(n_groups, c0 + chunk_groups)
        g = torch.arange(group_start + c0, group_start + c1, dtype=torch.int64, device=dev)
        gg = g[:, None].expand(-1, n_per_group)
        rr = r[None, :].expand(c1 - c0, -1)
        sl = slice(c0 * n_per_group, c1 * n_per_group)
        beta0 = _uniform01(_stream(seed, g, torch.zeros_like(g) - 1, 0)) * 20.0 - 10.0
        acc = beta0[:, None].expand(-1, n_per_group).clone()
        for j in range(p):
            beta_j = _uniform01(_stream(seed, g, torch.zeros_like(g) - 1, 1 + j)) * 10.0 - 5.0
            xj = _uniform01(_stream(seed, gg, rr, 1 + j)) * 20.0 - 10.0
            x_cols[j][sl] = xj.reshape(-1)
            acc += beta_j[:, None] * xj
        u1 = _uniform01(_stream(seed, gg, rr, 100))
        u2 = _uniform01(_stream(seed, gg, rr, 101))
        noise = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)
        y[sl] = (acc + 2.0 * noise).reshape(-1)
        if weights:
            w[sl] = (_uniform01(_stream(seed, gg, rr, 102)) + 0.5).reshape(-1)
    return offsets, y, x_cols, w
