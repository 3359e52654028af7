"""Seeded synthetic slabs in the reference's chunk format (SURVEY.md 8d "kernel-only
microbench"): raw ~ N(0,1) [B,T_r,1] with the last U{0..15} positions zero-padded, events
~ N(0,1) [B,T_e,5] with U{0..10} trailing zero rows.  No signal data ships with the reference
(only FASTA), so every measurement in this repo runs on these."""
from __future__ import annotations

import numpy as np


def make_slab(B, T_r, T_e, seed=0, max_raw_pad=15, max_event_pad=10, L=48):
    rng = np.random.default_rng(seed)
    raw = rng.standard_normal((B, T_r, 1)).astype(np.float32)
    ev = rng.standard_normal((B, T_e, 5)).astype(np.float32)
    rp = rng.integers(0, max_raw_pad + 1, B)
    ep = rng.integers(0, max_event_pad + 1, B)
    for b in range(B):
        if rp[b]:
            raw[b, T_r - min(rp[b], T_r):] = 0.0
        if ep[b]:
            ev[b, T_e - min(ep[b], T_e):] = 0.0
    nuc = np.zeros((B, L), np.int64)    # target tokens are only used for their length L
    nuc[:, 0] = 2
    return raw, ev, nuc


def hash_slab(B, T_r, T_e, seed=0):
    """Exactly reproducible variant (splitmix64) for committed golden fixtures."""
    from .weights import _splitmix_uniform
    raw = (_splitmix_uniform(seed * 7 + 1, B * T_r) * 2.0).reshape(B, T_r, 1).astype(np.float32)
    ev = (_splitmix_uniform(seed * 7 + 2, B * T_e * 5) * 2.0).reshape(B, T_e, 5).astype(np.float32)
    for b in range(B):
        rp, ep = (3 * b + seed) % 16, (2 * b + seed) % 11
        if rp:
            raw[b, T_r - min(rp, T_r):] = 0.0
        if ep:
            ev[b, T_e - min(ep, T_e):] = 0.0
    return raw, ev


def make_read(n_bases=600, seed=0, mean_dwell=9.0):
    """A synthetic nanopore read in the reference's file form: int signal + labels
    (start, end, base).  6-mer -> level table ~ N(0,1) from seed 1234 (DeepSimulator's pore model
    is not available), dwell 5 + Geometric clipped to [3,40], Gaussian noise, quantised to int."""
    rng = np.random.default_rng(seed)
    bases = rng.integers(0, 4, n_bases)
    table = np.random.default_rng(1234).standard_normal(4096)
    kmer = np.zeros(n_bases, np.int64)
    for i in range(n_bases):
        k = 0
        for j in range(6):
            k = k * 4 + bases[min(max(i - 2 + j, 0), n_bases - 1)]
        kmer[i] = k
    dwell = np.clip(5 + rng.geometric(1.0 / max(mean_dwell - 5.0, 1.0), n_bases) - 1, 3, 40)
    levels = np.repeat(table[kmer], dwell)
    signal = np.round(500 + 80 * (levels + 0.15 * rng.standard_normal(levels.size))).astype(int)
    ends = np.cumsum(dwell)
    labels = np.column_stack([ends - dwell, ends, np.array(list("ACGT"))[bases]]).astype(object)
    return signal, labels
