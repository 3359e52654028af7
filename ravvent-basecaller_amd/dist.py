"""Multi-GPU form of the hot path: chunks shard embarrassingly.

One process per GPU; a read's (or a queue's) chunk range is cut into `world` contiguous, nearly
equal pieces (contiguity keeps the order the merger needs); weights are replicated; every rank
decodes its piece with its own `Basecaller`; ONE fixed-shape all-gather (RCCL over xGMI on
MI355X, gloo on CPU in the tests) of `[n_max, L-1]` tokens + scores ends the call.  No collective
touches the data path before that.  The reference has no distributed code at all
(SURVEY.md 2.1); this module is new, not a port.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of rank's chunks; the first n % world ranks take one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_calls(tokens: torch.Tensor, scores: torch.Tensor, steps: int, n_total: int, max_steps: int,
                 end_token: int = 1, group=None):
    """All-gather per-rank results into read order.

    tokens/scores: this rank's [n_local, S_local] (S_local <= max_steps); returns
    (tokens [n_total, S], scores [n_total, S]) with S = max over ranks of S_local.  The reference
    loop runs until EVERY row of the slab is finished, so a rank whose shard
    finished earlier is extended with exactly what that loop would have emitted for its rows had
    it run on: all beams finished => the end token, at an unchanged top-1 score.  The gathered
    result is therefore identical to the single-GPU result for the whole slab."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = tokens.device
    n_max = -(-n_total // world)
    pt = torch.full((n_max, max_steps), end_token, dtype=torch.int32, device=dev)
    ps = torch.zeros((n_max, max_steps), dtype=torch.float32, device=dev)
    n_loc, s_loc = tokens.shape
    pt[:n_loc, :s_loc] = tokens
    ps[:n_loc, :s_loc] = scores
    if 0 < s_loc < max_steps:
        ps[:n_loc, s_loc:] = scores[:, s_loc - 1:s_loc]
    gt = torch.empty((world * n_max, max_steps), dtype=torch.int32, device=dev)
    gs = torch.empty((world * n_max, max_steps), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(gt, pt, group=group)
    dist.all_gather_into_tensor(gs, ps, group=group)
    s = torch.tensor([steps], dtype=torch.int32, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.MAX, group=group)
    S = int(s.item())
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rows.append(torch.arange(r * n_max, r * n_max + (hi - lo), device=dev))
    idx = torch.cat(rows)
    return gt[idx, :S], gs[idx, :S]


def sharded_beam_search(basecaller, raw, event, beam_width: int, max_output_len: int, group=None):
    """Decode this rank's contiguous shard of the slab and gather everyone's calls.
    raw / event: the FULL slab (host arrays or tensors); every rank holds the same inputs."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = (raw if raw is not None else event).shape[0]
    lo, hi = shard_range(n, rank, world)
    pick = lambda x: None if x is None else x[lo:hi]
    mode = basecaller.input_data_type
    inp = {"joint": (pick(raw), pick(event)), "raw": pick(raw), "event": pick(event)}[mode]
    tok, sc = basecaller.beam_search_prediction(inp, beam_width=beam_width, max_output_len=max_output_len)
    return gather_calls(tok, sc, tok.shape[1], n, max(int(max_output_len) - 1, 0),
                        end_token=int(basecaller.output_end_token), group=group)
