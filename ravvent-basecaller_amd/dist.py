"""Multi-GPU form of the hot path: chunks shard embarrassingly.

One process per GPU; a read's (or a queue's) chunk range is cut into `world` contiguous, nearly
equal pieces (contiguity keeps the order the merger needs); weights are replicated; every rank
decodes its piece with its own `Basecaller`; ONE fixed-shape all-gather (RCCL over xGMI on
MI355X, gloo on CPU in the tests) ends the call: tokens, score bits and the shard's step count
travel in a single int32 tensor `[n_max, 2 (L-1) + 1]`.  No collective touches the data path
before that.  The reference has no distributed code at all; this module is new,
not a port.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of rank's chunks; the first n % world ranks take one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _collective_device(like: torch.Tensor, group=None, device=None) -> torch.device:
    """Where the gather buffers live: RCCL moves device memory only, gloo host memory."""
    if device is not None:
        return torch.device(device)
    if dist.get_backend(group) == "nccl":
        return like.device if like.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


_INDEX: dict = {}


def _read_order_index(n_total: int, world: int, n_max: int, dev) -> torch.Tensor:
    """Rows of the rank-major gathered buffer in read order (cached: it only depends on the shapes)."""
    key = (n_total, world, n_max, str(dev))
    if key not in _INDEX:
        if len(_INDEX) > 64:
            _INDEX.clear()
        rows = []
        for r in range(world):
            lo, hi = shard_range(n_total, r, world)
            rows.append(torch.arange(r * n_max, r * n_max + (hi - lo), device=dev))
        _INDEX[key] = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.long, device=dev)
    return _INDEX[key]


_BUFFERS: dict = {}


def _buffers(kind, rows, cols, world, dev):
    key = (kind, rows, cols, world, str(dev))
    if key not in _BUFFERS:
        if len(_BUFFERS) > 16:
            _BUFFERS.clear()
        _BUFFERS[key] = (torch.empty((rows, cols), dtype=torch.int32, device=dev),
                         torch.empty((world * rows, cols), dtype=torch.int32, device=dev))
    return _BUFFERS[key]


def gather_calls(tokens: torch.Tensor, scores: torch.Tensor, steps: int, n_total: int, max_steps: int,
                 end_token: int = 1, group=None, device=None):
    """All-gather per-rank results into read order with ONE collective.

    tokens/scores: this rank's [n_local, S_local] (S_local <= max_steps); returns
    (tokens [n_total, S], scores [n_total, S]) with S = max over ranks of S_local.  The reference
    loop runs until EVERY row of the slab is finished, so a rank whose shard
    finished earlier is extended with exactly what that loop would have emitted for its rows had
    it run on: all beams finished => the end token, at an unchanged top-1 score.  The gathered
    result is therefore identical to the single-GPU result for the whole slab.

    Wire format, one int32 row per chunk: [tokens (L-1) | score bits (L-1) | S_local]."""
    world = dist.get_world_size(group)
    dev = _collective_device(tokens, group, device)
    n_max = -(-n_total // world) if n_total else 0
    cols = 2 * max_steps + 1
    packed, gathered = _buffers("calls", max(n_max, 1), cols, world, dev)
    n_loc, s_loc = tokens.shape
    full = n_loc == packed.shape[0] and s_loc == max_steps      # the usual case: every cell of `packed` is overwritten below
    if not full:
        packed[:, :max_steps] = end_token
        packed[:, max_steps:2 * max_steps] = 0
    packed[:, 2 * max_steps] = int(steps)
    if n_loc and s_loc:
        packed[:n_loc, :s_loc] = tokens.to(dev, torch.int32)
        sc = scores.to(dev, torch.float32).contiguous()
        packed[:n_loc, max_steps:max_steps + s_loc] = sc.view(torch.int32)
        if s_loc < max_steps:      # finished shard: unchanged top-1 score on the steps the slab-wide loop would still run
            packed[:n_loc, max_steps + s_loc:2 * max_steps] = sc[:, s_loc - 1:s_loc].view(torch.int32)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    S = int(gathered[:, 2 * max_steps].max().item()) if n_total else 0
    if n_total == world * packed.shape[0]:                      # equal shards: the gathered rows already are in read order
        out = gathered
    else:
        out = gathered[_read_order_index(n_total, world, packed.shape[0], dev)]
    # fresh tensors: `gathered` is a cached buffer the next call overwrites
    return out[:, :S].clone(), out[:, max_steps:max_steps + S].clone().view(torch.float32)


def _decode_shard(basecaller, raw, event, lo, hi, beam_width, max_output_len, slab):
    """This rank's [lo, hi) in slabs of at most `slab` chunks -> ([n_local, S_local] tokens, scores, S_local); slabs that
    stop earlier are extended like a shard that stops earlier (see gather_calls)."""
    mode = basecaller.input_data_type
    steps = max(int(max_output_len) - 1, 0)
    end_token = int(basecaller.output_end_token)
    slab = int(slab) if slab else max(hi - lo, 1)
    def cut(a, b):
        pick = lambda x: None if x is None else x[a:b]
        return {"joint": (pick(raw), pick(event)), "raw": pick(raw), "event": pick(event)}[mode]
    cuts = [(a, min(a + slab, hi)) for a in range(lo, hi, slab)]
    if len(cuts) > 1 and hasattr(basecaller, "beam_search_stream"):      # the shard's slabs in flight together (identical results)
        parts = list(basecaller.beam_search_stream((cut(a, b) for a, b in cuts), beam_width, max_output_len))
    else:
        parts = [basecaller.beam_search_prediction(cut(a, b), beam_width=beam_width, max_output_len=max_output_len) for a, b in cuts]
    if len(parts) == 1:
        tok, sc = parts[0]
        return tok, sc, tok.shape[1]
    if not parts:
        return torch.zeros((0, 0), dtype=torch.int32), torch.zeros((0, 0)), 0
    S = max(t.shape[1] for t, _ in parts)
    dev = parts[0][0].device
    tok = torch.full((hi - lo, S), end_token, dtype=torch.int32, device=dev)
    sc = torch.zeros((hi - lo, S), dtype=torch.float32, device=dev)
    row = 0
    for t, s in parts:
        n, si = t.shape
        tok[row:row + n, :si] = t
        sc[row:row + n, :si] = s
        if 0 < si < S:
            sc[row:row + n, si:] = s[:, si - 1:si]
        row += n
    assert S <= steps
    return tok, sc, S


def sharded_beam_search(basecaller, raw, event, beam_width: int, max_output_len: int, group=None, slab: int | None = None):
    """Decode this rank's contiguous shard of the slab and gather everyone's calls (one collective).
    raw / event: the FULL slab (host arrays or tensors, host or device); every rank holds the same inputs.
    `slab`: decode the shard in pieces of at most that many chunks (a shard larger than the handle's max_batch)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = (raw if raw is not None else event).shape[0]
    lo, hi = shard_range(n, rank, world)
    tok, sc, S = _decode_shard(basecaller, raw, event, lo, hi, beam_width, max_output_len, slab)
    dev = getattr(basecaller, "device", None) if dist.get_backend(group) == "nccl" else None
    return gather_calls(tok, sc, S, n, max(int(max_output_len) - 1, 0),
                        end_token=int(basecaller.output_end_token), group=group, device=dev)


def sharded_beam_search_stream(basecaller, slabs, beam_width: int, max_output_len: int, group=None, slab: int | None = None):
    """Generator over an iterable of FULL slabs (raw, event): every rank decodes its shard of slab k + 1 (asynchronous calls) while
    slab k's results are gathered -- the pipelined form of a loop over `sharded_beam_search`, one collective per slab, identical
    results.  Shards larger than `slab` (or than the handle's max_batch) fall back to the slab-at-a-time form."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    steps = max(int(max_output_len) - 1, 0)
    end_token = int(basecaller.output_end_token)
    dev = getattr(basecaller, "device", None) if dist.get_backend(group) == "nccl" else None
    depth = max(int(getattr(basecaller, "async_depth", 2)), 1)
    mode = basecaller.input_data_type
    queue = []

    def finish(item):
        call, n = item
        if call is None:
            tok, sc = torch.zeros((0, 0), dtype=torch.int32), torch.zeros((0, 0))
        else:
            tok, sc = basecaller.collect(call)
        return gather_calls(tok, sc, tok.shape[1], n, steps, end_token=end_token, group=group, device=dev)

    # (a consumer that abandons the generator early leaves collective calls unmatched on the other ranks: iterate it to the end)
    for raw, event in slabs:
        n = (raw if raw is not None else event).shape[0]
        lo, hi = shard_range(n, rank, world)
        if not hasattr(basecaller, "submit_beam_search") or (slab and hi - lo > slab):
            while queue:
                yield finish(queue.pop(0))
            yield sharded_beam_search(basecaller, raw, event, beam_width, max_output_len, group=group, slab=slab)
            continue
        if len(queue) >= depth:
            yield finish(queue.pop(0))
        pick = lambda x: None if x is None else x[lo:hi]
        inp = {"joint": (pick(raw), pick(event)), "raw": pick(raw), "event": pick(event)}[mode]
        queue.append((basecaller.submit_beam_search(inp, beam_width, max_output_len) if hi > lo else None, n))
    while queue:
        yield finish(queue.pop(0))


_MANY: dict = {}


def _many_buffers(K, n_max, steps, world, dev, reuse):
    """(packed [K, F], gathered [world, K, F], tokens [K, world n_max, L-1], score bits likewise), F = 2 n_max (L-1) + n_max."""
    F = 2 * n_max * steps + n_max
    def make():
        return (torch.empty((K, F), dtype=torch.int32, device=dev), torch.empty((world, K, F), dtype=torch.int32, device=dev),
                torch.empty((K, world * n_max, steps), dtype=torch.int32, device=dev),
                torch.empty((K, world * n_max, steps), dtype=torch.int32, device=dev))
    if not reuse:
        return make()
    key = (K, n_max, steps, world, str(dev))
    if key not in _MANY:
        if len(_MANY) > 8:
            _MANY.clear()
        _MANY[key] = make()
    return _MANY[key]


def reserve_many_buffers(basecaller, n_slabs: int, chunks_per_slab: int, max_output_len: int, group=None):
    """Allocate the gather buffers `sharded_beam_search_many(..., reuse_buffers=True)` will use for a queue of `n_slabs` slabs of
    `chunks_per_slab` chunks, ahead of time (a service sizes them once; a first-use allocation of a few MB inside a timed region costs
    more than the collective)."""
    world = dist.get_world_size(group)
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device(getattr(basecaller, "device", None) or torch.device("cuda", torch.cuda.current_device())) if nccl else torch.device("cpu")
    for t in _many_buffers(int(n_slabs), max(-(-int(chunks_per_slab) // world), 1), max(int(max_output_len) - 1, 0), world, dev, True):
        t.zero_()                                              # (touched once: the first kernels to write them do not find them cold)


def sharded_beam_search_many(basecaller, slabs, beam_width: int, max_output_len: int, group=None, slab: int | None = None,
                             reuse_buffers: bool = False):
    """A whole queue of FULL slabs (raw, event) with ONE collective at the end (north_star: "read-chunks shard across the GPUs with a
    single RCCL gather at the end"): every rank streams its shards of all the slabs through the asynchronous calls and one
    `all_gather_into_tensor` returns everything; no host synchronisation with the other ranks before that.  Returns a list of
    (tokens [n_k, S_k], scores [n_k, S_k]) in slab order, each identical to `sharded_beam_search` of that slab.
    `reuse_buffers`: gather and result buffers are kept per shape (see `reserve_many_buffers`); the returned tensors are then views
    that stay valid until the next call with the same shapes.

    Wire format, one int32 row per slab: [token plane n_max x (L-1) | score-bit plane n_max x (L-1) | steps per row n_max] with
    n_max = the largest shard of any slab.  With device inputs the library writes its results straight into the planes (the
    `out` tensors of `Basecaller.submit_beam_search`): the host issues no copy per slab.  A row that stopped at step s < S_k (its
    piece or its shard finished earlier than the slab-wide loop would) is extended after the gather with what that loop emits:
    the end token at an unchanged score."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    slabs = list(slabs)
    steps = max(int(max_output_len) - 1, 0)
    end_token = int(basecaller.output_end_token)
    mode = basecaller.input_data_type
    ns = [(raw if raw is not None else event).shape[0] for raw, event in slabs]
    K = len(slabs)
    if K == 0:
        return []
    n_max = max(max(-(-n // world) for n in ns), 1)
    plane = n_max * steps
    limit = int(slab) if slab else None
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device(getattr(basecaller, "device", None) or torch.device("cuda", torch.cuda.current_device())) if nccl else torch.device("cpu")
    packed, gathered, tok, sc = _many_buffers(K, n_max, steps, world, dev, reuse_buffers)
    tokP = packed[:, :plane].view(K, n_max, steps)             # views built once: the per-slab host work below is what delays the
    scP = packed[:, plane:2 * plane].view(K, n_max, steps)     # next submit while the GPU waits for it
    tok_plane = lambda k: tokP[k]
    sc_plane = lambda k: scP[k]
    cuts = []                                                  # (slab, first row, rows, first chunk) of this rank's calls, in order
    for k, n in enumerate(ns):
        lo, hi = shard_range(n, rank, world)
        step = limit if limit else max(hi - lo, 1)
        cuts += [(k, a - lo, min(a + step, hi) - a, a) for a in range(lo, hi, step)]

    def inputs():
        for k, _, rows, a in cuts:
            raw, event = slabs[k]
            whole = a == 0 and rows == ns[k]
            pick = lambda x: None if x is None else (x if whole else x[a:a + rows])
            yield {"joint": (pick(raw), pick(event)), "raw": pick(raw), "event": pick(event)}[mode]
    first = next((x for x in slabs[0] if x is not None))
    direct = (getattr(basecaller, "supports_out", False) and torch.is_tensor(first) and first.is_cuda and dev.type == "cuda" and steps > 0)
    steps_of = []                                              # (slab, first row, rows, steps) per call
    if direct:
        # device inputs: the library writes every call's tokens and scores straight into the planes -- raw addresses, so that the host
        # does nothing per slab but submit and collect (its time per slab is what the GPU waits for whenever the queue runs low)
        base, depth, queue = packed.data_ptr(), max(int(getattr(basecaller, "async_depth", 2)), 1), []
        def finish(item):
            c, call = item
            steps_of.append((c[0], c[1], c[2], basecaller.collect(call)))
        try:
            for c, x in zip(cuts, inputs()):
                if len(queue) >= depth:
                    finish(queue.pop(0))
                k, r0 = c[0], c[1]
                off = 4 * (k * (2 * plane + n_max) + r0 * steps)
                queue.append((c, basecaller.submit_beam_search(x, beam_width, max_output_len, out_ptrs=(base + off, base + off + 4 * plane))))
            while queue:
                finish(queue.pop(0))
        finally:                                               # (an error: no ticket may stay uncollected on the handle)
            while queue:
                try:
                    basecaller.collect(queue.pop(0)[1])
                except Exception:
                    pass
    else:
        if hasattr(basecaller, "beam_search_stream"):
            results = basecaller.beam_search_stream(inputs(), beam_width, max_output_len)
        else:
            results = (basecaller.beam_search_prediction(x, beam_width=beam_width, max_output_len=max_output_len) for x in inputs())
        for (k, r0, rows, _), (tk, ss) in zip(cuts, results):
            s_loc = int(tk.shape[1])
            if rows and s_loc:
                tok_plane(k)[r0:r0 + rows, :s_loc] = tk.to(dev, torch.int32)
                sc_plane(k)[r0:r0 + rows, :s_loc] = ss.to(dev, torch.float32).contiguous().view(torch.int32)
            steps_of.append((k, r0, rows, s_loc))
    srow_host = torch.zeros((K, n_max), dtype=torch.int32)     # steps per row, filled on the host: ONE copy to the device
    for k, r0, rows, s_loc in steps_of:
        srow_host[k, r0:r0 + rows] = s_loc
    packed[:, 2 * plane:] = srow_host.to(dev)
    dist.all_gather_into_tensor(gathered.view(world * K, 2 * plane + n_max), packed, group=group)
    # read order: [K, world * n_max, ...], rank-major inside a slab
    srow = gathered[:, :, 2 * plane:].permute(1, 0, 2).reshape(K, world * n_max)
    smax = srow.amax(dim=1)
    ragged = ((srow != smax[:, None]) & (srow > 0)).any()      # some row stopped before its slab's longest one
    info = torch.cat([smax, ragged[None].to(torch.int32)]).tolist()          # the call's ONE host synchronisation after the gather
    S, ragged = info[:K], bool(info[K])
    tok.view(K, world, n_max, steps).copy_(gathered[:, :, :plane].view(world, K, n_max, steps).permute(1, 0, 2, 3))
    sc.view(K, world, n_max, steps).copy_(gathered[:, :, plane:2 * plane].view(world, K, n_max, steps).permute(1, 0, 2, 3))
    if steps and ragged:
        col = torch.arange(steps, device=dev, dtype=torch.int32)
        past = col[None, None, :] >= srow[:, :, None]          # columns the row's own loop never wrote
        last = torch.gather(sc, 2, (srow.long() - 1).clamp(min=0)[:, :, None])
        tok.masked_fill_(past, end_token)
        sc.copy_(torch.where(past, last.expand_as(sc), sc))
    out = []
    for k, n in enumerate(ns):
        if n == world * n_max:
            tk, ss = tok[k], sc[k]
        else:
            idx = _read_order_index(n, world, n_max, dev)
            tk, ss = tok[k][idx], sc[k][idx]
        out.append((tk[:, :S[k]], ss[:, :S[k]].view(torch.float32) if S[k] else ss[:, :0].float()))
    return out


def pack_call_arrays(bases, probs, lens, n_total: int, max_steps: int, world: int):
    """This rank's rows of the read-level wire format, one int32 row per chunk, padded to the common shard size
    n_max = ceil(n_total / world): [prob bits (L-1) | base bytes, 4 per word | length]."""
    import numpy as np
    n_max = max(-(-n_total // world) if n_total else 0, 1)
    bw = -(-max_steps // 4)
    cols = max_steps + bw + 1
    row = np.zeros((n_max, cols), np.int32)
    n_loc = int(len(lens))
    if n_loc:
        row[:n_loc, :max_steps] = np.ascontiguousarray(probs, np.float32).view(np.int32)
        bb = np.zeros((n_loc, 4 * bw), np.uint8)
        bb[:, :max_steps] = bases
        row[:n_loc, max_steps:max_steps + bw] = bb.view(np.int32)
        row[:n_loc, cols - 1] = lens
    return row


def unpack_call_arrays(gathered, n_total: int, max_steps: int, world: int):
    """[world * n_max, cols] gathered rows (rank-major, as all_gather_into_tensor lays them out) -> read order:
    (bases u8 [n_total, L-1], probs f32 [n_total, L-1], lens i32 [n_total])."""
    import numpy as np
    n_max = max(-(-n_total // world) if n_total else 0, 1)
    bw = -(-max_steps // 4)
    cols = max_steps + bw + 1
    idx = _read_order_index(n_total, world, n_max, "cpu").numpy()
    g = np.ascontiguousarray(np.asarray(gathered).reshape(world * n_max, cols)[idx])
    out_probs = np.ascontiguousarray(g[:, :max_steps]).view(np.float32)
    out_bases = np.ascontiguousarray(g[:, max_steps:max_steps + bw]).view(np.uint8)[:, :max_steps]
    return np.ascontiguousarray(out_bases), out_probs, np.ascontiguousarray(g[:, cols - 1])


def gather_call_arrays(bases, probs, lens, n_total: int, max_steps: int, group=None, device=None):
    """Read-level form (BASELINE configs 4/5): every rank's per-chunk base letters, per-base probabilities and string
    lengths (`Basecaller.beam_search_call_arrays`) gathered into read order with ONE collective; returns numpy arrays
    as `merger.Merger.merge_arrays` takes them."""
    world = dist.get_world_size(group)
    ref = torch.zeros(0, device=device) if device is not None else torch.zeros(0)
    dev = _collective_device(ref, group, device)
    packed = torch.from_numpy(pack_call_arrays(bases, probs, lens, n_total, max_steps, world)).to(dev)
    gathered = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    return unpack_call_arrays(gathered.cpu().numpy(), n_total, max_steps, world)
