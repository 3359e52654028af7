"""Speed harness counterpart of `RavventPerformanceEvaluator.run`
(/root/reference/ravvent_performance_evaluator.py:24-87): same call sequence -- slabs of
`chunk_size` chunks -> `beam_search_prediction(beam_width=5)` -> per-base probabilities and
strings -- and the same result-dict keys and phase timers.  The reference file itself imports
TensorFlow and Biopython and cannot run here.  Read-level merging is the C++
merger behind `merger.Merger` (csrc/merger.cpp; `t_merge`, ravvent_performance_evaluator.py:73-75).
"""
from __future__ import annotations

from timeit import default_timer as timer

import numpy as np

from . import merger, utils


class PerformanceEvaluator:
    def __init__(self, basecaller, stride: int = 6, fused_postprocessing: bool = True, pipelined_merge: bool = False,
                 concurrent_slabs: int = 1):
        self.basecaller = basecaller
        self.stride = stride           # ravvent_performance_evaluator.py:16
        # True (default since round 4): strings + per-base probabilities come straight from the device (rv_beam_search_calls) as
        # arrays, and the merge takes those arrays: no per-chunk Python object is built inside the timed phases (`nuc_preds` is put
        # together afterwards, for callers that want the per-chunk view).  False: the reference's host sequence, chunk by chunk --
        # calc_prob_logits_beam_search_scores, tokens_to_nuc_sequences, a list of SeqLogitsPair into Merger.merge
        # (ravvent_performance_evaluator.py:66-75)
        self.fused_postprocessing = fused_postprocessing
        self.merger = merger.Merger()  # ravvent_performance_evaluator.py:17
        # True: slab k is stitched on a host thread (rv_merger_append) while the GPU decodes slab k+1; implies the
        # fused post-processing.  t_merge is then the part of the merge the GPU work did not hide.
        self.pipelined_merge = pipelined_merge
        # > 1: at least that many slabs in flight on the handle's contexts (Basecaller.set_async_depth; the default depth of 2
        # applies otherwise): slab k + 1's encoders fill the CUs slab k's decode leaves as its chunks finish, and no launch gap
        # or host copy leaves the GPU idle.  Results are identical (slabs are independent).
        self.concurrent_slabs = max(1, int(concurrent_slabs))
        self._clones = []

    def close(self):
        """Release the extra handles of `concurrent_slabs` (the caller's own Basecaller stays open)."""
        for bc in self._clones:
            bc.close()
        self._clones = []

    @staticmethod
    def _split_into_chunks(arr, def_chunk_size):
        """ravvent_performance_evaluator.py:19-22"""
        return np.array_split(arr, np.arange(1, arr.shape[0] // def_chunk_size + 1) * def_chunk_size)

    def run(self, signal_data_source, chunk_size: int = 1024):
        """ravvent_performance_evaluator.py:24-87 for a `.signal` / `.label` file pair."""
        from pathlib import Path
        label_path = Path(signal_data_source).with_suffix(".label")
        signal = np.loadtxt(signal_data_source, dtype=int)
        labels = np.loadtxt(label_path, dtype=object)
        return self.run_read(signal, labels, chunk_size=chunk_size)

    def run_read(self, signal, labels, chunk_size: int = 1024, beam_width: int = 5):
        """Same as run() from in-memory arrays: labels rows are (start, end, base)."""
        from . import data_loader as dl
        ranges_ids = np.asarray(labels)[:, :2].astype(int)
        ref_seq = "".join(list(np.asarray(labels)[:, 2]))
        samples_num = int(ranges_ids[-1, 1] - ranges_ids[0, 0])
        start = timer()
        slab = dl.snippets_to_slab(*dl.prepare_snippets(signal, ranges_ids, np.asarray(labels)[:, 2], self.stride))
        t_chunking = timer() - start
        res = self.run_slabs(*slab, bases_num=len(ref_seq), samples_num=samples_num, chunk_size=chunk_size,
                             beam_width=beam_width)
        res["t_data_loading"] += t_chunking
        res["total"] += t_chunking
        return res

    def run_slabs(self, raw_snippets, event_snippets, nuc_tk_snippets, bases_num=None, samples_num=None,
                  chunk_size: int = 1024, beam_width: int = 5):
        """The body of `run()` after data loading: inputs are the padded snippet arrays
        `load_data_from_single_signal_label` would return (data_loader.py:113-126)."""
        start = timer()
        data_chunks = list(zip(self._split_into_chunks(raw_snippets, chunk_size),
                               self._split_into_chunks(event_snippets, chunk_size),
                               self._split_into_chunks(nuc_tk_snippets, chunk_size)))
        data_chunks = [d for d in data_chunks if d[0].shape[0] > 0]
        t_data_loading = timer() - start
        nuc_preds = []
        t_predicting = t_postprocessing = 0.0
        if self.pipelined_merge:
            return self._run_pipelined(data_chunks, raw_snippets, event_snippets, bases_num, samples_num, beam_width,
                                       t_data_loading)
        # The reference decodes slab k, post-processes it, then decodes slab k + 1 (ravvent_performance_evaluator.py:51-70).  Here the
        # slabs go through the asynchronous calls (Basecaller.beam_search_stream: identical results): slab k + 1 is already on the
        # GPU while slab k is post-processed on the host.  t_predicting = the time this loop waits for the GPU.
        unpacked = [utils.unpack_data_to_input_target(data, self.basecaller.input_data_type) for data in data_chunks]
        L = unpacked[0][1].shape[1] if unpacked else 0
        stream_ok = hasattr(self.basecaller, "beam_search_stream") and all(t.shape[1] == L for _, t in unpacked)
        if stream_ok:
            results = self.basecaller.beam_search_stream((inp for inp, _ in unpacked), beam_width, L, calls=self.fused_postprocessing)
        elif self.fused_postprocessing:
            results = (self.basecaller.beam_search_call_arrays(inp, beam_width=beam_width, max_output_len=t.shape[1]) for inp, t in unpacked)
        else:
            results = (self.basecaller.beam_search_prediction(inp, beam_width=beam_width, max_output_len=t.shape[1]) for inp, t in unpacked)
        kept = []                      # fused form: the slabs' call arrays (bases, probs, lengths)
        while True:
            start = timer()
            res = next(results, None)
            t_predicting += timer() - start
            if res is None:
                break
            start = timer()
            if self.fused_postprocessing:
                kept.append(res)       # the post-processing happened on the device; nothing per chunk to do here
            else:
                pred_tokens, beam_scores = res
                scores = utils.calc_prob_logits_beam_search_scores(beam_scores).numpy()
                seqs = self.basecaller.tokens_to_nuc_sequences(pred_tokens)
                nuc_preds.extend((seq, sc[:len(seq)]) for seq, sc in zip(seqs, scores))
            t_postprocessing += timer() - start
        start = timer()                # ravvent_performance_evaluator.py:73-75
        if self.fused_postprocessing:
            merged_seq = ""
            if kept:
                steps = max(b.shape[1] for b, _, _ in kept)
                cat = lambda k, dt: np.concatenate([np.pad(a[k], ((0, 0), (0, steps - a[k].shape[1]))) if a[k].shape[1] < steps else a[k] for a in kept]).astype(dt, copy=False)
                merged_seq = self.merger.merge_arrays(cat(0, np.uint8), cat(1, np.float32), np.concatenate([a[2] for a in kept]))[0]
        else:
            merged_seq = self.merger.merge([merger.SeqLogitsPair(seq, lg) for seq, lg in nuc_preds]).seq if nuc_preds else ""
        t_merge = timer() - start
        for bases, probs, lens in kept:                          # per-chunk view for callers that want it (untimed)
            flat, steps = bases.tobytes(), bases.shape[1]
            nuc_preds.extend((flat[i * steps:i * steps + int(n)].decode("ascii"), probs[i, :int(n)]) for i, n in enumerate(lens))
        n_chunks = int(raw_snippets.shape[0] if raw_snippets is not None else event_snippets.shape[0])
        if bases_num is None:
            bases_num = n_chunks * self.stride
        return {
            "bases_num": int(bases_num), "samples_num": samples_num, "chunks_num": n_chunks,
            "t_data_loading": t_data_loading, "t_predicting": t_predicting,
            "t_postprocessing": t_postprocessing, "t_merge": t_merge,
            "total": t_data_loading + t_predicting + t_postprocessing + t_merge,
            "total_processing": t_predicting + t_postprocessing + t_merge,
            "nuc_preds": nuc_preds, "merged_seq": merged_seq,
        }

    # ------------------------------------------------------------------ BASELINE configs 4/5: one read over N GPUs
    def run_read_sharded(self, signal, labels, chunk_size: int = 1024, beam_width: int = 5, group=None):
        """`run_read` with the read's chunks sharded over the ranks of `group` (one process per GPU): every rank cuts the
        read into chunks (host pre-processing, outside the metric, deterministic), decodes ITS contiguous range
        (`dist.shard_range`) in slabs of `chunk_size`, ONE all-gather returns every chunk's call in read order, and rank 0
        stitches them with the C++ merger (ravvent_performance_evaluator.py:24-87 with :51-55 sharded and :73-75 on rank 0).
        `merged_seq` is None on the other ranks."""
        from . import data_loader as dl
        ranges_ids = np.asarray(labels)[:, :2].astype(int)
        ref_seq = "".join(list(np.asarray(labels)[:, 2]))
        samples_num = int(ranges_ids[-1, 1] - ranges_ids[0, 0])
        start = timer()
        slab = dl.snippets_to_slab(*dl.prepare_snippets(signal, ranges_ids, np.asarray(labels)[:, 2], self.stride))
        t_chunking = timer() - start
        res = self.run_slabs_sharded(*slab, bases_num=len(ref_seq), samples_num=samples_num, chunk_size=chunk_size,
                                     beam_width=beam_width, group=group)
        res["t_data_loading"] += t_chunking
        res["total"] += t_chunking
        return res

    def run_slabs_sharded(self, raw_snippets, event_snippets, nuc_tk_snippets, bases_num=None, samples_num=None,
                          chunk_size: int = 1024, beam_width: int = 5, group=None):
        """Body of `run_read_sharded` from the padded snippet arrays every rank holds."""
        import torch.distributed as tdist
        from . import dist as rdist
        world, rank = tdist.get_world_size(group), tdist.get_rank(group)
        n = int(raw_snippets.shape[0] if raw_snippets is not None else event_snippets.shape[0])
        L = int(nuc_tk_snippets.shape[1])
        steps = max(L - 1, 0)
        lo, hi = rdist.shard_range(n, rank, world)
        start = timer()
        bases, probs, lens = self.decode_range(raw_snippets, event_snippets, lo, hi, L, chunk_size, beam_width)
        t_predicting = timer() - start
        start = timer()
        dev = getattr(self.basecaller, "device", None) if tdist.get_backend(group) == "nccl" else None
        g_bases, g_probs, g_lens = rdist.gather_call_arrays(bases, probs, lens, n, steps, group=group, device=dev)
        t_gather = timer() - start
        start = timer()
        merged_seq = None
        if rank == 0:
            merged_seq = self.merger.merge_arrays(g_bases, g_probs, g_lens)[0] if n else ""
        t_merge = timer() - start
        if bases_num is None:
            bases_num = n * self.stride
        return {
            "bases_num": int(bases_num), "samples_num": samples_num, "chunks_num": n, "chunks_local": hi - lo,
            "t_data_loading": 0.0, "t_predicting": t_predicting, "t_postprocessing": 0.0, "t_gather": t_gather,
            "t_merge": t_merge, "total": t_predicting + t_gather + t_merge,
            "total_processing": t_predicting + t_gather + t_merge,
            "call_arrays": (g_bases, g_probs, g_lens), "merged_seq": merged_seq,
        }

    def decode_range(self, raw_snippets, event_snippets, lo: int, hi: int, max_output_len: int, chunk_size: int = 1024,
                     beam_width: int = 5):
        """Calls of chunks [lo, hi) in slabs of `chunk_size`: (bases u8 [hi-lo, L-1], probs f32, lens i32) -- one rank's
        share of `run_slabs_sharded`."""
        mode = self.basecaller.input_data_type
        steps = max(int(max_output_len) - 1, 0)
        bases = np.zeros((hi - lo, steps), np.uint8)
        probs = np.zeros((hi - lo, steps), np.float32)
        lens = np.zeros(hi - lo, np.int32)
        cuts = [(a, min(a + chunk_size, hi)) for a in range(lo, hi, chunk_size)]
        pick = lambda a, b: {"joint": lambda: (raw_snippets[a:b], event_snippets[a:b]), "raw": lambda: raw_snippets[a:b],
                             "event": lambda: event_snippets[a:b]}[mode]()
        if hasattr(self.basecaller, "beam_search_stream"):     # slabs in flight on the handle's contexts (identical results)
            results = self.basecaller.beam_search_stream((pick(a, b) for a, b in cuts), beam_width, max_output_len, calls=True)
        else:
            results = (self.basecaller.beam_search_call_arrays(pick(a, b), beam_width=beam_width, max_output_len=max_output_len) for a, b in cuts)
        for (a, b), (bs, pr, ln) in zip(cuts, results):
            bases[a - lo:b - lo] = bs; probs[a - lo:b - lo] = pr; lens[a - lo:b - lo] = ln
        return bases, probs, lens

    def _run_pipelined(self, data_chunks, raw_snippets, event_snippets, bases_num, samples_num, beam_width, t_data_loading):
        from concurrent.futures import ThreadPoolExecutor
        sm = merger.StreamingMerger(self.merger.scores_id, self.merger.overlap_seq_len)
        kept, pending = [], []
        mode = self.basecaller.input_data_type
        if self.concurrent_slabs > 1 and hasattr(self.basecaller, "set_async_depth"):
            self.basecaller.set_async_depth(max(self.concurrent_slabs, getattr(self.basecaller, "async_depth", 2)))
        unpacked = [utils.unpack_data_to_input_target(data, mode) for data in data_chunks]
        L = unpacked[0][1].shape[1] if unpacked else 0
        if hasattr(self.basecaller, "beam_search_stream") and all(t.shape[1] == L for _, t in unpacked):
            results = self.basecaller.beam_search_stream((inp for inp, _ in unpacked), beam_width, L, calls=True)
        else:
            results = (self.basecaller.beam_search_call_arrays(inp, beam_width=beam_width, max_output_len=t.shape[1]) for inp, t in unpacked)
        with ThreadPoolExecutor(max_workers=1) as pool:          # merge worker: slabs appended in read order while the GPU decodes on
            start = timer()
            for arrays in results:
                kept.append(arrays)
                pending.append(pool.submit(sm.append, *arrays))
            t_predicting = timer() - start
            start = timer()
            for f in pending:
                f.result()
            merged_seq, _ = sm.result()
            t_merge = timer() - start
        sm.close()
        nuc_preds = []                                           # per-chunk view for callers that want it (untimed)
        for bases, probs, lens in kept:
            flat, steps = bases.tobytes(), bases.shape[1]
            nuc_preds.extend((flat[i * steps:i * steps + int(n)].decode("ascii"), probs[i, :int(n)])
                             for i, n in enumerate(lens))
        n_chunks = int(raw_snippets.shape[0] if raw_snippets is not None else event_snippets.shape[0])
        if bases_num is None:
            bases_num = n_chunks * self.stride
        return {
            "bases_num": int(bases_num), "samples_num": samples_num, "chunks_num": n_chunks,
            "t_data_loading": t_data_loading, "t_predicting": t_predicting, "t_postprocessing": 0.0, "t_merge": t_merge,
            "total": t_data_loading + t_predicting + t_merge, "total_processing": t_predicting + t_merge,
            "nuc_preds": nuc_preds, "merged_seq": merged_seq,
        }

    def run_many(self, reads, chunk_size: int = 1024, beam_width: int = 5, max_output_len: int | None = None,
                 merge_threads: int = 4):
        """Many reads through shared slabs (short reads leave a GPU's slab half empty; the chunks of all
        reads are queued into full slabs, decoded, and stitched back per read).

        reads: list of (raw_snippets [n_i,T_r,1], event_snippets [n_i,T_e,5], nuc_tk_snippets [n_i,L_i]) as
        `load_data_from_single_signal_label` returns them (one tuple per read).  Returns one dict per read with
        `merged_seq`, `chunks_num`, plus the shared timers under key "timing" of the first entry.  The reference has no
        counterpart (it evaluates one read at a time); per-read results equal `run_slabs` of that read alone when
        `max_output_len` is given or all reads share one target length."""
        from concurrent.futures import ThreadPoolExecutor
        start = timer()
        n = [int(r[0].shape[0]) for r in reads]
        off = np.concatenate([[0], np.cumsum(n)])
        L = int(max_output_len) if max_output_len is not None else max(int(r[2].shape[1]) for r in reads)
        raw = np.concatenate([r[0] for r in reads], axis=0) if self.basecaller.input_data_type != "event" else None
        ev = np.concatenate([r[1] for r in reads], axis=0) if self.basecaller.input_data_type != "raw" else None
        total = int(off[-1])
        t_data_loading = timer() - start
        bases = np.zeros((total, max(L - 1, 1)), np.uint8)
        probs = np.zeros((total, max(L - 1, 1)), np.float32)
        lens = np.zeros(total, np.int32)
        t_predicting = 0.0
        cuts = [(b0, min(b0 + chunk_size, total)) for b0 in range(0, total, chunk_size)]
        pick = lambda b0, b1: (raw[b0:b1], ev[b0:b1]) if self.basecaller.input_data_type == "joint" else (raw[b0:b1] if raw is not None else ev[b0:b1])
        start = timer()
        if hasattr(self.basecaller, "beam_search_stream"):
            results = self.basecaller.beam_search_stream((pick(b0, b1) for b0, b1 in cuts), beam_width, L, calls=True)
        else:
            results = (self.basecaller.beam_search_call_arrays(pick(b0, b1), beam_width=beam_width, max_output_len=L) for b0, b1 in cuts)
        for (b0, b1), (bs, pr, ln) in zip(cuts, results):
            bases[b0:b1, :bs.shape[1]] = bs; probs[b0:b1, :pr.shape[1]] = pr; lens[b0:b1] = ln
        t_predicting += timer() - start
        start = timer()

        def merge_one(i):
            if n[i] == 0:
                return ""
            return self.merger.merge_arrays(bases[off[i]:off[i + 1]], probs[off[i]:off[i + 1]], lens[off[i]:off[i + 1]])[0]
        with ThreadPoolExecutor(max_workers=max(1, merge_threads)) as pool:      # reads are independent; ctypes drops the GIL
            merged = list(pool.map(merge_one, range(len(reads))))
        t_merge = timer() - start
        out = [{"merged_seq": m, "chunks_num": n[i]} for i, m in enumerate(merged)]
        if out:
            out[0]["timing"] = {"t_data_loading": t_data_loading, "t_predicting": t_predicting, "t_postprocessing": 0.0,
                                "t_merge": t_merge, "total_processing": t_predicting + t_merge, "chunks_num": total}
        return out
