"""`Basecaller`: the reference's class API (/root/reference/basecaller.py:156-416) in front of
the HIP path.  Same constructor signature, same method names, argument meaning and return
shapes as the two evaluators rely on (ravvent_performance_evaluator.py:91-107,51-67;
SURVEY.md 8b); every number is produced by libravvent_hip.so on the GPU.

Returned tensors are ``torch.Tensor`` (they support ``.numpy()`` like the TF tensors the
callers expect).  Inputs may be numpy arrays or torch tensors, host or device; host inputs go
through the host-buffer C entry points (PCIe copies included), CUDA/HIP tensors through the
``*_dev`` entry points with no host round trip.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import _capi, weights as _weights
from .config import RvConfig, RAW_FEATURES, EVENT_FEATURES
from .data_loader import tokens_to_strings


def _as_int(x) -> int:
    """max_output_len arrives as tf.shape(target)[1] in the reference (a 0-d tensor)."""
    if isinstance(x, torch.Tensor):
        return int(x.item())
    return int(np.asarray(x).reshape(()))


class Basecaller:
    def __init__(self, enc_units: int, dec_units: int, batch_sz: int, tokenizer, input_data_type: str,
                 input_padding_value, encoder_depth: int = 2, decoder_depth: int = 1,
                 rnn_type: str = "bilstm", teacher_forcing=True, attention_type: str = "luong",
                 beam_width: int = 5, *, device: int | None = None, max_batch: int = 1024,
                 max_raw_len: int = 300, max_event_len: int = 45, max_output_len: int = 64,
                 honor_attention_type: bool = False):
        """Positional / keyword arguments are those of the reference constructor
        (basecaller.py:158).  Keyword-only extras size device memory.

        The reference hard-codes Luong attention when it builds its Decoder
        (basecaller.py:194) and only stores ``attention_type`` (:201); that behaviour is kept
        unless ``honor_attention_type=True`` selects the Bahdanau mechanism that
        ``Decoder(attention_type='bahdanau')`` builds (basecaller.py:131-132)."""
        if rnn_type != "bilstm":
            raise NotImplementedError(
                f"rnn_type={rnn_type!r}: only the BiLSTM encoder / LSTM decoder of the north-star path is built")
        if input_data_type not in ("raw", "event", "joint"):
            raise ValueError(f"input_data_type {input_data_type!r}")
        self.batch_sz = batch_sz
        self.rnn_type = rnn_type
        self.tokenizer = tokenizer
        self.teacher_forcing = teacher_forcing
        self.input_data_type = input_data_type
        self.input_padding_value = input_padding_value
        self.attention_type = attention_type
        self.beam_width = beam_width
        self.max_input_len = {"raw": 200, "event": 30, "joint": 230}[input_data_type]   # basecaller.py:180-185
        self.output_start_token = np.int32(tokenizer.word_index["$"])
        self.output_end_token = np.int32(tokenizer.word_index["^"])
        self.output_padding_token = np.int32(tokenizer.word_index[""])

        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.cfg = RvConfig(
            enc_units=enc_units, dec_units=dec_units, enc_depth=encoder_depth, dec_depth=decoder_depth,
            mode=input_data_type,
            attention=attention_type if honor_attention_type else "luong",
            vocab=len(tokenizer.word_index),
            start_token=int(self.output_start_token), end_token=int(self.output_end_token),
            pad_token=int(self.output_padding_token), padding_value=float(input_padding_value),
            max_batch=max_batch, max_raw_len=max_raw_len, max_event_len=max_event_len,
            max_output_len=max_output_len, max_beam=8, device=device)
        self.device = torch.device("cuda", device)
        self._lib = _capi.load_library()
        self._h = ctypes.c_void_p()
        ccfg = self.cfg.to_c()
        rc = self._lib.rv_create(ctypes.byref(ccfg), ctypes.byref(self._h))
        _capi.check(self._lib, None, rc, "rv_create")
        self.optimizer = None

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rv_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def compile(self, optimizer=None, **kwargs):
        """Keras ``compile`` (ravvent_performance_evaluator.py:104): inference needs no
        optimizer; accepted and ignored."""
        self.optimizer = optimizer

    def _check(self, rc, what):
        _capi.check(self._lib, self._h, rc, what)

    # ------------------------------------------------------------------ weights
    def set_weights_flat(self, flat: dict):
        blob = _weights.pack(self.cfg, flat)
        self._check(self._lib.rv_load_weights(self._h, blob.ctypes.data_as(ctypes.c_void_p), blob.size),
                    "rv_load_weights")
        self._blob = blob

    def clone(self):
        """A second handle on the same device with the same configuration and weights (own stream, own buffers): two handles
        driven from two host threads keep the GPU fed across slab boundaries (evaluator.PerformanceEvaluator(concurrent_slabs=2))."""
        other = object.__new__(Basecaller)
        for k, v in self.__dict__.items():
            if k not in ("_h", "_outs"):
                setattr(other, k, v)
        other._h = ctypes.c_void_p()
        ccfg = self.cfg.to_c()
        rc = self._lib.rv_create(ctypes.byref(ccfg), ctypes.byref(other._h))
        _capi.check(self._lib, None, rc, "rv_create")
        if getattr(self, "_blob", None) is not None:
            other._check(self._lib.rv_load_weights(other._h, self._blob.ctypes.data_as(ctypes.c_void_p), self._blob.size),
                         "rv_load_weights")
        return other

    def load_weights(self, path):
        """ravvent_performance_evaluator.py:107.  `path` is either a TF-format checkpoint prefix as Keras writes it
        (`prefix.index` + `prefix.data-*`: read by checkpoint.py without TensorFlow, variables matched by their
        attribute paths, weights_manifest.json) or this build's own ``.npz`` weight file (weights.py)."""
        from . import checkpoint
        if checkpoint.is_tf_checkpoint(path):
            self.set_weights_flat(checkpoint.flat_from_checkpoint(str(path), self.cfg))
        else:
            self.set_weights_flat(_weights.load(str(path), self.cfg))
        return self

    def init_random_weights(self, seed: int = 22, scheme: str = "keras", gain: float = 1.0):
        flat = _weights.init_weights(self.cfg, seed=seed, scheme=scheme, gain=gain)
        self.set_weights_flat(flat)
        return flat

    def set_option(self, key: str, value: int):
        rc = self._lib.rv_set_option(self._h, key.encode(), int(value))
        if rc > 0:      # a warning code (include/ravvent_hip.h): the option took effect, the message says what to look at
            import warnings
            msg = self._lib.rv_last_error(self._h)
            warnings.warn(f"rv_set_option({key}): {msg.decode() if msg else rc}", RuntimeWarning, stacklevel=2)
            return
        self._check(rc, f"rv_set_option({key})")

    # ------------------------------------------------------------------ input plumbing
    def _split_inputs(self, input_data):
        if self.input_data_type == "joint":
            raw, ev = input_data
        elif self.input_data_type == "raw":
            raw, ev = input_data, None
        else:
            raw, ev = None, input_data
        return raw, ev

    def _prep(self, x, feat):
        """-> (keepalive, pointer, B, T, on_device)"""
        if x is None:
            return None, None, None, 0, None
        if isinstance(x, torch.Tensor):
            if x.dim() != 3 or x.shape[-1] != feat:
                raise ValueError(f"expected [B,T,{feat}], got {tuple(x.shape)}")
            if x.is_cuda:
                if x.device != self.device:
                    raise ValueError(f"input on {x.device}, basecaller on {self.device}")
                # (the usual case costs nothing: a submit's host time is what the GPU waits for when the queue runs low)
                t = x if (x.dtype == torch.float32 and x.is_contiguous() and not x.requires_grad) else x.detach().to(torch.float32).contiguous()
                return t, ctypes.c_void_p(t.data_ptr()), t.shape[0], t.shape[1], True
            x = x.detach().cpu().numpy()
        a = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
        if a.ndim != 3 or a.shape[-1] != feat:
            raise ValueError(f"expected [B,T,{feat}], got {a.shape}")
        return a, a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1], False

    def _gather_inputs(self, input_data):
        raw, ev = self._split_inputs(input_data)
        kr, pr, Br, Tr, dr = self._prep(raw, RAW_FEATURES)
        ke, pe, Be, Te, de = self._prep(ev, EVENT_FEATURES)
        if kr is not None and ke is not None:
            if Br != Be:
                raise ValueError(f"raw batch {Br} != event batch {Be}")
            if dr != de:   # mixed residency: bring both to the host path
                if dr:
                    return self._gather_inputs((kr.cpu(), ke))
                return self._gather_inputs((kr, ke.cpu()))
        B = Br if kr is not None else Be
        on_dev = dr if kr is not None else de
        return (kr, ke), pr, pe, B, Tr, Te, on_dev

    # ------------------------------------------------------------------ the hot path
    def beam_search_prediction(self, input_data, beam_width, max_output_len):
        """basecaller.py:296-315 -> (predicted_ids[:,:,0] [B,S] int32, scores[:,:,0] [B,S] f32)."""
        keep, pr, pe, B, Tr, Te, on_dev = self._gather_inputs(input_data)
        L, W = _as_int(max_output_len), int(beam_width)
        steps = max(L - 1, 0)
        S = ctypes.c_int32(0)
        if on_dev:
            torch.cuda.current_stream(self.device).synchronize()   # inputs ready before the library's stream reads them
            tokens, scores = self._out_buffers("beam", B, steps, 1)
            rc = self._lib.rv_beam_search_dev(self._h, pr, pe, B, Tr, Te, W, L,
                                              ctypes.c_void_p(tokens.data_ptr()), ctypes.c_void_p(scores.data_ptr()),
                                              ctypes.byref(S))
        else:
            tk = np.empty((B, steps), np.int32)
            sc = np.empty((B, steps), np.float32)
            rc = self._lib.rv_beam_search(self._h, pr, pe, B, Tr, Te, W, L,
                                          tk.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.byref(S))
            tokens, scores = torch.from_numpy(tk), torch.from_numpy(sc)
        self._check(rc, "rv_beam_search")
        self.last_steps = S.value
        return tokens[:, :S.value], scores[:, :S.value]

    # ------------------------------------------------------------------ the hot path, several slabs in flight
    def set_async_depth(self, depth: int):
        """Slab contexts the submit_* calls rotate through (1..16; default 2).  With several slabs in flight the GPU never idles
        at a slab boundary, and the encoder recurrences run 16 chunks per workgroup on the matrix pipe (option wide_recurrence)."""
        self.set_option("async_depth", int(depth))
        self.async_depth = int(depth)

    supports_out = True       # submit_beam_search / beam_search_stream take caller-provided device outputs

    def submit_beam_search(self, input_data, beam_width, max_output_len, out=None, out_ptrs=None):
        """Queue `beam_search_prediction(input_data, ...)` without waiting for the GPU (rv_beam_search_submit / _submit_dev);
        returns a ticket for `collect`.  Results are byte-identical to the synchronous call.  Device inputs must stay untouched
        until the ticket is collected (the ticket keeps them alive).  `out` (device inputs only): a pair of contiguous device
        tensors (int32 [B, L-1], float32 [B, L-1]) the library writes into instead of fresh ones -- e.g. views into a gather buffer.
        `out_ptrs` (device inputs only): the same as two raw device addresses (the caller vouches for room, type and lifetime; `collect`
        then returns only the step count) -- a loop that submits thousands of slabs into one buffer saves the per-slab tensor views."""
        keep, pr, pe, B, Tr, Te, on_dev = self._gather_inputs(input_data)
        L, W = _as_int(max_output_len), int(beam_width)
        steps = max(L - 1, 0)
        t = ctypes.c_int32(-1)
        call = {"kind": "dev" if on_dev else "host", "keep": keep, "B": B, "steps": steps}
        if on_dev:
            torch.cuda.current_stream(self.device).synchronize()
            if out_ptrs is not None:
                rc = self._lib.rv_beam_search_submit_dev(self._h, pr, pe, B, Tr, Te, W, L, ctypes.c_void_p(int(out_ptrs[0])),
                                                         ctypes.c_void_p(int(out_ptrs[1])), ctypes.byref(t))
                self._check(rc, "rv_beam_search_submit")
                call["kind"] = "dev_ptrs"; call["ticket"] = t.value
                return call
            if out is not None:
                tokens, scores = out
                if (tuple(tokens.shape) != (B, steps) or tuple(scores.shape) != (B, steps) or tokens.dtype != torch.int32 or
                        scores.dtype != torch.float32 or not tokens.is_contiguous() or not scores.is_contiguous() or
                        tokens.device != self.device or scores.device != self.device):
                    raise ValueError(f"out: contiguous int32 / float32 tensors of shape {(B, steps)} on {self.device}")
            else:
                both = torch.empty((2, B, steps), dtype=torch.int32, device=self.device)      # one allocation: tokens | score bits
                tokens, scores = both[0], both[1].view(torch.float32)
            call["out"] = (tokens, scores)
            rc = self._lib.rv_beam_search_submit_dev(self._h, pr, pe, B, Tr, Te, W, L, ctypes.c_void_p(tokens.data_ptr()),
                                                     ctypes.c_void_p(scores.data_ptr()), ctypes.byref(t))
        else:
            rc = self._lib.rv_beam_search_submit(self._h, pr, pe, B, Tr, Te, W, L, ctypes.byref(t))
        self._check(rc, "rv_beam_search_submit")
        call["ticket"] = t.value
        return call

    def collect(self, call):
        """Wait for a `submit_beam_search` ticket -> (predicted_ids[:,:,0] [B,S] int32, scores[:,:,0] [B,S] f32)."""
        S = ctypes.c_int32(0)
        if call["kind"] == "dev_ptrs":     # results went to the caller's addresses: only the step count comes back
            self._check(self._lib.rv_beam_search_collect_dev(self._h, call["ticket"], ctypes.byref(S)), "rv_beam_search_collect_dev")
            call["keep"] = None
            self.last_steps = S.value
            return S.value
        if call["kind"] == "dev":
            self._check(self._lib.rv_beam_search_collect_dev(self._h, call["ticket"], ctypes.byref(S)), "rv_beam_search_collect_dev")
            tokens, scores = call["out"]
        elif call["kind"] == "host":
            tk = np.empty((call["B"], call["steps"]), np.int32)
            sc = np.empty((call["B"], call["steps"]), np.float32)
            self._check(self._lib.rv_beam_search_collect(self._h, call["ticket"], tk.ctypes.data_as(ctypes.c_void_p),
                                                         sc.ctypes.data_as(ctypes.c_void_p), ctypes.byref(S)), "rv_beam_search_collect")
            tokens, scores = torch.from_numpy(tk), torch.from_numpy(sc)
        else:
            raise ValueError("collect_calls() takes the tickets of submit_calls()")
        call["keep"] = None
        self.last_steps = S.value
        return tokens[:, :S.value], scores[:, :S.value]

    def _call_lut(self):
        lut = np.zeros(8, np.uint8)
        for idx, word in self.tokenizer.index_word.items():
            if 0 <= idx < 8 and word not in ("", "^", "$", " "):
                lut[idx] = ord(word.upper())
        return lut

    def submit_calls(self, input_data, beam_width, max_output_len):
        """Asynchronous `beam_search_call_arrays` (rv_beam_search_submit_calls): host inputs, fused on-device post-processing."""
        keep, pr, pe, B, Tr, Te, on_dev = self._gather_inputs(input_data)
        if on_dev:
            host = tuple(None if k is None else k.cpu().numpy() for k in keep)
            return self.submit_calls(host if self.input_data_type == "joint" else (host[0] if host[0] is not None else host[1]),
                                     beam_width, max_output_len)
        L = _as_int(max_output_len)
        t = ctypes.c_int32(-1)
        lut = self._call_lut()
        self._check(self._lib.rv_beam_search_submit_calls(self._h, pr, pe, B, Tr, Te, int(beam_width), L,
                                                          lut.ctypes.data_as(ctypes.c_void_p), ctypes.byref(t)), "rv_beam_search_submit_calls")
        return {"kind": "calls", "ticket": t.value, "B": B, "steps": max(L - 1, 0)}

    def collect_calls(self, call):
        """-> (bases u8 [B, L-1], probs f32 [B, L-1], lengths i32 [B]) of a `submit_calls` ticket."""
        B, steps = call["B"], call["steps"]
        bases = np.zeros((B, steps), np.uint8); lens = np.zeros(B, np.int32); probs = np.zeros((B, steps), np.float32)
        S = ctypes.c_int32(0)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self._lib.rv_beam_search_collect_calls(self._h, call["ticket"], p(bases), p(lens), p(probs), ctypes.byref(S)),
                    "rv_beam_search_collect_calls")
        self.last_steps = S.value
        return bases, probs, lens

    def beam_search_stream(self, slabs, beam_width, max_output_len, calls: bool = False, outs=None):
        """Generator: decode an iterable of slabs with `async_depth` of them in flight, yielding each slab's result in order
        ((tokens, scores), or (bases, probs, lengths) with calls=True) -- the overlapped form of a loop over
        `beam_search_prediction` / `beam_search_call_arrays`, with identical results."""
        depth = getattr(self, "async_depth", 2)
        outs = iter(outs) if outs is not None else None
        sub = self.submit_calls if calls else self.submit_beam_search
        col = self.collect_calls if calls else self.collect
        queue = []
        try:
            for x in slabs:
                if len(queue) >= depth:
                    yield col(queue.pop(0))
                if outs is not None:       # device inputs: the caller's output tensors, one pair per slab (submit_beam_search's `out`)
                    queue.append(sub(x, beam_width, max_output_len, out=next(outs)))
                else:
                    queue.append(sub(x, beam_width, max_output_len))
            while queue:
                yield col(queue.pop(0))
        finally:                       # a consumer that stops early (or an error): no ticket may stay uncollected on the handle
            while queue:
                try:
                    col(queue.pop(0))
                except Exception:
                    pass

    def greedy_search_prediction(self, input_data, max_output_len):
        """basecaller.py:317-330 -> (sample_id [B,S] int32, rnn_output logits [B,S,V] f32)."""
        keep, pr, pe, B, Tr, Te, on_dev = self._gather_inputs(input_data)
        L, V = _as_int(max_output_len), self.cfg.vocab
        steps = max(L - 1, 0)
        S = ctypes.c_int32(0)
        if on_dev:
            torch.cuda.current_stream(self.device).synchronize()
            tokens, logits = self._out_buffers("greedy", B, steps, V)
            rc = self._lib.rv_greedy_search_dev(self._h, pr, pe, B, Tr, Te, L,
                                                ctypes.c_void_p(tokens.data_ptr()), ctypes.c_void_p(logits.data_ptr()),
                                                ctypes.byref(S))
        else:
            tk = np.empty((B, steps), np.int32)
            lg = np.empty((B, steps, V), np.float32)
            rc = self._lib.rv_greedy_search(self._h, pr, pe, B, Tr, Te, L,
                                            tk.ctypes.data_as(ctypes.c_void_p), lg.ctypes.data_as(ctypes.c_void_p),
                                            ctypes.byref(S))
            tokens, logits = torch.from_numpy(tk), torch.from_numpy(lg)
        self._check(rc, "rv_greedy_search")
        self.last_steps = S.value
        return tokens[:, :S.value], logits[:, :S.value]

    def _out_buffers(self, kind, B, steps, V):
        """Device outputs of the `*_dev` entry points.  Default: FRESH tensors per call, like the reference's return
        values (callers may keep one result per slab in a list); torch's caching allocator makes that cheap.
        `reuse_output_buffers = True` (explicit opt-in, bench.py's timed loop) returns views into one buffer per shape,
        valid only until the next call with that shape."""
        shape2 = (B, steps, V) if kind == "greedy" else (B, steps)
        if not getattr(self, "reuse_output_buffers", False):
            return (torch.empty((B, steps), dtype=torch.int32, device=self.device),
                    torch.empty(shape2, dtype=torch.float32, device=self.device))
        key = (kind, B, steps, V)
        cache = self.__dict__.setdefault("_outs", {})
        if key not in cache:
            cache.clear()
            cache[key] = (torch.empty((B, steps), dtype=torch.int32, device=self.device),
                          torch.empty(shape2, dtype=torch.float32, device=self.device))
        return cache[key]

    def beam_search_calls(self, input_data, beam_width, max_output_len, arrays: bool = False):
        """beam_search_prediction + the evaluators' post-processing fused on the device
        (ravvent_performance_evaluator.py:55,66-70): returns (seqs: list[str], probs: list[np.ndarray])
        where probs[i] = calc_prob_logits_beam_search_scores(scores)[i][:len(seqs[i])]."""
        keep, pr, pe, B, Tr, Te, on_dev = self._gather_inputs(input_data)
        if on_dev:   # host-buffer entry point: bring device tensors down (inputs are ~2 KB per chunk)
            return self.beam_search_calls(tuple(None if k is None else k.cpu().numpy() for k in keep)
                                          if self.input_data_type == "joint" else
                                          (keep[0] if keep[0] is not None else keep[1]).cpu().numpy(),
                                          beam_width, max_output_len, arrays)
        L = _as_int(max_output_len)
        steps = max(L - 1, 0)
        lut = self._call_lut()
        bases = np.zeros((B, steps), np.uint8); lens = np.zeros(B, np.int32); probs = np.zeros((B, steps), np.float32)
        S = ctypes.c_int32(0)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self._lib.rv_beam_search_calls(self._h, pr, pe, B, Tr, Te, int(beam_width), L, p(lut), p(bases),
                                                   p(lens), p(probs), ctypes.byref(S)), "rv_beam_search_calls")
        self.last_steps = S.value
        if arrays:
            return bases, probs, lens
        flat = bases.tobytes()
        seqs = [flat[i * steps:i * steps + int(n)].decode("ascii") for i, n in enumerate(lens)]
        return seqs, [probs[i, :int(n)] for i, n in enumerate(lens)]

    def beam_search_call_arrays(self, input_data, beam_width, max_output_len):
        """beam_search_calls without per-chunk Python objects: (bases u8 [B, L-1], probs f32 [B, L-1], lengths i32 [B]),
        the layout `merger.Merger.merge_arrays` / `StreamingMerger.append` (rv_merge_calls) take."""
        return self.beam_search_calls(input_data, beam_width, max_output_len, arrays=True)

    def tokens_to_nuc_sequences(self, result_tokens):
        """basecaller.py:289-294"""
        if isinstance(result_tokens, torch.Tensor):
            result_tokens = result_tokens.detach().cpu().numpy()
        return tokens_to_strings(result_tokens)

    # ------------------------------------------------------------------ debug taps / profile
    def get_tensor(self, name: str) -> np.ndarray:
        n = ctypes.c_size_t(0)
        self._lib.rv_get_tensor(self._h, name.encode(), None, 0, ctypes.byref(n))
        out = np.empty(n.value, np.float32)
        self._check(self._lib.rv_get_tensor(self._h, name.encode(), out.ctypes.data_as(ctypes.c_void_p),
                                            out.size, ctypes.byref(n)), f"rv_get_tensor({name})")
        return out

    def profile(self) -> dict:
        buf = ctypes.create_string_buffer(4096)
        self._check(self._lib.rv_profile_names(self._h, buf, len(buf)), "rv_profile_names")
        out = {}
        for name in filter(None, buf.value.decode().split(";")):
            ms, n = ctypes.c_double(0), ctypes.c_int64(0)
            self._check(self._lib.rv_get_profile(self._h, name.encode(), ctypes.byref(ms), ctypes.byref(n)),
                        "rv_get_profile")
            out[name] = (ms.value, n.value)
        return out

    def reset_profile(self):
        self._check(self._lib.rv_reset_profile(self._h), "rv_reset_profile")
