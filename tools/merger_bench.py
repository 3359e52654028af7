"""t_merge of one synthetic read: C++ merger (rv_merge_calls) vs the pure-Python restatement (oracle/merger_oracle.py,
the stand-in for the reference's Biopython path).  Host only.  usage: python tools/merger_bench.py [n_chunks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ravvent_basecaller_amd as rv
import ravvent_basecaller_amd.merger as merger
from oracle import merger_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(0)
read = "".join(rng.choice(list("ACGT"), 6 * n + 40))
stride = 47
bases = np.zeros((n, stride), np.uint8); probs = np.zeros((n, stride), np.float32); lengths = np.zeros(n, np.int32)
snips = []
for i in range(n):                                   # stride-6 chunks of ~31 bases like the evaluator's slabs
    s = read[max(0, 6 * i - 25):6 * i + 6]
    s = "".join(c if rng.random() > 0.05 else rng.choice(list("ACGT")) for c in s)
    lengths[i] = len(s); bases[i, :len(s)] = np.frombuffer(s.encode(), np.uint8)
    probs[i, :len(s)] = rng.random(len(s)).astype(np.float32)
    snips.append((s, list(probs[i, :len(s)])))
m = merger.Merger()
m.merge_arrays(bases, probs, lengths)
t = time.perf_counter(); reps = 20
for _ in range(reps):
    seq, lg = m.merge_arrays(bases, probs, lengths)
t_cpp = (time.perf_counter() - t) / reps
t = time.perf_counter()
ref_seq, ref_lg = merger_oracle.merge(snips)
t_py = time.perf_counter() - t
assert seq == ref_seq
print(f"chunks {n}  merged {len(seq)} bases  C++ {t_cpp*1e3:.3f} ms ({t_cpp/n*1e6:.2f} us/chunk)  "
      f"python restatement {t_py*1e3:.1f} ms  ratio {t_py/t_cpp:.0f}x")
