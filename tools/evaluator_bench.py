"""End-to-end `total_processing` of the evaluator call sequence (t_predicting + t_postprocessing + t_merge,
/root/reference/ravvent_performance_evaluator.py:86) on synthetic slabs: host form, fused post-processing, and
fused + pipelined merge.  usage: python tools/evaluator_bench.py [n_chunks] [chunk_size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ravvent_basecaller_amd as rv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T_r, T_e, L = 200, 30, 32                     # the reference evaluator's native slab (data_loader.py:12-17)
bc = rv.Basecaller(128, 128, 128, rv.data_loader.nuc_tk, "joint", 0.0, max_batch=cs, max_raw_len=T_r, max_event_len=T_e,
                   max_output_len=L)
flat = rv.weights.init_weights(bc.cfg, seed=22)
flat["b_fc"][3:7] += 1.5; flat["b_fc"][bc.cfg.end_token] -= 3.0     # every chunk yields ~L-1 = 31 bases: full-length decode, and the
                                                                       # merger finds an alignment for every pair instead of stopping early
bc.set_weights_flat(flat)
raw, ev, nuc = rv.synthetic.make_slab(n, T_r, T_e, seed=0, L=L)
for name, kw in (("host post-processing + merge", {}), ("fused post-processing + merge", {"fused_postprocessing": True}),
                 ("fused + pipelined merge", {"pipelined_merge": True})):
    e = rv.evaluator.PerformanceEvaluator(bc, **kw)
    e.run_slabs(raw[:cs], ev[:cs], nuc[:cs], chunk_size=cs)          # warm-up
    t = time.perf_counter()
    r = e.run_slabs(raw, ev, nuc, chunk_size=cs)
    wall = time.perf_counter() - t
    print(f"{name:32s} chunks {n} slab {cs}: t_predicting {r['t_predicting']*1e3:8.2f} ms  t_postprocessing "
          f"{r['t_postprocessing']*1e3:7.2f} ms  t_merge {r['t_merge']*1e3:7.2f} ms  total_processing "
          f"{r['total_processing']*1e3:8.2f} ms = {n / r['total_processing']:9.0f} chunks/s  merged {len(r['merged_seq'])} bases"
          f"  (wall {wall*1e3:.1f} ms)")
