"""Build-level guard (CPU, hipcc cross-compiles gfx950 here): the hot kernels sit at the edge of the register file -- the persistent
decode keeps a chunk's attention memory in 184 of its 256 VGPRs, the fused recurrence + projection kernel has 168 -- and a change
that tips the register allocator over does not fail any parity test: it silently spills the resident rows to scratch and the
kernel runs 1.5x slower (seen three times in round 2).  This test compiles the two kernel files for the device only and bounds the
scratch of the instantiations the benchmark workloads launch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ravvent-basecaller_amd", "csrc")

# kernel-name regex -> max bytes of scratch per lane
LIMITS = {
    r"k_dec_persistILi5ELi11ELi1ELi0E": 64,      # C3: Luong, beam 5, T_m <= 352 (36 B today)
    r"k_dec_persistILi5ELi8ELi1ELi0E": 0,        # R: T_m <= 256
    r"k_dec_persistILi5ELi11ELi1ELi1E": 64,      # C3 with Bahdanau (28 B today)
    r"k_dec_persistILi5ELi11ELi1ELi2E": 24,      # C3 with the attention on the matrix pipe (20 B today; 100 B in round 2)
    r"k_dec_persistILi5ELi8ELi1ELi2E": 0,        # R
    r"k_dec_persistILi5ELi11ELi1ELi3E": 8,       # C3, attention, cell product and output layer on the matrix pipe: the default since round 3 (0 B today)
    r"k_dec_persistILi5ELi8ELi1ELi3E": 0,        # R, the same
    r"k_dec_persistILi5ELi11ELi1ELi4E": 192,     # C3 with Bahdanau scores on the VALU and everything else on the matrix pipe, the Bahdanau default since round 4: 176 B today,
                                                 # of which the step loop touches 56 (three U' fragments reloaded in the context phase); the rest is the prologue's (keys -> 2^k')
    r"k_dec_persistILi5ELi8ELi1ELi4E": 96,       # R, the same (60-76 B: single registers of the unrolled score loop)
    r"k_dec_persistILi5ELi11ELi2ELi3E": 96,      # C3 with two decoder cells on the matrix pipe, the enc3/dec2 default since round 4 (80 B today: five U' fragments reloaded in the context phase)
    r"k_dec_persistILi5ELi8ELi2ELi3E": 0,        # R, the same
    r"k_lstm_rec_projILi2ELi[012]EE": 0,         # C3 fused recurrence + projection (f32, split-bf16 and split-f16 MFMA forms)
    r"k_gemm_mem_split3": 0,                     # attention-memory projection on split-f16 MFMAs (compute waves + loader waves)
    r"k_lstm_recILi2ELi1EE": 0, r"k_lstm_recILi2ELi5EE": 0,
    r"k_lstm_rec_twILi2ELi1EE": 0, r"k_lstm_rec_twILi2ELi5EE": 0,   # C3 layer 0 (tail-wave variant)
    r"k_lstm_rec_mxILi[015]ELi(16|8)EE": 0,      # matrix-pipe recurrence (inputs pre-projected / raw / events; 16 / 8 chunks per workgroup): U^T resident as A fragments (128 VGPRs)
}


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_hot_kernels_do_not_spill(tmp_path):
    found = {}
    for src, extra in (("decode.hip", []), ("lstm_rec.hip", ["-fno-slp-vectorize"]), ("lstm_mx.hip", []), ("gemm_f32.hip", [])):
        out = tmp_path / (src + ".s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *extra,
                        os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True)
        text = out.read_text()
        for m in re.finditer(r"^(_Z\w+):", text, re.M):
            tail = text[text.index(".Lfunc_end", m.start()):][:4000]
            sc = re.search(r"; ScratchSize: (\d+)", tail)
            vg = re.search(r"; NumVgprs: (\d+)", tail)
            if sc:
                found[m.group(1)] = (int(sc.group(1)), int(vg.group(1)))
    for pat, limit in LIMITS.items():
        hits = {k: v for k, v in found.items() if re.search(pat, k)}
        assert hits, f"no kernel matches {pat}"
        for name, (scratch, vgpr) in hits.items():
            assert scratch <= limit, f"{name}: {scratch} B of scratch per lane ({vgpr} VGPRs) > {limit}: the resident rows are being spilled"
