"""Accuracy harness counterpart of `MappingEvaluator` (/root/reference/ravvent_mapping_evaluator.py:19-174):
same call chain as the speed harness (slabs -> beam search -> per-base probabilities -> merger), then
FASTA / FASTQ files -> `minimap2 -x map-ont -c` -> PAF identity.  SURVEY.md 8f next #4.

minimap2 is an external binary that is not part of this image: `_run_minimap` raises `RuntimeError` when it is
not on PATH (no substitute aligner is used); everything up to the files and the PAF parser runs without it."""
from __future__ import annotations

import json
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np

from . import data_loader as dl
from . import merger, utils

BEAM_WIDTH = 5          # ravvent_mapping_evaluator.py:15


class MappingEvaluator():
    def __init__(self, merger_scores_id=0, basecaller=None, beam_width: int = BEAM_WIDTH, workdir: str = "temp"):
        self.merger = merger.Merger(scores_id=merger_scores_id)     # :22
        self.stride = 6                                              # :23
        self.basecaller = basecaller
        self.beam_width = beam_width
        self.workdir = workdir

    @staticmethod
    def _split_into_chunks(arr, def_chunk_size):
        """ravvent_mapping_evaluator.py:26-29"""
        return np.array_split(arr, np.arange(1, arr.shape[0] // def_chunk_size + 1) * def_chunk_size)

    def basecall_read(self, raw_snippets, event_snippets, nuc_tk_snippets, chunk_size=1024) -> str:
        """ravvent_mapping_evaluator.py:36-57: slabs -> calls -> merged read."""
        nuc_preds = []
        for data in zip(self._split_into_chunks(raw_snippets, chunk_size), self._split_into_chunks(event_snippets, chunk_size),
                        self._split_into_chunks(nuc_tk_snippets, chunk_size)):
            if data[0].shape[0] == 0:
                continue
            input_data, target_data = utils.unpack_data_to_input_target(data, self.basecaller.input_data_type)
            pred_tokens, beam_scores = self.basecaller.beam_search_prediction(
                input_data, beam_width=self.beam_width, max_output_len=target_data.shape[1])
            scores = utils.calc_prob_logits_beam_search_scores(beam_scores).numpy()
            seqs = self.basecaller.tokens_to_nuc_sequences(pred_tokens)
            nuc_preds.extend(merger.SeqLogitsPair(seq, list(sc[:len(seq)])) for seq, sc in zip(seqs, scores))
        return self.merger.merge(nuc_preds).seq

    def run(self, signal_data_source, chunk_size=1024):
        """ravvent_mapping_evaluator.py:31-72"""
        label_path = Path(signal_data_source).with_suffix('.label')
        ref_seq = ''.join(list(np.loadtxt(label_path, dtype='object')[:, 2]))
        snippets = dl.load_data_from_single_signal_label(signal_data_source, label_path, self.stride)
        return self.map_read(ref_seq, self.basecall_read(*snippets, chunk_size=chunk_size))

    def map_read(self, ref_seq: str, merged_seq: str):
        os.makedirs(self.workdir, exist_ok=True)
        fasta_path = os.path.join(self.workdir, 'ref.fasta')
        fastq_path = os.path.join(self.workdir, 'pred.fastq')
        mapping_path = os.path.join(self.workdir, 'mapping.paf')
        self._create_fasta(ref_seq, fasta_path)
        self._create_fastq(merged_seq, fastq_path)
        self._run_minimap(fasta_path, fastq_path, mapping_path)
        return self._read_mapping_identity(mapping_path)

    def _create_fasta(self, seq, fname):
        """One record named by the first 10 bases, no trailing newline (ravvent_mapping_evaluator.py:74-76)."""
        Path(fname).write_text(">" + seq[:10] + "\n" + seq)

    def _create_fastq(self, seq, fname):
        """Same record name; every base gets the lowest quality '!' (ravvent_mapping_evaluator.py:78-83)."""
        Path(fname).write_text("\n".join(("@" + seq[:10], seq, "+", "!" * len(seq))))

    def _run_minimap(self, ref_path, pred_path, out_path):
        exe = shutil.which("minimap2")
        if exe is None:
            raise RuntimeError("minimap2 is not on PATH: the mapping identity cannot be computed "
                               "(ravvent_mapping_evaluator.py:85-88)")
        with open(out_path, "wt") as out:
            subprocess.run([exe, "-x", "map-ont", "-c", str(ref_path), str(pred_path)], stdout=out)

    def _read_mapping_identity(self, mapping_path):
        """PAF columns 2 (query length), 10 (residue matches) and 11 (alignment block length), the latter two summed
        over every alignment record of the read; records with fewer than 11 columns are skipped; the query length
        is the last record's (ravvent_mapping_evaluator.py:90-108)."""
        rows = [ln.strip().split("\t") for ln in Path(mapping_path).read_text().splitlines()]
        rows = [r for r in rows if len(r) >= 11]
        matches = sum(int(r[9]) for r in rows)
        blocks = sum(int(r[10]) for r in rows)
        return {"read_length": int(rows[-1][1]) if rows else 0, "matches": matches, "total_block_len": blocks,
                "identity": matches / blocks if blocks else 0.}

    def compute_total_results(self, results_path):
        """Reference-length-weighted identity over all reads / over mapped reads, and the share of unmapped reads,
        all in percent, rounded to 3 places (ravvent_mapping_evaluator.py:130-167; what follows the first `return`
        there is dead code).  An unmapped read (read_length 0) counts with identity 0 in the first figure."""
        results = json.loads(Path(results_path).read_text())
        ref_len = np.array([r["ref_length"] for r in results], np.float64)
        mapped = np.array([r["read_length"] != 0 for r in results], bool)
        ident = np.array([r["matches"] / r["total_block_len"] if r["read_length"] != 0 else 0.0 for r in results], np.float64)
        w_valid = ref_len[mapped].sum()
        total = (ident * ref_len).sum() / ref_len.sum() * 100 if w_valid > 0 else 0
        valid = (ident * ref_len)[mapped].sum() / w_valid * 100 if w_valid > 0 else 0
        return round(float(total), 3), round(float(valid), 3), round(float((~mapped).mean() * 100), 3)
