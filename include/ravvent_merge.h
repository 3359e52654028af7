/* libravvent_hip.so -- host entry points for the read-level merger.
 *
 * Replaces, for a reference maintainer, the body of Merger.merge (/root/reference/merger.py:155-248),
 * which the evaluator times as t_merge (/root/reference/ravvent_performance_evaluator.py:73-75), and
 * the Bio.pairwise2 call inside it (merger.py:168-180).  Plain host code (no GPU involved): the chain
 * of 25x25 alignments is sequential in the read.  Inputs are laid out exactly like the outputs of
 * rv_beam_search_calls (include/ravvent_hip.h): one row of `stride` bytes / floats per chunk. */
#ifndef RAVVENT_MERGE_H
#define RAVVENT_MERGE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { RV_MERGE_EINVAL = -1, RV_MERGE_ESPACE = -2, RV_MERGE_EALPHABET = -3 };

/* Merger(scores_id).merge(nuc_pred_snippets) (merger.py:155-248).
 *   bases   [n_chunks, stride]  upper-case letters of chunk i's call in its first lengths[i] bytes
 *   probs   [n_chunks, stride]  per-base probability ("logits" of SeqLogitsPair), same layout
 *   scores_id 0|1: localms(match, mismatch, open, extend) tables of merger.py:125-136; 2: localds matrix :137-146
 *   overlap  = Merger.overlap_seq_len (25, merger.py:150)
 * Writes the merged read and its per-base values; *out_len is always set to the merged length.
 * Returns 0, RV_MERGE_ESPACE if out_cap is too small (nothing written), RV_MERGE_EALPHABET for a letter outside
 * ACGT under scores_id 2 (the reference raises KeyError there), RV_MERGE_EINVAL for bad arguments.
 * Like the reference: a pair without any alignment replaces the read so far while nothing has been merged yet,
 * and ends the merge (returning what has been merged) afterwards (merger.py:181-200). */
int rv_merge_calls(const uint8_t* bases, const float* probs, const int32_t* lengths, int64_t stride, int32_t n_chunks,
                   int32_t scores_id, int32_t overlap, uint8_t* out_seq, float* out_probs, int64_t out_cap, int64_t* out_len);

/* The same merge as a resumable loop, so that the host can stitch slab k while the GPU decodes slab k+1: create,
 * append the chunks of each slab in read order, read the result.  State = merged read so far + merge_flag + whether
 * the early return of merger.py:195-200 was taken (later appends are then ignored, like the reference's return).
 * A handle is not thread-safe; different handles are independent. */
typedef struct RvMerger* rv_merger;
int rv_merger_create(int32_t scores_id, int32_t overlap, rv_merger* out);
int rv_merger_append(rv_merger m, const uint8_t* bases, const float* probs, const int32_t* lengths, int64_t stride,
                     int32_t n_chunks);
int rv_merger_result(rv_merger m, uint8_t* out_seq, float* out_probs, int64_t out_cap, int64_t* out_len);
void rv_merger_destroy(rv_merger m);

/* pairwise2.align.localms / localds (a, b, <scores_id tables>)[0]: the two gapped, equal-length strings (both
 * sequences in full, '-' for gaps), score, begin, end.  Returns 1 with *out_len columns written, 0 when the
 * reference's list would be empty (*out_len = 0), negative on error (cap too small: RV_MERGE_ESPACE). */
int rv_local_align(const char* a, int32_t len_a, const char* b, int32_t len_b, int32_t scores_id, char* out_a, char* out_b,
                   int32_t cap, int32_t* out_len, double* score, int32_t* begin, int32_t* end);

#ifdef __cplusplus
}
#endif
#endif
