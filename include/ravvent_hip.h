/* ravvent_hip.h -- C-ABI of the MI355X-native Ravvent inference hot path (libravvent_hip.so).
 *
 * The reference has no FFI / plugin interface for this path: callers use the Python methods
 * of `Basecaller` (a Keras Model subclass).  This header is therefore the boundary a ctypes
 * binding of that class needs; each entry point names the reference method it stands behind
 * (file:line into /root/reference).  INTEGRATION.md shows the reference-side ctypes stub.
 *
 * Conventions: plain pointers + sizes, no C++/torch types.  Every function returns 0 on
 * success and a negative RV_E* code on failure; the message is available through
 * rv_last_error().  Nothing throws across the boundary.  A handle is bound to ONE device and
 * is NOT thread-safe (the reference is single-threaded and synchronous:
 * ravvent_performance_evaluator.py:51-55); use one handle per GPU / per process.
 *
 * Synchronous calls -- rv_beam_search, rv_beam_search_dev, rv_beam_search_calls, rv_greedy_search,
 * rv_greedy_search_dev, rv_load_weights, rv_get_tensor: results are complete when the call
 * returns, so a caller's wall-clock timers mean what they mean around the reference's methods.
 * Asynchronous calls -- rv_beam_search_submit, _submit_dev, _submit_calls: they return once the
 * slab is queued; the matching rv_beam_search_collect* waits for it.  A synchronous call may be
 * made while tickets are in flight: it runs on an idle slab context of the handle (its own when
 * that one is idle) and fails with RV_ESTATE, touching nothing, when every context holds an
 * uncollected ticket.  rv_get_tensor and the debug-tap options refer to the handle's OWN context:
 * with taps on, a synchronous call needs that context idle.
 *
 * Hardware queues: every slab context is one HIP stream, and the HIP runtime multiplexes a
 * process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says
 * otherwise when the runtime starts, i.e. at the process's first HIP call).  Loading this
 * library sets GPU_MAX_HW_QUEUES=16 if the variable is unset (RAVVENT_KEEP_ENV=1 opts out): load
 * it before the first HIP call, or export the variable yourself.  rv_set_option("async_depth", n)
 * returns the warning RV_WQUEUES when n exceeds the setting it finds.
 *
 * Host-buffer entry points copy inputs H2D / outputs D2H themselves.  The *_dev variants
 * take device addresses valid on the handle's device (e.g. torch tensor data_ptr()) and move
 * no data across PCIe.
 */
#ifndef RAVVENT_HIP_H
#define RAVVENT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RV_ABI_VERSION 1

enum { RV_OK = 0, RV_EINVAL = -1, RV_ENOMEM = -2, RV_EHIP = -3, RV_ESTATE = -4, RV_EUNSUPPORTED = -5 };
/* Positive codes are warnings: the call took effect, rv_last_error() says what to look at. */
enum { RV_WQUEUES = 1 };   /* rv_set_option("async_depth", n): n exceeds the HIP runtime's hardware queues (GPU_MAX_HW_QUEUES) */
enum { RV_MODE_RAW = 0, RV_MODE_EVENT = 1, RV_MODE_JOINT = 2 };      /* input_data_type, basecaller.py:180-185 */
enum { RV_ATT_LUONG = 0, RV_ATT_BAHDANAU = 1 };                       /* basecaller.py:131-134 */

/* Mirrors the arguments of Basecaller.__init__ (basecaller.py:158-206) that shape the
 * inference path, plus the maximum shapes device memory is sized for. */
typedef struct RvConfig {
  int32_t enc_units;      /* LSTMCell units per direction (128)          basecaller.py:22 */
  int32_t dec_units;      /* decoder LSTMCell units = attention depth    basecaller.py:86 */
  int32_t enc_depth;      /* stacked BiLSTM layers per encoder           basecaller.py:31 */
  int32_t dec_depth;      /* StackedRNNCells depth                       basecaller.py:85-91 */
  int32_t mode;           /* RV_MODE_*                                                    */
  int32_t attention;      /* RV_ATT_*                                                     */
  int32_t vocab;          /* len(tokenizer.word_index) = 7               basecaller.py:189 */
  int32_t start_token;    /* '$' = 2   basecaller.py:204 */
  int32_t end_token;      /* '^' = 1   basecaller.py:205 */
  int32_t pad_token;      /* ''  = 0   basecaller.py:206 */
  float   padding_value;  /* INPUT_PADDING = 0.   data_loader.py:14 */
  int32_t max_batch;      /* chunks per call (slab; evaluator uses 1024) */
  int32_t max_raw_len;    /* T_r upper bound */
  int32_t max_event_len;  /* T_e upper bound */
  int32_t max_output_len; /* L upper bound (decode runs at most L-1 steps) */
  int32_t max_beam;       /* beam width upper bound (<= 8) */
  int32_t device;         /* HIP device ordinal */
} RvConfig;

typedef struct RvContext* rv_handle;

int rv_abi_version(void);

/* Basecaller.__init__: allocates every device buffer, streams and graphs for the max shapes. */
int rv_create(const RvConfig* cfg, rv_handle* out);
void rv_destroy(rv_handle h);

/* Message of the last failing call on this handle (h == NULL: last rv_create failure). */
const char* rv_last_error(rv_handle h);

/* Number of fp32 values rv_load_weights expects for this handle's config. */
size_t rv_weight_count(rv_handle h);

/* Basecaller.load_weights (ravvent_performance_evaluator.py:107): flat little-endian fp32
 * blob, order documented in ravvent-basecaller_amd/weights.py. Host pointer. */
int rv_load_weights(rv_handle h, const float* blob, size_t n_floats);

/* Basecaller.beam_search_prediction (basecaller.py:296-315).
 *   raw   [B,T_r,1] f32 (NULL in event mode), event [B,T_e,5] f32 (NULL in raw mode),
 *   W beam width, L = max_output_len (decode runs S <= L-1 steps, S returned in *S_out).
 *   tokens [B, L-1] i32 and scores [B, L-1] f32, row stride L-1; columns [0,S) are
 *   predicted_ids[:,:,0] and beam_search_decoder_output.scores[:,:,0]; columns >= S are
 *   pad_token / 0. */
int rv_beam_search(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r,
                   int32_t T_e, int32_t W, int32_t L, int32_t* tokens, float* scores, int32_t* S_out);
int rv_beam_search_dev(rv_handle h, const float* d_raw, const float* d_event, int32_t B, int32_t T_r,
                       int32_t T_e, int32_t W, int32_t L, int32_t* d_tokens, float* d_scores,
                       int32_t* S_out);

/* Beam search with the callers' post-processing fused on the device (scope row 8f#3):
 * tokens_to_nuc_sequences (basecaller.py:289-294) and calc_prob_logits_beam_search_scores
 * (utils.py:123-128) of the best beam, as ravvent_performance_evaluator.py:66-70 consumes them.
 *   lut[vocab]: token id -> upper-case ASCII letter, 0 for ids the string form drops ('', '^', '$');
 *   bases [B, L-1] u8: the chunk's letters compacted to the front (zero-filled tail);
 *   lengths [B] i32: letters per chunk;  probs [B, L-1] f32: exp(score_t - score_{t-1}), score_{-1} = 0
 *   (the reference pairs a chunk's string with probs[:len(string)]).  Host buffers. */
int rv_beam_search_calls(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r,
                         int32_t T_e, int32_t W, int32_t L, const uint8_t* lut, uint8_t* bases,
                         int32_t* lengths, float* probs, int32_t* S_out);

/* Asynchronous form of rv_beam_search* (no counterpart in the reference, whose evaluator decodes one slab at a time:
 * ravvent_performance_evaluator.py:51-55).  The handle owns "async_depth" slab contexts (own stream and buffers, one set of
 * weights); rv_beam_search_submit* queues a slab's whole path on an idle context and returns a ticket without waiting for the
 * GPU, rv_beam_search_collect* waits for that slab and hands out its results, which are byte-identical to the synchronous
 * call's (with option "wide_recurrence" at 1, the default, 2 or 0; at -1 the recurrence form is chosen per call from the slabs in flight,
 * and the forms agree to f32 rounding only).  Up to async_depth slabs are in flight: slab k+1's encoders run beside the tail of slab k's decode (chunks leave the
 * decode as their beams finish), and with several slabs in flight the encoder recurrences switch to 16 chunks per workgroup on
 * the matrix pipe (option "wide_recurrence").  Tickets may be collected in any order; submit fails with RV_ESTATE when every
 * context holds an uncollected call.  Same single-thread rule as the rest of the handle.
 *   submit       : host inputs, copied to pinned staging before the call returns (the caller may reuse its buffers at once)
 *   submit_dev   : device inputs and outputs; they must stay valid and untouched until the ticket is collected
 *   submit_calls : the fused post-processing of rv_beam_search_calls */
int rv_beam_search_submit(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e,
                          int32_t W, int32_t L, int32_t* ticket);
int rv_beam_search_collect(rv_handle h, int32_t ticket, int32_t* tokens, float* scores, int32_t* S_out);
int rv_beam_search_submit_dev(rv_handle h, const float* d_raw, const float* d_event, int32_t B, int32_t T_r, int32_t T_e,
                              int32_t W, int32_t L, int32_t* d_tokens, float* d_scores, int32_t* ticket);
int rv_beam_search_collect_dev(rv_handle h, int32_t ticket, int32_t* S_out);
int rv_beam_search_submit_calls(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e,
                                int32_t W, int32_t L, const uint8_t* lut, int32_t* ticket);
int rv_beam_search_collect_calls(rv_handle h, int32_t ticket, uint8_t* bases, int32_t* lengths, float* probs, int32_t* S_out);

/* Basecaller.greedy_search_prediction (basecaller.py:317-330): tokens = sample_id [B, L-1],
 * logits = rnn_output [B, L-1, vocab]; columns >= S are pad_token / 0. */
int rv_greedy_search(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r,
                     int32_t T_e, int32_t L, int32_t* tokens, float* logits, int32_t* S_out);
int rv_greedy_search_dev(rv_handle h, const float* d_raw, const float* d_event, int32_t B, int32_t T_r,
                         int32_t T_e, int32_t L, int32_t* d_tokens, float* d_logits, int32_t* S_out);

/* Options.  The DEFAULT path is: matrix-pipe recurrences ("wide_recurrence" 1) + split-f16 GEMMs + the one-launch persistent decode with
 * attention and cell product on the matrix pipe ("persistent_decode", "matrix_attention", "matrix_cell" 1).  Options marked [fma form] only
 * act with "wide_recurrence" 0 (or -1 when it picks that form); options marked [per-step decode] only with "persistent_decode" 0, debug
 * taps, or a configuration the persistent decode does not cover (more than two decoder cells; Bahdanau with two cells; beam > 5 with two cells).
 *          "debug_taps" (0/1: keep per-step logits/alignments for rv_get_tensor),
 *          "use_graph"  (0/1 [per-step decode]: replay the decode loop from a captured hipGraph),
 *          "decode_split" (1..4, default 1 [per-step decode]: beam-search decode of one slab runs as that many
 *                       concurrent sub-slabs; results are identical),
 *          "attend_threads" (0 auto | 256 | 512 [per-step decode]: workgroup size of the single-pass attend; auto uses
 *                       256 (two workgroups per CU) for slabs larger than the CU count),
 *          "flash_attend" (0/1, default 1 [per-step decode]: single-pass Luong attention over values only;
 *                       0 = two-pass keys-then-values dataflow of the reference),
 *          "concurrent_encoders" (0/1, default 1: in joint mode the event encoder runs on a side stream beside the raw encoder -- on the default
 *                       path for ONE isolated slab (a synchronous call: its matrix-pipe encoder launches use an eighth of the chip each), on the
 *                       fma form with "fused_projection" 0 under the raw encoder's input-projection GEMM; results are identical),
 *          "fused_projection" (0/1, default 1 [fma form]: encoder layers >= 1 compute their input projection inside the
 *                       recurrence kernel, on MFMA waves of the same workgroup; 0 = separate GEMM launch + pre-projected
 *                       tensor; results agree to fp32 rounding),
 *          "split_projection" (0/1/2, default 2 [fma form for the encoder; 0 also moves the memory projection to f32 MFMAs on every path]: the fused projection runs on 16-bit MFMAs with both operands cut into parts
 *                       whose products are exact in f32.  2 = two f16 parts of the scaled operands (2^14 x, and W scaled per
 *                       column by a power of two), three products: operands held to 2^-23, what f32 holds; 1 = three bf16
 *                       parts, six products; 0 = v_mfma_f32_16x16x4_f32 on the f32 operands.  Results agree to f32 rounding),
 *          "tail_wave"  (0/1, default 1 [fma form]: encoder layer 0 with two or more chunks per workgroup leaves its cell update to a
 *                       ninth wave and runs its rows as two groups half a step apart; 0 = every wave does its own; results
 *                       agree to fp32 rounding),
 *          "lane_projection" (0/1, default 1: with the matrix-pipe recurrence the EVENT encoder's layer 0 takes its five-feature input
 *                       projection in the lane, from LDS windows of its chunks' events, as the raw encoder's layer 0 always does with its
 *                       one feature: no projection launch, no pre-projected tensor; 0 = k_inproj_small + pre-projected inputs.
 *                       Identical results: the same fused multiply-adds in the same order),
 *          "wide_recurrence" (1/2/0/-1, default 1: the encoder recurrences as ONE split-f16 MFMA product per step for 16 chunks of a
 *                       direction per workgroup (raw layer 0 with its input projection in the lane, the other layers on inputs
 *                       pre-projected by an elementwise kernel / a split-f16 GEMM); 2 = the same with EIGHT chunks per workgroup,
 *                       both f16 parts of a chunk's state in the product's columns: a shorter step (0.29 vs 0.40 ms per C3 layer)
 *                       on twice the workgroups -- the latency form; 0 = packed fp32 FMAs on 1-8 chunks per workgroup with the
 *                       projection fused in ("fused_projection", "split_projection", "tail_wave" configure that form).  The
 *                       matrix forms cost per workgroup, the FMA form per chunk.  -1 = choose per call from the chunks in flight
 *                       (this call's slab x the calls in flight): FMA below 160, form 2 up to 512, form 1 above -- fastest (one
 *                       isolated C3 slab: 1.13 ms against 1.34 with form 1), but the forms agree to f32 rounding only, so results
 *                       then depend on the slab size and on what else is in flight; with 1, 2 or 0 a chunk's result never depends on
 *                       the slab or shard it travels in),
 *          "async_depth" (1..16, default 2: contexts the rv_beam_search_submit* calls rotate through; returns the warning RV_WQUEUES -- the
 *                       option is set -- when the value exceeds GPU_MAX_HW_QUEUES as found in the environment, see "Hardware queues" above),
 *          "slab_graph" (0/1, default 0: a call on the default path -- matrix-pipe recurrences, persistent decode, no profiling, no taps --
 *                       replays as ONE hipGraphLaunch per slab (captured per slab context and call shape; the caller's input / output
 *                       addresses reach the kernels through a table in mapped pinned memory) instead of nine kernel launches (ten with "lane_projection" 0): half the
 *                       host time per call (24 us against 46), but the replayed slabs stream 1.5-3 % slower on ROCm 7.2, so it is off
 *                       unless a caller's host thread is the bottleneck; results are identical),
 *          "persistent_decode" (0/1, default 1: Luong beam search (beam <= 8; <= 5 with two decoder cells) and greedy search, no
 *                       debug taps runs its whole decode loop in ONE launch, one workgroup per chunk, the
 *                       chunk's attention memory resident in registers; 0 = per-step kernels in a hipGraph.
 *                       The step_ids / parent_ids / step_scores taps of a chunk then end at its own last
 *                       step instead of the slab's),
 *          "matrix_attention" (0/1, default 1: the persistent decode with Luong attention and one decoder cell takes its scores
 *                       and its context on the matrix pipe -- the chunk's keys and attention-layer image stay resident as f16
 *                       MFMA fragments, two parts per value, three exact part products per block as in "split_projection";
 *                       0 = packed fp32 FMAs on fp32 rows.  Results agree to f32 rounding),
 *          "matrix_cell" (0/1, default 1; needs "matrix_attention": the same decode also takes the decoder cell's product
 *                       [attention | h] . [W_a ; U + A_h W_a] on the matrix pipe -- the kernel streamed as f16 MFMA fragments, two
 *                       parts per value, three exact part products per block; 0 = packed fp32 FMAs.  Results agree to f32 rounding.
 *                       With Bahdanau attention (one decoder cell) the same option moves the context, the processed query h . W_q, the
 *                       cell product and the output layer to the matrix pipe; the tanh scores stay on the vector ALU),
 *          "persist_taps" (0/1, default 0: the persistent decode also records every step's logits [S,B,W,V] for
 *                       rv_get_tensor("step_logits"); rows of a chunk beyond its own last step ("chunk_steps") are not written),
 *          "profile"    (0 off; 1: hipEvents around every launch outside the decode graph and around
 *                       the graph as a whole; 2: no graph, events around every kernel; 3: events around the decode
 *                       launch only -- the persistent decode kernel or the decode graph). */
int rv_set_option(rv_handle h, const char* key, int32_t value);

/* Debug taps of the LAST call, copied to host as fp32 (bool/int tensors are converted):
 *   "enc_output" [B,T_m,2u]   _encode_input's output              basecaller.py:405
 *   "mask"       [B,T_m]      input_mask (1.0 / 0.0)              basecaller.py:406
 *   "keys"       [B,T_m,d]    attention keys after setup_memory   basecaller.py:303
 *   "projected_memory" [B,T_m,2u] enc_output . [W_mem | A_c]: keys (columns 0..u-1) and the attention layer's image of the values
 *                              (A_c = rows u..3u-1 of the attention layer), what the persistent decode keeps on chip (last call
 *                              must have run the persistent decode)
 *   "step_logits"     [S,B,W,V]   (needs debug_taps, or persist_taps on the persistent decode)
 *   "chunk_steps"     [B]         steps each chunk ran in the persistent decode (its beams all finished there)
 *   "step_alignments" [S,B,W,T_m] (needs debug_taps)
 *   "step_ids" / "parent_ids" / "step_scores" [S,B,W]
 * n_written receives the element count; fails with RV_EINVAL if dst is too small. */
int rv_get_tensor(rv_handle h, const char* name, float* dst, size_t dst_floats, size_t* n_written);

/* Profile read-out ("profile" option): accumulated device time and launch count of the named
 * kernel since the last rv_reset_profile; rv_profile_names fills a ';'-separated list. */
int rv_get_profile(rv_handle h, const char* kernel, double* total_ms, int64_t* launches);
int rv_profile_names(rv_handle h, char* dst, size_t dst_bytes);
int rv_reset_profile(rv_handle h);

/* ---- host-side pre-processing ("next" row of the scope table; pure CPU, needs no handle) ----
 * EventDetector.run (event_detection/event_detector.py:75-210): streaming two-window t-statistic
 * event detection over one read's raw samples.  Outputs are parallel arrays of `capacity`
 * entries; *n_events receives the number of events found (call with capacity 0 / NULL arrays to
 * count).  Returns RV_EINVAL if capacity was too small. */
int rv_detect_events(const double* raw, size_t n, int32_t window_length1, int32_t window_length2,
                     double threshold1, double threshold2, double peak_height, int64_t* start,
                     int64_t* length, double* mean, double* stdv, size_t capacity, size_t* n_events);

#ifdef __cplusplus
}
#endif
#endif /* RAVVENT_HIP_H */
