/* etude_hip.h -- C ABI of libetude_hip.so: the MI355X (gfx950) implementation of Etude's two compute
 * hot paths.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (Xiugapurin/Etude) is pure Python and has no FFI; the "interface each entry point
 * replaces" is therefore the Python call it stands in for (file:line under /root/reference):
 *
 *   etd_frontend_*        AMTAPC_Extractor._wav2feature               etude/data/extractor.py:178-197
 *                         (torchaudio Resample + MelSpectrogram + log)
 *   etd_extractor_create  _load_model + _Spec2MIDI construction        etude/data/extractor.py:78-113
 *   etd_transcript        AMTAPC_Extractor._transcript                 etude/data/extractor.py:199-253
 *   etd_transcript_windows  _Spec2MIDI.forward on [B,n_bin,n_frame+2m] etude/data/extractor.py:53-56,
 *                         = Model_SPEC2MIDI.forward                    etude/models/amt_apc.py:29-49
 *   etd_mpe2note, etd_mpe2note_dev  AMTAPC_Extractor._mpe2note (host / device)  etude/data/extractor.py:256-418
 *   etd_decoder_create    load_etude_decoder + EtudeDecoder.__init__   etude/utils/model_loader.py:12-60,
 *                                                                      etude/models/etude_decoder.py:94-123
 *   etd_decoder_begin_bar / etd_decoder_step / etd_decoder_poll / etd_decoder_read_tokens
 *                         the body of EtudeDecoder.generate's token loop: forward (embeddings +
 *                         GPT-NeoX + lm_head) + greedy argmax + KV cache  etude/models/etude_decoder.py:300-343,148-206
 *   etd_decoder_generate_bar  one bar of generate(): prefill + <=limit greedy steps, stop at Bar_EOS
 *                                                                      etude/models/etude_decoder.py:291-343
 *
 * Conventions: every function returns 0 on success or a negative errno-style code (ETD_E*); the
 * message is available from etd_last_error() (thread-local).  "dev" pointers are device (HBM)
 * addresses valid in the calling process's HIP context, "host" pointers are ordinary memory.  The
 * caller owns every buffer it passes; the library owns only what *_create allocated and frees it in
 * *_destroy.  `stream` is a hipStream_t (NULL = default stream).  Calls on one handle are not
 * re-entrant; launches are asynchronous on `stream` unless stated otherwise.
 */
#ifndef ETUDE_HIP_H
#define ETUDE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history: 1 = rounds 1-2 (etd_ext_cfg without `precision` / etd_sched_cfg without the job keys in its first builds);
 * 2 = round 3: struct_bytes leads every config struct, so a caller compiled against another layout is refused instead of misread;
 * etd_decoder_stats / _stats_reset / _stamp added; the diagnostic hooks moved to etude_hip_debug.h;
 * 3 = round 6 (the change itself is round 5's): the element type of every 16-bit buffer reachable through this API -- KV cache rows, activation taps and peeks of the
 * debug header -- is IEEE half (was bf16; `etd_decoder_operand_type` / `etd_extractor_operand_type` name the type of a given build); `etd_frontend_run(feat = NULL)`
 * sizes its output; no struct layout changed. */
#define ETD_ABI_VERSION 3
#define ETD_OK 0
#define ETD_EINVAL (-22)
#define ETD_ENOMEM (-12)
#define ETD_EHIP (-5)
#define ETD_EIO (-6)      /* host file I/O failed (etd_midi_write) */

int etd_version(void);
const char* etd_last_error(void);
/* Hash of the sources this binary was built from (etude_amd/build.py: src_hash); the Python binding refuses a library whose
 * id differs from the tree it sits in, so a stale in-tree .so cannot stand in for the current kernels. */
const char* etd_build_id(void);

/* ------------------------------------------------------------------ launch profiler (measurement only)
 * When enabled, every kernel launch of the library is bracketed by HIP events on its own stream;
 * etd_prof_collect() synchronises them and accumulates per-kernel totals. */
int etd_prof_enable(int on);
int etd_prof_reset(void);
int etd_prof_collect(void);
int etd_prof_count(void);
int etd_prof_entry(int i, char* name, int name_cap, double* total_ms, long long* launches, double* flops, double* bytes);

/* ------------------------------------------------------------------ audio front end */
typedef struct etd_frontend etd_frontend;
/* Tables are built by the host layer exactly as torchaudio builds them:
 *   kernT_host [K][nw]   polyphase sinc kernel, transposed (k-major); orig/nw = sr_in/gcd, sr_out/gcd
 *   window_host[n_fft]   periodic Hann
 *   mel filterbank in CSR form: filter m covers power-spectrum bins [mel_start[m], +mel_len[m]) with
 *   weights mel_w[sum(mel_len[:m]) ...].  */
int etd_frontend_create(int sr_in, int sr_out, int orig, int nw, int K, int width, const float* kernT_host,
                        int n_fft, int hop, const float* window_host, int n_mels, const int* mel_start,
                        const int* mel_len, const float* mel_w_host, float log_offset, etd_frontend** out);
void etd_frontend_destroy(etd_frontend*);
/* STFT centre padding: 0 = reflect (AMTAPC_Extractor, torchaudio default), 1 = zeros (HFT_Transformer: pad_mode="constant",
 * etude/models/hft_transformer.py:124-131) */
int etd_frontend_set_pad_mode(etd_frontend*, int constant_zero);
/* frame-wise RMS of a mono device signal: out[t] = sqrt(mean(x[t*hop - frame/2 .. + frame)^2)), zeros outside the signal
 * (librosa.feature.rms, center=True) -- the volume contour of analyze_volume, etude/utils/preprocess.py:116-152 */
int etd_rms_frames(const float* x_dev, long long n, int frame_length, int hop_length, float* out_dev, long long n_frames, void* stream);
long long etd_frontend_resampled_len(const etd_frontend*, long long n_in);
long long etd_frontend_num_frames(const etd_frontend*, long long n_in);
/* wav_dev: planar [channels][n_in] fp32.  resampled_dev: scratch >= resampled_len floats.
 * feat_dev: [T][n_mels] fp32 log-mel, T = 1 + resampled_len / hop (written to *n_frames_out); NULL = channel mean + resample only (what
 * analyze_volume needs of this stage: etude/utils/preprocess.py:135), *n_frames_out = 0. */
int etd_frontend_run(etd_frontend*, const float* wav_dev, int channels, long long n_in, float* resampled_dev,
                     float* feat_dev, long long feat_capacity_frames, long long* n_frames_out, void* stream);

/* ------------------------------------------------------------------ extractor (hFT-Transformer) */
typedef struct etd_ext etd_ext;
typedef struct {
  int struct_bytes;       /* sizeof(etd_ext_cfg) of the caller: a mismatch is ETD_EINVAL */
  int n_margin, n_frame, n_bin, cnn_channel, cnn_kernel, hid_dim, pf_dim, n_heads;
  int n_layers_enc, n_layers_dec, n_note, n_velocity;
  float min_value;        /* -18.0: padding value of _transcript */
  int max_windows;        /* windows processed per internal batch (workspace size) */
  int chunk_frames;       /* frames per encoder/freq-decoder chunk (0 = default) */
  int precision;          /* 0 = the 16-bit serving mode: IEEE-half operands (etd_extractor_operand_type), fp32 accumulate / LayerNorm / softmax /
                             sigmoid (default, the fast path);
                             1 = exact-parity mode: fp32 weights and activations, fp32-grade products (two-plane f16 splits on the matrix cores:
                             csrc/gemm3.h, csrc/ext_fp32.hip), one window at a time -- what the note-level parity tests run on.
                             Architectures (etude/config/schema.py:100-112): mode 0 is built for the reference's default one (hid 256 / 4 heads / pf 512 /
                             256 bins / margin 32 / conv 4 x 5 / 3 + 3 layers / 128 velocities, n_frame % 32 == 0, n_note % 4 == 0 <= 128) and refuses any
                             other; mode 1 takes hid_dim = 64 * n_heads <= 512, pf_dim % 32 == 0, n_bin % 32 == 0, n_margin <= 64,
                             cnn_kernel <= 2 * n_margin + 1, any layer / note / frame counts, n_velocity <= 128 */
} etd_ext_cfg;
/* element type of the 16-bit serving mode's operands and activation buffers (debug taps): 1 = IEEE half (the default build), 0 = bf16 (-DETD_EXT_BF16) */
int etd_extractor_operand_type(void);
/* Weights: n named fp32 host tensors with the reference checkpoint's own keys ("encoder.*",
 * "decoder.*"); every key the model needs must be present with the right element count. */
int etd_extractor_create(const etd_ext_cfg* cfg, const char* const* names, const float* const* host_ptrs,
                         const int64_t* numels, int n, etd_ext** out);
void etd_extractor_destroy(etd_ext*);
/* feat_dev [T][n_bin] fp32 -> outputs over T_pad = ceil(T/n_frame)*n_frame rows of n_note:
 * onset/offset/mpe fp32 probabilities and int8 velocity argmax of the time ("B") heads; the "A"
 * (frequency) head outputs are produced only when all four *_A pointers are non-NULL.
 * PRECONDITION on the input (both entry points, both precisions): log-mel features lie in [-F, F], F = max(|min_value|, 32) -- what
 * log(mel + 1e-8) of audio in [-1, 1] gives (>= -18.4, < 15) and what the wrappers pad with (-18 / -80).  The library's 16-bit operand
 * planes (IEEE half; in the exact-parity mode the two-plane splits of csrc/gemm3.h) are scaled from bounds that assume it; finite features
 * beyond it (a spectrogram of int16-scale samples) can overflow a plane into Inf / NaN probabilities.  Not checked here (the call is
 * asynchronous); the Python mirror checks it (AMTAPC_Extractor.check_feature_range). */
int etd_transcript(etd_ext*, const float* feat_dev, long long T,
                   float* onset_B, float* offset_B, float* mpe_B, int8_t* vel_B,
                   float* onset_A, float* offset_A, float* mpe_A, int8_t* vel_A, void* stream);
/* spec_dev [B][n_bin][n_frame + 2*n_margin] fp32 (the model's own input layout) -> [B*n_frame][n_note]. */
int etd_transcript_windows(etd_ext*, const float* spec_dev, int B,
                           float* onset_B, float* offset_B, float* mpe_B, int8_t* vel_B,
                           float* onset_A, float* offset_A, float* mpe_A, int8_t* vel_A, void* stream);
/* algorithmic FLOPs of one n_frame window (SURVEY.md 8d formula) */
double etd_extractor_window_flops(const etd_ext*);

/* ------------------------------------------------------------------ notes (host) */
typedef struct { double onset, offset; int32_t pitch, velocity; } etd_note;
/* onset/offset/mpe [T][n_note] fp32 host, velocity [T][n_note] int8 host -> notes sorted by (onset, pitch).
 * Keeps the reference's numerics as it runs under numpy>=2 (see oracle/mpe2note.py).  Returns the
 * number of notes in *n_out; fails with ETD_ENOMEM if cap is too small (n_out = needed). */
int etd_mpe2note(const float* onset, const float* offset, const float* mpe, const int8_t* velocity, long long T,
                 int n_note, float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr,
                 int note_min, etd_note* out, long long cap, long long* n_out);
/* The same with the reference's two mode switches (extractor.py:256-258): mode_velocity "ignore_zero" (default) drops notes whose
 * velocity argmax is 0, "org" keeps them; mode_offset picks, when both an offset peak and an mpe drop exist, the earlier one
 * ("shorter", default), the later one ("longer") or always the offset peak ("offset") (:386-404).  The device path has the same
 * switches (etd_mpe2note_dev_modes); etd_mpe2note / etd_mpe2note_dev are the defaults, which is all the reference's callers use. */
enum { ETD_M2N_VEL_IGNORE_ZERO = 0, ETD_M2N_VEL_ORG = 1 };
enum { ETD_M2N_SHORTER = 0, ETD_M2N_LONGER = 1, ETD_M2N_OFFSET = 2 };
int etd_mpe2note_modes(const float* onset, const float* offset, const float* mpe, const int8_t* velocity, long long T, int n_note,
                       float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr, int note_min,
                       int mode_velocity, int mode_offset, etd_note* out, long long cap, long long* n_out);
/* The same conversion on the DEVICE: the four frame-wise arrays stay in HBM ([T][n_note], as
 * etd_transcript wrote them), only the notes come back, already in the reference's order.  Bit-identical to etd_mpe2note.
 * The handle owns scratch that grows to the largest T seen; calls on one handle are not re-entrant. */
typedef struct etd_m2n etd_m2n;
int etd_mpe2note_dev_create(int n_note, etd_m2n** out);
void etd_mpe2note_dev_destroy(etd_m2n*);
int etd_mpe2note_dev(etd_m2n*, const float* onset_dev, const float* offset_dev, const float* mpe_dev, const int8_t* vel_dev,
                     long long T, float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr, int note_min,
                     etd_note* out_host, long long cap, long long* n_out, void* stream);
int etd_mpe2note_dev_modes(etd_m2n*, const float* onset_dev, const float* offset_dev, const float* mpe_dev, const int8_t* vel_dev,
                           long long T, float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr, int note_min,
                           int mode_velocity, int mode_offset, etd_note* out_host, long long cap, long long* n_out, void* stream);

/* ------------------------------------------------------------------ decoder (EtudeDecoder / GPT-NeoX) */
typedef struct etd_dec etd_dec;
typedef struct {
  int struct_bytes;       /* sizeof(etd_dec_cfg) of the caller */
  int vocab_size, hidden_size, num_hidden_layers, num_attention_heads, intermediate_size;
  int max_position_embeddings, num_classes, num_attribute_bins, attribute_emb_dim;
  float rotary_pct, rope_theta, layer_norm_eps;
  int max_streams;        /* concurrent token streams (KV slots) */
  int max_ctx;            /* KV positions per stream */
  int precision;          /* 0 = exact-parity mode: fp32 weights, activations and KV cache, fp32-grade products (csrc/gemm3.h) -- the reference's token ids;
                             1 = the 16-bit serving mode: IEEE-half weights, LayerNorm rows and KV cache (etd_decoder_operand_type), fp32 residual stream /
                             accumulators / softmax / logits */
  int max_prefill_rows;   /* prompt rows one etd_decoder_begin_bars call may carry (0 = max(max_ctx, max_streams)) */
} etd_dec_cfg;
int etd_decoder_create(const etd_dec_cfg* cfg, const char* const* names, const float* const* host_ptrs,
                       const int64_t* numels, int n, etd_dec** out);
/* element type of the 16-bit serving mode (weights, KV cache, debug peeks): 1 = IEEE half (the default build), 0 = bf16 (-DETD_DEC_BF16) */
int etd_decoder_operand_type(void);
/* 1 when built with -DETD_EXPERIMENTS (measured dead ends compiled in, their environment switches live); the shipped build returns 0 and has one path per precision */
int etd_has_experiments(void);
void etd_decoder_destroy(etd_dec*);
/* A second engine over the SAME weights: own KV cache, workspaces and stream state (same cfg), the weight buffers of `src`
 * (or of the handle `src` was cloned from) are shared, not copied -- concurrent engines then stream one weight set through
 * the caches instead of one copy each.  Handles may be destroyed in any order and from any thread (the family's reference count
 * is kept under a lock); the weights go with the last one.  Using the handles concurrently, one host thread per handle, is what
 * they are for. */
int etd_decoder_clone(etd_dec* src, etd_dec** out);
/* Start one bar on stream `slot` (etude_decoder.py:291-297 + first loop iteration): reset the slot's KV
 * cache and generation state, run the prompt (ids/cls: int32 host [T]; attrs4: int32 host [4][T] in the
 * concat order of etude_decoder.py:171-176 = pitch_overlap, polyphony, note_sustain, rhythm_intensity)
 * through the model and leave the greedy first token in the slot's device-side state.  tgt_attrs4 (same
 * order) condition the generated tokens; generation stops at eos_id or after `limit` tokens.  The bar needs T + limit - 1
 * KV positions: ETD_EINVAL when that exceeds max_ctx (the reference's dynamic cache / on-the-fly rotary has no such bound,
 * etude_decoder.py:285-300; EtudeDecoder sizes max_ctx so that every legal generate() argument set fits). */
/* Sampling branch of generate() (etude_decoder.py:321-331): temperature > 0 -> softmax(logits / T), top-p filter when
 * 0 < top_p < 1, one draw per token; temperature == 0 -> greedy argmax (the default).  Draws are a pure function of (seed, the
 * stream's key, index of the token inside its bar) -- reproducible, independent of slot / engine placement; the reference
 * draws from torch's global generator instead, so parity is distributional (tests/test_gpu_sampling.py).  Keys default to
 * the slot index; etd_decoder_run_jobs sets key = (job index, bar index). */
int etd_decoder_set_sampling(etd_dec*, float temperature, float top_p, unsigned long long seed, void* stream);
int etd_decoder_set_keys(etd_dec*, int n, const int32_t* slots, const unsigned long long* keys);
int etd_decoder_begin_bar(etd_dec*, int slot, const int32_t* ids, const int32_t* cls, const int32_t* attrs4, int T,
                          const int32_t* tgt_attrs4, int eos_id, int limit, void* stream);
/* Batched form: n bars at once (distinct slots).  T[i] = prompt length of bar i; ids/cls hold the prompts back to
 * back (sum(T) rows), attrs4 is [4][sum(T)]; tgt_attrs4 [n][4], eos_ids [n], limits [n].  All prompts go through the
 * model as ONE pass (big-tile MFMA GEMMs over sum(T) rows in bf16 mode). */
int etd_decoder_begin_bars(etd_dec*, int n, const int32_t* slots, const int32_t* T, const int32_t* ids, const int32_t* cls,
                           const int32_t* attrs4, const int32_t* tgt_attrs4, const int32_t* eos_ids, const int32_t* limits, void* stream);
/* n_steps greedy decode steps for the n_active streams listed in `slots` (host array): each step feeds every
 * stream's current token (class TGT=2, its target attrs), appends K/V, and writes the argmax back as the
 * stream's current token and into its output ring -- all on the device: no host sync, no allocation.
 * Streams that already finished (EOS / limit) idle. */
int etd_decoder_step(etd_dec*, const int32_t* slots, int n_active, int n_steps, void* stream);
/* done flag and number of generated tokens of each listed stream (synchronises the stream). */
int etd_decoder_poll(etd_dec*, const int32_t* slots, int n, int32_t* done_out, int32_t* n_out_out, void* stream);
/* Copy out the tokens generated so far by `slot` (synchronises). *n = count. */
int etd_decoder_read_tokens(etd_dec*, int slot, int32_t* out, int cap, int* n, void* stream);
/* The same for n streams with ONE synchronisation: out is [n][cap], counts [n]. */
int etd_decoder_read_many(etd_dec*, int n, const int32_t* slots, int32_t* out, int cap, int32_t* counts, void* stream);
/* The whole bar loop of EtudeDecoder.generate (etude_decoder.py:246-354: prompt assembly, history window, truncation,
 * token budget, Bar_EOS stop) for MANY independent jobs, scheduled natively as concurrent device streams.  A job is
 * one (song, attribute tuple): x_ids = its condition bars back to back, x_offsets [n_bars+1], attrs4 [n_bars][4] in
 * C-ABI attribute order.  Result: out[job_offsets[j] ..] = [n_bars_done, len_0 .., tokens of bar 0 ([Bar_BOS]+generated), ...].
 * `ready` (may be null = ready now) points at a host int another thread sets non-zero once the job's condition bars are
 * valid -- the upstream pipeline stages (infer.py:82-163: extract .. tokenize) of that song have finished; the scheduler
 * admits jobs in list order and never reads x_ids/x_offsets/attrs4 of a job before its flag is set. */
typedef struct { const int32_t* x_ids; const int32_t* x_offsets; int n_bars; const int32_t* attrs4; const int32_t* ready; } etd_job;
typedef struct {
  int struct_bytes;       /* sizeof(etd_sched_cfg) of the caller */
  int bar_bos_id, bar_eos_id, n_ctx_pairs, max_position_embeddings, max_output_tokens, max_bar_token_limit;
  float context_overlap_ratio;
  int force_bar_tokens;   /* >0 (benchmarks): suppress Bar_EOS, every bar is exactly this many tokens */
  int max_streams, max_prefill_rows, steps_per_poll;
  float temperature, top_p;            /* etude_decoder.py:213-214; temperature 0 = greedy */
  unsigned long long seed;
  int job_key_offset, job_key_stride;  /* sampling keys: job k of THIS call is global job job_key_offset + k * max(job_key_stride, 1), so that several
                                          engines sharing one job list (run_engines deals it round-robin) draw independent streams */
} etd_sched_cfg;
int etd_decoder_run_jobs(etd_dec*, const etd_sched_cfg* cfg, const etd_job* jobs, int n_jobs, int32_t* out, long long out_cap,
                         long long* job_offsets, long long* n_steps_out, void* stream);
/* begin_bar + steps until done + read_tokens for one stream.  Synchronous. */
int etd_decoder_generate_bar(etd_dec*, int slot, const int32_t* ids, const int32_t* cls, const int32_t* attrs4, int T,
                             const int32_t* tgt_attrs4, int eos_id, int limit, int32_t* out, int* n_out, void* stream);
/* test hook: full logits [T][vocab] fp32 of a prompt (EtudeDecoder.forward), copied to the host. */
int etd_decoder_prefill_logits(etd_dec*, int slot, const int32_t* ids, const int32_t* cls, const int32_t* attrs4, int T,
                               float* logits_host, void* stream);
/* algorithmic HBM bytes of one decode step for n_streams at context `ctx` (SURVEY.md 8d formula) */
double etd_decoder_step_bytes(const etd_dec*, int n_streams, int ctx);
/* Exact host-side accounting of the decode steps issued on this handle since the last reset (graph replays included):
 *   out[0] steps, out[1] rows x steps (= tokens generated by steps), out[2] algorithmic K/V bytes the steps' attention read (all layers),
 *   out[3] attention launches, out[4] launches measured by the device stamps, out[5] their summed duration in seconds,
 *   out[6] algorithmic bytes (K/V + streamed weights) of the stamped launches, out[7] weight bytes one step streams (SURVEY 8d "W").
 * etd_decoder_stamp(on, skip_steps): while on, every k_dstep_attn_down launch of this handle records its own span on the device
 * (s_memrealtime of its first workgroup's start and last workgroup's end) -- the kernel's duration in the configuration it
 * actually runs in, other engines included, which HIP events cannot give inside hipGraph replays.  The first `skip_steps` decode
 * steps after switching on are left out (bytes and spans alike), e.g. the bars in which a job's 4-pair history is still filling up.
 * Stamped steps use their own captured graphs; production graphs carry no stamp code path.  All three synchronise `stream`. */
int etd_decoder_stats(etd_dec*, double* out, int n, void* stream);
int etd_decoder_stats_reset(etd_dec*, void* stream);
int etd_decoder_stamp(etd_dec*, int on, int skip_steps, void* stream);
/* (start, end) of every stamped attention launch since the last reset, in 100 MHz ticks of the device's s_memrealtime -- ONE clock for every queue of the chip, so the
 * logs of several engines can be merged into the union of the times an attention launch was running (bench.py: the roofline of concurrent engines is bytes of all
 * launches / that union, not a per-launch fraction).  out_pairs [cap][2]; *n = launches written (the log keeps the first 131 072 per handle). */
int etd_decoder_stamp_log(etd_dec*, unsigned long long* out_pairs, long long cap, long long* n, void* stream);
/* ---- TinyREMITokenizer glue on either side of the decoder (host code, no GPU) ----
 * etd_tok_create      TinyREMITokenizer.__init__ / _create_measures      etude/data/tokenizer.py:24-41,166-229
 * etd_tok_encode      encode (+ _assign_notes, grace-note linking)        :231-252, :78-116, :265-297
 * etd_tok_split_bars  split_sequence_into_bars                            :43-76
 * etd_tok_decode      decode_to_notes (+ glissandos, velocities, sort)    :300-496
 * Results are bit-identical to the reference (same double arithmetic, tie-breaking and summation orders). */
typedef struct { double bpm; int time_sig; double start; const double* downbeats; int n_downbeats; } etd_tempo_region;
enum { ETD_EV_BAR = 0 /* value 1 = BOS, 0 = EOS */, ETD_EV_POS = 1, ETD_EV_NOTE = 2, ETD_EV_DURATION = 3, ETD_EV_GRACE = 4, ETD_EV_OTHER = 5 };
typedef struct { int32_t type; int32_t value; } etd_event;
typedef struct etd_tok etd_tok;
int etd_tok_create(const etd_tempo_region* regions, int n_regions, etd_tok** out);
void etd_tok_destroy(etd_tok*);
int etd_tok_num_measures(const etd_tok*);
int etd_tok_measures(const etd_tok*, double* start, double* end, double* bpm, int32_t* time_sig);
int etd_tok_encode(const etd_tok*, const etd_note* notes, long long n, int with_grace_note, etd_event* out, long long cap, long long* n_out);
int etd_tok_split_bars(const int32_t* ids, long long n, int bar_bos_id, int bar_eos_id, int32_t* out_ids, long long cap_ids,
                       long long* bar_offsets, long long cap_bars, long long* n_bars);
int etd_tok_decode(const etd_tok*, const etd_event* events, long long n, const double* volume /* or NULL */, long long n_volume,
                   etd_note* out, long long cap, long long* n_out);

/* ------------------------------------------------------------------ MIDI output (host)
 * TinyREMITokenizer.note_to_midi, etude/data/tokenizer.py:499-524 (infer.py:207): the notes as a format-1 Standard MIDI
 * File exactly as `pretty_midi.PrettyMIDI()` + one `Instrument(program=0)` + `.write()` lays it out (220 ticks per beat,
 * 120 bpm, tick = round(time * 440)).  pitch / velocity outside 0..127 fail with ETD_EINVAL (mido raises there). */
int etd_midi_write(const etd_note* notes, long long n, const char* path);

#ifdef __cplusplus
}
#endif
#endif
