/*
 * kws.h -- batch extension of the drop-in boundary (last row): the same hot
 * path as run_classifier() (include/kws/ei_compat.h), for B clips resident in HBM, plus model
 * loading (the reference compiles its model in; here it is a .kwsm blob made by tools/eon_import.py
 * from the reference's generated MODEL/tflite-model/trained_model_compiled.cpp:70-328 and
 * MODEL/model-parameters/model_metadata.h:38-132).
 *
 * Streams and concurrency: a handle owns one set of scratch buffers (the int8 input tensor / the cepstra of the combined entry
 * points, the fast mode's clip list).  Calls on ONE handle are ordered by the library: a call enqueued on a different stream than
 * the handle's previous call first makes its stream wait (hipStreamWaitEvent) for that previous call's work, so two streams on
 * one handle are safe but do not overlap; for overlap use one handle per stream.  Growing the scratch (a larger batch than any
 * before) synchronises the device.
 *
 * Plain C ABI: pointers and sizes only.  `*_device` entry points take DEVICE pointers and a
 * hipStream_t passed as void* (NULL = default stream) and are asynchronous; the others take host
 * pointers and synchronise.  All return EI_IMPULSE_ERROR values (0 = EI_IMPULSE_OK).
 */
#ifndef KWS_H
#define KWS_H

#include <stddef.h>
#include <stdint.h>

#include "ei_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kws_handle kws_handle;

/* Build the execution plan for a model blob on HIP device `device` (tables are computed on the host
 * exactly as the reference computes them per clip -- filterbank feature.hpp:54-171, twiddles
 * kiss_fft.cpp:351-357, requantisation multipliers kernel_util_lite.cc:47-120 ... -- and uploaded once). */
EI_IMPULSE_ERROR kws_create(const void *model_blob, size_t nbytes, int device, kws_handle **out);
EI_IMPULSE_ERROR kws_create_from_file(const char *path, int device, kws_handle **out);
void kws_destroy(kws_handle *h);
const char *kws_last_error(void);          /* thread-local detail string for the last failing call */

int kws_label_count(const kws_handle *h);                /* EI_CLASSIFIER_LABEL_COUNT */
const char *kws_label(const kws_handle *h, int i);       /* ei_classifier_inferencing_categories[i] */
int kws_feature_count(const kws_handle *h);              /* EI_CLASSIFIER_NN_INPUT_FRAME_SIZE */
int kws_clip_samples(const kws_handle *h);               /* EI_CLASSIFIER_RAW_SAMPLE_COUNT */
int kws_frame_count(const kws_handle *h);                /* MFCC rows (49) */
int kws_filter_count(const kws_handle *h);               /* mel filters of the DSP block (32) */
int kws_pooled_tap_bytes(const kws_handle *h);           /* bytes/clip of the pooled-activation tap */
const char *kws_nn_kernel_name(const kws_handle *h);    /* which network kernel serves this model (diagnostics) */
const char *kws_mfcc_kernel_name(const kws_handle *h);  /* "kws_mfcc8_kernel" / "kws_mfcc_kernel" (tuned shapes), "kws_spectral_lds_kernel" (general shapes), "kws_spectral_generic_kernel" (those whose arrays exceed the LDS) */
int kws_model_is_float(const kws_handle *h);             /* 1: float32 graph (EI_CLASSIFIER_TFLITE_INPUT_QUANTIZED == 0) */

/* ---- arithmetic mode of the device-resident batch hot path (kws_run_classifier_batch_device, kws_extract_mfcc_batch_device,
 * kws_cmvn_inference_batch_device, kws_streams_step_device).  The host-buffer entry point kws_run_classifier_batch always runs the exact kernels: it is bound by
 * PCIe, the mode would buy nothing there.
 * KWS_MODE_EXACT (default): every floating-point operation replays the reference's order: MFCC features bit-identical, int8 graphs
 *   bit-identical end to end, float32 scores within 1e-6.
 * KWS_MODE_FAST: the tolerance BASELINE.json grants (1e-4 on float32 scores) is spent where the reference's operation order is
 *   expensive: fp32 power spectrum, DCT on the matrix cores, O(1) running-sum cmvnw, and -- float32 graphs of CONV_2D blocks --
 *   the network on v_mfma_f32_16x16x4_f32 in the same launch (the feature matrix never leaves the chip).  The FFT keeps
 *   KissFFT's order.  Clips whose cmvnw is ill-conditioned (near-constant column) are detected and re-run by the exact
 *   kernels inside the same call, so their results are the exact mode's.  int8 graphs: fast MFCC + the exact int8 network;
 *   an int8 input value may then differ by one step where a feature sits on a rounding boundary.
 * kws_streams_step_device follows the mode too (the slice's MFCC stays exact; the whole-window cmvnw + network take the fast
 * kernel).  The SDK entry points (run_classifier ...) and the other stage entry points always run the exact kernels.
 * Models whose DSP block is MFE (extract_mfe_features of the newer SDK copy): KWS_MODE_FAST runs the block's front end -- FFT in
 *   KissFFT's order, fp32 power, fused mel products -- in the fast kernel and keeps the block's normalisation (it divides by the
 *   matrix's range, not by a deviation: nothing is ill-conditioned, no clip is handed back) and the network on their exact kernels;
 *   kws_run_classifier_batch_device and kws_extract_mfcc_batch_device follow the mode, the entry points that start from mel matrices
 *   (streams, kws_cmvn_inference_batch_device) have nothing left to relax. */
#define KWS_MODE_EXACT 0
#define KWS_MODE_FAST 1
EI_IMPULSE_ERROR kws_set_mode(kws_handle *h, int mode);   /* KWS_ERROR_UNSUPPORTED_MODEL if the model's DSP configuration is outside the fast kernel (general-shape kernels) */
int kws_get_mode(const kws_handle *h);
/* 1: KWS_MODE_FAST runs this model's network fused behind the MFCC block (float32 CONV_2D graphs); 0: features go through HBM */
int kws_fast_is_fused(const kws_handle *h);
/* What the last KWS_MODE_FAST batch call on this handle did with its clips (both synchronise the device):
 *   kws_fast_fallback_count  clips the fast kernel (first tier) handed back.  They went to the SECOND tier: cepstra from the exact kernels
 *                            (bit-identical to the reference's), then the fast cmvnw + network from those -- about 0.4 x the exact path;
 *   kws_fast_exact_count     clips the second tier handed back in turn: they were finished by the exact kernels and carry the exact
 *                            mode's bits.
 * A float32 graph whose gain leaves the fast DSP tiers no room (entry_tier >= 2) takes another route since round 5: EVERY clip's feature
 * matrix comes from the exact kernels (bit for bit) and the fused network runs from it on the matrix cores; what is left to guard is the
 * network's own arithmetic against the clip's own scores.  Both counts then are the (normally zero) clips that guard sent through the
 * exact network as well.
 * When a tier hands a clip back -- the rule follows from the LOADED MODEL: cmvnw divides whatever the fast arithmetic
 * moved in a cepstral coefficient, or in a window's mean, by the window's deviation; the graph carries a feature error into its logits with
 * a gain that depends on its weights.  kws_create measures that gain per cepstral column (reverse differentiation of the float graph on a
 * calibration set: kws_fast_gain) and the kernels sum, per clip,
 *     V = sigma_net^2 + sum over cmvnw windows (row r, column c) of ((abs[c] + lev[c] x level + rel[c] x |window mean|) / (deviation + eps))^2
 * -- an estimate of the variance of the error of a logit difference; abs / lev / rel = gain[c] x the rms error of a coefficient (absolute,
 * per unit of the clip's log-mel level = mean over its frames of |mean over the filters of the log-mel energies|, per unit of |mean|:
 * kws_fast_guard), level = 0 for tier 2.  Tier 1, clips with digitally silent frames (kws_fast_tolerance::silent_rows_exact): those frames' rows are
 * the reference's own, so abs and lev are multiplied by sqrt(live frames / frames) and level is taken over the live frames.  The clip stays in its tier iff
 *     V x max(g_c1 x P^2, g_c2) <= 1,    g_c1 = (k_sigma x lin_margin / score_tol)^2,  g_c2 = (k_sigma / logit_cap)^2,
 * P = the largest p (1 - p) among the clip's own scores where the network runs in the same launch (|d score| <= p (1 - p) x the error of
 * a logit difference: a saturated softmax passes nothing on), 1/4 otherwise.  In words: k_sigma standard deviations of the estimated logit
 * error, through the clip's own softmax, must stay below the score tolerance of 1e-4.  A calibrated statistical estimate, not a bound
 * (a worst-case bound through the weights' row sums would refuse every model); tests/test_gpu_fast_families.py re-evaluates the rule
 * from the oracle's cepstra and holds scores AND logits to it on eleven input families.
 * How many standard deviations k_sigma is worth depends on how well the calibrated gain covers the clip at hand: gain[c] is the largest value
 * over 48 calibration matrices x 1.25.  On real clips' Jacobians (tests/test_gain_calibration.py) no clip's total gain exceeds the calibrated
 * one -- the full 4.5 sigma for an error spread over the columns -- and a single column's gain reaches at most 1.3 x the calibrated value:
 * 4.5 / 1.3 = 3.4 sigma for a clip whose whole error sat in that column (kws_fast_tolerance::k_sigma_worst_column).  For a graph whose
 * activation patterns on its real inputs differ from anything the calibration set reaches, nothing bounds the underestimate: the numbers
 * are what was measured on the shipped graphs.
 * int8 graphs have no float logits to protect (the network is bit-exact from its int8 input tensor on): gain[c] is the constant for which
 * the rule reads "k_sigma x the rms of the clip's feature error estimates <= 1e-4", calibrated = 0.
 *   kws_fast_guard   coef [4][n_columns]: abs, lev, rel, and the alternative rel: column 0 -- when its window means were replayed in the reference's
 *                    order (a decision of the kernel; always for a clip with digitally silent frames); the other columns -- where the reference's
 *                    sequential window sums round systematically: every column of a clip with digitally silent frames (a frame energy of exactly 0),
 *                    and (float32 graphs) a column whose deviation is below kws_fast_tolerance::systematic_ratio x |mean| in its lane's first window
 *   kws_fast_gain    gain [n_columns] of a float32 graph (logit-difference error per unit of feature error, rms over a column's rows)
 * kws_streams_step_device and kws_cmvn_inference_batch_device start from exact cepstra: their one fast tier is tier 2. */
typedef struct {
    float score_tol, k_sigma, lin_margin, logit_cap;   /* 1e-4, 4.5, 1.1, 0.1 */
    float g_c1, g_c2;                                  /* as above */
    float sigma_net;                                   /* sqrt of the clip-independent part of V: the matrix cores' summation order in a fused float32 graph, the
                                                          relative error of a window's deviation (x total_gain) */
    float total_gain;                                  /* sqrt(sum over all features of gain^2) */
    float uniform_feature_tol;                         /* the feature error of random sign, the same size on every feature, that exactly meets the rule at P = 1/4 */
    int calibrated, n_columns, n_frames;
    int entry_tier;                                    /* where kws_run_classifier_batch_device starts in KWS_MODE_FAST: 1 the fast kernel; 2 / 3: the graph's gain
                                                          leaves tier 1 (and tier 2) no room -- a typical clip would be handed on anyway.  float32 graphs of the
                                                          fused shapes then get the exact kernels' feature matrix + the network on the matrix cores; other graphs:
                                                          2 = exact cepstra for every clip, then the fast cmvnw + network; 3 = the exact kernels.  Routing only:
                                                          every tier applies its guard */
    int dev_overrides;                                 /* non-zero: a KWS_DEV_FAST_* development switch (guard off / scaled, no re-run) was set in the environment when
                                                          the model was created -- KWS_MODE_FAST results are then outside the documented tolerance */
    float k_sigma_worst_column;                        /* k_sigma / 1.3: what k_sigma is worth for a clip whose whole error sits in the column where a real clip's gain was
                                                          measured furthest above the calibrated one (see above); = k_sigma for int8 graphs (no gain is calibrated) */
    int silent_rows_exact;                             /* 1 (round 6): the rows of digitally silent frames (frame energy exactly 0) carry the reference's own cepstral
                                                          row in tier 1 -- recorded at kws_create from the exact kernels on an all-zero window --, so the rule's abs / lev
                                                          terms are multiplied by sqrt(live frames / frames) and `level` is the mean over the LIVE frames only */
    float systematic_ratio;                            /* a column whose deviation is below this x |mean| in a lane's first window (rows 0, cr, 2 cr ... of the kernel's
                                                          row groups, cr = 13 for up to 16 columns and in the three-waves-per-SIMD build, else 17) takes the alternative rel coefficient, like a clip with silent frames (column 0 of
                                                          such a clip always has its window means replayed) */
    int fused_waves_per_simd, fused_waves;             /* (round 6) the build of the fast kernel a float32 graph's batch calls enter through: 2 waves per SIMD (8 per
                                                          workgroup, 256 registers) or 3 (12 / 11 per workgroup, <= 168 registers, clips dealt out by ticket: plans that
                                                          enter through the PCM form and hold at least eleven waves in the LDS block); 0, 0: no fused float32 form */
} kws_fast_tolerance;
EI_IMPULSE_ERROR kws_fast_fallback_count(kws_handle *h, size_t *count);
EI_IMPULSE_ERROR kws_fast_exact_count(kws_handle *h, size_t *count);
EI_IMPULSE_ERROR kws_fast_guard(const kws_handle *h, int tier, float *coef);
EI_IMPULSE_ERROR kws_fast_gain(const kws_handle *h, float *gain);
EI_IMPULSE_ERROR kws_fast_tolerance_info(const kws_handle *h, kws_fast_tolerance *out);

/* float32 graphs: the device-resident batch entry points that produce scores (kws_run_classifier_batch_device,
 * kws_cmvn_inference_batch_device) also write every clip's FULLY_CONNECTED outputs -- the logits the SOFTMAX reads -- to
 * logits [B][label_count] (device) until the tap is cleared with NULL; both modes, every tier of KWS_MODE_FAST.  A score near 0 or 1 hides
 * its logit (d score = p (1 - p) d logit): the parity tests of the fast mode hold the logits themselves.  B is the calling batch's. */
EI_IMPULSE_ERROR kws_set_logits_tap(kws_handle *h, float *logits);

/* The model used by the SDK-style entry points run_classifier()/run_inference().  If none was set,
 * the first call loads the file named by the environment variable KWS_MODEL on device KWS_DEVICE (0). */
EI_IMPULSE_ERROR kws_set_default_model(kws_handle *h);
kws_handle *kws_default_model(void);

/* ---- the hot path, batch form: replaces run_classifier() for B clips ------------------------------
 * pcm      [B][clip_samples] int16, device
 * scores   [B][label_count]  float, device  (classification[].value of each clip)
 * features [B][feature_count] float, device, optional (NULL to skip): extract_mfcc_features output
 * q_in     [B][feature_count] int8,  device, optional: the quantised input tensor              */
EI_IMPULSE_ERROR kws_run_classifier_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *scores,
                                                 float *features, int8_t *q_in, void *stream);
/* host pointers; copies in, runs, copies out, synchronises */
EI_IMPULSE_ERROR kws_run_classifier_batch(kws_handle *h, const int16_t *pcm, size_t B, float *scores,
                                          float *features, int8_t *q_in);

/* ---- the stages, for callers that hold intermediate data already and for parity tests ------------- */
/* speechpy::feature::mfcc for B windows (dsp/speechpy/feature.hpp:370-439): cepstra BEFORE cmvnw,
 * mfcc [B][feature_count] float, device */
EI_IMPULSE_ERROR kws_mfcc_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *mfcc, void *stream);
/* processing::cmvnw (processing.hpp:326-389) + run_inference (ei_run_classifier.h:293-493) on B cepstral
 * matrices; features / q_in optional outputs as above */
EI_IMPULSE_ERROR kws_cmvn_inference_batch_device(kws_handle *h, const float *mfcc, size_t B, float *scores,
                                                 float *features, int8_t *q_in, void *stream);
/* speechpy::feature::mfe for B clips (dsp/speechpy/feature.hpp:193-318; the MFE block's front end):
 * mel [B][frames][filters] filterbank energies and energy [B][frames] frame energies (may be NULL), both after
 * zero handling, before any log. */
EI_IMPULSE_ERROR kws_mfe_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *mel, float *energy, void *stream);
/* extract_mfe_features for B clips -- the MFE DSP block of the newer SDK copy (nucleo-l432 .../edge-impulse-sdk/classifier/
 * ei_run_dsp.h:369-418): speechpy::feature::mfe on the raw signal (no pre-emphasis), processing::cmvnw(win_size, false, true)
 * (dsp/speechpy/processing.hpp:327-399) and numpy::normalize (dsp/numpy.hpp:1391-1429).  Frame / filter / window settings are
 * the model's DSP settings; features [B][frames * filters] float, device. */
EI_IMPULSE_ERROR kws_extract_mfe_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *features, void *stream);
/* extract_mfcc_features for B clips (classifier/ei_run_dsp.h:256-308) */
EI_IMPULSE_ERROR kws_extract_mfcc_batch_device(kws_handle *h, const int16_t *pcm, size_t B, float *features,
                                               int8_t *q_in, void *stream);
/* run_inference for B feature vectors (ei_run_classifier.h:293-493) */
EI_IMPULSE_ERROR kws_run_inference_batch_device(kws_handle *h, const float *features, size_t B, float *scores,
                                                void *stream);
/* network only, from int8 input tensors; optional int8 taps (device, may be NULL):
 *   tap_pooled [B][kws_pooled_tap_bytes]  every MAX_POOL_2D output, in graph order
 *   tap_fc     [B][label_count]           FULLY_CONNECTED output
 *   tap_out    [B][label_count]           SOFTMAX output                                         */
EI_IMPULSE_ERROR kws_nn_batch_device(kws_handle *h, const int8_t *q_in, size_t B, float *scores, int8_t *tap_pooled,
                                     int8_t *tap_fc, int8_t *tap_out, void *stream);
EI_IMPULSE_ERROR kws_nn_batch(kws_handle *h, const int8_t *q_in, size_t B, float *scores, int8_t *tap_pooled,
                              int8_t *tap_fc, int8_t *tap_out);
/* float32 models (the reference's float kernels: TFL/kernels/internal/reference/conv.h:28-99, add.h:179-215,
 * pooling.h:189-237, fully_connected.h:26-60, softmax.h:31-63): network only, from float feature vectors; optional tap
 *   tap_logits [B][label_count]  FULLY_CONNECTED output (bit-identical to the reference; softmax uses the device expf).
 * The int8 entry points above return KWS_ERROR_UNSUPPORTED_MODEL for a float model, and this one for an int8 model;
 * kws_run_classifier_batch*, kws_run_inference_batch_device, kws_cmvn_inference_batch_device, the stream API and the
 * SDK entry points serve both kinds (int8 outputs must then be NULL). */
EI_IMPULSE_ERROR kws_nn_f32_batch_device(kws_handle *h, const float *features, size_t B, float *scores,
                                         float *tap_logits, void *stream);

/* ---- continuous mode for S streams in lock step ----------------------------------------------------
 * Each stream follows run_classifier_continuous() (classifier/ei_run_classifier.h:184-282): one slice of audio per
 * step, a rolling cepstra buffer, whole-window cmvnw + network once it is full, 2-tap moving average per class
 * (ei_run_classifier.h:134-145).  All per-stream state lives in HBM.  As in the reference, every step of a batch
 * but its very first claims one extra frame length (ei_run_dsp.h:319-325; not reset by kws_streams_init) and
 * pre-emphasis then needs the sample one frame beyond the slice: pass those S floats in end_of_signal (device),
 * or NULL for 0 (what the reference sees when the application's get_data refuses the read).
 *   slices [S][slice_samples] int16 (device), scores [S][label_count] float (device), *produced = inference ran. */
typedef struct kws_stream_batch kws_stream_batch;
EI_IMPULSE_ERROR kws_streams_create(kws_handle *h, size_t S, kws_stream_batch **out);
void kws_streams_destroy(kws_stream_batch *sb);
EI_IMPULSE_ERROR kws_streams_init(kws_stream_batch *sb);            /* run_classifier_init, ei_run_classifier.h:164 */
EI_IMPULSE_ERROR kws_streams_step_device(kws_stream_batch *sb, const int16_t *slices, size_t slice_samples,
                                         const float *end_of_signal, float *scores, int *produced, void *stream);

/* ---- multi-GPU: clips shard contiguously over the GPUs of one node (rank r owns clips [r*B, (r+1)*B)), tables are
 * replicated, nothing is exchanged inside the pipeline; the one collective is the all-gather of the per-clip scores over xGMI.
 * It goes through RCCL's C API (librccl is opened on first use: single-GPU applications do not need it).  One process per GPU:
 * rank 0 obtains the 128-byte id and hands it to the other ranks with whatever started them (environment, file, MPI, a socket --
 * bench.py uses torch.distributed); every rank then creates its communicator on its own device.
 *   all_scores [world_size * clips_per_rank][label_count] float, device: rank-major = global clip order.  Asynchronous on
 * `stream` (the same stream as the batch call that produced local_scores: no synchronisation in between). */
#define KWS_COMM_ID_BYTES 128
typedef struct kws_comm kws_comm;
EI_IMPULSE_ERROR kws_comm_unique_id(void *id, size_t nbytes);
EI_IMPULSE_ERROR kws_comm_create(const void *id, size_t nbytes, int world_size, int rank, int device, kws_comm **out);
int kws_comm_world_size(const kws_comm *c);
int kws_comm_rank(const kws_comm *c);
int kws_comm_ranks_seen(const kws_comm *c);     /* ncclCommCount of the communicator: what RCCL itself says (kws_comm_create refuses a mismatch) */
int kws_comm_rccl_version(void);                /* ncclGetVersion of the loaded librccl (0: not loadable); its major version must be the rccl.h's this library was built with */
/* Waits until everything enqueued on `stream` (the batch call and its all-gather) has completed -- against a deadline (environment variable
 * KWS_COMM_TIMEOUT_MS, default 120 000; kws_comm_create's wait for the other ranks uses the same one): if a peer has failed or nothing moves,
 * the communicator is aborted (ncclCommAbort) and KWS_ERROR_HIP returned instead of hanging. */
EI_IMPULSE_ERROR kws_comm_wait(kws_comm *c, void *stream);
EI_IMPULSE_ERROR kws_allgather_scores(kws_comm *c, const float *local_scores, float *all_scores, size_t clips_per_rank, int label_count,
                                      void *stream);
void kws_comm_destroy(kws_comm *c);

/* ---- the step before the path: mix_audio of /root/reference/dataset-curation.py:93-137, batched on the GPU, so that a
 * harness can feed real keyword / background recordings instead of synthetic clips.  Inputs are float32 waveforms already at the
 * model's sampling rate (what librosa.load(sr = 16000, mono = True) returns: resampling is NOT part of this call):
 *   words [n_clips] waveforms of word_len[b] samples at words + b * word_stride (device; NULL = background noise only),
 *   noise one background track (device; NULL = no background), start[b] = first sample of clip b's window in it (the reference
 *   draws it with random.randint(0, len(noise) - n); the caller supplies it),
 *   out [n_clips][n] int16 = PCM16(0.5 * word_vol * word + 0.5 * bg_vol * noise[start .. start + n)), words padded with zeros or
 *   truncated to n samples.  PARITY UNPINNED (librosa / soundfile unavailable when this was written): held to the restatement in
 *   oracle/ only. */
EI_IMPULSE_ERROR kws_mix_audio_device(const float *words, const int *word_len, size_t word_stride, const float *noise, size_t noise_len,
                                      const int *start, float word_vol, float bg_vol, size_t n_clips, size_t n, int16_t *out, void *stream);

/* ---- the rest of that step: what librosa.load(path, sr = 16000, mono = True) (dataset-curation.py:111,126) does with a WAV file before
 * mix_audio sees it -- decode, mix down to mono, resample.  PARITY UNPINNED like kws_mix_audio_device (no librosa / soundfile / resampy in
 * reach): the decoder follows libsndfile's published conversion rules (integer PCM / 2^(bits - 1); 8-bit WAV is unsigned), the mono mix-down
 * NumPy's mean over the channels, the resampler the published design of resampy's "kaiser_best" filter (Kaiser-windowed sinc, 64 zero
 * crossings); tests hold them to Python's `wave` / scipy.io.wavfile and to scipy.signal.resample_poly within a stated tolerance.
 *   kws_wav_info_from_memory  container facts of a RIFF/WAVE image in memory (PCM 8 / 16 / 24 / 32 bit, IEEE float 32, also as
 *                             WAVE_FORMAT_EXTENSIBLE; unknown chunks are skipped)
 *   kws_wav_decode_mono       host: samples -> float32 in [-1, 1), channels averaged; out == NULL only reports *frames / *sample_rate
 *   kws_resample_length       ceil(n_in * sr_out / sr_in), librosa's output length
 *   kws_resample_device       device -> device; sr_in == sr_out copies (librosa.load leaves such a file alone).  The reference's behaviour:
 *                             resampy's published loop (a wing walks the table in truncated integer steps with one interpolation factor,
 *                             float32 accumulation, floor(n ratio) samples, then librosa's fix_length zero padding up to ceil(n ratio))
 *   kws_resample_device_ex    flags = KWS_RESAMPLE_EXACT_POSITIONS: every tap's table position exact, products summed in double, every
 *                             sample computed -- 5e-8 .. 7e-7 from the analytic signal where the reference's stepping is 6e-4 .. 2e-3 away */
#define KWS_RESAMPLE_EXACT_POSITIONS 1
typedef struct {
    int channels, sample_rate, bits_per_sample, is_float;
    size_t frames, data_offset;
} kws_wav_info;
EI_IMPULSE_ERROR kws_wav_info_from_memory(const void *bytes, size_t nbytes, kws_wav_info *info);
EI_IMPULSE_ERROR kws_wav_decode_mono(const void *bytes, size_t nbytes, float *out, size_t out_cap, size_t *frames, int *sample_rate);
size_t kws_resample_length(size_t n_in, int sr_in, int sr_out);
EI_IMPULSE_ERROR kws_resample_device(const float *in, size_t n_in, int sr_in, float *out, size_t n_out, int sr_out, void *stream);
EI_IMPULSE_ERROR kws_resample_device_ex(const float *in, size_t n_in, int sr_in, float *out, size_t n_out, int sr_out, int flags, void *stream);

/* deterministic synthetic clips generated directly in HBM (include/kws/kws_synth.h) */
EI_IMPULSE_ERROR kws_synth_clips_device(uint32_t seed, uint32_t first_clip, uint32_t n_clips, uint32_t clip_len,
                                        int16_t *out, void *stream);

/* device memory helpers so that a pure-C caller needs no HIP headers */
EI_IMPULSE_ERROR kws_device_malloc(void **ptr, size_t nbytes);
EI_IMPULSE_ERROR kws_device_free(void *ptr);
EI_IMPULSE_ERROR kws_memcpy_h2d(void *dst, const void *src, size_t nbytes);
EI_IMPULSE_ERROR kws_memcpy_d2h(void *dst, const void *src, size_t nbytes);
EI_IMPULSE_ERROR kws_device_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif
