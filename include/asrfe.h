/*
 * libasrfe - MI355X (gfx950) speech frontend + Paraformer SAN-M encoder, C ABI.
 *
 * The reference (terrense/toolbox-for-ASR-and-TTS) has no FFI of its own for this
 * path: it calls Python objects of third-party packages.  Each entry point below
 * names the Python-level interface it stands in for; reference citations are
 * relative to /root/reference/.
 *
 *   funasr.AutoModel(...) / modelscope pipeline construction
 *       voice-service/app/services/voice_interface.py:686-700, 832-841   -> asrfe_create,
 *                                                                           asrfe_load_cmvn,
 *                                                                           asrfe_load_encoder_weights
 *   pipeline(wav, batch_size_s=..., ...) -> WavFrontend.forward(input[B,N], lengths[B])
 *       voice_interface.py:2049-2053 (ASR), :1585-1590 (VAD), :1370-1374 (KWS)
 *                                                                        -> asrfe_fbank,
 *                                                                           asrfe_lfr_cmvn,
 *                                                                           asrfe_frontend
 *   ... -> SANMEncoder.forward(xs_pad[B,T',560], ilens[B])
 *       voice_interface.py:2049-2053                                     -> asrfe_encoder_forward
 *
 * Conventions
 *   - every function returns an int32 status (ASRFE_OK == 0, negative on error); no C++
 *     exception crosses the boundary; asrfe_last_error() holds the detail text;
 *   - all data pointers except the *_host ones are DEVICE pointers owned by the caller;
 *     the library never frees caller memory and never allocates in the compute calls
 *     (asrfe_fbank, asrfe_lfr_cmvn, asrfe_frontend, asrfe_encoder_forward): scratch
 *     comes in through the workspace argument;
 *   - compute calls are asynchronous with respect to the host and ordered on the HIP
 *     stream passed as void* (a hipStream_t; NULL = the default stream);
 *   - a context is bound to the device that was current in asrfe_create and is not
 *     internally synchronised: one context per (process, device);
 *   - PCM is mono 16 kHz, row b of a [B, pcm_stride] array holding pcm_lens[b] samples:
 *     int16 (canonical, voice_interface.py:1997-2023) or float32 in [-1, 1) which is
 *     multiplied by 32768 (voice_interface.py:1009-1010, 1046).
 */
#ifndef ASRFE_H
#define ASRFE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASRFE_OK 0
#define ASRFE_ERR_BAD_ARG (-1)
#define ASRFE_ERR_SHAPE (-2)
#define ASRFE_ERR_HIP (-3)
#define ASRFE_ERR_UNSUPPORTED_ARCH (-4)
#define ASRFE_ERR_NOT_LOADED (-5)
#define ASRFE_ERR_WORKSPACE (-6)
#define ASRFE_ERR_RANGE (-7) /* fp16 operands cannot hold this model's activations: use ASRFE_DT_BF16 */

#define ASRFE_PCM_I16 0
#define ASRFE_PCM_F32 1
/* float32 in [-1, 1] as the reference's streaming session buffer holds it; the kernel applies the session's int16
 * hand-off while loading: int16(clip(x, -1, 1) * 32767), truncating (voice-service/app/services/
 * voice_interface.py:1997-2010), so the features equal those of the WAV file the reference writes for its pipeline. */
#define ASRFE_PCM_F32_WIRE 2

#define ASRFE_DT_F32 0
#define ASRFE_DT_BF16 1
#define ASRFE_DT_F16 2

/* Threading.  A context is immutable once its tables and weights are loaded: asrfe_fbank, asrfe_lfr_cmvn,
 * asrfe_frontend and asrfe_encoder_forward only enqueue kernels on the caller's stream and may be called from
 * several threads at once on one context, provided every call has its own outputs and workspace.  The last-error
 * text is kept per thread: asrfe_last_error(ctx) returns the text of the calling thread's last failed call on ctx.
 * asrfe_load_*, asrfe_profile_* and asrfe_destroy must not run beside other calls on the same context
 * (asrfe_load_cmvn waits for the device to go idle before it replaces the table).  The
 * reference calls its models synchronously from one thread (voice-service/app/api/voice.py:613). */
typedef struct asrfe_ctx asrfe_ctx;

/* Frontend + encoder hyper-parameters (FunASR config.yaml: frontend_conf / encoder_conf). */
typedef struct asrfe_config {
    int32_t sample_rate;     /* 16000 */
    int32_t frame_length;    /* samples per frame: 400 (25 ms)   - only 400 is built  */
    int32_t frame_shift;     /* hop in samples:   160 (10 ms)   - only 160 is built  */
    int32_t n_fft;           /* 512 */
    int32_t n_mels;          /* 80  (<= 80) */
    float preemph;           /* 0.97 */
    float low_freq;          /* 20 Hz */
    float high_freq;         /* 0 or negative: offset from Nyquist, Kaldi style */
    int32_t window_type;     /* 0 hamming, 1 hanning, 2 povey, 3 rectangular */
    int32_t lfr_m;           /* 7 */
    int32_t lfr_n;           /* 6 */
    int32_t enc_input_size;  /* 560 */
    int32_t enc_output_size; /* 512 - only 512 is built */
    int32_t enc_heads;       /* 4   - only d_k = 128 is built */
    int32_t enc_linear_units;/* 2048 */
    int32_t enc_num_blocks;  /* 50 (1 layer input_size->output_size + num_blocks-1 layers) */
    int32_t enc_kernel_size; /* 11 */
    int32_t enc_sanm_shift;  /* 0 */
    float enc_ln_eps;        /* 1e-12 */
    int32_t enc_compute_dtype; /* 16-bit type of the MFMA operands (weights, activations, P): ASRFE_DT_F16
                                  (default: meets the 1e-2 parity bar) or ASRFE_DT_BF16 (same MFMA rate, ~8x the
                                  rounding error); accumulation, residual stream, LayerNorm and softmax are fp32 */
    int32_t enc_small_batch_rows; /* encoder calls of at most this many rows (batch * tp; one short utterance) run a
                                  launch plan built for latency: every product is cut into 32 x 32 output tiles spread
                                  over the chip (2048 by default: measured faster than the throughput plan up to ~2400
                                  rows; 0 switches it off and saves the row-major weight copies it reads, ~316 MB
                                  for paraformer-large) */
} asrfe_config;

/* One named host tensor of an encoder state_dict (FunASR key names without the "encoder." prefix). */
typedef struct asrfe_tensor_desc {
    const char* name;       /* e.g. "encoders.3.self_attn.linear_q_k_v.weight" */
    const float* data_host; /* contiguous row-major float32 on the HOST */
    int32_t ndim;
    int64_t shape[4];
} asrfe_tensor_desc;

/* Fills *cfg with the paraformer-large defaults listed above. */
int32_t asrfe_default_config(asrfe_config* cfg);

/* Stands in for funasr.AutoModel(model=..., device="cuda:0") (voice_interface.py:686-700):
 * builds window / mel / twiddle tables on the current device. */
int32_t asrfe_create(const asrfe_config* cfg, asrfe_ctx** out);
int32_t asrfe_destroy(asrfe_ctx* ctx);

/* am.mvn of the model directory: y = (x + shift) * scale over lfr_m * n_mels dims (host pointers). */
int32_t asrfe_load_cmvn(asrfe_ctx* ctx, const float* shift_host, const float* scale_host, int32_t dim);

/* model.pt encoder tensors (host float32); converted to the device layout the kernels use.
 * With ASRFE_DT_F16 operands the call first proves that no activation the kernels round to fp16 can leave the fp16
 * range: q, k, v, the attention output, the FSMN memory and the feed-forward hidden layer are linear images of a
 * normalised row n (||n||_2 = sqrt(K)), so |w . n + c| <= ||w||_2 sqrt(K) + |c| bounds them from the weights alone.
 * Where a bound fails and the function allows it, the call first moves a power of two between partners, which changes
 * no result and rounds nothing: between dimension j of q and of k (a score is sum_j q_j k_j), and between a hidden
 * unit's row of W1 / b1 and its column of W2 (ReLU is positively homogeneous).  Weights that pass keep their bits.
 * What still fails - v, the attention output and the FSMN memory have no such partner - is refused with
 * ASRFE_ERR_RANGE (load it into a context created with ASRFE_DT_BF16); ordinary checkpoints stay orders of magnitude
 * below the bound.  Independently of the proof the kernels
 * convert to fp16 with saturation (+-65504, never an infinity). */
int32_t asrfe_load_encoder_weights(asrfe_ctx* ctx, const asrfe_tensor_desc* descs, int32_t n);

/* snip_edges frame count and LFR frame count for a clip of n_samples (host helpers). */
int32_t asrfe_num_frames(const asrfe_ctx* ctx, int64_t n_samples);
int32_t asrfe_num_lfr_frames(const asrfe_ctx* ctx, int32_t n_frames);

/* kaldi.fbank over a batch: logmel[b, t, :] for t < n_frames[b]; rows t >= n_frames[b] (< t_max) are zero.
 * logmel: [B, t_max, n_mels] float32, n_frames_out: [B] int32 (device).
 * Truncation: a clip with more than t_max frames is processed as if it had been cut after frame t_max - 1, and
 * n_frames_out[b] = t_max.  asrfe_frontend does the same at tp_max * lfr_n frames (lens_out[b] <= tp_max), and
 * asrfe_lfr_cmvn reads at most t_max frames per clip and reports at most tp_max rows. */
int32_t asrfe_fbank(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride,
                    const int32_t* pcm_lens, int32_t batch, int32_t t_max, float* logmel,
                    int32_t* n_frames_out, void* stream);

/* apply_lfr + apply_cmvn on [B, t_max, n_mels] log-mel -> feats [B, tp_max, lfr_m * n_mels]
 * (float32 or bf16 per out_dtype), zero-padded rows past lens_out[b]. */
int32_t asrfe_lfr_cmvn(asrfe_ctx* ctx, const float* logmel, const int32_t* n_frames, int32_t batch,
                       int32_t t_max, int32_t tp_max, void* feats, int32_t out_dtype,
                       int32_t* lens_out, void* stream);

/* WavFrontend.forward fused in one kernel: PCM -> feats [B, tp_max, 560], lens_out [B]. */
int32_t asrfe_frontend(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride,
                       const int32_t* pcm_lens, int32_t batch, int32_t tp_max, void* feats,
                       int32_t out_dtype, int32_t* lens_out, void* stream);

/* Bytes of device scratch asrfe_encoder_forward needs for a [batch, tp] call. */
int64_t asrfe_encoder_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t tp);

/* SANMEncoder.forward: feats [B, tp, 560] float32, lens [B] int32 -> out [B, tp, 512]
 * (ASRFE_DT_F32, _BF16 or _F16), rows t >= lens[b] zero.  workspace: >= asrfe_encoder_workspace_bytes, 256-B aligned. */
int32_t asrfe_encoder_forward(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int32_t batch,
                              int32_t tp, void* out, int32_t out_dtype, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* ---- the step behind the encoder: CIF predictor + Paraformer SAN-M decoder + greedy token choice ----------------
 * Stands in for what the reference's pipeline call (voice_interface.py:2049-2053) does with the encoder output before
 * it returns text (consumed at voice_interface.py:2056-2067, 2108-2181): FunASR's CifPredictorV2.forward,
 * ParaformerSANMDecoder.forward and the arg-max of Paraformer.inference.  Turning token ids into characters needs the
 * model directory's token list and is host code (the Python mirror takes it as an argument). */
typedef struct asrfe_decoder_config {
    int32_t num_blocks;      /* 16 decoder blocks with FSMN + cross-attention (att_layer_num == num_blocks), then one
                                feed-forward-only block, after_norm and the output layer */
    int32_t linear_units;    /* 2048 (a multiple of 512 up to 4096) */
    int32_t kernel_size;     /* 11 - only 11 is built */
    int32_t vocab_size;      /* 8404 */
    float ln_eps;            /* 1e-12 */
    float threshold;         /* CIF firing threshold, 1.0 - only 1.0 is built (the reference subtracts a literal 1) */
    float tail_threshold;    /* 0.45 */
    float smooth_factor;     /* 1.0 */
    float noise_threshold;   /* 0.0 */
} asrfe_decoder_config;
int32_t asrfe_default_decoder_config(asrfe_decoder_config* cfg);

/* model.pt tensors "predictor.*" and "decoder.*" (host float32, FunASR names); needs an encoder width of 512.  With
 * fp16 operands the same range proof as asrfe_load_encoder_weights runs over the linear layers behind a LayerNorm. */
int32_t asrfe_load_decoder_weights(asrfe_ctx* ctx, const asrfe_decoder_config* cfg, const asrfe_tensor_desc* descs, int32_t n);

/* Device scratch for asrfe_cif_predictor + asrfe_decoder_forward on [batch, tp] encoder frames and up to n_max tokens per
 * utterance, and the row stride of the logits (vocab_size rounded up to a multiple of 256). */
int64_t asrfe_decoder_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t tp, int32_t n_max);
int32_t asrfe_decoder_logits_stride(const asrfe_ctx* ctx);

/* CifPredictorV2.forward: enc_out [B, tp, 512] float32 (rows >= lens[b] are not read), lens [B] ->
 * embeds [B, n_max, 512] float32 (token n's integrate-and-fire sum of frames; rows past the fired tokens zero),
 * token_num [B] int32 = floor of the summed alphas incl. the tail weight, capped at n_max (tokens beyond n_max are
 * dropped: size n_max for the longest utterance, tp + 1 is always enough), and optionally alphas_out [B, tp + 1]. */
int32_t asrfe_cif_predictor(asrfe_ctx* ctx, const float* enc_out, const int32_t* lens, int32_t batch, int32_t tp,
                            int32_t n_max, float* embeds, int32_t* token_num, float* alphas_out, void* workspace,
                            int64_t workspace_bytes, void* stream);

/* Token time stamps: alphas [B, tp + 1] float32 as asrfe_cif_predictor leaves them in alphas_out (tail weight in place)
 * -> fire_frames [B, n_max] int32, the frame at which token n fires (the predictor's integrate-and-fire recurrence on its
 * own), -1 behind an utterance's fires.  A frame is lfr_n * frame_shift samples (60 ms), so token n spans
 * (fire_frames[n - 1], fire_frames[n]] * 60 ms: what the reference reads as "start" / "end" of a sentence
 * (voice-service/app/services/voice_interface.py:2064-2067, 2124-2126). */
int32_t asrfe_cif_fire_frames(asrfe_ctx* ctx, const float* alphas, int32_t batch, int32_t tp, int32_t n_max,
                              int32_t* fire_frames, void* stream);

/* ParaformerSANMDecoder.forward + arg-max: memory = enc_out [B, tp, 512] float32 with lens [B], embeds [B, n_max, 512]
 * and token_num [B] as asrfe_cif_predictor leaves them -> token_ids [B, n_max] int32 (arg-max over the vocabulary for
 * rows < token_num[b], -1 behind them, 0 for a row whose logits have no maximum: all NaN or all -inf; sos / eos / blank
 * are left in: dropping them is the host's one-line filter) and optionally logits [B * n_max,
 * asrfe_decoder_logits_stride] float32 (columns >= vocab_size are padding).  The call builds its own 16-bit copy of
 * enc_out (16-byte aligned): any workspace of asrfe_decoder_workspace_bytes will do, it need not be the predictor's. */
int32_t asrfe_decoder_forward(asrfe_ctx* ctx, const float* enc_out, const int32_t* lens, int32_t batch, int32_t tp,
                              const float* embeds, const int32_t* token_num, int32_t n_max, int32_t* token_ids,
                              float* logits, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- HiFi-GAN vocoder (mel spectrogram -> waveform) ---------------------------------------------------------------
 * Stands in for the vocoder half of the reference's TTS pipeline call
 * pipeline_instance(input=seg, voice=voice, forward_params=...) (tts-service/app/services/tts_service.py:617-625; the
 * pipeline is modelscope's Sambert-HiFiGAN, tts_service.py:34-38, 164-177).  The acoustic model that turns text into
 * the mel spectrogram is not part of this library.  Generator structure as published (conv_pre, per stage a transposed
 * convolution and the mean of n_resblocks residual blocks of three dilated / plain convolution pairs, conv_post, tanh);
 * every convolution runs on the matrix cores as an implicit GEMM. */
typedef struct asrfe_vocoder_config {
    int32_t n_mels;                   /* 80 (a multiple of 16) */
    int32_t initial_channel;          /* 512; halves at every stage, must stay a multiple of 32 */
    int32_t n_upsamples;              /* 4 (1..8) */
    int32_t upsample_rates[8];        /* 8, 8, 2, 2: their product is the hop (samples per mel frame) */
    int32_t upsample_kernels[8];      /* 16, 16, 4, 4: a multiple of the stride each, padding (k - stride) / 2 */
    int32_t n_resblocks;              /* 3 residual blocks per stage (1..4) */
    int32_t resblock_kernels[4];      /* 3, 7, 11 (odd) */
    int32_t resblock_dilations[4][3]; /* 1, 3, 5 each */
    float lrelu_slope;                /* 0.1 */
    float final_slope;                /* 0.01: the leaky ReLU in front of conv_post */
} asrfe_vocoder_config;
int32_t asrfe_default_vocoder_config(asrfe_vocoder_config* cfg);
/* Generator tensors (host float32) under the published names with weight norm removed: conv_pre.{weight,bias},
 * ups.I.{weight,bias}, resblocks.J.convs1.K.{weight,bias}, resblocks.J.convs2.K.{weight,bias}, conv_post.{weight,bias}. */
int32_t asrfe_load_vocoder_weights(asrfe_ctx* ctx, const asrfe_vocoder_config* cfg, const asrfe_tensor_desc* descs, int32_t n);
int64_t asrfe_vocoder_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t t_mel);
int32_t asrfe_vocoder_hop(const asrfe_ctx* ctx);
/* mel [B, n_mels, t_mel] float32 (device) -> wav [B, t_mel * hop] float32 in (-1, 1). */
int32_t asrfe_vocoder_forward(asrfe_ctx* ctx, const float* mel, int32_t batch, int32_t t_mel, float* wav, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* ---- FSMN voice-activity network (frame scores for the speech / silence detector) ----------------------------------
 * Stands in for the network inside the reference's per-chunk call vad_model.generate(input=audio_np, cache=self.vad_cache,
 * is_final=False, chunk_size=...) (voice-service/app/services/voice_interface.py:1585-1590; the model is loaded at
 * :723-727): FunASR's FSMN encoder of speech_fsmn_vad_zh-cn-16k-common-pytorch.  The features come from asrfe_fbank +
 * asrfe_lfr_cmvn of a context created with lfr_m = 5, lfr_n = 1 and the model's own CMVN; the detector that turns frame
 * scores into [begin, end] segments is sequential host logic (the Python mirror carries it). */
typedef struct asrfe_vad_config {
    int32_t input_dim;         /* 400 = lfr_m 5 x 80 mel (a multiple of 8) */
    int32_t input_affine_dim;  /* 140 */
    int32_t fsmn_layers;       /* 4 */
    int32_t linear_dim;        /* 250 */
    int32_t proj_dim;          /* 128 (<= 256) */
    int32_t lorder;            /* 20 causal taps (<= 33); rorder 0 and stride 1 only */
    int32_t output_affine_dim; /* 140 */
    int32_t output_dim;        /* 248 classes; class 0 is silence */
} asrfe_vad_config;
int32_t asrfe_default_vad_config(asrfe_vad_config* cfg);
/* model.pt tensors under FunASR's names (host float32): encoder.in_linear1.linear.{weight,bias}, encoder.in_linear2...,
 * encoder.fsmn.I.linear.linear.weight, encoder.fsmn.I.fsmn_block.conv_left.weight [proj, 1, lorder, 1],
 * encoder.fsmn.I.affine.linear.{weight,bias}, encoder.out_linear1..., encoder.out_linear2....  The products run with
 * operands in the context's enc_compute_dtype (conversions to fp16 saturate). */
int32_t asrfe_load_vad_weights(asrfe_ctx* ctx, const asrfe_vad_config* cfg, const asrfe_tensor_desc* descs, int32_t n);
int64_t asrfe_vad_workspace_bytes(const asrfe_ctx* ctx, int32_t batch, int32_t t);
/* floats of cache one stream needs: fsmn_layers x (lorder - 1) x proj_dim (the rows in front of the next chunk) */
int64_t asrfe_vad_cache_floats(const asrfe_ctx* ctx);
/* FSMN.forward: feats [B, t, input_dim] float32, lens [B] int32 or NULL (= t rows each) -> scores [B, t, output_dim]
 * float32 (softmax; rows >= lens[b] hold no meaning).  cache [B, asrfe_vad_cache_floats] float32 or NULL: the memory
 * blocks read it as the rows before the chunk and the call leaves the last lorder - 1 rows of (cache ++ the chunk's
 * lens[b] rows) in it; zeros start a stream; with NULL every call starts from silence. */
int32_t asrfe_vad_forward(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int32_t batch, int32_t t, float* cache,
                          float* scores, void* workspace, int64_t workspace_bytes, void* stream);
/* The level the detector reads beside the scores (E2EVadModel.ComputeDecibel): decibel [B, t_max] float32 =
 * 10 log10(sum over the frame's 400 samples of x^2 + 1e-6) with x in [-1, 1]; 0 for rows past a clip's frames. */
int32_t asrfe_frame_decibel(asrfe_ctx* ctx, const void* pcm, int32_t pcm_dtype, int64_t pcm_stride, const int32_t* pcm_lens,
                            int32_t batch, int32_t t_max, float* decibel, void* stream);

/* asrfe_encoder_forward for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): no workspace argument.  No counterpart in the reference.  The scratch is a
 * device array inside the library, ASRFE_STATIC_SCRATCH_BYTES per GPU for the life of the process; a [batch, tp]
 * call needs asrfe_encoder_workspace_bytes(ctx, batch, tp) of it (about 8.3 KiB per row: up to ~3970 rows, e.g.
 * 7 clips of 30 s or 79 of 3 s) and larger calls are refused with ASRFE_ERR_WORKSPACE.  The call allocates nothing
 * and only enqueues kernels on `stream`.  All calls on one GPU share the array: calls that overlap on one GPU
 * (different streams without ordering, or threads) race; calls ordered on one stream do not. */
#define ASRFE_STATIC_SCRATCH_BYTES (32ll << 20)
int32_t asrfe_encoder_forward_static(asrfe_ctx* ctx, const float* feats, const int32_t* lens, int32_t batch,
                                     int32_t tp, void* out, int32_t out_dtype, void* stream);

/* Host-only table builders (double precision, no device needed): the Kaldi mel banks over bins
 * 0 .. n_fft/2 - 1 as [n_mels][n_fft/2], and the analysis window [frame_length].  These are the
 * tables asrfe_create uploads; they stand in for torchaudio's get_mel_banks / _feature_window_function
 * underneath the WavFrontend call sites cited above and let CPU-only tests check them. */
int32_t asrfe_host_mel_banks(const asrfe_config* cfg, double* out);
int32_t asrfe_host_window(const asrfe_config* cfg, double* out);

/* Measurement support (no counterpart in the reference).  Between asrfe_profile_begin and
 * asrfe_profile_end every kernel the encoder / frontend calls enqueue is bracketed by a pair of HIP
 * events on the call's stream; asrfe_profile_end waits for them and returns, per kernel kind, the
 * number of launches and the summed launch duration in milliseconds.  Events are created in
 * asrfe_profile_begin (max_launches pairs), never in the compute calls. */
#define ASRFE_PROF_KINDS 12
typedef struct asrfe_profile {
    int32_t launches[ASRFE_PROF_KINDS];
    float total_ms[ASRFE_PROF_KINDS];
} asrfe_profile;
/* kinds: 0 fbank, 1 frontend(fused), 2 lfr_cmvn, 3 pe_table, 4 layernorm, 5 gemm_qkv, 6 fsmn,
 *        7 attention, 8 gemm_out, 9 gemm_ffn1, 10 gemm_ffn2, 11 post_attn (output projection + FSMN + residual, norm2, feed-forward, next norm1) */
int32_t asrfe_profile_begin(asrfe_ctx* ctx, int32_t max_launches);
int32_t asrfe_profile_end(asrfe_ctx* ctx, asrfe_profile* out);

/* Kernel-level test hook (no counterpart in the reference): one launch of the encoder's GEMM,
 * out[M][N] = epilogue(A[M][K] * W[N][K]^T + bias), 16-bit operands in the context's enc_compute_dtype.
 * epi: 0 = 16-bit out, 1 = 16-bit out with ReLU, 2 = f32 out + add16 (16-bit, optional) + add32 (f32, optional).
 * N must be a multiple of 256 and K of 128; all pointers are device pointers.  flags bit 0 skips the stores
 * (timing experiments only); bits 8-9 force the output tile height (1: 256 rows, 2: 128 rows; 0: the launch
 * takes 128 rows where tiles of 256 would leave compute units idle; the result is the same bit for bit). */
int32_t asrfe_debug_gemm(asrfe_ctx* ctx, const void* a, int32_t lda, const void* w, int32_t ldw, const float* bias,
                         int32_t m, int32_t n, int32_t k, void* out, int32_t ldo, const void* add16,
                         const float* add32, int32_t epi, int32_t flags, void* stream);

/* Kernel-level test hook (no counterpart in the reference): one launch of the fused post-attention kernel of
 * encoder layer `layer` on m rows: att, fsmn [m][512] in the 16-bit compute dtype, x [m][512] float32 or NULL,
 *     x_mid = x + fsmn + att Wo^T + b_o,   x_out = x_mid + W2 relu(W1 norm2(x_mid) + b1) + b2   (x_out may alias x)
 * and, when y is not NULL, y [m][512] = x_out normalised to zero mean / unit variance per row, in the compute
 * dtype (the next norm1's scale and shift are folded into the next layer's QKV weights).  When qkv is not NULL
 * (y must be NULL then, and `layer` must not be the last) the kernel also projects the next layer's
 * qkv [m][1536] = norm1(x_out) Wqkv^T + b_qkv in the compute dtype, as asrfe_encoder_forward has it do.
 * flags bit 9 (512): x is given in the order asrfe_encoder_forward keeps the residual stream in between two launches
 * of this kernel, and x_out is written in it when qkv is set (row-major otherwise; y must be NULL): buffers of whole
 * 128-row tiles.  The order is that of the kernel form the context was created for.  Default (v_mfma_f32_16x16x32):
 * per 32-row group [block 16][a 2][mh 2][q 4][c 16][4 floats] with the 4 floats = columns 32 block + 16 a + 4 q .. + 3 of
 * row 16 mh + c; that form serves only calls with bit 9 set and y == NULL (others return ASRFE_ERR_BAD_ARG).
 * ASRFE_MFMA16=0 (or a plan with a row-major stream) at asrfe_create selects the v_mfma_f32_32x32x16 form: per 32-row
 * group [tile 16][quad 4][h 2][row 32][4 floats] with the 4 floats = columns 32 tile + 8 quad + 4 h .. + 3 of the row.
 * The other flag bits are timing experiments that produce wrong results. */
int32_t asrfe_debug_post_attn(asrfe_ctx* ctx, int32_t layer, const void* att, const void* fsmn, const float* x,
                              float* x_out, void* y, void* qkv, int32_t m, int32_t flags, void* stream);

/* Kernel-level test hook (no counterpart in the reference): one launch of the attention + FSMN kernel of encoder
 * layer `layer` on qkv [batch * tp][1536] (16-bit compute dtype; q | k | v, 4 heads of 128) with lens [batch]:
 * att [batch * tp][512] = softmax(q k^T / sqrt(128), keys < lens[b]) v per head, and fsmn [batch * tp][512] = the FSMN
 * memory block of v (11 taps, rows >= lens[b] zero).  flags != 0 are timing experiments that produce wrong results. */
int32_t asrfe_debug_attention(asrfe_ctx* ctx, int32_t layer, const void* qkv, const int32_t* lens, int32_t batch,
                              int32_t tp, void* att, void* fsmn, int32_t flags, void* stream);

/* Text of the last error on this context (or of the last failed asrfe_create when ctx is NULL). */
const char* asrfe_last_error(const asrfe_ctx* ctx);
const char* asrfe_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ASRFE_H */
