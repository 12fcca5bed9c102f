/*
 * thewhisper.h - C ABI of libthewhisper_gfx950.so, the MI355X (gfx950 / CDNA4) Whisper hot path.
 *
 * The reference (TheStageAI/TheWhisper) has NO C ABI / FFI for this path: its NVIDIA backend hands
 * the whole computation to a Python model object (Hugging Face `WhisperForConditionalGeneration`, or
 * the closed TensorRT wheel) under `ASRPipeline` (R:thestage_speechkit/nvidia/asr_pipeline.py:47-60),
 * and its Apple backend swaps encoder/decoder modules under the same object
 * (R:thestage_speechkit/apple/model.py:601-614).  Each entry point below therefore cites the
 * Python-level interface it replaces (R: = /root/reference, HF: = transformers 5.15.0).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0 on success
 *    and a negative TW_E* code on failure; tw_last_error() returns a human-readable message.
 *  - "dev" pointers are HIP device pointers owned by the caller (torch); the library never frees
 *    them and never retains them after the call returns, except nothing: weights are COPIED
 *    (converted / repacked) into the context at tw_load_weight time.
 *  - "host" pointers are ordinary host memory.  Calls that return host data synchronise the
 *    stream; all other calls only enqueue work on `stream` (a hipStream_t passed as void*; pass
 *    torch's current stream so ordering with torch ops is automatic; NULL = default stream).
 *  - A context is bound to one device, owns its workspace / KV arenas (allocated in tw_create, freed
 *    in tw_destroy) and is NOT thread-safe: one context per (GPU, worker thread).  Every call makes the
 *    context's device current for its duration and restores the calling thread's device on return.
 *  - Batch entries ("streams") of one call occupy slots 0..B-1 of the context, B <= max_batch.
 */
#ifndef THEWHISPER_H
#define THEWHISPER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TW_VERSION_STRING "thewhisper-gfx950 0.1.0"

/* element types of caller buffers and of the context's compute mode */
enum { TW_F32 = 0, TW_BF16 = 1, TW_F16 = 2,
       TW_BF16_MXFP8 = 3, /* context dtype only: bf16 activations / encoder, decoder projection weights as MXFP8 (OCP e4m3 +
                            one power-of-two scale per 32 values) AND the activations quantised the same way in registers, on
                            v_mfma_scale_f32_16x16x128_f8f6f4 ("W8A8"); fp8 cross-attention K / V caches; BASELINE config 5 */
       TW_BF16_W8A16 = 4  /* context dtype only: the same MXFP8 weights and fp8 cross K / V caches, but the weight fragments are
                            widened to bf16 in registers and contracted with the UNQUANTISED bf16 activations on the bf16 MFMA
                            ("W8A16": same bytes from HBM, no activation-quantisation error); up to 64 streams */ };

/* error codes */
enum {
  TW_OK = 0,
  TW_EINVAL = -1,   /* bad argument / unsupported configuration */
  TW_EHIP = -2,     /* HIP runtime error (message has the hipError string) */
  TW_ESTATE = -3,   /* call order violated (e.g. decode before encode / weights not finalized) */
  TW_ENOMEM = -4,
  TW_ENAME = -5     /* unknown weight name or shape mismatch */
};

#define TW_MAX_ALIGN_HEADS 32

typedef struct tw_ctx tw_ctx;

/* Model + capacity description.  Mirrors the fields of HF `WhisperConfig` that the hot path uses
 * (HF:models/whisper/configuration_whisper.py) plus the per-context capacities. */
typedef struct tw_config {
  int32_t d_model;
  int32_t enc_layers;
  int32_t dec_layers;
  int32_t heads;                 /* head_dim = d_model / heads must be 64 */
  int32_t ffn;
  int32_t vocab;
  int32_t n_mels;
  int32_t source_positions;      /* T: encoder frames per chunk = 50 * chunk_seconds (<= 1500).  When
                                    < 1500 the 1500-row positional table is linearly interpolated at
                                    load time exactly like patch_hf_model
                                    (R:thestage_speechkit/nvidia/asr_pipeline.py:15-27). */
  int32_t target_positions;      /* decoder positions, 448 */
  int32_t max_batch;             /* concurrent streams per call: 1..64 (1..16 with TW_BF16_MXFP8) */
  int32_t dtype;                 /* TW_BF16 (production), TW_F16 (the reference's streaming default dtype,
                                    R:thestage_speechkit/streaming/streaming_pipeline.py:369-370: same MFMA rate, 10 mantissa bits,
                                    activations saturate at 65504 as HF clamps its float16 hidden states), TW_F32 (strict-parity
                                    mode), TW_BF16_MXFP8 / TW_BF16_W8A16 (fp8 decoder weights) */
  int32_t n_align_heads;         /* alignment heads for word timestamps (generation_config.alignment_heads) */
  int32_t align_heads[2 * TW_MAX_ALIGN_HEADS]; /* (layer, head) pairs */
  int32_t device;                /* HIP device ordinal */
  int32_t use_graph;             /* 1: replay the decode step from a captured hipGraph */
} tw_config;

/* Options of one greedy decode (A9/A10).  Mirrors what HF's generate derives from
 * generation_config + the three Whisper logits processors
 * (HF:models/whisper/generation_whisper.py:1774-1812, HF:generation/logits_process.py:1816-2047). */
typedef struct tw_greedy_opts {
  int32_t eos_id;
  int32_t pad_id;
  int32_t max_new_tokens;
  int32_t min_new_tokens;        /* MinNewTokensLength: eos masked until this many new tokens */
  int32_t max_length;            /* prompt + new tokens cap (generation_config.max_length) */
  int32_t timestamps;            /* 1: apply WhisperTimeStampLogitsProcessor */
  int32_t no_timestamps_id;      /* timestamp_begin = no_timestamps_id + 1 */
  int32_t max_initial_timestamp_index; /* < 0: unset */
  int32_t n_begin_suppress;
  const int32_t* begin_suppress; /* host */
  int32_t n_suppress;
  const int32_t* suppress;       /* host */
  int32_t want_alignment;        /* 1: record alignment-head cross-attention rows for tw_token_timestamps */
  int32_t n_forced;              /* the last n_forced tokens of every prompt row are FORCED OUTPUT, not prompt: they count as generated
                                    (begin index = n_prompt - n_forced: max_new_tokens, min_new_tokens and the timestamp grammar see
                                    them as tokens the loop produced), and positions 0 .. n_prompt-2 are processed by a batched
                                    prefill (rows = streams x positions per launch) instead of one step each.  What it is for:
                                    SURVEY.md section 8f-3 - a streaming backend re-decodes every 0.5 s a buffer most of whose text
                                    it emitted a moment ago (R:thestage_speechkit/streaming/streaming_pipeline.py:770-796) and may
                                    force that text and decode only the tail (thewhisper_amd/streaming.py, opt-in).  0 = off. */
  int32_t n_draft;               /* the last n_draft tokens of every prompt row are a DRAFT of the output (exact form):
                                    GUESSES, e.g. what the previous 0.5 s tick of the same stream produced
                                    (R:thestage_speechkit/streaming/streaming_pipeline.py:388-435 decodes the rolling buffer from scratch
                                    every tick, :770-796).  The call returns exactly what it returns with the same prompt WITHOUT
                                    them - same ids, same alignment rows - however good the guesses are: prompt and draft are run
                                    through the decoder in launches of up to 64 rows (streams x consecutive positions) WITH the logits
                                    and the logits processors of every row; the draft is accepted up to the first position where
                                    some stream's arg-max differs from its guess, that arg-max becomes the token there, the rest of
                                    the draft is offered again behind it, and the one-token-per-step loop takes over where nothing
                                    more is confirmed.  Sound because a row's result does not depend on the other rows of its launch
                                    (one reduction order for every launch width, k_decode.hip).  Begin index = n_prompt - n_draft.
                                    Not together with n_forced.  0 = off.  tw_last_draft reports what was accepted. */
} tw_greedy_opts;

const char* tw_version(void);
/* ctx may be NULL: returns the message of the last failed tw_create on this thread. */
const char* tw_last_error(const tw_ctx* ctx);

/* Replaces: model construction under ASRPipeline.__init__
 * (R:thestage_speechkit/nvidia/asr_pipeline.py:47-60). */
int tw_create(const tw_config* cfg, tw_ctx** out);
int tw_destroy(tw_ctx* ctx);
/* A second context of the SAME model on the same device that shares `src`'s finalized weights (no copy: `src` owns them and must be
 * destroyed after the sibling) and has its own workspace, K/V arenas (for `max_batch` streams), graphs and timing events.  What it
 * is for: two stages of a serving pipeline on one GPU at the same time - the encoder stage of new chunks on a CU-masked stream
 * while the decode loop of the current pass runs (a context is not thread-safe, two contexts are independent).  No reference
 * counterpart (one pipeline object, one request at a time: R:examples/server.py:22-115). */
int tw_create_sibling(const tw_ctx* src, int32_t max_batch, tw_ctx** out);

/* Replaces: `from_pretrained` weight materialisation (R:thestage_speechkit/nvidia/asr_pipeline.py:58-60).
 * `name` is the HF state_dict key (layout table in SURVEY.md section 8b), `dev_ptr` a contiguous device tensor
 * of element type `dtype` (TW_F32/TW_BF16/TW_F16) and shape `shape[ndim]`.  The tensor is converted to
 * the context dtype and copied/repacked; the caller may free it afterwards. */
int tw_load_weight(tw_ctx* ctx, const char* name, const void* dev_ptr, int32_t dtype, int32_t ndim,
                   const int64_t* shape, void* stream);
/* Checks that every tensor arrived, interpolates the encoder positions (A0) and folds each decoder pre-LayerNorm
 * into the projection that consumes it (W <- W*gain plus two per-row vectors).  Call exactly once. */
int tw_finalize_weights(tw_ctx* ctx, void* stream);

/* A1.  Replaces: WhisperFeatureExtractor._torch_extract_fbank_features
 * (HF:models/whisper/feature_extraction_whisper.py:135-168) as called from the ASR pipeline
 * (HF:pipelines/automatic_speech_recognition.py:67-72).
 * pcm_dev: float32 [B, pcm_stride] mono 16 kHz; n_valid_host[b] (may be NULL = n_samples) samples
 * are real, the rest up to n_samples is treated as zero padding; out_dev: [B, n_mels, n_samples/160]
 * of element type out_dtype (TW_F32 or the context dtype). */
int tw_logmel(tw_ctx* ctx, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_valid_host, int32_t B,
              int32_t n_samples, void* out_dev, int32_t out_dtype, void* stream);

/* A2-A4.  Replaces: WhisperEncoder.forward (HF:models/whisper/modeling_whisper.py:540-646).
 * mel_dev: [B, n_mels, 2T] of mel_dtype (TW_F32 / TW_BF16 / TW_F16).  The encoder output is kept inside the context (slots
 * 0..B-1); if out_hidden_dev != NULL it is also written there as [B, T, d_model] in out_dtype (TW_F32 / TW_BF16). */
int tw_encode(tw_ctx* ctx, const void* mel_dev, int32_t mel_dtype, int32_t B, void* out_hidden_dev,
              int32_t out_dtype, void* stream);

/* A5.  Replaces: the lazy cross-attention K/V projection + EncoderDecoderCache.is_updated
 * (HF:models/whisper/modeling_whisper.py:312-335).  Projects the context's encoder output of slots
 * 0..B-1 into the per-layer cross K/V arenas, once per chunk. */
int tw_cross_kv(tw_ctx* ctx, int32_t B, void* stream);

/* The same two stages for slots slot0 .. slot0+B-1 of the context, leaving the other slots as they are: a serving pass is
 * then assembled from several groups of chunks (the ones that need a further Whisper seek iteration first, late arrivals
 * while those are being encoded), and ONE tw_generate_greedy runs over all of them.  There is no reference counterpart
 * (HF's generate() encodes a fixed batch, HF:models/whisper/generation_whisper.py:785-903); results per slot are
 * independent of the grouping.  tw_encode / tw_cross_kv are the slot0 = 0 forms and reset the number of filled slots. */
int tw_encode_at(tw_ctx* ctx, const void* mel_dev, int32_t mel_dtype, int32_t B, int32_t slot0, void* stream);
int tw_cross_kv_at(tw_ctx* ctx, int32_t B, int32_t slot0, void* stream);

/* Cross K/V of slots src_slot0 .. src_slot0+B-1 of `src` become slots dst_slot0 .. dst_slot0+B-1 of `dst` (device-to-device copies
 * on `stream`; both contexts: same model dimensions, same source_positions, same dtype, same device; dst_slot0 must be the
 * number of slots `dst` holds, like tw_cross_kv_at).  `src_stream` = the stream `src`'s encoder stage was enqueued on (NULL = its
 * default): the copies are ordered after everything enqueued there so far, and whatever is enqueued there next after the copies
 * (so the source may refill the slots at once).  The caller must not run another call on `src` concurrently with this one.  What it is for: a serving loop encodes the chunks that ARRIVE while a pass is
 * decoding on a second context of the same weights (its own workspace and arenas, a CU-masked stream), and the next pass adopts
 * them - their encoder stage is then hidden under the previous pass's decode loop (thewhisper_amd/serving.py).  No reference
 * counterpart (R:examples/server.py:22-115 processes one request at a time).  Results are the ones tw_encode_at + tw_cross_kv_at
 * on `dst` would have produced: same kernels, same weights. */
int tw_adopt_cross_kv(tw_ctx* dst, int32_t dst_slot0, tw_ctx* src, int32_t src_slot0, int32_t B, void* stream, void* src_stream);

/* A6-A8.  Replaces: WhisperDecoder.forward + proj_out for ONE new token per stream
 * (HF:models/whisper/modeling_whisper.py:649-795, :1080).  tw_decoder_reset rewinds the self-attention
 * cache to position 0.  ids_host: int32 [B].  logits_dev: float32 [B, vocab] or NULL. */
int tw_decoder_reset(tw_ctx* ctx, int32_t B, void* stream);
int tw_decode_step(tw_ctx* ctx, int32_t B, const int32_t* ids_host, float* logits_dev, void* stream);

/* A9+A10.  Replaces: GenerationMixin._sample (HF:generation/utils.py:2783-2946) for num_beams=1,
 * do_sample=False with Whisper's logits processors, as invoked by
 * WhisperGenerationMixin.generate_with_fallback (HF:models/whisper/generation_whisper.py:1027).
 * prompt_host: int32 [B, n_prompt]; out_ids_host: int32 [B, max_length] receives prompt + generated
 * tokens padded with pad_id; out_len_host[0] = common sequence length (HF pads finished rows).
 * Requires tw_encode + tw_cross_kv for the same B. */
int tw_generate_greedy(tw_ctx* ctx, int32_t B, const int32_t* prompt_host, int32_t n_prompt,
                       const tw_greedy_opts* opts, int32_t* out_ids_host, int32_t* out_len_host, void* stream);

/* Of the most recent tw_generate_greedy with n_draft > 0 (all streams together): draft tokens offered, draft tokens confirmed, rows-mode
 * launches and verify rounds it took.  Any pointer may be NULL.  No reference counterpart (see tw_greedy_opts::n_draft). */
int tw_last_draft(tw_ctx* ctx, int32_t* offered, int32_t* accepted, int32_t* launches, int32_t* rounds);

/* A11.  Replaces: _extract_token_timestamps + _median_filter + _dynamic_time_warping
 * (HF:models/whisper/generation_whisper.py:241-381, :43-61, :64-115) on the alignment rows recorded by the
 * last tw_generate_greedy(want_alignment=1).  num_frames_host[b] = valid mel frames: the time columns of row b are
 * cropped ONCE with Python-slice semantics `[: num_frames // 2]` (floor division; a negative bound counts from the end;
 * nothing left = HF's DTW on the empty matrix: every generated token at -1 * time_precision).  HF itself applies that
 * slice once or twice depending on the type and uniformity of its `num_frames` argument (:310-330, :357-359); which
 * bound reproduces HF's result for a given generate() batch is the host mirror's business
 * (thewhisper_amd/shortform.py::hf_kept_columns).  out_ts_host: float32 [B, seq_len] seconds (seq_len = out_len of the greedy call). */
int tw_token_timestamps(tw_ctx* ctx, int32_t B, int32_t n_prompt, int32_t seq_len, const int32_t* num_frames_host,
                        double time_precision, float* out_ts_host, void* stream);
/* Debug/parity access: copy the recorded alignment rows [B, n_align_heads, n_rows, T] (float32) to host. */
int tw_get_alignment(tw_ctx* ctx, int32_t B, int32_t n_rows, float* out_host, void* stream);

/* Per-stage device timings (milliseconds, HIP events on the call's stream) of the most recent call
 * of each kind: [0]=logmel [1]=encode [2]=cross_kv [3]=greedy loop [4]=token_timestamps; also the
 * number of decode steps of the last greedy call in steps_out.  Used by bench.py for the roofline. */
int tw_last_timings(tw_ctx* ctx, float* ms_out5, int32_t* steps_out);

/* SURVEY 8f-2.  Stands where the reference calls silero-vad: `prob = vad_model(frame_512, 16000).item()` on consecutive
 * 512-sample frames with state kept between calls (R:thestage_speechkit/streaming/streaming_pipeline.py:533-538, :589-622).
 * NOT silero (its weights are not obtainable offline): an adaptive-noise-floor energy detector with the same contract, rule in
 * thewhisper_amd/csrc/k_vad.hip.  pcm_dev: float32 [B, stream_stride] (16-byte aligned), the first n_frames*512 samples of
 * every row are processed in order; state_dev: float32 [B, 2] (noise floor dB, started flag; zero-initialise to reset a
 * stream); prob_dev: float32 [B, n_frames].  Asynchronous on `stream`. */
int tw_vad_energy(int32_t device, const float* pcm_dev, int64_t stream_stride, int32_t B, int32_t n_frames, float* state_dev,
                  float* prob_dev, void* stream);

/* Host-side schedule helpers (no reference counterpart: the reference processes one request at a time,
 * R:examples/server.py:22-115).  Streams and events created by the SAME HIP runtime the library and torch use, for the
 * encoder / decoder stage overlap (thewhisper_amd/overlap.py): a stream whose kernels may only run on the compute units
 * set in `cu_mask` (n_words x 32 bits), plain events, record / wait.  Handles are hipStream_t / hipEvent_t as void*. */
int tw_stream_create_masked(int32_t device, const uint32_t* cu_mask, int32_t n_words, void** out_stream);
int tw_stream_destroy(void* stream);
int tw_stream_synchronize(void* stream);
int tw_event_create(int32_t device, void** out_event);
int tw_event_destroy(void* event);
int tw_event_record(void* event, void* stream);
int tw_stream_wait_event(void* stream, void* event);

#ifdef __cplusplus
}
#endif
#endif /* THEWHISPER_H */
