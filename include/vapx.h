/*
 * vapx.h — C ABI of libvapx.so: MI355X-native many-stream engine for the Realtime-VAP
 * streaming forward pass (CPC encoder -> LSTM -> downsample -> per-stream context ring ->
 * 1 self + 3 self/cross GPT layers -> VAP head -> p_now / p_future / VAD).
 *
 * The reference (inokoj/VAP-Realtime) has NO FFI / plugin interface for this path: it is plain
 * Python attribute calls.  Each entry point below therefore cites the reference
 * Python call it stands in for; the ctypes binding a maintainer would add on the reference side is
 * shown in INTEGRATION.md and implemented in vap-realtime_amd/engine.py.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (hipStream_t travels as void*).
 *   - every function returns 0 on success or a negative VAPX_E_* code; no exceptions cross the
 *     ABI; vapx_last_error() returns a human-readable message for the last failure on a handle
 *     (or for a failed vapx_create when called with NULL).
 *   - ownership: the caller owns audio / output / blob memory (the blob is copied at create);
 *     the library owns device weights, per-stream state and scratch.
 *   - threading: calls on one handle must be serialised by the caller (the reference runs
 *     inference on a single thread, rvap/vap_main/vap_main.py:520-521).  Work is ordered on the
 *     HIP stream passed in; device outputs are valid after that stream is synchronised, host
 *     outputs are valid on return.
 *   - one handle per GPU; streams (dialogues) are independent, so multi-GPU = one handle per
 *     device with the stream ids partitioned by the caller (no collective).
 */
#ifndef VAPX_H_
#define VAPX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAPX_ABI_VERSION 2

/* error codes */
#define VAPX_OK 0
#define VAPX_E_INVAL (-1)    /* bad argument */
#define VAPX_E_HIP (-2)      /* HIP runtime error (see vapx_last_error) */
#define VAPX_E_NOMEM (-3)
#define VAPX_E_RANGE (-4)    /* stream id / batch size out of range */
#define VAPX_E_NODEVICE (-5) /* no gfx950 device visible */
#define VAPX_E_NUMERIC (-6)  /* host-output vapx_step only: at least one stream produced non-finite p_now / p_future / VAD / aux
                                probabilities (a poisoned LSTM / ring state, Inf audio, or |activation| >= 65504 on the
                                split-precision path).  PER STREAM, not per call: the out block is complete, every other row is
                                valid, the offending rows carry VAPX_OUT_STATUS = 1 and vapx_bad_slots() lists them; reset those
                                streams and keep serving the rest.  A NaN / Inf audio SAMPLE triggers it exactly as it poisons the
                                reference: ChannelNorm and torch.relu propagate it (encoder_components.py:64-66,103), the LSTM state
                                keeps it for good (tests/golden/poison20.npz records the unmodified reference doing so). */

/* model variants: which heads are evaluated (vap_main.py:290-307, vap_bc_main.py:272-277,
 * vap_nod_main.py:273-279) */
#define VAPX_MODE_VAP 0
#define VAPX_MODE_BC 1
#define VAPX_MODE_NOD 2

/* memory-space flags for vapx_step */
#define VAPX_AUDIO_HOST 0
#define VAPX_AUDIO_DEVICE 1
#define VAPX_OUT_HOST 0
#define VAPX_OUT_DEVICE 2
#define VAPX_IDS_DEVICE 4 /* stream_ids points to device memory (default: host) */
#define VAPX_DEFER_JOIN 8 /* with overlap groups > 1: do not make hip_stream wait for the groups at the end of the step; the
                             caller orders consumers of `out` with vapx_join.  Lets group g's next tick start while other
                             groups still finish this one.  Honoured only when the whole step is device-resident
                             (VAPX_AUDIO_DEVICE | VAPX_OUT_DEVICE, ids NULL or VAPX_IDS_DEVICE) — otherwise ignored; a step
                             whose n (or group split) differs from the previous one, or that has resets pending, first
                             joins the previous tick's groups itself. */

/* vapx_config.flags */
#define VAPX_FLAG_GROUPS_MASK 0xF     /* bits 0-3: intra-tick overlap groups (0 = default 1 = none, max 8) */
#define VAPX_FLAG_MATERIALIZE_X0 64   /* copy the context window chronologically each tick ("x0" peekable); default: layer 0 reads
                                         the embedding / Q|K|V rings in place (short and long windows alike) */
#define VAPX_FLAG_SPLIT_F16 512        /* opt-in: every contraction with bounded or scalable operands — the fused flat-row blocks (FFN,
                                         projections), the attention blocks' projections, the conv / LSTM-input GEMMs, and for windows longer
                                         than 64 frames the attention itself (Q.K^T and P.V) — as fp32-accurate 3-term split products on the
                                         f16 matrix cores (x s = hi + lo; hi.hi + lo.hi + hi.lo, fp32 accumulate, power-of-two operand scales
                                         s from row / tile maxima or static weight bounds: no input can overflow f16); same 1e-4 parity bar,
                                         same error against float64 as the default fp32-MFMA path.  vapx_create refuses the flag
                                         (VAPX_E_INVAL) for a checkpoint whose transformer weights break the static bounds (|w| >= 255) */
#define VAPX_FLAG_UNFUSED_PROJ 1024    /* long windows (T > 64): attention output projections (+ residual + LN, + cross-attention query
                                         projection) as separate GEMM launches instead of riding in the fused blocks; kept for A/B tests */
#define VAPX_FLAG_SPLIT_QKV_IN_FFN 2048 /* with VAPX_FLAG_SPLIT_F16 and windows longer than 64 frames: the self-attention Q|K|V of layers 1-2 come from the
                                         previous layer's flat-row block through HBM (rounds 2-4) instead of being projected inside the attention
                                         kernel (csrc/attention_proj_f16x3.hip); kept for A/B tests */
#define VAPX_FLAG_UNFUSED_LAST_ROW 256 /* last layer's newest-row path as ten launches (gathers, M = 2B GEMMs, single-query
                                         attention) instead of the fused last_block_kernel; kept for A/B parity tests */
#define VAPX_FLAG_UNFUSED_CONV 32     /* conv2-4 as three GEMM launches (materialises "h2","h3" for vapx_peek) */
#define VAPX_FLAG_FULL_LAST_LAYER 16  /* compute every row of the last layer (default: only the newest row,
                                         which is all process_vap consumes; needed to vapx_peek "stereo2") */

/* Layout of one output row (floats).  Row stride is VAPX_OUT_STRIDE. */
#define VAPX_OUT_P_NOW 0     /* [2]  result_p_now        vap_main.py:316 */
#define VAPX_OUT_P_FUTURE 2  /* [2]  result_p_future     vap_main.py:317 */
#define VAPX_OUT_VAD 4       /* [2]  result_vad          vap_main.py:313-320 */
#define VAPX_OUT_AUX 6       /* [4]  bc: {-, p_bc_react, p_bc_emo, -}; nod: {-, short, long, long_p} */
#define VAPX_OUT_NVALID 10   /* [1]  n = rows in the context window this frame (as float) */
#define VAPX_OUT_VAD_LOGIT 11 /* [2] va_classifier outputs before the sigmoid (what the training-style forward() returns) */
#define VAPX_OUT_STATUS 13   /* [1]  0 = ok, 1 = this row's probabilities are not finite (see VAPX_E_NUMERIC); written on device,
                                      so the device-output path carries it too */
#define VAPX_OUT_LOGITS 16   /* [256] vap_head logits of the newest row  vap_main.py:290;
                                      nod mode: p_bc of rows 0..n-1 instead (vap_nod_main.py:276 quirk) */
#define VAPX_OUT_E 272       /* [2*256] this frame's embeddings e1,e2   vap_main.py:272 */
#define VAPX_OUT_STRIDE 784

typedef struct vapx_engine* vapx_handle;

typedef struct vapx_config {
  int32_t struct_size;  /* sizeof(vapx_config), for ABI evolution */
  int32_t device_id;    /* HIP device ordinal */
  int32_t frame_hz;     /* 5, 10, 20 or 50: VAPRealTime frame_rate   vap_main.py:192,219 */
  int32_t ctx_frames;   /* T = int(context_len_sec*frame_rate)        vap_main.py:221.  1 <= T <= 512 (vapx_create returns VAPX_E_INVAL beyond).
                         * Every published checkpoint fits in 256 (largest: 20 Hz x 10 s = 200 rows, README.md; BASELINE configs[2]: 50 Hz x 5 s = 250)
                         * and that is what the long-window attention kernels are tuned for; the reference's ALiBi transformer itself takes any T
                         * (modules.py:303-308), so windows of 257 .. 512 rows run too — through a plain fp32 attention kernel (K / V from L2, no
                         * on-chip tile; on the split-precision path as well) that is correct, tested against the oracle, and not tuned */
  int32_t max_streams;  /* stream slots whose state lives in HBM */
  int32_t max_batch;    /* max streams per vapx_step call (sizes scratch) */
  int32_t mode;         /* VAPX_MODE_* */
  int32_t flags;        /* VAPX_FLAG_* */
} vapx_config;

/* Number of floats the weight blob must have for a frame rate (layout:
 * vap-realtime_amd/weights.py:blob_layout).  Returns 0 for an unsupported rate. */
size_t vapx_blob_floats(int32_t frame_hz);

/* Stands in for VAPRealTime.__init__ (vap_main.py:192-247): builds device weights from the packed
 * blob (host memory, fp32), allocates per-stream state (context ring, LSTM h/c, 320-sample
 * carry) zero-initialised, and scratch for max_batch streams. */
int vapx_create(const vapx_config* cfg, const float* weights_blob, size_t n_floats, vapx_handle* out);

void vapx_destroy(vapx_handle h);

/* Stands in for VAPRealTime.process_vap (vap_main.py:249-335) for n streams at once.
 *   stream_ids : n ids in [0,max_streams), all distinct (host ids are checked: VAPX_E_RANGE / VAPX_E_INVAL for a duplicate;
 *                device ids are the caller's responsibility); NULL means 0..n-1.
 *   audio      : fp32 [n][2][samples_per_ch].  samples_per_ch == hop (=16000/frame_hz): new
 *                samples only, the engine prepends its own 320-sample carry like proc_serv_in
 *                (vap_main.py:397-409).  samples_per_ch == hop+320: a complete frame exactly as
 *                process_vap receives x1/x2 (carry supplied by the caller, vap_offline.py:51-61);
 *                the engine's carry is then set to the frame's last 320 samples.
 *   out        : fp32 [n][VAPX_OUT_STRIDE].
 *   flags      : VAPX_AUDIO_* | VAPX_OUT_* | VAPX_IDS_DEVICE | VAPX_DEFER_JOIN
 *   hip_stream : hipStream_t to order the work on (NULL = default stream).
 * Host audio / out in vapx_host_alloc memory is copied with true async DMA; pageable memory is staged through the engine's
 * own pinned buffers (one extra memcpy each way). */
int vapx_step(vapx_handle h, int32_t n, const int32_t* stream_ids, const float* audio,
              int32_t samples_per_ch, float* out, int32_t flags, void* hip_stream);

/* Batch slots (row indices of `out`) of the latest host-output vapx_step whose results were not finite; returns their number
 * (writes at most max_slots of them; slots may be NULL to just count). */
int32_t vapx_bad_slots(vapx_handle h, int32_t* slots, int32_t max_slots);

/* Page-locked host memory for audio / out blocks (hipHostMalloc): vapx_step then DMAs straight from / into it.
 * NULL on failure.  Not tied to a handle. */
void* vapx_host_alloc(size_t bytes);
void vapx_host_free(void* p);

/* Make hip_stream wait for every overlap group of the latest vapx_step (see VAPX_DEFER_JOIN). */
int vapx_join(vapx_handle h, void* hip_stream);

/* Multi-model serving on one shared CPC trunk.  The bc / nod / vap programs of the reference
 * (rvap/vap_bc/vap_bc_main.py, rvap/vap_nod/vap_nod_main.py, rvap/vap_main/vap_main.py) each load the SAME cpc_model
 * file for the CNN + LSTM (vap_main.py:199-201 skips the state dict's encoder.* keys) and differ only in the
 * downsample, the transformer and the heads.  After vapx_attach_trunk(follower, leader) the follower never runs the
 * encoder: each tick, step the leader with the audio, then step every follower with audio == NULL (same n, same
 * hip_stream; stream_ids is ignored, the leader's are used).  The follower applies its own downsample to the
 * leader's LSTM outputs and runs its own rings / transformer / heads.  Requirements: both engines freshly created
 * (no step yet), same device, frame_hz, ctx_frames, max_streams, max_batch, and bit-identical CPC weights in the two
 * blobs.  A leader can have several followers; vapx_reset_stream on the leader resets them too (and is refused on
 * a follower); LSTM / carry state import / export goes through the leader, ring state through each engine.
 * Destroy followers before their leader. */
int vapx_attach_trunk(vapx_handle follower, vapx_handle leader);

/* Zero one stream's state (context ring fill, LSTM h/c, carry).  The reference never resets
 * model state on reconnect (vap_main.py:368-369 re-zeroes only the carry); this is the explicit
 * equivalent of constructing a fresh VAPRealTime for that stream.
 * Stream-ordered and free for everybody else: the call only queues the request (no device work, no synchronisation); the
 * next vapx_step / vapx_encode_audio applies it on its HIP stream before touching any state, and vapx_get_state /
 * vapx_set_state / vapx_peek apply it before they look. */
int vapx_reset_stream(vapx_handle h, int32_t stream_id);

/* Zero only the 320-sample carry of a stream: exactly what the reference does when an input client (re)connects
 * (vap_main.py:368-369: current_x1 / current_x2 restart from zeros, LSTM and context are kept).  Queued and stream-ordered
 * like vapx_reset_stream. */
int vapx_reset_carry(vapx_handle h, int32_t stream_id);

/* The configuration a handle was created with. */
int vapx_get_config(vapx_handle h, vapx_config* out);

/* State export / import for one stream (tests, migration between GPUs).  Host pointers, any may
 * be NULL to skip.  ring: [2][T][256] oldest->newest, rows >= n_frames undefined;
 * lstm: [2 ch][2 (h,c)][256]; carry: [2][320]. */
int vapx_get_state(vapx_handle h, int32_t stream_id, float* ring, int32_t* n_frames, float* lstm, float* carry);
int vapx_set_state(vapx_handle h, int32_t stream_id, const float* ring, int32_t n_frames, const float* lstm,
                   const float* carry);

/* Stage-level entry points: the model-attribute surface process_vap calls (SURVEY.md §8b level 1).
 * All pointers are DEVICE memory, fp32, contiguous.  They use the handle's weights and scratch
 * but touch no stream state except vapx_encode_audio (LSTM h/c of the given stream ids). */

/* VapGPT.encode_audio (vap_main.py:175-180): frames [n][2][hop+320] -> e [n][2][256].
 * Stateful: advances the LSTM state of stream_ids (host pointer, NULL = 0..n-1). */
int vapx_encode_audio(vapx_handle h, int32_t n, const int32_t* stream_ids, const float* frames, float* e,
                      void* hip_stream);

/* GPT.forward (ar_channel, vap_main.py:285-286) then GPTStereo.forward (ar, :287) on explicit
 * context tensors x [n][2][rows][256] (rows <= ctx_frames).  Any output pointer may be NULL.
 *   o    [n][2][rows][256]  ar_channel(x_c)["x"]
 *   x12  [n][2][rows][256]  ar(...)["x1"], ["x2"]
 *   comb [n][rows][256]     ar(...)["x"]  (Combinator output, all rows)
 *   stage 0: both; 1: ar_channel only (x -> o); 2: ar only (x is then o1,o2 -> x12, comb). */
int vapx_transformer(vapx_handle h, int32_t n, int32_t rows, const float* x, float* o, float* x12, float* comb,
                     int32_t stage, void* hip_stream);

/* The head callables process_vap applies to tensors of any row count (vap_main.py:290-307); device pointers, rows x 256 fp32:
 *   vapx_vap_head       logits = vap_head(x)            Linear(256, 256) + bias   (vap_main.py:131,290)   [rows][256]
 *   vapx_va_classifier  y = va_classifier(x)            Linear(256, 1) + bias, BEFORE the sigmoid (:142,292-293)   [rows]
 *   vapx_aux_head       y = bc_head(x) / nod_head(x)    the bc / nod variants' extra Linear heads as vap_realtime/model.py:197,217-218
 *                       applies them to out["x"] (vap_realtime/vap_models.py:220 Linear(256, 3); :328-329 Linear(256, 4) + Linear(256, 1));
 *                       which = VAPX_AUX_BC_HEAD | VAPX_AUX_NOD_HEAD; raw outputs BEFORE softmax / sigmoid   [rows][3 | 1 | 4]
 *   vapx_softmax256     probs = logits.softmax(-1)      (:295)
 *   vapx_aggregate      objective.probs_next_speaker_aggregate(probs, from_bin, to_bin) (objective.py:186-206)   [rows][2] */
int vapx_vap_head(vapx_handle h, int64_t rows, const float* x, float* logits, void* hip_stream);
int vapx_va_classifier(vapx_handle h, int64_t rows, const float* x, float* y, void* hip_stream);
#define VAPX_AUX_BC_HEAD 0
#define VAPX_AUX_NOD_HEAD 1
int vapx_aux_head(vapx_handle h, int32_t which, int64_t rows, const float* x, float* y, void* hip_stream);
int vapx_softmax256(int64_t rows, const float* x, float* y, void* hip_stream);
int vapx_aggregate(int64_t rows, const float* probs, int32_t from_bin, int32_t to_bin, float* out, void* hip_stream);

/* Copy an internal scratch buffer of the LAST vapx_step to the host (per-layer parity tests).
 * name: "h0".."h3","z","lstm_out","e","x0","o","stereo0".."stereo2"; returns the number of
 * floats written (<= max_floats) or a negative error.  "h2"/"h3" need VAPX_FLAG_UNFUSED_CONV and
 * "stereo2" needs VAPX_FLAG_FULL_LAST_LAYER (the default path never materialises them).
 * Debug: "guard_violations" writes one float, the number of canary bytes overwritten around the process's engine allocations (-1 unless
 * the library was started with VAPX_GUARD_ZONES=1; INTEGRATION.md "Debug and experiment knobs"). */
int64_t vapx_peek(vapx_handle h, const char* name, float* dst, size_t max_floats);

/* Standalone fp32-MFMA GEMM used by every dense contraction of the path (kernel unit tests):
 * C[M][N] = epilogue(A[M][K] . W[N][K]^T); device pointers.  epi: 0 store(+bias) 1 gelu
 * 2 +resid 3 +resid & LN copy -> C2  4 bias+ChannelNorm+ReLU  5 bias+LN+GELU.  N must be a
 * multiple of 256, K a multiple of 32; epilogues 3,4,5 need N == 256. */
int vapx_gemm(void* hip_stream, int32_t M, int32_t N, int32_t K, const float* A, const float* W, float* C,
              int32_t epi, const float* bias, const float* gamma, const float* beta, const float* resid,
              float* C2, int32_t tile_rows);

/* Per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * class ids: 0..4 = the GEMM by epilogue (vapx_gemm's epi 0..4), 5 fused conv2-4 tail, 6 fused FFN block,
 * 7 last-row path of the final layer, 8 conv0, 9 lstm,
 * 10 ring gather+LN, 11 attention, 12 heads, 13 the GEMM with epilogue 5 (bias + LayerNorm + GELU: a trunk follower's
 * downsample, nod's Combinator on all rows), 14 the long-window mode-2 flat-row block (attention output projection + ln_src_attn + cross-attention
 * query projection; the FFN block proper stays class 6).  enable(mask) selects classes (0 = off);
 * read() synchronises the device, sums the elapsed time and launch count per class since the
 * last read into total_ms[n_classes] / launches[n_classes], and recycles the events. */
#define VAPX_PROF_CLASSES 15
int vapx_profile_enable(vapx_handle h, uint32_t class_mask);
int vapx_profile_read(vapx_handle h, double* total_ms, int64_t* launches, int32_t n_classes);

/* ------------------------------------------------------------------------------------------------------------------
 * Native many-stream TCP front-end with the reference's packet framing.
 *
 * Stands in for proc_serv_in / proc_serv_out / proc_serv_out_dist (rvap/vap_main/vap_main.py:338-457) and the codec
 * rvap/common/util.py:52-237, for MANY dialogues in one process: the reference accepts exactly one input client
 * (`s.listen(1)`, :360-366) and decodes every sample with a Python struct.unpack.  Here: epoll receive threads decode the
 * 2560-byte packets (160 x {f64 ch1, f64 ch2}, little-endian) straight into page-locked staging (f64 -> f32 cast as
 * vap_main.py:266-270; optional gain multiplied in float64 first, :393-395), a tick thread steps every stream whose frame is
 * complete (vapx_step, host in / host out), and sender threads write the length-prefixed result packets
 * (u32 len | f64 t | u32 n | x1 | u32 n | x2 | u32 2 | p_now | u32 2 | p_future | u32 2 | vad, util.py:122-143; bc / nod variants
 * :193-237) byte-identical to the reference codec.
 *   - every connection accepted on port_in becomes a stream (lowest free slot); the 320-sample carry lives on the device
 *     and is re-zeroed for a new connection like vap_main.py:368-369 (reset_on_connect additionally clears the LSTM /
 *     context state, which the reference keeps);
 *   - every connection accepted on port_out is attached to the stream with the fewest listeners (lowest index first), i.e.
 *     the k-th output connection hears the k-th input stream; broadcast = 1 sends every result to every output
 *     connection (the reference's behaviour; default for a 1-stream engine);
 *   - a tick runs when min_batch streams are ready (0: all connected ones), or max_wait_us after the first became ready
 *     (ragged batches: only the ready streams are stepped), but never sooner than the pacing rule allows (target_util_pct:
 *     small back-to-back batches would keep the GPU 100 % busy at its least efficient operating point);
 *   - output sockets are non-blocking like the reference's (:346-347): a listener that cannot take a whole packet is dropped;
 *   - a stream whose results are not finite (VAPX_OUT_STATUS) is reset and skipped for that tick, the others are served.
 * The engine handle must outlive the front-end and must not be stepped by anyone else while it runs. */
typedef struct vapx_ingest* vapx_ingest_handle;

typedef struct vapx_ingest_config {
  int32_t struct_size;      /* sizeof(vapx_ingest_config) */
  int32_t port_in;          /* 50007 in the reference (vap_main.py:470); 0 = ephemeral, see vapx_ingest_ports; -1 (both ports) = passive
                               shard of a vapx_frontdoor: no listening sockets of its own */
  int32_t port_out;         /* 50008 */
  int32_t rx_threads;       /* 0 = 2 */
  int32_t tx_threads;       /* 0 = 2 */
  int32_t max_wait_us;      /* 0 = 2000 */
  int32_t min_batch;        /* 0 = every connected stream */
  int32_t reset_on_connect; /* 1 = a new input connection starts from a fresh stream state */
  int32_t broadcast;        /* -1 = auto (1 for a single-stream engine), 0, 1 */
  int32_t bind_any;         /* 0 = 127.0.0.1 like the reference, 1 = 0.0.0.0 */
  double gain;              /* audio_gain, 1.0 = off */
  int32_t target_util_pct;  /* pacing: the next tick starts no earlier than (previous tick's start + its duration * 100 / pct), so the
                               engine stays at most pct % busy and batches grow instead of the queue (0 = 90; 100 = back-to-back) */
  int32_t flags;            /* VAPX_INGEST_* (0 = defaults) */
  /* thread placement (optional; a caller that passes the shorter struct of ABI 2 as first shipped gets 0 / 0 = no pinning): with
     cpu_count > 0 the front-end pins its threads to the cores cpu_first .. cpu_first + cpu_count - 1 (wrapping): tick (and accept) thread on
     the first, then one core per receive thread, then one per sender thread.  Give the cores of the GPU's NUMA node, and keep load
     generators / other tenants off them: with 4096 connections the kernel's softirq work for the sockets otherwise lands on whatever core a
     front-end thread happens to run on and its tail latency follows the neighbours'. */
  int32_t cpu_first;
  int32_t cpu_count;
} vapx_ingest_config;
#define VAPX_INGEST_CORE_SET 2      /* with cpu_count > 0: every front-end thread may run on ANY core of the range (one affinity set) instead of
                                       one core each: keeps other processes' work off the range without nailing a thread to a core that the
                                       kernel then borrows for softirq work */
#define VAPX_INGEST_KEEP_NOFILE 1   /* never touch RLIMIT_NOFILE.  Default (flag clear): vapx_ingest_open* needs one descriptor per dialogue
                                       and one per listener; if the process's SOFT limit is below 2 x streams + 256 it is raised towards the
                                       hard limit with setrlimit() — a process-wide change the host should know about (select()-based code
                                       with FD_SETSIZE tables must not be handed descriptors >= 1024) */

typedef struct vapx_ingest_stats {
  int64_t frames_done;      /* stream-frames stepped and answered */
  int64_t ticks;            /* vapx_step calls */
  int64_t rx_bytes, tx_bytes;
  int64_t in_connections, out_connections;   /* currently open */
  int64_t dropped_listeners;                 /* output connections closed because they could not take a packet */
  int64_t numeric_resets;                    /* streams reset after non-finite results */
  int64_t overruns;                          /* frames a sender delivered faster than the engine consumed (input paused) */
  double mean_batch;                         /* streams per tick */
  double lat_mean_ms, lat_p50_ms, lat_p99_ms, lat_max_ms;   /* frame complete on the host -> result packet handed to the kernel */
  double step_mean_ms;                       /* time inside vapx_step per tick */
} vapx_ingest_stats;

int vapx_ingest_open(vapx_handle engine, const vapx_ingest_config* cfg, vapx_ingest_handle* out);
/* The same front-end over a caller-supplied step function instead of an engine (host-logic tests without a GPU):
 * step(user, n, stream_ids, audio[n][2][hop], out[n][VAPX_OUT_STRIDE]) returns 0 or a negative VAPX_E_* code;
 * reset(user, stream_id) may be NULL. */
typedef int (*vapx_ingest_step_fn)(void* user, int32_t n, const int32_t* stream_ids, const float* audio, float* out);
typedef void (*vapx_ingest_reset_fn)(void* user, int32_t stream_id);
int vapx_ingest_open_fn(vapx_ingest_step_fn step, vapx_ingest_reset_fn reset, void* user, int32_t n_streams, int32_t max_batch,
                        int32_t frame_hz, int32_t mode, const vapx_ingest_config* cfg, vapx_ingest_handle* out);
int vapx_ingest_ports(vapx_ingest_handle g, int32_t* port_in, int32_t* port_out);   /* (0, 0) for a passive shard */
int vapx_ingest_stats_read(vapx_ingest_handle g, vapx_ingest_stats* out, int32_t reset_latency_window);
/* Exact server-side count of late answers in the current latency window (since the last stats_read(.., 1)): result packets handed to the
 * kernel more than 10 ms after their frame was complete on the host — the north-star bound — and the packets of the window. */
int vapx_ingest_late_read(vapx_ingest_handle g, int64_t* over_10ms, int64_t* answered);
void vapx_ingest_close(vapx_ingest_handle g);

/* ---- one front door for N GPUs -----------------------------------------------------------------------------------------------------
 * The reference serves ONE port pair (proc_serv_in / proc_serv_out_dist bind port_num_in / port_num_out, vap_main.py:338-366,470-471).
 * For N GPUs in one process: create one engine per device (vapx_create, device_id = r), open one PASSIVE front-end per engine
 * (vapx_ingest_config.port_in = port_out = -1: it listens on nothing) and put them behind vapx_frontdoor_open, which owns the single
 * port pair and hands every accepted connection to a shard.  Dialogue slots are numbered globally g = local_slot * N + shard:
 *   - an input connection takes the lowest free global slot (GPUs fill evenly; a dialogue that reconnects while its slot is still the
 *     lowest free one returns to the GPU that holds its state);
 *   - the k-th output connection hears the k-th dialogue (fewest listeners, lowest global slot), as with a single front-end.
 * All shards must have the same frame rate and mode.  Close the front door first, then the shards, then the engines. */
typedef struct vapx_frontdoor* vapx_frontdoor_handle;
int vapx_frontdoor_open(vapx_ingest_handle* shards, int32_t n_shards, int32_t port_in, int32_t port_out, int32_t bind_any,
                        vapx_frontdoor_handle* out);
/* The same door with every shard in a process OF ITS OWN (one process per GPU: the process that owns the engine also owns its front-end's
 * threads and descriptors - a container's RLIMIT_NOFILE of 20 000 holds ~9 800 dialogues, BASELINE config 4 has 32 768).  Door and worker are
 * joined by one AF_UNIX / SOCK_SEQPACKET socket pair (`link`), created by whoever starts the workers:
 *   worker process:  vapx_create, vapx_ingest_open (passive: port_in = port_out = -1), vapx_ingest_attach_link(front_end, its end of the link)
 *   door process:    vapx_frontdoor_open_links(the other ends, n, port_in, port_out, ...) - it waits for every worker's greeting (slots, frame
 *                    rate, mode), then owns the reference's ONE port pair (vap_main.py:338-366,470-471)
 * The door keeps a mirror of every shard's slot / listener occupancy and is the only allocator: it names the slot, passes the accepted socket to
 * the worker (SCM_RIGHTS) and closes its own copy; workers report released slots and dropped listeners back over the link.  Placement is the
 * in-process door's: lowest free global slot g = local_slot * N + shard for inputs, fewest listeners / lowest global slot for outputs.  A worker
 * that dies takes its dialogues with it and receives no new ones; the others keep serving.  The link descriptors stay the caller's: close them
 * after vapx_frontdoor_close / vapx_ingest_close. */
int vapx_ingest_attach_link(vapx_ingest_handle g, int32_t link_fd);
int vapx_frontdoor_open_links(const int32_t* link_fds, int32_t n_links, int32_t port_in, int32_t port_out, int32_t bind_any,
                              vapx_frontdoor_handle* out);
int vapx_frontdoor_ports(vapx_frontdoor_handle d, int32_t* port_in, int32_t* port_out);
int vapx_frontdoor_counts(vapx_frontdoor_handle d, int64_t* accepted_in, int64_t* accepted_out, int64_t* refused);
void vapx_frontdoor_close(vapx_frontdoor_handle d);

/* The wire codec on its own (byte-parity tests against the reference's util.py output, tests/golden/wire.npz).
 * decode: n_bytes (a multiple of 16) of input packets -> n_bytes / 16 samples per channel as the engine sees them (f32) and as
 * the result packet echoes them (f64, gain applied); either destination may be NULL.  Returns the sample count or < 0.
 * encode: one result packet INCLUDING the 4-byte length prefix for an output row of vapx_step; x1 / x2 = the frame's n
 * echoed samples; returns the packet size (or the size needed if dst is NULL / cap too small, nothing written). */
int64_t vapx_wire_decode_input(const uint8_t* bytes, size_t n_bytes, double gain, float* x1_f32, float* x2_f32, double* x1_f64,
                               double* x2_f64);
int64_t vapx_wire_encode_result(int32_t mode, double t, const double* x1, const double* x2, int32_t n, const float* out_row,
                                uint8_t* dst, size_t cap);

const char* vapx_last_error(vapx_handle h);
int32_t vapx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VAPX_H_ */
