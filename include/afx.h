/*
 * afx.h -- C ABI of libafx.so, the MI355X (gfx950) native anti-spoof inference path.
 *
 * This is the drop-in boundary for the reference's model forward
 *   waveform (B,L) fp32  ->  logits (B,2) fp32   (index 1 = bonafide score)
 * i.e. what the reference obtains from `model(batch_x)` at main.py:210 and
 * trainer.py:106, for the model families of models/xlsr_aasist.py:5-177,
 * models/conformer_baseline.py:31-99 and the bare SSL extractor of
 * models/fe.py:8-40,53-99 / models/models.py:13-41.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer that is documented "device" is a
 *     HIP device pointer on the current device; `stream` is a hipStream_t passed as
 *     void* (NULL = the default stream).  Calls are asynchronous on that stream.
 *   - every function returns 0 on success, non-zero on error; afx_last_error() gives
 *     a thread-local message (mirrors the Python exceptions of the reference:
 *     ValueError text for bad layer counts etc.).
 *   - the caller owns inputs, outputs and the workspace; the handle owns only the
 *     packed weights.  A handle is re-entrant across streams as long as each
 *     concurrent call has its own workspace.
 *
 * The Python host side (real-time-deepfake-speech-detection_amd/afx/_lib.py) binds
 * exactly these symbols with ctypes; INTEGRATION.md shows the stub.
 */
#ifndef AFX_H_
#define AFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct afx_engine* afx_handle;

enum { AFX_ARCH_SSL = 0, AFX_ARCH_XLSR_AASIST = 1, AFX_ARCH_CONFORMER = 2,
       AFX_ARCH_CONFORMER_HEAD = 3 /* MyConformer alone (models/conformer_baseline.py:8-29): class token, Conformer blocks, fc5; no trunk */ };
enum { AFX_EXTRACTOR_LAYER_NORM = 0, /* XLS-R: every conv layer conv+bias -> LayerNorm(512) -> GELU (what the reference loads) */
       AFX_EXTRACTOR_GROUP_NORM = 1  /* wav2vec2-base: bias-free convs, GroupNorm(512,512) on layer 0 only, GELU */ };
enum { AFX_DT_BF16 = 0, AFX_DT_FP16 = 1, AFX_DT_FP32 = 2, /* operand type (FP32: exact mode, fp32 MFMA); accumulation is always fp32 */
       AFX_DT_FP16X3 = 3 /* split precision: activations, LayerNorms, attention in fp32 as in exact mode; every dense product on
                            the fp16 matrix pipe as xh.wh + xl.wh + xh.wl (fp16 hi / lo pairs of both operands, ~22 significant
                            bits): the unconditional-parity mode at ~1/3 of the fp16 rate instead of 1/16 */ };

typedef struct afx_config {
  int arch;          /* AFX_ARCH_* */
  int dtype;         /* AFX_DT_* */
  int n_layers;      /* transformer layers in the SSL trunk, 1..24 (models/fe.py:60-62) */
  int conf_emb;      /* Conformer: emb_size   (models/conformer_baseline.py:38) */
  int conf_heads;    /*            heads      (:39) */
  int conf_kernel;   /*            kernel_size(:40) */
  int conf_blocks;   /*            n_encoders (:41) */
  int pre_emphasis;  /* 1: apply data/preprocess.py:16-29 inside the first kernel */
  float pre_emphasis_coef;
  int extractor_mode; /* AFX_EXTRACTOR_* (fairseq extractor_mode "layer_norm" / "default"); checkpoint keys
                         feature_extractor.conv_layers.0.2.{weight,bias} instead of ....{i}.2.1.* and no conv biases */
} afx_config;

/* ---- lifecycle ---------------------------------------------------------------- */
int afx_create(const afx_config* cfg, afx_handle* out);
void afx_destroy(afx_handle h);
const char* afx_last_error(void);
const char* afx_version(void);   /* "afx <ver> (gfx950) build <id> hip <toolchain version>" */
const char* afx_build_id(void);  /* hash of the sources the library was built from (tools stamp profiles/ with it) */
/* HIP_VERSION the library was compiled with / the runtime this process resolved (hipRuntimeGetVersion); afx/_lib.py refuses a
 * different major version (no reference counterpart: the reference has no native code) */
int afx_hip_versions(int* build, int* runtime);

/* ---- weights: reference checkpoint key names, without
 * the optional "module." prefix (utils.py:13-43).  `dev_ptr` is a contiguous fp32
 * device tensor; it is copied / repacked, the caller may free it afterwards.
 * Unknown-but-harmless keys (quantizer.*, project_q.*, final_proj.*, mask_emb,
 * *.bn1.*, num_batches_tracked) return 0 and are ignored. ------------------------- */
int afx_load_weight(afx_handle h, const char* name, const float* dev_ptr, const int64_t* shape, int ndim,
                    void* stream);
/* fold BatchNorms / weight-norm, check that every required tensor arrived */
int afx_finalize(afx_handle h, void* stream);

/* ---- forward ------------------------------------------------------------------ */
int afx_num_frames(int n_samples);                       /* T after the 7 conv layers */
size_t afx_workspace_bytes(afx_handle h, int B, int L);  /* scratch needed by one call */
/* wave: device (B,L) fp32.  logits: device (B,2) fp32.  L is free (test_duration_sec, config.py:75): at least
 * 400 samples (one SSL frame); the AASIST head needs >= 6 frames and holds at most 630 temporal graph nodes
 * (clips up to about 37 s); beyond either bound the call fails with a message, it never truncates. */
int afx_forward(afx_handle h, const float* wave, int B, int L, float* logits, void* ws, size_t ws_bytes,
                void* stream);
/* The same forward in two calls, for scoring loops that overlap the back-end of one batch with the trunk of the next
 * (main.py:199-221 scores batch after batch; nothing orders batch i's head before batch i+1's trunk but the one stream):
 * afx_trunk_forward(ws) on stream A leaves the SSL features inside `ws`; afx_head_from_workspace(ws) on stream B -- after an
 * event the caller records behind the trunk -- runs the back-end on them.  Two workspaces alternate; a workspace is reused
 * for a trunk only after its head has finished.  Same kernels, same results as afx_forward, bit for bit. */
int afx_trunk_forward(afx_handle h, const float* wave, int B, int L, void* ws, size_t ws_bytes, void* stream);
int afx_head_from_workspace(afx_handle h, int B, int L, float* logits, void* ws, size_t ws_bytes, void* stream);
/* SSL features only: feats device (B,T,1024) fp32 == extract_feat() of models/fe.py:17-21 */
int afx_ssl_forward(afx_handle h, const float* wave, int B, int L, float* feats, void* ws, size_t ws_bytes,
                    void* stream);
/* Ragged batches (clips of different lengths in one call; the length policy of data/test_set.py:201-248 is what makes
 * the reference's batches uniform -- un-cropped clips are the case it leaves to batch size 1).  wave: device (B,Lmax) fp32,
 * row b holds clip b's n_samples[b] samples followed by ZEROS; n_samples: HOST int[B].  Each clip is scored exactly as if it
 * were alone: key-padding masks in both attentions, zero padding past its own frames in the positional / depthwise convs,
 * per-length sub-batches for the AASIST graphs.  logits (B,2).  Synchronises the stream once (length upload). */
size_t afx_ragged_workspace_bytes(afx_handle h, int B, int Lmax);
int afx_forward_ragged(afx_handle h, const float* wave, int B, int Lmax, const int* n_samples, float* logits, void* ws,
                       size_t ws_bytes, void* stream);
/* feats device (B,Tmax,1024) fp32 with Tmax = afx_num_frames(Lmax), rows past a clip's own frames zeroed;
 * n_frames (HOST int[B], may be NULL) receives the frame counts */
int afx_ssl_forward_ragged(afx_handle h, const float* wave, int B, int Lmax, const int* n_samples, float* feats,
                           int* n_frames, void* ws, size_t ws_bytes, void* stream);
/* The path from the OUTPUT OF CONV LAYER 5 on: conv5_h device (B,T5,512) in the operand type (fp16 / bf16; fp32 in exact
 * mode) -> conv layer 6, feature LayerNorm, projection, positional conv, transformer layers, head -> logits (B,2).
 * Same kernels in the same order as afx_forward from that point, so a caller that keeps conv layers 0-5 incrementally
 * (they are causal-local: a new 250-ms chunk adds 800/400/200/100/50/25 frames, the rest of the 4-s window is unchanged)
 * reproduces afx_forward(window) bit for bit at a sixteenth of the conv cost (afx/streaming.py, BASELINE config 5). */
size_t afx_tail_workspace_bytes(afx_handle h, int B, int T5);
int afx_tail_forward(afx_handle h, const void* conv5_h, int B, int T5, float* logits, void* ws, size_t ws_bytes,
                     void* stream);
/* the same with the window inside a longer per-stream buffer: utterance b's T5 rows start at conv5_h + b * batch_stride
 * elements (0 = packed; a multiple of 8, at least T5 * 512) -- a streaming caller appends the new frames to a ring and
 * passes a view, no copy per hop */
int afx_tail_forward_strided(afx_handle h, const void* conv5_h, long batch_stride, int B, int T5, float* logits, void* ws,
                             size_t ws_bytes, void* stream);
/* ---- KV-cached streaming mode (BASELINE config 5 as named: "250 ms chunks with cached SSL-encoder KV state") ----------
 * NOT A REFERENCE FUNCTION: the reference's trunk is bidirectional over the clip, an encoder that sees every frame once
 * -- when its chunk arrives -- is a different model (block-causal: a chunk's frames attend to the chunk and to the cached
 * keys / values of the 15 chunks before it, the positional conv sees no frame beyond the chunk).  Parity target: the
 * build's own offline restatement oracle/streaming.py.  An afx_kv holds the per-stream state of
 * n_streams lock-stepped streams (K / V rings of every layer, the positional conv's left context, the feature window);
 * afx_kv_step consumes the NEW frames of conv layer 6 of every stream, (n_streams, n, 512) fp32 with 1 <= n <= 16, and
 * returns the back-end's logits on the window that ends with this chunk.  The handle must outlive its afx_kv objects; weights
 * reloaded into the handle do not refresh keys / values already cached. */
typedef struct afx_kv afx_kv;
int afx_kv_create(afx_handle h, int n_streams, afx_kv** out);
void afx_kv_destroy(afx_kv* kv);
size_t afx_kv_state_bytes(const afx_kv* kv);
size_t afx_kv_workspace_bytes(const afx_kv* kv, int n_frames);
int afx_kv_step(afx_kv* kv, const float* feats6, int n_frames, float* logits, void* ws, size_t ws_bytes, void* stream);
/* back-end alone from given SSL features (B,T,1024) fp32 -> logits (B,2) */
int afx_head_forward(afx_handle h, const float* feats, int B, int T, float* logits, void* ws, size_t ws_bytes,
                     void* stream);
size_t afx_head_workspace_bytes(afx_handle h, int B, int T);
/* MyConformer.forward (models/conformer_baseline.py:22-29) alone: tokens device (B,T,emb) fp32 (what Model.forward hands it
 * after LL / BatchNorm / SELU, :58-63) -> class token prepended, the n_encoders Conformer blocks -> logits (B,2) = fc5(token 0)
 * and, when `embedding` is not NULL, token 0 itself (B,emb).  Handles of arch CONFORMER or CONFORMER_HEAD; workspace
 * afx_head_workspace_bytes(h, B, T). */
int afx_conformer_forward(afx_handle h, const float* tokens, int B, int T, float* logits, float* embedding, void* ws,
                          size_t ws_bytes, void* stream);
/* Overflow guard (no reference counterpart: the reference computes in fp32, models/fe.py:11-21).  The half-precision
 * engines keep operand copies in fp16 / bf16 (fp16x3: fp16 hi / lo pairs); a checkpoint with outlier channels can push one
 * past the format's range, after which the scores are NaN -- or, behind the AASIST head's max-pooling and top-k, finite
 * garbage.  Every forward counts, on the device and for free, the rows of the trunk's final LayerNorm whose statistics are
 * not finite and the logits that are not finite; this call waits for `stream`, returns non-zero (afx_last_error names the
 * counts and the precision) when anything was counted since the last check, and clears the counters.  A scoring loop calls
 * it once before it writes its scores (afx/harness.py does). */
int afx_check_finite(afx_handle h, void* stream);
/* debug taps (off by default; when on, forward keeps fp32 copies of intermediates) */
int afx_enable_taps(afx_handle h, int on);
/* debug taps: copy an intermediate of the LAST forward on this workspace into `out`
 * (device fp32).  Names: "conv", "proj", "pos", "layer<N>", "ssl", "tokens",
 * "block<N>", "e_S", "e_T", "hidden".  Returns the element count through n_out. */
int afx_tap(afx_handle h, const char* name, float* out, size_t cap_elems, size_t* n_out, void* stream);

/* ---- per-kernel-class timing (hipEvents on the launch stream around every launch of
 * the forwards issued between begin and end; off otherwise).  afx_profile_end waits
 * for the recorded events and returns, per class, summed milliseconds, algorithmic
 * FLOPs and launch counts (arrays of afx_profile_num_classes() entries). ------------ */
int afx_profile_begin(afx_handle h);
int afx_profile_end(afx_handle h, int n_classes, double* ms, double* flops, long long* launches);
int afx_profile_num_classes(void);
const char* afx_profile_class_name(int cls);

/* per-handle switches between two forms of the same op (A/B measurements and the per-op reference paths of the tests):
 * "posconv_sliding" 1 (default) sliding-window positional conv / 0 chunked-K GEMM; "conf_attn_mfma" 1 matrix-core Shaw
 * attention / 0 the fp32 VALU kernel; "fuse_conformer" 1 fused row chains / 0 one kernel per op; "fuse_conv_ln" 1 conv +
 * LayerNorm + GELU in one kernel / 0 two.  They act on THIS handle only. */
int afx_engine_set(afx_handle h, const char* key, int value);
/* tuning knobs for A/B measurements (process-wide; not part of the drop-in surface).
 * "gemm_map": workgroup->tile order of the MFMA GEMM, -1 default, 0 linear, 1 XCD-
 * contiguous, 2 XCD-contiguous + grouped.  "gemm_tile": -1 auto, 0 128x128, 1 256x256.
 * "fuse_conv_ln": 1 (default) conv layers 1-6 use the fused LayerNorm epilogue, 0 two kernels.
 * None of these changes WHAT is computed; the timing-only switches that do ("gemm_nodma") exist only in the
 * attribution build (make attr), the product library refuses them. */
int afx_debug_set(const char* key, int value);

/* ---- single-kernel entry points (unit parity tests; operand pointers are bf16 or
 * fp16 device arrays according to `dtype`; fp32 arrays for AFX_DT_FP32 and, in afx_k_gemm / afx_k_mhsa, for
 * AFX_DT_FP16X3: the split-precision forms take and return fp32 -- afx_k_gemm then builds the hi / lo operand forms
 * per call in temporary device memory and synchronises: a test hook) --------------------------------------------- */
int afx_k_gemm(int dtype, const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
               int act, float alpha, const float* resid, long ldr, float* out_f, long ldo_f, void* out_h, long ldo_h,
               void* stream);
/* Conv1d(Cin->N, k, stride s) on channel-last input (B,Tin,Cin) as one GEMM; Wp is
 * the tap-major packed weight [N][k*Cin]; out_f (B,Tout,N) fp32 */
int afx_k_conv_gemm(int dtype, const void* in_h, const void* Wp, int B, int Tin, int Tout, int Cin, int k, int s,
                    int N, const float* bias, float* out_f, void* stream);
/* the same conv with LayerNorm over the 512 output channels + activation fused into the GEMM
 * epilogue (row-complete tile); N is fixed at 512; out_f and/or out_h (B,Tout,512) */
int afx_k_conv_ln_act(int dtype, const void* in_h, const void* Wp, int B, int Tin, int Tout, int Cin, int k, int s,
                      const float* bias, const float* gamma, const float* beta, float eps, int act, float* out_f,
                      void* out_h, void* stream);
int afx_k_pack_linear(int dtype, const float* w, int N, int K, int Kpad, void* out_h, void* stream);
int afx_k_pack_conv(int dtype, const float* w, int N, int Cin, int k, void* out_h, void* stream);
int afx_k_conv0(int dtype, const float* wave, int B, int L, const float* w, const float* bias, const float* gamma,
                const float* beta, int pre_emph, float coef, void* out_h, void* stream);
/* the same without the per-call operand build of the test hook above (device allocation + stream synchronisation): the
 * layer's split-precision fp16 operand block is built once per checkpoint into afx_k_conv0_pack_bytes() bytes and passed in;
 * asynchronous, allocates nothing -- what the streaming scorer runs every hop */
size_t afx_k_conv0_pack_bytes(void);
int afx_k_conv0_pack(const float* w, const float* bias, void* pack, void* stream);
int afx_k_conv0_packed(int dtype, const float* wave, int B, int L, const void* pack, const float* w, const float* bias,
                       const float* gamma, const float* beta, int pre_emph, float coef, void* out_h, void* stream);
/* data/preprocess.py:16-29 as a stand-alone op: y[t] = x[t] - coef*x[t-1], reflect pad */
int afx_k_pre_emphasis(const float* x, int B, int L, float coef, float* y, void* stream);
/* Utterance length policy, batched (data/test_set.py:139-146 pad, :201-227 adjustDuration, :229-248
 * adjustDuration_random_start): x = the ragged clips packed back to back (device), offs[B+1] their sample
 * offsets (device, int64), starts[B] crop starts or NULL; out (B, duration): out[b][i] = x_b[(start_b + i) mod n_b]. */
int afx_k_tile_crop(const float* x, const long long* offs, const long long* starts, int B, int duration, float* out,
                    void* stream);
int afx_k_rownorm(int dtype, const float* x, long ldx, int rows, int C, const float* gamma, const float* beta,
                  float eps, int act, float* out_f, long ldo_f, void* out_h, long ldo_h, void* stream);
int afx_k_mhsa(int dtype, const void* qkv, void* out, int B, int T, int H, void* stream);
int afx_k_conf_attn(int dtype, const float* q, long ldq, const float* kv, long ldkv, const float* rel, int max_pos,
                    int B, int N, int H, int dh, void* out_h, long ldo, void* stream);
/* the same on the matrix cores (operand-type q/k/E/P, fp32 accumulation): rel_h is the embedding table
 * packed by afx_k_pack_linear with Kpad = 64; head dim 36, at most 209 tokens */
int afx_k_conf_attn_mfma(int dtype, const float* q, long ldq, const float* kv, long ldkv, const void* rel_h, int max_pos,
                         int B, int N, int H, int dh, void* out_h, long ldo, void* stream);
int afx_k_conf_dwconv(int dtype, const float* x, long ldx, const float* w, const float* bias, const float* bn_scale,
                      const float* bn_shift, int B, int N, int C, int k, void* out_h, long ldo, void* stream);

/* ---- AASIST graph modules alone (models/aasist_modules.py:17-110, 112-294, 296-338);
 * all fp32, BatchNorm passed folded (scale = w/sqrt(var+eps), shift = b - mean*scale).
 * Supported (in,out) dims: (64,64), (64,32), (32,32); N <= 80 nodes. ------------------ */
const char* afx_aasist_error(void);
int afx_k_gat(const float* x, int B, int N, int din, int dout, const float* att_w, const float* att_b,
              const float* att_vec, const float* w1, const float* b1, const float* w2, const float* b2,
              const float* bn_scale, const float* bn_shift, float temp, float* y, void* stream);
/* wts: t1w t1b t2w t2b att_w att_b attM_w attM_b v11 v22 v12 vM w1 b1 w2 b2 w1M b1M w2M b2M bn_scale bn_shift
 * master == NULL -> mean of the projected nodes; xp_scratch: B*(n1+n2)*din + B*din floats */
int afx_k_hgat(const float* x1, int n1, const float* x2, int n2, int B, int din, int dout, const float* const* wts,
               float temp, const float* master, long master_bstride, float* xp_scratch, float* y1, float* y2,
               float* mout, void* stream);
int afx_k_graph_pool(const float* h, int B, int N, int D, int keep, const float* w, const float* b, float* out,
                     void* stream);

/* Residual_block alone (models/aasist_modules.py:340-397: conv1 (2,3) pad (1,1) ON x (Q2) -> bn2 -> SELU -> conv2 (2,3)
 * pad (0,1), + conv_downsample (1,3) of x when the channel count changes).  x, y: device NCHW fp32 (B,cin,H,W) /
 * (B,cout,H,W); conv weights in their checkpoint layout [cout][cin][kh][kw]; bn2 folded; down_w/down_b NULL exactly when
 * cin == cout.  cin 1 or a multiple of 16, cout 32 / 64 / 128.  scratch: afx_k_resblock_scratch_floats() floats. */
size_t afx_k_resblock_scratch_floats(int B, int cin, int cout, int H, int W);
int afx_k_resblock(const float* x, int B, int cin, int cout, int H, int W, const float* conv1_w, const float* conv1_b,
                   const float* bn2_scale, const float* bn2_shift, const float* conv2_w, const float* conv2_b,
                   const float* down_w, const float* down_b, float* scratch, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AFX_H_ */
