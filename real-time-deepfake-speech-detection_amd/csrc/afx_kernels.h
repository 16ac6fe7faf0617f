// Host-visible launchers of the gfx950 kernels (internal to libafx; the public
// C ABI is include/afx.h).  Every launcher is asynchronous on the given stream,
// allocates nothing, and returns nullptr on success or a static error string.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afx {

// DT_FP16X3: split precision -- activations live in fp32 (every non-GEMM kernel runs its DT_FP32 form), every dense product
// runs on the fp16 matrix pipe as x.w ~ xh.wh + xl.wh + xh.wl with fp16 hi / lo pairs of both operands (GemmArgs::k1)
enum DType { DT_BF16 = 0, DT_FP16 = 1, DT_FP32 = 2, DT_FP16X3 = 3 };
inline size_t dtype_size(int dt) { return dt == DT_FP32 || dt == DT_FP16X3 ? 4 : 2; }
// Split-precision A operands are scaled by a power of two before the hi / lo split, per operand KIND (the product divides it
// out again: GemmArgs::a_inv).  The matrix instruction keeps fp16 subnormals (tools/denorm_probe.hip), so the scale only
// decides where the lo half of a small entry stops being exact to 2^-22 relative and becomes exact to 2^-25 / scale absolute:
//   * LayerNorm outputs are bounded by sqrt(C) |gamma| + |beta| -- x 16 (|y| < 4094; entries above 0.008 fully precise);
//   * everything else (GELU'd FFN hidden, attention output, projections, a caller's features) is unbounded in a trained
//     checkpoint -- x 1: the same range as the fp16 mode (65 504), entries above 0.125 fully precise.
// Measured on the 48-utterance teacher gate: all x 16: features 1.1e-6 / logits 1.4e-6; all x 1: 4.4e-6 / 2.8e-6
// (profiles/r04_s3_scale_ab.txt); no GraphPool decision moves either way.
constexpr float kS3ScaleBounded = 16.f;
constexpr float kS3ScaleFree = 1.f;
// PAIR FORM (round 4): the hi / lo halves of a split-precision operand are interleaved in groups of 32 elements -- element k
// of a row lives at half (k / 32) * 64 + k % 32 (hi) and 32 halfs further (lo) -- so a row of K values is 2 K consecutive
// halfs in the bytes its fp32 form would take (row-local: any row stride that is a multiple of 32 elements works), and one
// 128-byte piece of it is exactly one K-step of the matrix instruction for BOTH halves: the tile kernels walk it like a plain
// fp16 operand of 2 K columns and issue, per 64-half K-tile, lo.hi + hi.lo + hi.hi on the fragments they have read anyway.
__host__ __device__ inline long s3_pair_index(long k) { return ((k >> 5) << 6) + (k & 31); }

struct GemmArgs {
  const void* A;  // matrix-core operand type (bf16/fp16), K contiguous
  const void* W;  // [N][K] (nn.Linear layout), K contiguous
  int M, N, K;    // N is per group; K % 64 == 0
  int k_algo;     // un-padded K for FLOP accounting (0: same as K); ignored by the kernel
  // A row m starts at (m / rpb) * a_batch + (m % rpb) * a_row   (elements)
  int rpb;
  long a_batch, a_row;
  // k -> (k / kchunk) * kchunk_stride + (k % kchunk); plain GEMM: kchunk = K
  int kchunk;
  long kchunk_stride;
  long ldw;
  // grid.z groups: element offsets per group into A and W; column offset per group
  long g_a, g_w;
  int g_n;
  const float* bias;  // [groups * N] or null
  int act;
  float alpha;
  const float* resid;  // fp32, indexed like the output, or null
  long ldr;
  float* out_f;  // fp32 output or null
  long ldo_f;
  void* out_h;  // operand-type output or null
  long ldo_h;
  // output row of A-row m: (m / rpb) * o_batch_rows + (m % rpb) + o_row_off
  // (out_f and resid use o_*, out_h uses oh_* -- e.g. the fp32 residual stream is
  // unpadded while the operand copy feeds the time-padded positional-conv buffer)
  int o_batch_rows, o_row_off;
  int oh_batch_rows, oh_row_off;
  int map_mode;  // workgroup->tile order, set by launch_gemm (0 linear, 1 XCD-contiguous, 2 + grouped)
  // fused row LayerNorm epilogue (needs N == 512, one row-complete tile per M-tile):
  // out = act(LayerNorm(acc + bias) * ln_gamma + ln_beta); null ln_gamma = off
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int a_nt;  // 1: non-temporal LDS-DMA for the A panel (set by launch_gemm)
  int no_deep;  // 1: never the deep form of the 128x64 tile (per-engine A/B switch "gemm_small_deep" = 0; bit-identical rows either way)
  int m_lo;  // 8-wave kernels only: the launch covers rows [m_lo, M) (set by launch_gemm for a remainder launch; 0 otherwise)
  // ---- split precision (launch_gemm with DT_FP16X3; set by the engine's wrapper, 0 / null otherwise) ---------------------
  // k1 = the algorithmic K; A and W are in PAIR FORM (above) and every length / stride on the K side of this struct counts
  // HALFS of it: K = 2 k1, a_row / a_batch / g_a / kchunk / kchunk_stride / ldw / g_w are twice their fp32-element values.
  // Per 64-half K-tile (32 k values: hi | lo) a tile kernel issues w_lo.a_hi + w_hi.a_lo + w_hi.a_hi.  The epilogue
  // multiplies the accumulator of column n by pre_scale[n] x a_inv (the inverse of the row's power-of-two weight scale x
  // the inverse of the A operand's activation scale) before the bias, and writes `out_h` as FP32 (the engine's operand
  // buffers are fp32 in that mode).
  int k1;
  const float* pre_scale;
  float a_inv;
  // split precision, output side: oh_pairs != 0 = `out_h` receives the result AS the next product's A operand -- the pair
  // form of oh_scale x value, row by row in place of the fp32 row (ldo_h % 32 == 0) -- instead of fp32 (the producer
  // writes the operand, no separate split launch: launch_split_pairs)
  int oh_pairs;
  float oh_scale;
  int dbg_nodma;  // attribution build (-DAFX_ATTR) only, ignored otherwise: epilogue bits 8 no activation, 16 narrow stores, 32 no stores, 64 no epilogue
};
const char* launch_gemm(const GemmArgs& p, int dtype, int groups, hipStream_t s);
const char* launch_gemm_f32(const GemmArgs& p, int groups, hipStream_t s);  // afx_gemm_f32.hip (DT_FP32 operands)
bool gemm_is_narrow(int N);  // true: the 128x64 tile instance serves this N
int gemm_tile_of(const GemmArgs& p, int groups);  // tile instance id (afx_gemm.hip)
void gemm_set_s3_small(int v);  // A/B knob: split-precision small-M products (0 default, 1 = 128x128, 2 = 128x64 2-stage)
void gemm_set_map_mode(int m);  // A/B knob: -1 default, else force map_mode
void gemm_set_small_deep(int v);  // A/B knob: 1 (default) = deep form of the 128x64 tile at <= two tiles per CU
void gemm_set_tile(int t);      // A/B knob: -1 default, 0: 128x128 tile, 1: 256x256 tile
void gemm_set_a_nt(int v);      // A/B knob: -1 auto, 0/1 non-temporal A-panel loads
void gemm_set_conv_split(int v);  // A/B knob: 1 (default) = multi-round conv layers as whole rounds of 128-row tiles + a 64-row remainder launch
void gemm_set_ph4(int v);       // A/B knob: 1 = 4-phase K-tile of the 8-wave kernels (default 0: two phases of 32 MFMAs)
void gemm_set_fit(int v);       // A/B knob: 1 (default) = 8-phase tile height fitted to one round of the CUs (160..256 rows)
void gemm_set_split(int v);     // A/B knob: 1 (default) = whole rounds on the 8-phase kernel + 128x128 remainder rows
void mhsa_set_zsplit(int v);       // A/B knob: workgroups per (utterance, head) of the one-pass trunk attention (0 = automatic)
void mhsa_set_waves(int v);        // A/B knob: 4 or 7 (default) waves per workgroup of the one-pass trunk attention
void mhsa_set_force_long(int v);   // test knob: the blocked any-length trunk attention kernel at every length
void conf_attn_mfma_set_waves(int v);       // A/B knob: 4 or 7 (default) waves per workgroup of the one-pass Shaw attention
void conf_attn_mfma_set_force_long(int v);  // test knob: the blocked matrix-core Shaw attention at every length
void conf_attn_set_block(int v);   // test knob: keys per LDS block of the fp32 attention (0 = automatic)
void gemm_set_deep(int v);      // A/B knob, conv tile: 0 = 2-stage kernel, -1/2 = 8-phase kernel (default)
bool gemm_set_nodma(int v);     // timing-only knob (attribution build, -DAFX_ATTR, only): epilogue parts off; false = not in this build

// ---- frontend / row kernels (afx_frontend.hip) ---------------------------------
// conv layer 0 (Cin=1,k=10,s=5) + LayerNorm(512) + erf-GELU; optional pre-emphasis.
const char* launch_conv0(const float* wave, int B, int L, int T0, const float* w /*[512][10]*/,
                         const float* bias, const float* gamma, const float* beta, int pre_emph,
                         float pre_coef, void* out_h, int dtype, hipStream_t s, const void* wpack = nullptr, float pair_scale = 0.f);
// dtype DT_FP16X3 (the split-precision engines): the packed-operand kernel with fp32 rows out, or -- pair_scale > 0 -- the rows as conv
// layer 1's pair-form operand scaled by that power of two
// wpack: the layer's weights + bias as the split-precision fp16 MFMA operand (conv0_pack_bytes() bytes, built once per
// checkpoint by launch_conv0_pack); with it the half-precision engines run the layer on the fp16 matrix pipe at fp32 accuracy
size_t conv0_pack_bytes();
const char* launch_conv0_pack(const float* w /*[512][10]*/, const float* bias, void* pack, hipStream_t s);
// the wav2vec2-base form of layer 0: bias-free conv -> GroupNorm(512,512) (per utterance and channel, over time) -> GELU;
// stats: scratch of conv0_groupnorm_stats_floats(B, T0) floats
const char* launch_conv0_groupnorm(const float* wave, int B, int L, int T0, const float* w, const float* gamma,
                                   const float* beta, float eps, float* stats, void* out_h, int dtype, hipStream_t s);
size_t conv0_groupnorm_stats_floats(int B, int T0);
// ragged batch (packed, offs[B+1]) -> (B, duration): out[b][i] = x_b[(start_b + i) mod n_b]; starts may be null
const char* launch_tile_crop(const float* x, const long long* offs, const long long* starts, int B, int duration,
                             float* out, hipStream_t s);
void conv0_set_mfma(int v);  // A/B knob: 1 (default) = matrix-core forms (split-precision fp16 when packed), 2 = fp32 MFMA form, 0 = VALU form
// y[t] = x[t] - coef * x[t-1] with a reflect pad on the left; (B,L) fp32 -> (B,L) fp32
const char* launch_pre_emphasis(const float* x, int B, int L, float coef, float* y, hipStream_t s);
// rows x C fp32 -> LayerNorm (optional activation) -> fp32 and/or operand-type outputs.
// Output row r goes to (r / rpb) * o_batch_rows + (r % rpb) + o_row_off (row strides ld*).
struct RowNormArgs {
  const float* x;
  long ldx;
  int rows, C;
  const float* gamma;
  const float* beta;
  float eps;
  int act;
  float* out_f;
  long ldo_f;
  void* out_h;
  long ldo_h;
  int rpb, o_batch_rows, o_row_off;
  int* nonfinite;  // device counter or null: += 1 for every row whose statistics are not finite (the engine's overflow guard on the trunk's final LayerNorm)
  int oh_pairs;  // split precision (with DT_FP32): != 0 = out_h receives the PAIR FORM of oh_scale x y in place of the fp32 row (ldo_h % 32 == 0)
  float oh_scale;
};
const char* launch_rownorm(const RowNormArgs& a, int dtype, hipStream_t s);
// zero the time padding rows of the positional-conv operand buffer (B, T+128, C)
// (lens: ragged batch -- rows [pad_front + lens[b], end) of utterance b are zeroed as well)
const char* launch_zero_pad_rows(void* buf_h, int B, int T, int C, int pad_front, int pad_back, int dtype,
                                 hipStream_t s, const int* lens = nullptr);
// fp32 -> operand type conversion with optional layout permutes (weight packing)
const char* launch_pack_linear(const float* w, int N, int K, int Kpad, void* out_h, int dtype, hipStream_t s);
const char* launch_pack_conv(const float* w, int N, int Cin, int k, void* out_h, int dtype, hipStream_t s);
const char* launch_pack_posconv(const float* v, const float* g, int C, int cpg, int k, float* norm_tmp /*[k]*/,
                                void* out_h, int dtype, hipStream_t s);
// Split precision (DT_FP16X3).  Weights: a matrix packed as fp32 rows [N][K] (the pack launchers above with DT_FP32) is
// rewritten IN PLACE as the pair form of w * 2^s_n, s_n the power of two that brings the row's largest magnitude
// to [8192, 16384); row_scale[n] = 2^-s_n.  K <= 12288, K % 32 == 0.
const char* launch_split_weight_rows(void* w_f32_inplace, int N, int K, float* row_scale, hipStream_t s);
// Activations: n fp32 values (n % 32 == 0: whole 32-element groups; 16-byte aligned) -> the pair form of x * scale
// at `pairs` (2 n halfs; distinct buffers)
const char* launch_split_pairs(const float* x, long n, void* pairs, float scale, hipStream_t s);

// ---- positional conv of the encoder as a sliding-window kernel (afx_posconv.hip) ---------------
// x (B*T, 1024) fp32 += GELU(grouped conv over the time-padded operand copy xpad (B, T+128, 1024)); T <= 224
struct PosConvArgs {
  const void* xpad;  // operand type, rows [64, 64+T) = x, zero rows around
  long xpad_batch;   // elements between utterances ((T + 128) * 1024)
  const void* W;     // packed [1024][128 taps x 64 in] (launch_pack_posconv)
  const float* bias;
  float* x;
  int B, T, slab_rows;  // slab_rows is filled in by the launcher
};
const char* launch_posconv(const PosConvArgs& p, int dtype, hipStream_t s);

// ---- transformer self-attention (afx_attn.hip) -----------------------------------
// qkv: (B*T, 3*H*64) operand type [q | k | v]; out: (B*T, H*64) operand type.
// lens (device int32[B], or null): ragged batch -- utterance b has lens[b] valid frames of its T rows (key-padding mask)
const char* launch_mhsa(const void* qkv, void* out, int B, int T, int H, int dtype, hipStream_t s, const int* lens = nullptr);
void mhsa_set_vtr(int v);  // A/B knob: V row-major in LDS + ds_read_b64_tr_b16 (0 = the V^T image)

// the same attention with fp32 rows in / out and split-precision products on the fp16 matrix pipe (dtype "fp16x3"); T <= 224
// out_pairs: `out` receives the pair form of out_scale x the result (row by row, in place of the fp32 rows) instead of fp32 rows
const char* launch_mhsa_split(const float* qkv, float* out, int B, int T, int H, hipStream_t s, const int* lens = nullptr,
                              bool out_pairs = false, float out_scale = 1.f);
// KV-cached streaming attention (afx_kv_step; not a reference function): ring (S, 256, 3*H*64) rows [q | k | v] in 16-slot
// groups, cnt[16] valid frames per group, the queries are group q_tile's slots; out (S, 16, H*64)
const char* launch_mhsa_ring(const void* ring, void* out, int S, int H, int q_tile, const int* cnt, int dtype, hipStream_t s);
// the same in split precision (dtype "fp16x3"): fp32 [q | k | v] slots in; fp32 rows or (out_pairs) pair-form rows out
const char* launch_mhsa_ring_split(const float* ring, float* out, int S, int H, int q_tile, const int* cnt, hipStream_t s,
                                   bool out_pairs = false, float out_scale = 1.f);

// ---- Conformer student head (afx_conformer.hip) ----------------------------------
// y = selu(bn(x)) for rows 1..T of each utterance, row 0 = class token; x is the LL
// output (B*T, E) fp32; out (B*(T+1), E) fp32 residual stream.
// raw: rows are copied as they are (no BatchNorm / SELU): MyConformer.forward's own input (models/conformer_baseline.py:22-24)
const char* launch_conf_tokens(const float* ll, const float* cls, float bn_scale, float bn_shift, int B, int T,
                               int E, float* out, hipStream_t s, bool raw = false);
// Shaw relative-position attention; q (B*N, H*dh) fp32, kv (B*N, 2*H*dh) fp32,
// rel (2*max_pos+1, dh) fp32; out operand-type rows of stride ldo.
const char* launch_conf_attn(const float* q, long ldq, const float* kv, long ldkv, const float* rel, int max_pos,
                             int B, int N, int H, int dh, void* out_h, long ldo, int dtype, hipStream_t s,
                             const int* lens = nullptr, int len_add = 0);  // ragged: tokens of utterance b = lens[b] + len_add
// the same on the matrix cores: rel_h is the embedding table in operand type, rows padded to 64
// (pack_linear with Kpad = 64); N <= 209 tokens, head dim 36
// the same in split precision (dtype "fp16x3"): fp32 rows in / out, rel64 = the table as fp32 rows padded to 64; N <= 209, head dim 36
const char* launch_conf_attn_split(const float* q, long ldq, const float* kv, long ldkv, const float* rel64, int max_pos, int B, int N,
                                   int H, int dh, float* out, long ldo, hipStream_t s, const int* lens = nullptr, int len_add = 0);
const char* launch_conf_attn_mfma(const float* q, long ldq, const float* kv, long ldkv, const void* rel_h, int max_pos,
                                  int B, int N, int H, int dh, void* out_h, long ldo, int dtype, hipStream_t s,
                                  const int* lens = nullptr, int len_add = 0);
// GLU -> depthwise conv (same pad) -> BatchNorm(eval) -> Swish.  x (B*N, 2*C) fp32.
const char* launch_conf_dwconv(const float* x, long ldx, const float* w /*[C][k]*/, const float* bias,
                               const float* bn_scale, const float* bn_shift, int B, int N, int C, int k,
                               void* out_h, long ldo, int dtype, hipStream_t s, const int* lens = nullptr, int len_add = 0);
// Row-local chains of a Conformer block, 16 token rows per wave, registers only
// (afx_conformer_fused.hip).  stage 0: x += 1/2 FF1(x), out2 = W_a LN2(x) (q|k|v, no bias);
// stage 1: x += W_a in_h + b_a, out2 = W_b LN2(x) + b_b (pointwise conv 1, GLU input);
// stage 2: x += W_a in_h + b_a, x += 1/2 FF(x), x_out = LN2(x) (post-norm).
// float offsets inside a chain's parameter block (per-column vectors, 8 KB, zero-filled where unused):
// FF LayerNorm gamma/beta, FF bias 1 (576) and 2, second LayerNorm gamma/beta, bias of W_a, bias of W_b (576)
enum ChainParamOffsets { CP_FF_G = 0, CP_FF_B = 144, CP_FF_B1 = 288, CP_FF_B2 = 864, CP_LN2_G = 1008, CP_LN2_B = 1152,
                         CP_BA = 1296, CP_BB = 1440 };
constexpr int kChainParamFloats = 2048;
// split precision (DT_FP16X3): a second block behind it, the per-output-column powers of two that undo the weight rows' scaling
// (afx_frontend.hip::split_weight_rows_kernel): feed-forward W1 (576) and W2 (144), W_a (<= 432), W_b (576)
constexpr int kChainScaleFloats = 2048;
enum ChainScaleOffsets { CS_FF1 = 0, CS_FF2 = 576, CS_A = 720, CS_B = 1152 };
static_assert(CS_B + 576 <= kChainScaleFloats, "scale block");
struct ConfChainArgs {
  int M, E, Ep, FFp;        // rows; emb (144); weight row strides of the K = E and K = 4E matrices
  const float* x_in;        // (M, E) fp32 residual rows
  float* x_out;
  const float* params;      // the chain's parameter block (ChainParamOffsets)
  const void *ff_w1, *ff_w2;                       // feed-forward module (stages 0 and 2): packed [4E][Ep], [E][FFp]
  const void* w_a;                                 // stage 0: W_qkv [3E][Ep]; 1: W_out [E][Ep]; 2: W_pw2 [E][ld_w_a]
  int ld_w_a;
  const void* w_b;                                 // stage 1: W_pw1 [4E][Ep]
  const void* in_h;                                // stage 1: attention output (M, ld_in_h); 2: depthwise-conv output
  long ld_in_h;
  float* out2;                                     // stage 0: q|k|v (M, 3E); 1: GLU input (M, 4E)
  long ld_out2;
};
const char* launch_conf_chain(const ConfChainArgs& p, int stage, int dtype, hipStream_t s);
// logits = fc5(token0):  x (B*N, E) fp32 rows, token row = b*N.
// nonfinite (device counter or null): += 1 for every output that is not finite
const char* launch_small_linear(const float* x, long row_stride, int rows, int K, const float* w, const float* b,
                                int N, float* out, hipStream_t s, int* nonfinite = nullptr);

}  // namespace afx
