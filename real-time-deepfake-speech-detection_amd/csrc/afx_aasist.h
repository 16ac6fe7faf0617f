// AASIST graph-attention back-end -- internal interface
// between the engine and afx_aasist.hip.  Everything after the SSL trunk runs in
// fp32: top-k graph pooling is discontinuous, so operand rounding is kept out of it.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>

namespace afx {

// Activations of the 2-D residual encoder live channel-last in a zero-padded image
// (B, AAS_HP, AAS_WP, C): logical pixel (h, w) sits at (h+1, w+1).
constexpr int AAS_F = 42;       // spectral bins after max_pool2d(3,3): 128 // 3
constexpr int AAS_HP = 46;      // padded rows (43 conv1 rows + 1 top + slack)

struct AasistWeights {
  bool ready = false;
  // true (set by the engine before aasist_finalize for fp16 / bf16 engines): the dense products of the head run in
  // the split-precision form on the fp16 matrix pipe (afx_aasist.hip, f32s_gemm_kernel: fp32 accuracy, ~5x the speed
  // of the fp32 MFMA); false = exact mode, true-fp32 matrix instruction
  bool split = false;
  struct Split {
    _Float16 *hi = nullptr, *lo = nullptr;
  };
  Split LLs, att_s0, att_s3;
  // pointers into engine-owned fp32 tensors / prepared buffers
  const float *LLw, *LLb;
  float bn0_scale, bn0_shift;  // first_bn (1 channel)
  struct Block {
    int cin, cout;
    float *w1, *w2, *wd;  // tap-major packed [cout][taps*cin]
    const float *b1, *b2, *bd;
    float *bn2_scale, *bn2_shift;
    Split s1, s2, sd;
  } blk[6];
  float *bn1_scale, *bn1_shift;  // first_bn1 (64)
  float *att_w0, *att_w3;        // 1x1 convs as [128][64], [64][128]
  const float *att_b0, *att_b3;
  float *att_bn_scale, *att_bn_shift;
  const float *pos_S, *master1, *master2;
  struct Gat {
    const float *att_w, *att_b, *att_vec, *w1, *b1, *w2, *b2;
    float *bn_scale, *bn_shift;
  } gatS, gatT;
  struct HGat {
    int din, dout;
    const float *t1w, *t1b, *t2w, *t2b, *att_w, *att_b, *attM_w, *attM_b;
    const float *v11, *v22, *v12, *vM;
    const float *w1, *b1, *w2, *b2, *w1M, *b1M, *w2M, *b2M;
    float *bn_scale, *bn_shift;
  } h11, h12, h21, h22;
  struct Pool {
    const float *w, *b;
  } pS, pT, phS1, phT1, phS2, phT2;
  const float *out_w, *out_b;
};

struct AasistWs {
  float *ll;                 // (B*T, 128)
  float *imgA, *imgB, *imgC; // padded images, 64 channels capacity each
  float *wmap1, *wmap2;      // attention 1x1 conv intermediates (128 / 64 channels)
  float *eS, *eT;            // (B,42,64), (B,Tt,64)
  float *gS, *gT;            // GAT outputs
  float *oS, *oT;            // pooled
  float *br;                 // per-branch scratch
  float *hidden;             // (B,160)
  // debug views of the last forward (names/pointers into the workspace)
  const char* dbg_name[16];
  const float* dbg_ptr[16];
  size_t dbg_n[16];
  int dbg_count = 0;
};

using GetF = std::function<const float*(const std::string&)>;
using Alloc = std::function<void*(size_t)>;

// prepare derived tensors (BN folds, tap-major conv weights); nullptr or error text
const char* aasist_finalize(AasistWeights& w, const GetF& get, const Alloc& alloc, hipStream_t s);
void aasist_set_stop(int v);  // diagnostic knob: return after that many stages of the back-end (0 = all)
void aasist_carve(int B, int T, const Alloc& take, AasistWs* ws);
// feats (B,T,1024) fp32 -> logits (B,2); nonfinite (device counter or null): += 1 for every logit that is not finite
const char* aasist_forward(const AasistWeights& w, const float* feats, int B, int T, AasistWs& ws, float* logits,
                           hipStream_t s, int* nonfinite = nullptr);

}  // namespace afx
