// ViS (SummaryMixing aggregator) forward/backward sequencing on top of the GEMM engine.
#pragma once
#include "../../include/sequoia_hip.h"
#include "sq_common.h"

struct Arena {
    char* base;
    size_t off;
    void* take(size_t bytes) {
        off = sq_align_up(off, 256);
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

// Workspace of one ViS forward (+ what backward re-reads when save_for_backward).
// T = compute dtype (f32 or bf16).  M = B*N tokens, HD = nheads*64.
struct VisBufs {
    int nsave;                          // depth when saving, else 1
    float* Xin[SQ_MAX_DEPTH + 1];       // [M, D] f32 layer inputs (index 0 = x + pos); 1 buffer when not saving
    void* Xin_lp[SQ_MAX_DEPTH + 1];     // bf16 copies (bf16 mode) else == Xin
    float* X1[SQ_MAX_DEPTH];            // [M, D] f32 after the mixer block
    void* X1_lp[SQ_MAX_DEPTH];          // bf16 copy (unused by forward, kept for symmetry) or == X1
    float* Xbar32[SQ_MAX_DEPTH];        // [B, D] f32 token mean
    void* Xbar[SQ_MAX_DEPTH];           // [B, D] T
    float* F[SQ_MAX_DEPTH];             // [M, HD] f32  f(x)+b (pre-LN)
    void* Lf[SQ_MAX_DEPTH];             // [M, HD] T    GELU(LN64(F))
    float* Sm[SQ_MAX_DEPTH];            // [B, HD] f32  s(mean x)+b (pre-LN)
    void* Ts[SQ_MAX_DEPTH];             // [B, HD] T    GELU(LN64(Sm))
    float* Cs[SQ_MAX_DEPTH];            // [B, HD] f32  summary half of the combiner + bias
    void* P[SQ_MAX_DEPTH];              // [M, HD] T    combiner pre-activation (saved only when training)
    void* O[SQ_MAX_DEPTH];              // [M, HD] T    GELU(P)
    void* Y[SQ_MAX_DEPTH];              // [M, D] T     LayerNorm(X1)
    void* U[SQ_MAX_DEPTH];              // [M, D] T     FF pre-activation (saved only when training)
    void* H1[SQ_MAX_DEPTH];             // [M, D] T     GELU(U)
    float* skws; size_t skws_bytes;     // split-K scratch for the skinny (M = B) GEMMs
    float* x1m;                         // [B, D] f32 token mean of the last layer's X1 (inference: the last linear map runs behind the mean)
    float* xm;                          // [B, D] f32 token mean of the last layer output
    void* xn;                           // [B, D] T   LayerNorm(xm)
    size_t bytes;
};

void sq_vis_bufs(const sq_vis_config& c, int dtype, int B, int save, char* base, VisBufs* out);

// summary branch of one layer, forward, in one launch (summary.hip; bf16 only)
int sq_launch_summary_fwd(const void* Xbar, const void* Ws, const float* bs, const float* lng, const float* lnb, const void* Wc,
                          const float* bc, float* Sm, void* Ts, float* Cs, int B, int D, int H, hipStream_t stream);

// bf16 mode keeps the residual stream, the saved activations and the gradient stream in bf16 only (forward AND backward must
// agree on it: the backward pass re-reads what the forward pass stored); SQ_VIS_FP32_STREAM=1: fp32 streams + bf16 operand copies
inline bool sq_vis_lean_stream(int dtype) {
    return dtype == SQ_BF16 && !sq_env_flag("SQ_VIS_FP32_STREAM");      // (read per call: tests flip it inside one process)
}

// What a save_for_backward forward pass stored in `workspace` (lean bf16 rows or fp32 rows): noted by sq_vis_forward_ex, checked
// by sq_vis_backward_buckets -- a switch flipped between the two calls is an SQ_ERR_ARG, not bf16 rows re-read as fp32.
void sq_vis_note_saved_stream(const void* workspace, bool lean);
int sq_vis_saved_stream(const void* workspace);            // 1 lean, 0 fp32, -1 nothing noted for this workspace

// dtype of the saved GELU pre-activations (U, P): the operand dtype
inline int sq_vis_preact_dtype(int dtype) { return dtype; }
