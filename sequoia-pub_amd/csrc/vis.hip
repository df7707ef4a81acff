// ViS forward (src/tformer_lin.py) as a fixed launch sequence over the MFMA GEMM engine.
//
// Per layer (M = B*100 tokens, HD = nheads*64):
//   Xbar = mean_n X                                   [B, D]    token_mean
//   F    = X . Wf^T + bf        (16 heads as ONE GEMM) [M, HD]   gemm
//   Lf   = GELU(LN64(F))                               [M, HD]   ln64_gelu        (tformer_lin.py:20)
//   Sm   = Xbar . Ws^T + bs     (= mean_n s(x): the linear map commutes with the token mean,
//                                 100x fewer MACs than :21-22)    [B, HD]   gemm
//   Ts   = GELU(LN64(Sm))                              [B, HD]   ln64_gelu        (:22)
//   Cs   = Ts_h . Wc_h[:, 64:]^T + bc_h   per head     [B, HD]   batched gemm     (:24, summary half of cat)
//   O    = GELU(Lf_h . Wc_h[:, :64]^T + Cs[b])         [M, HD]   batched gemm, per-slide row bias (:24)
//   X1   = O . Wp^T + bp + X                            [M, D]    gemm + residual  (:45-46,75)
//   Y    = LN_D(X1)                                     [M, D]    ln_rows          (:54)
//   H1   = GELU(Y . W1^T + b1)                          [M, D]    gemm             (:55-56)
//   X    = H1 . W2^T + b2 + X1                          [M, D]    gemm + residual  (:57,76)
// Head:  out = LN_D(mean_n X) . Wh^T + bh               [B, G]                     (:103-106)
#include <mutex>
#include <unordered_map>
#include "vis.h"
#include "elementwise.h"
#include "gemm.h"

void sq_vis_bufs(const sq_vis_config& c, int dtype, int B, int save, char* base, VisBufs* o) {
    Arena a{base, 0};
    const size_t es = sq_dtype_size(dtype);
    const size_t pes = sq_dtype_size(sq_vis_preact_dtype(dtype));
    const size_t M = (size_t)B * c.num_clusters, D = c.input_dim, HD = (size_t)c.nheads * SQ_HEAD_DIM;
    const int L = save ? c.depth : 1;
    o->nsave = L;
    const int nX = save ? c.depth + 1 : 1;
    for (int l = 0; l <= SQ_MAX_DEPTH; ++l) {
        const int src = l < nX ? l : 0;
        if (l < nX) {
            o->Xin[l] = (float*)a.take(M * D * 4);
            o->Xin_lp[l] = dtype == SQ_BF16 ? a.take(M * D * 2) : (void*)o->Xin[l];
        } else {
            o->Xin[l] = o->Xin[src];
            o->Xin_lp[l] = o->Xin_lp[src];
        }
    }
    for (int l = 0; l < SQ_MAX_DEPTH; ++l) {
        if (l < L) {
            o->X1[l] = (float*)a.take(M * D * 4);
            o->X1_lp[l] = o->X1[l];
            o->Xbar32[l] = (float*)a.take((size_t)B * D * 4);
            o->Xbar[l] = dtype == SQ_BF16 ? a.take((size_t)B * D * 2) : (void*)o->Xbar32[l];
            o->F[l] = (float*)a.take(M * HD * 4);
            o->Lf[l] = a.take(M * HD * es);
            o->Sm[l] = (float*)a.take((size_t)B * HD * 4);
            o->Ts[l] = a.take((size_t)B * HD * es);
            o->Cs[l] = (float*)a.take((size_t)B * HD * 4);
            o->P[l] = save ? a.take(M * HD * pes) : nullptr;     // bf16 mode: the GELU' source is kept in bf16 (half the traffic)
            o->O[l] = a.take(M * HD * es);
            o->Y[l] = a.take(M * D * es);
            o->U[l] = save ? a.take(M * D * pes) : nullptr;
            o->H1[l] = a.take(M * D * es);
        } else {
            o->X1[l] = o->X1[0]; o->X1_lp[l] = o->X1_lp[0]; o->Xbar32[l] = o->Xbar32[0]; o->Xbar[l] = o->Xbar[0];
            o->F[l] = o->F[0]; o->Lf[l] = o->Lf[0]; o->Sm[l] = o->Sm[0]; o->Ts[l] = o->Ts[0]; o->Cs[l] = o->Cs[0];
            o->P[l] = o->P[0]; o->O[l] = o->O[0]; o->Y[l] = o->Y[0]; o->U[l] = o->U[0]; o->H1[l] = o->H1[0];
        }
    }
    o->skws_bytes = (size_t)16 * B * (HD > D ? HD : D) * 4;
    o->skws = (float*)a.take(o->skws_bytes);
    o->x1m = (float*)a.take((size_t)B * D * 4);
    o->xm = (float*)a.take((size_t)B * D * 4);
    o->xn = a.take((size_t)B * D * es);
    o->bytes = sq_align_up(a.off, 256);
}

namespace {
std::mutex g_saved_mu;
std::unordered_map<const void*, int> g_saved_stream;       // forward workspace -> 1 lean bf16 rows / 0 fp32 rows (a handful of entries)
}  // namespace

void sq_vis_note_saved_stream(const void* workspace, bool lean) {
    std::lock_guard<std::mutex> lk(g_saved_mu);
    if (g_saved_stream.size() > 4096) g_saved_stream.clear();       // workspaces come and go with the caller's allocator
    g_saved_stream[workspace] = lean ? 1 : 0;
}

int sq_vis_saved_stream(const void* workspace) {
    std::lock_guard<std::mutex> lk(g_saved_mu);
    auto it = g_saved_stream.find(workspace);
    return it == g_saved_stream.end() ? -1 : it->second;
}

static int check_cfg(const sq_vis_config* c) {
    SQ_REQUIRE(c != nullptr, "vis: null config");
    SQ_REQUIRE(c->input_dim > 0 && c->input_dim % 64 == 0 && c->input_dim <= 4096, "vis: input_dim=%d must be a multiple of 64, <= 4096", c->input_dim);
    SQ_REQUIRE(c->depth >= 1 && c->depth <= SQ_MAX_DEPTH, "vis: depth=%d out of [1,%d]", c->depth, SQ_MAX_DEPTH);
    SQ_REQUIRE(c->nheads >= 1 && c->nheads <= 64, "vis: nheads=%d out of [1,64]", c->nheads);
    SQ_REQUIRE(c->num_outputs >= 1, "vis: num_outputs=%d", c->num_outputs);
    SQ_REQUIRE(c->num_clusters >= 1, "vis: num_clusters=%d", c->num_clusters);
    return SQ_OK;
}

extern "C" int sq_vis_layout_init(const sq_vis_config* c, sq_vis_layout* out) {
    if (int e = check_cfg(c)) return e;
    SQ_REQUIRE(out != nullptr, "vis: null layout");
    const int64_t D = c->input_dim, H = c->nheads, HD = H * SQ_HEAD_DIM, G = c->num_outputs;
    int64_t off = 0;
    auto take = [&](int64_t n) { off = (off + 7) / 8 * 8; const int64_t o = off; off += n; return o; };
    out->pos = take((int64_t)c->num_clusters * D);
    for (int l = 0; l < SQ_MAX_DEPTH; ++l) {
        sq_vis_layer_offsets& L = out->layer[l];
        if (l >= c->depth) { L = sq_vis_layer_offsets{-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1}; continue; }
        L.f_w = take(HD * D);  L.f_b = take(HD);
        L.s_w = take(HD * D);  L.s_b = take(HD);
        L.lnf_g = take(HD);    L.lnf_b = take(HD);
        L.lns_g = take(HD);    L.lns_b = take(HD);
        L.c_w = take(HD * 2 * SQ_HEAD_DIM);  L.c_b = take(HD);
        L.proj_w = take(D * HD);  L.proj_b = take(D);
        L.ffln_g = take(D);    L.ffln_b = take(D);
        L.ff1_w = take(D * D); L.ff1_b = take(D);
        L.ff2_w = take(D * D); L.ff2_b = take(D);
    }
    out->head_ln_g = take(D);
    out->head_ln_b = take(D);
    out->head_w = take(G * D);
    out->head_b = take(G);
    out->total = (off + 7) / 8 * 8;
    return SQ_OK;
}

extern "C" size_t sq_vis_workspace_bytes(const sq_vis_config* c, int dtype, int batch, int save_for_backward) {
    if (check_cfg(c) != SQ_OK || batch < 1) return 0;
    VisBufs b;
    sq_vis_bufs(*c, dtype, batch, save_for_backward, nullptr, &b);
    return b.bytes;
}

namespace {

struct Lin {      // one nn.Linear on the flat parameter buffer
    const void* w; int ldw; size_t w_bytes; const float* b;
};

}  // namespace

extern "C" int sq_vis_forward(const sq_vis_config* c, int dtype, const float* params, const void* params_lp, const float* x,
                              float* out, int B, int save, void* workspace, size_t workspace_bytes, sq_stream_t stream_) {
    SQ_REQUIRE(x && out, "vis_forward: null pointer");
    return sq_vis_forward_ex(c, dtype, params, params_lp, x, nullptr, nullptr, 0, out, nullptr, B, save, workspace, workspace_bytes, stream_);
}

static int vis_forward_impl(const sq_vis_config* c, int dtype, const float* params, const void* params_lp, const float* x,
                            const float* gather_src, const int32_t* gather_idx, int gather_rows, const float* f_tile, const float* f_pos,
                            float* out, float* head_in, int B, int save, void* workspace, size_t workspace_bytes, sq_stream_t stream_);

extern "C" int sq_vis_forward_ex(const sq_vis_config* c, int dtype, const float* params, const void* params_lp, const float* x,
                                 const float* gather_src, const int32_t* gather_idx, int gather_rows, float* out, float* head_in,
                                 int B, int save, void* workspace, size_t workspace_bytes, sq_stream_t stream_) {
    return vis_forward_impl(c, dtype, params, params_lp, x, gather_src, gather_idx, gather_rows, nullptr, nullptr, out, head_in, B, save,
                            workspace, workspace_bytes, stream_);
}

// The gather form of sq_vis_forward_ex with the FIRST layer's local projection taken from per-tile projections: f(x) is linear in
// x = tile feature + position, so F[(b, n)] = f_tile[idx[b, n]] + f_pos[n] with f_tile = cache . Wf^T computed once per TILE (50 000 rows
// at BASELINE config 5) instead of once per window token (4.78 M rows) -- 1/24 of the sliding-window path's large products.
extern "C" int sq_vis_forward_tiles(const sq_vis_config* c, int dtype, const float* params, const void* params_lp, const float* gather_src,
                                    const int32_t* gather_idx, int gather_rows, const float* f_tile, const float* f_pos, float* head_in,
                                    int B, void* workspace, size_t workspace_bytes, sq_stream_t stream_) {
    SQ_REQUIRE(f_tile && f_pos && gather_src && gather_idx && head_in, "vis_forward_tiles: null pointer");
    return vis_forward_impl(c, dtype, params, params_lp, nullptr, gather_src, gather_idx, gather_rows, f_tile, f_pos, nullptr, head_in, B, 0,
                            workspace, workspace_bytes, stream_);
}

static int vis_forward_impl(const sq_vis_config* c, int dtype, const float* params, const void* params_lp, const float* x,
                            const float* gather_src, const int32_t* gather_idx, int gather_rows, const float* f_tile, const float* f_pos,
                            float* out, float* head_in, int B, int save, void* workspace, size_t workspace_bytes, sq_stream_t stream_) {
    if (int e = check_cfg(c)) return e;
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(dtype == SQ_F32 || dtype == SQ_BF16, "vis_forward: dtype %d", dtype);
    SQ_REQUIRE(params && workspace, "vis_forward: null pointer");
    SQ_REQUIRE((x != nullptr) != (gather_src != nullptr && gather_idx != nullptr), "vis_forward: give either x or (gather_src, gather_idx)");
    SQ_REQUIRE(x || gather_rows >= 1, "vis_forward: gather_rows=%d", gather_rows);
    SQ_REQUIRE((out != nullptr) != (head_in != nullptr), "vis_forward: give either out (predictions) or head_in (the head's input)");
    SQ_REQUIRE(dtype == SQ_F32 || params_lp, "vis_forward: bf16 mode needs the bf16 parameter shadow");
    SQ_REQUIRE(B >= 1, "vis_forward: batch=%d", B);
    sq_vis_layout lay;
    if (int e = sq_vis_layout_init(c, &lay)) return e;
    VisBufs w;
    sq_vis_bufs(*c, dtype, B, save, (char*)workspace, &w);
    if (w.bytes > workspace_bytes) {
        sq_set_error("vis_forward: workspace %zu < required %zu", workspace_bytes, w.bytes);
        return SQ_ERR_WORKSPACE;
    }
    const int N = c->num_clusters, D = c->input_dim, H = c->nheads, HD = H * SQ_HEAD_DIM, G = c->num_outputs;
    const int M = B * N;
    const size_t es = sq_dtype_size(dtype);
    const bool lp = dtype == SQ_BF16;
    const char* wbase = lp ? (const char*)params_lp : (const char*)params;
    auto W = [&](int64_t off) { return (const void*)(wbase + (size_t)off * es); };
    auto Wrem = [&](int64_t off) { return (size_t)(lay.total - off) * es; };
    auto Pf = [&](int64_t off) { return params + off; };

    const bool stream16 = sq_vis_lean_stream(dtype);
    float* const x32 = stream16 ? nullptr : w.Xin[0];      // bf16-only stream (inference and training): the fp32 rows have no reader
    if (save) sq_vis_note_saved_stream(workspace, stream16);
    if (x) {
        if (int e = sq_k_add_pos(x, Pf(lay.pos), x32, lp ? (bf16_t*)w.Xin_lp[0] : nullptr, B, N, D, st)) return e;
    } else {
        if (int e = sq_k_add_pos_gather(gather_src, gather_idx, Pf(lay.pos), x32, lp ? (bf16_t*)w.Xin_lp[0] : nullptr, B, N, D, st)) return e;
    }
    // Inference in bf16 mode keeps the residual stream x in bf16 only (the operand copy IS the stream): the two residual
    // products of a layer then read and write 2-byte rows instead of 4-byte rows plus a 2-byte copy, LayerNorm and the
    // token mean read half the bytes -- about a third of a layer's HBM traffic.  The fp32 stream stays for training
    // (save_for_backward), for fp32 mode, and on request (SQ_VIS_FP32_STREAM=1).
    // Training in bf16 mode (save_for_backward) does the same since round 5 -- BASELINE config 2 is "forward+backward bf16": the
    // layer inputs, X1 and the pre-LayerNorm(64) tensor F are stored in bf16 only (what the backward pass re-reads), fp32 master
    // weights and AdamW state stay.  SQ_VIS_FP32_STREAM=1 brings the fp32 stream back for both.
    // the summary branch (token mean + three small products per layer) beside the f projection on a helper stream -- while the
    // branch is small.  At the spatial path's batches (M = 204 800 rows and more) the token mean is a 0.4 GB pass that takes more
    // from the projection than running it first costs: 437 -> 430 ms per 50 000-tile slide on ONE stream (round 5)
    SqSideStream* fs = (lp && !sq_env_flag("SQ_FWD_ONE_STREAM") && M < 65536) ? sq_side_stream(1, 4 * SQ_MAX_DEPTH) : nullptr;
    hipStream_t s2 = fs ? fs->stream : st;
    int ev_next = 0;

    for (int l = 0; l < c->depth; ++l) {
        const sq_vis_layer_offsets& L = lay.layer[l];
        const int s = save ? l : 0;
        float* Xin = w.Xin[s];
        const void* Xin_t = w.Xin_lp[s];
        float* Xout = w.Xin[save ? l + 1 : 0];
        void* Xout_lp = w.Xin_lp[save ? l + 1 : 0];

        // summary branch (per-slide tensors, four small launches) on a helper stream beside the f projection
        hipEvent_t ev_cs = nullptr;
        if (fs) {
            hipEvent_t ev = fs->events[ev_next++];
            SQ_HIP_CHECK(hipEventRecord(ev, st));
            SQ_HIP_CHECK(hipStreamWaitEvent(s2, ev, 0));
        }
        if (int e = sq_k_token_mean_any(stream16 ? Xin_t : (const void*)Xin, stream16 ? SQ_BF16 : SQ_F32, w.Xbar32[s],
                                        lp ? (bf16_t*)w.Xbar[s] : nullptr, B, N, D, s2)) return e;
        if (lp) {
            // Sm = Xbar Ws^T + bs;  Ts = GELU(LN64(Sm));  Cs = Ts_h Wc_h[:, 64:]^T + bc_h  -- one launch (summary.hip)
            if (int e = sq_launch_summary_fwd(w.Xbar[s], W(L.s_w), Pf(L.s_b), Pf(L.lns_g), Pf(L.lns_b), W(L.c_w), Pf(L.c_b), w.Sm[s],
                                              w.Ts[s], w.Cs[s], B, D, H, s2)) return e;
        } else {
            {   // Sm = Xbar Ws^T + bs
                GemmArgs g; g.A = w.Xbar[s]; g.lda = D; g.a_bytes = (size_t)B * D * es;
                g.B = W(L.s_w); g.ldb = D; g.b_bytes = Wrem(L.s_w); g.bias = Pf(L.s_b);
                g.C = w.Sm[s]; g.ldc = HD; g.M = B; g.N = HD; g.K = D; g.splitk_ws = w.skws; g.splitk_ws_bytes = w.skws_bytes;
                if (int e = sq_launch_gemm(g, dtype, s2)) return e;
            }
            if (int e = sq_k_ln64_gelu(w.Sm[s], Pf(L.lns_g), Pf(L.lns_b), w.Ts[s], dtype, B, HD, s2)) return e;
            {   // Cs[b, h] = Ts[b, h] . Wc_h[:, 64:128]^T + bc_h     (cat order: [local, summary], tformer_lin.py:24)
                GemmArgs g; g.A = w.Ts[s]; g.lda = HD; g.a_bytes = (size_t)B * HD * es; g.sA = SQ_HEAD_DIM;
                g.B = W(L.c_w + SQ_HEAD_DIM); g.ldb = 2 * SQ_HEAD_DIM; g.b_bytes = Wrem(L.c_w + SQ_HEAD_DIM); g.sB = SQ_HEAD_DIM * 2 * SQ_HEAD_DIM;
                g.bias = Pf(L.c_b); g.sBias = SQ_HEAD_DIM;
                g.C = w.Cs[s]; g.ldc = HD; g.sC = SQ_HEAD_DIM; g.M = B; g.N = SQ_HEAD_DIM; g.K = SQ_HEAD_DIM; g.batch = H;
                if (int e = sq_launch_gemm(g, dtype, s2)) return e;
            }
        }
        if (fs) { ev_cs = fs->events[ev_next++]; SQ_HIP_CHECK(hipEventRecord(ev_cs, s2)); }
        bool combined = false;
        if (l == 0 && f_tile) {
            // F = f_tile[idx] + f_pos (the projection of tile feature + position, taken per tile by the caller), LayerNorm(64) + GELU in
            // one streaming pass; the combiner follows as its own launch
            if (int e = sq_k_gather_ln64_gelu(f_tile, f_pos, gather_idx, Pf(L.lnf_g), Pf(L.lnf_b), w.Lf[s], dtype, B, N, HD, st)) return e;
        } else
        {   // Lf = GELU(LN64(F)),  F = X Wf^T + bf: LayerNorm + GELU in the epilogue (a head's 64 columns sit in 8
            // lanes of the staged tile); F itself is only written when the backward pass will need it
            GemmArgs g; g.A = Xin_t; g.lda = D; g.a_bytes = (size_t)M * D * es;
            g.B = W(L.f_w); g.ldb = D; g.b_bytes = Wrem(L.f_w); g.bias = Pf(L.f_b);
            g.ln64_g = Pf(L.lnf_g); g.ln64_b = Pf(L.lnf_b); g.act = SQ_ACT_GELU;
            if (save) { g.Cpre = w.F[s]; g.pre_dtype = stream16 ? SQ_BF16 : SQ_F32; g.ldpre = HD; }
            g.C = w.Lf[s]; g.out_dtype = dtype; g.ldc = HD; g.M = M; g.N = HD; g.K = D;
            // Inference at large batch in bf16 mode (the spatial path; the summary branch ran first, on this stream): the combiner
            //   O[m, h] = GELU(Lf[m, h] . Wc_h[:, 0:64]^T + Cs[slide(m), h])
            // runs in THIS launch's epilogue (gemm_p8.hip: the head's 16 x 64 LayerNorm + GELU slab is the A operand of eight more
            // MFMAs) -- Lf is neither written nor read and the batched 64 x 64 launch is gone (SQ_FWD_NO_FUSED_COMB=1: two launches)
            if (lp && !save && !fs && HD % 256 == 0 && !sq_env_flag("SQ_FWD_NO_FUSED_COMB") && sq_gemm_takes_p8_256(g)) {
                g.comb_w = W(L.c_w); g.comb_rb = w.Cs[s]; g.comb_ldrb = HD; g.comb_rpg = N;
                g.C = w.O[s];
                combined = true;
            }
            if (int e = sq_launch_gemm(g, dtype, st)) return e;
        }
        if (ev_cs) SQ_HIP_CHECK(hipStreamWaitEvent(st, ev_cs, 0));
        if (!combined) {   // O[m, h] = GELU(Lf[m, h] . Wc_h[:, 0:64]^T + Cs[slide(m), h])
            GemmArgs g; g.A = w.Lf[s]; g.lda = HD; g.a_bytes = (size_t)M * HD * es; g.sA = SQ_HEAD_DIM;
            g.B = W(L.c_w); g.ldb = 2 * SQ_HEAD_DIM; g.b_bytes = Wrem(L.c_w); g.sB = SQ_HEAD_DIM * 2 * SQ_HEAD_DIM;
            g.rowbias = w.Cs[s]; g.ldrb = HD; g.sRb = SQ_HEAD_DIM; g.rows_per_group = N;
            g.act = SQ_ACT_GELU;
            if (save) { g.Cpre = w.P[s]; g.pre_dtype = sq_vis_preact_dtype(dtype); g.ldpre = HD; g.sPre = SQ_HEAD_DIM; }   // pre-activations: backward only
            g.C = w.O[s]; g.out_dtype = dtype; g.ldc = HD; g.sC = SQ_HEAD_DIM;
            g.M = M; g.N = SQ_HEAD_DIM; g.K = SQ_HEAD_DIM; g.batch = H;
            if (int e = sq_launch_gemm(g, dtype, st)) return e;
        }
        {   // X1 = O Wp^T + bp + X
            GemmArgs g; g.A = w.O[s]; g.lda = HD; g.a_bytes = (size_t)M * HD * es;
            g.B = W(L.proj_w); g.ldb = HD; g.b_bytes = Wrem(L.proj_w); g.bias = Pf(L.proj_b);
            g.res = Xin; g.ldres = D; g.C = w.X1[s]; g.ldc = D; g.M = M; g.N = D; g.K = HD;
            if (stream16) { g.res = Xin_t; g.res_dtype = SQ_BF16; g.out_dtype = SQ_BF16; }     // X1 as bf16 in the same buffer
            if (int e = sq_launch_gemm(g, dtype, st)) return e;
        }
        if (int e = sq_k_ln_rows_any(w.X1[s], stream16 ? SQ_BF16 : SQ_F32, Pf(L.ffln_g), Pf(L.ffln_b), w.Y[s], dtype, M, D, nullptr, nullptr, st)) return e;
        {   // H1 = GELU(Y W1^T + b1)
            GemmArgs g; g.A = w.Y[s]; g.lda = D; g.a_bytes = (size_t)M * D * es;
            g.B = W(L.ff1_w); g.ldb = D; g.b_bytes = Wrem(L.ff1_w); g.bias = Pf(L.ff1_b);
            g.act = SQ_ACT_GELU;
            if (save) { g.Cpre = w.U[s]; g.pre_dtype = sq_vis_preact_dtype(dtype); g.ldpre = D; }
            g.C = w.H1[s]; g.out_dtype = dtype; g.ldc = D; g.M = M; g.N = D; g.K = D;
            if (int e = sq_launch_gemm(g, dtype, st)) return e;
        }
        if (!save && l == c->depth - 1) {
            // Inference, last layer: only the token mean of X = H1 W2^T + b2 + X1 is read (tformer_lin.py:103), and the mean commutes
            // with the linear map: mean_n(X) = mean_n(H1) W2^T + b2 + mean_n(X1).  One [B, D] x [D, D] product instead of the
            // [B N, D] x [D, D] one -- 1/24 of the forward pass's large products (config 5: 13 ms of 337 per 50 000-tile slide).
            // Means in fp32 (the kernel the summary branch uses); training (save_for_backward) keeps the per-token form.
            if (int e = sq_k_token_mean_any(w.H1[s], dtype, w.Xbar32[s], lp ? (bf16_t*)w.Xbar[s] : nullptr, B, N, D, st)) return e;
            if (int e = sq_k_token_mean_any(w.X1[s], stream16 ? SQ_BF16 : SQ_F32, w.x1m, nullptr, B, N, D, st)) return e;
            GemmArgs g; g.A = w.Xbar[s]; g.lda = D; g.a_bytes = (size_t)B * D * es;
            g.B = W(L.ff2_w); g.ldb = D; g.b_bytes = Wrem(L.ff2_w); g.bias = Pf(L.ff2_b);
            g.res = w.x1m; g.ldres = D; g.C = w.xm; g.ldc = D; g.M = B; g.N = D; g.K = D;
            // (no split-K workspace: the number of K slices would follow the batch size, and with it the fp32 summation order --
            // a window's result must not depend on how the windows are batched, tests/test_gpu_spatial.py)
            if (int e = sq_launch_gemm(g, dtype, st)) return e;
        } else {   // X = H1 W2^T + b2 + X1
            GemmArgs g; g.A = w.H1[s]; g.lda = D; g.a_bytes = (size_t)M * D * es;
            g.B = W(L.ff2_w); g.ldb = D; g.b_bytes = Wrem(L.ff2_w); g.bias = Pf(L.ff2_b);
            g.res = w.X1[s]; g.ldres = D; g.C = Xout; g.ldc = D;
            g.C2 = lp ? (bf16_t*)Xout_lp : nullptr; g.ldc2 = D; g.M = M; g.N = D; g.K = D;
            if (stream16) { g.res_dtype = SQ_BF16; g.C = Xout_lp; g.out_dtype = SQ_BF16; g.C2 = nullptr; }
            if (int e = sq_launch_gemm(g, dtype, st)) return e;
        }
    }
    if (save) {
        const float* Xfin = w.Xin[c->depth];
        if (int e = sq_k_token_mean_any(stream16 ? w.Xin_lp[c->depth] : (const void*)Xfin, stream16 ? SQ_BF16 : SQ_F32, w.xm, nullptr, B, N, D, st)) return e;
    }       // (inference: w.xm was written by the last layer's product above)
    if (head_in)        // the caller applies the (linear) head itself, e.g. after averaging over windows: LN output in fp32
        return sq_k_ln_rows(w.xm, Pf(lay.head_ln_g), Pf(lay.head_ln_b), head_in, SQ_F32, B, D, nullptr, nullptr, st);
    if (int e = sq_k_ln_rows(w.xm, Pf(lay.head_ln_g), Pf(lay.head_ln_b), w.xn, dtype, B, D, nullptr, nullptr, st)) return e;
    {   // out = xn Wh^T + bh
        GemmArgs g; g.A = w.xn; g.lda = D; g.a_bytes = (size_t)B * D * es;
        g.B = W(lay.head_w); g.ldb = D; g.b_bytes = Wrem(lay.head_w); g.bias = Pf(lay.head_b);
        g.C = out; g.ldc = G; g.M = B; g.N = G; g.K = D;
        if (int e = sq_launch_gemm(g, dtype, st)) return e;
    }
    return SQ_OK;
}

extern "C" int sq_cast_f32_to_bf16(const float* src, void* dst, size_t n, sq_stream_t stream) {
    SQ_REQUIRE(src && dst, "cast: null pointer");
    return sq_k_f32_to_bf16(src, (bf16_t*)dst, n, (hipStream_t)stream);
}

extern "C" int sq_cast_bf16_to_f32(const void* src, float* dst, size_t n, sq_stream_t stream) {
    SQ_REQUIRE(src && dst, "cast: null pointer");
    return sq_k_bf16_to_f32((const bf16_t*)src, dst, n, (hipStream_t)stream);
}

extern "C" int sq_linear(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual,
                         int ldres, int res_dtype, int act, void* C, int out_dtype, int ldc, int M, int N, int K, void* workspace,
                         size_t workspace_bytes, sq_stream_t stream) {
    SQ_REQUIRE(A && W && C, "linear: null pointer");
    SQ_REQUIRE(dtype == SQ_F32 || dtype == SQ_BF16, "linear: dtype %d", dtype);
    GemmArgs g;
    const size_t es = sq_dtype_size(dtype);
    g.A = A; g.lda = lda; g.a_bytes = ((size_t)(M - 1) * lda + K) * es;
    g.B = W; g.ldb = ldw; g.b_bytes = ((size_t)(N - 1) * ldw + K) * es;
    g.bias = bias; g.res = residual; g.ldres = ldres; g.res_dtype = res_dtype; g.act = act;
    g.C = C; g.out_dtype = out_dtype; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.splitk_ws = (float*)workspace; g.splitk_ws_bytes = workspace_bytes;
    return sq_launch_gemm(g, dtype, (hipStream_t)stream);
}

extern "C" int sq_linear_weight_grad(int dtype, const void* dY, int lddy, const void* X, int ldx, float* dW, int lddw, float* dbias,
                                     int n_out, int n_in, int n_tokens, void* workspace, size_t workspace_bytes, sq_stream_t stream) {
    SQ_REQUIRE(dY && X && dW, "linear_weight_grad: null pointer");
    GemmArgs g;
    const size_t es = sq_dtype_size(dtype);
    g.A = dY; g.lda = lddy; g.a_bytes = (size_t)n_tokens * lddy * es;
    g.B = X; g.ldb = ldx; g.b_bytes = (size_t)n_tokens * ldx * es;
    g.C = dW; g.ldc = lddw; g.M = n_out; g.N = n_in; g.K = n_tokens;
    g.colsum_a = dbias;
    g.splitk_ws = (float*)workspace; g.splitk_ws_bytes = workspace_bytes;
    return sq_launch_gemm_tn(g, dtype, (hipStream_t)stream);
}

extern "C" int sq_linear_weight_grad_group(int dtype, int n_members, const void* const* dY, const void* const* X, float* const* dW,
                                           float* const* dbias, int lddy, int ldx, int lddw, int n_out, int n_in, int n_tokens,
                                           sq_stream_t stream) {
    SQ_REQUIRE(dY && X && dW && n_members >= 1 && n_members <= 4, "linear_weight_grad_group: 1..4 members (got %d)", n_members);
    GemmArgs g;
    const size_t es = sq_dtype_size(dtype);
    g.lda = lddy; g.a_bytes = (size_t)n_tokens * lddy * es;
    g.ldb = ldx; g.b_bytes = (size_t)n_tokens * ldx * es;
    g.ldc = lddw; g.M = n_out; g.N = n_in; g.K = n_tokens;
    g.ngroup = n_members;
    for (int i = 0; i < n_members; ++i) {
        g.gA[i] = dY[i]; g.gB[i] = X[i]; g.gC[i] = dW[i];
        g.gcs[i] = dbias ? dbias[i] : nullptr;
    }
    return sq_launch_gemm_tn(g, dtype, (hipStream_t)stream);
}
