// ViS backward: gradients of every parameter (flat buffer, same layout as the parameters) and
// optionally of the input tokens, from d(loss)/d(out).  Replaces torch autograd over
// src/tformer_lin.py in the training loop src/vit.py:163-180.
//
// "dX = dY . W" products are NT GEMMs on a transposed copy of the (small) weight; "dW = dY^T . X"
// products contract over tokens and run on the TN kernel straight from the token-major activations
// the forward pass saved (no transposed activation copies); the same launch sums dY's columns into
// the bias gradient.
#include "elementwise.h"
#include "gemm.h"
#include "vis.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {

// Tensors the weight-gradient (TN) products read exist once per LAYER in bf16 mode, so those products can trail
// the dX chain on a second stream without write-after-read hazards; fp32 mode (one stream) aliases one buffer.
struct LayerG {
    void* dXin_lp;                   // [M, D] T   gradient entering the layer (operand copy)
    void* dX1_lp;                    // [M, D] T   gradient after the feed-forward block
    void* dU;                        // [M, D] T
    void* dP;                        // [M, HD] T
    void* dF;                        // [M, HD] T
    void* dSm;                       // [B, HD] T
    void* dCs_lp;                    // [B, HD] T
};

struct BwdBufs {
    float* dXa; float* dXb;          // [M, D] f32 running gradient (ping-pong, main stream only)
    LayerG lg[SQ_MAX_DEPTH];
    float* dY;                       // [M, D] f32
    float* dLf;                      // [M, HD] f32
    void* doutT;                     // [Gp, Bp] T  d(loss)/d(out) transposed (gene-major): the head's dX product contracts over genes
    struct LayerT { void *ff2, *ff1, *proj, *wc_lf, *wc_ts, *s, *f; } wt[SQ_MAX_DEPTH];
    float* dCs; float* dTs; float* dXbar;   // [B, HD] / [B, D]
    void* dout_lp;                   // [B, Gp] T
    float* dxn; float* dxm;          // [B, D]
    float* red_ws;                   // reduction scratch
    float* part_ws[3];               // partial [dg | db] rows of a layer's three LayerNorms (one reduction launch per layer)
    float* skws; size_t skws_bytes;  // split-K scratch (dW products have K = tokens and few output tiles)
    float* skws_side;                // the same for the second stream
    float* red_ws2; float* skws2; size_t skws2_bytes;   // scratch of the summary-branch stream
    size_t bytes;
};

void bwd_bufs(const sq_vis_config& c, int dtype, int B, char* base, BwdBufs* o) {
    Arena a{base, 0};
    const size_t es = sq_dtype_size(dtype);
    const bool lp = dtype == SQ_BF16;
    const size_t M = (size_t)B * c.num_clusters, D = c.input_dim, HD = (size_t)c.nheads * SQ_HEAD_DIM, G = c.num_outputs;
    const size_t Gp = sq_align_up(G, 8);
    const size_t W = D > HD ? D : HD;
    o->dXa = (float*)a.take(M * D * 4);
    o->dXb = (float*)a.take(M * D * 4);
    o->dY = (float*)a.take(M * D * 4);
    o->dLf = (float*)a.take(M * HD * 4);
    o->dCs = (float*)a.take((size_t)B * HD * 4);
    for (int l = 0; l < c.depth; ++l) {
        LayerG& g = o->lg[l];
        if (lp) {
            g.dXin_lp = a.take(M * D * 2); g.dX1_lp = a.take(M * D * 2); g.dU = a.take(M * D * 2);
            g.dP = a.take(M * HD * 2); g.dF = a.take(M * HD * 2);
            g.dSm = a.take((size_t)B * HD * 2); g.dCs_lp = a.take((size_t)B * HD * 2);
        } else if (l == 0) {
            // fp32: the operand "copies" are the f32 tensors themselves (dXin / dX1 are set per layer by the caller)
            g.dXin_lp = nullptr; g.dX1_lp = nullptr;
            g.dU = a.take(M * D * 4); g.dP = a.take(M * HD * 4); g.dF = a.take(M * HD * 4);
            g.dSm = a.take((size_t)B * HD * 4); g.dCs_lp = o->dCs;
        } else {
            g = o->lg[0];
        }
    }
    o->doutT = a.take(Gp * sq_align_up((size_t)B, 8) * es);
    for (int l = 0; l < c.depth; ++l) {
        o->wt[l].ff2 = a.take(D * D * es); o->wt[l].ff1 = a.take(D * D * es);
        o->wt[l].proj = a.take(HD * D * es);
        o->wt[l].wc_lf = a.take((size_t)c.nheads * 64 * 64 * es); o->wt[l].wc_ts = a.take((size_t)c.nheads * 64 * 64 * es);
        o->wt[l].s = a.take(D * HD * es); o->wt[l].f = a.take(D * HD * es);
    }
    o->dTs = (float*)a.take((size_t)B * HD * 4);
    o->dXbar = (float*)a.take((size_t)B * D * 4);
    o->dout_lp = a.take((size_t)B * Gp * es);
    o->dxn = (float*)a.take((size_t)B * D * 4); o->dxm = (float*)a.take((size_t)B * D * 4);
    size_t red = sq_ln_bwd_ws_floats((int)W);
    const size_t cs = sq_colsum_ws_floats((int)(G > W ? G : W));
    if (cs > red) red = cs;
    o->red_ws = (float*)a.take(red * 4);
    for (int i = 0; i < 3; ++i) o->part_ws[i] = (float*)a.take(red * 4);
    o->skws_bytes = (size_t)16 * W * W * 4;
    o->skws = (float*)a.take(o->skws_bytes);
    o->skws_side = lp ? (float*)a.take(o->skws_bytes) : o->skws;
    o->red_ws2 = lp ? (float*)a.take(red * 4) : o->red_ws;
    o->skws2_bytes = lp ? (size_t)32 * B * W * 4 : o->skws_bytes;
    o->skws2 = lp ? (float*)a.take(o->skws2_bytes) : o->skws;
    o->bytes = sq_align_up(a.off, 256);
}

}  // namespace

extern "C" size_t sq_vis_backward_workspace_bytes(const sq_vis_config* c, int dtype, int batch) {
    sq_vis_layout lay;
    if (sq_vis_layout_init(c, &lay) != SQ_OK || batch < 1) return 0;
    BwdBufs b;
    bwd_bufs(*c, dtype, batch, nullptr, &b);
    return b.bytes;
}

extern "C" int sq_vis_grad_buckets(const sq_vis_config* c, int64_t* lo, int64_t* hi, int cap) {
    sq_vis_layout lay;
    if (int e = sq_vis_layout_init(c, &lay)) return e;
    SQ_REQUIRE(lo && hi && cap >= c->depth + 1, "vis_grad_buckets: need room for %d buckets", c->depth + 1);
    lo[0] = lay.head_ln_g; hi[0] = lay.total;                     // the head's gradients are complete first
    for (int i = 0; i < c->depth; ++i) {                          // then the layers, last to first
        const int l = c->depth - 1 - i;
        lo[1 + i] = l == 0 ? 0 : lay.layer[l].f_w;                // pos_emb1D (finished last) rides with layer 0
        hi[1 + i] = l + 1 < c->depth ? lay.layer[l + 1].f_w : lay.head_ln_g;
    }
    return c->depth + 1;
}

extern "C" int sq_vis_backward(const sq_vis_config* c, int dtype, const float* params, const void* params_lp,
                               const float* grad_out, float* grad_params, float* grad_x, int B, void* fwd_workspace,
                               size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes,
                               sq_stream_t stream_) {
    return sq_vis_backward_buckets(c, dtype, params, params_lp, grad_out, grad_params, grad_x, B, fwd_workspace, fwd_workspace_bytes,
                                   bwd_workspace, bwd_workspace_bytes, stream_, nullptr, 0);
}

extern "C" int sq_vis_backward_buckets(const sq_vis_config* c, int dtype, const float* params, const void* params_lp,
                                       const float* grad_out, float* grad_params, float* grad_x, int B, void* fwd_workspace,
                                       size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes,
                                       sq_stream_t stream_, const sq_event_t* bucket_events, int n_bucket_events) {
    sq_vis_layout lay;
    if (int e = sq_vis_layout_init(c, &lay)) return e;
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(dtype == SQ_F32 || dtype == SQ_BF16, "vis_backward: dtype %d", dtype);
    SQ_REQUIRE(params && grad_out && grad_params && fwd_workspace && bwd_workspace, "vis_backward: null pointer");
    SQ_REQUIRE(dtype == SQ_F32 || params_lp, "vis_backward: bf16 mode needs the bf16 parameter shadow");
    SQ_REQUIRE(n_bucket_events == 0 || (bucket_events && n_bucket_events == c->depth + 1),
               "vis_backward: %d bucket events given, sq_vis_grad_buckets has %d", n_bucket_events, c->depth + 1);
    auto bucket_done = [&](int i) {                               // bucket i of sq_vis_grad_buckets is final from here on
        if (n_bucket_events == 0) return (int)SQ_OK;
        SQ_HIP_CHECK(hipEventRecord((hipEvent_t)bucket_events[i], st));
        return (int)SQ_OK;
    };
    VisBufs w;
    sq_vis_bufs(*c, dtype, B, 1, (char*)fwd_workspace, &w);
    BwdBufs b;
    bwd_bufs(*c, dtype, B, (char*)bwd_workspace, &b);
    if (w.bytes > fwd_workspace_bytes || b.bytes > bwd_workspace_bytes) {
        sq_set_error("vis_backward: workspaces %zu/%zu < required %zu/%zu", fwd_workspace_bytes, bwd_workspace_bytes, w.bytes, b.bytes);
        return SQ_ERR_WORKSPACE;
    }
    const int N = c->num_clusters, D = c->input_dim, H = c->nheads, HD = H * SQ_HEAD_DIM, G = c->num_outputs;
    const int M = B * N;
    const int Gp = (int)sq_align_up(G, 8), Bp = (int)sq_align_up((size_t)B, 8);
    const int es = sq_dtype_size(dtype);
    const bool lp = dtype == SQ_BF16;
    // lean bf16 stream (vis.h): X1 and F were saved in bf16, and the gradient stream between the layers (dXin, dX1, dY, dLf) is
    // bf16 only -- no fp32 running gradient beside the operand copies
    const bool lean = sq_vis_lean_stream(dtype);
    {
        const int saved = sq_vis_saved_stream(fwd_workspace);
        SQ_REQUIRE(saved < 0 || saved == (lean ? 1 : 0),
                   "vis_backward: the forward pass saved %s rows in this workspace, the backward pass would read %s rows (SQ_VIS_FP32_STREAM changed between the two calls)",
                   saved ? "bf16" : "fp32", lean ? "bf16" : "fp32");
    }
    const int sdt = lean ? SQ_BF16 : SQ_F32;       // dtype of the stream tensors
    const char* wbase = lp ? (const char*)params_lp : (const char*)params;
    auto W = [&](int64_t off) { return (const void*)(wbase + (size_t)off * es); };
    auto Pf = [&](int64_t off) { return params + off; };
    auto Gp_ = [&](int64_t off) { return grad_params + off; };
    float* gp = grad_params;

    // C[M_,N_] (f32, ldc) = A[M_,K_] (lda) . Bm[N_,K_]^T (ldb), operands in T
    auto gemm = [&](const void* A, int lda, const void* Bm, int ldb, float* C, int ldc, int M_, int N_, int K_) {
        GemmArgs g;
        g.A = A; g.lda = lda; g.a_bytes = ((size_t)(M_ - 1) * lda + K_) * es;
        g.B = Bm; g.ldb = ldb; g.b_bytes = ((size_t)(N_ - 1) * ldb + K_) * es;
        g.C = C; g.ldc = ldc; g.M = M_; g.N = N_; g.K = K_;
        g.splitk_ws = b.skws; g.splitk_ws_bytes = b.skws_bytes;
        return g;
    };
    // C[M_,N_] (f32, ldc) = sum_k A[k, 0:M_] * Bm[k, 0:N_]   (token-major operands, K_ rows)
    auto gemm_tn = [&](const void* A, int lda, const void* Bm, int ldb, float* C, int ldc, int M_, int N_, int K_) {
        GemmArgs g;
        g.A = A; g.lda = lda; g.a_bytes = (size_t)K_ * lda * es;
        g.B = Bm; g.ldb = ldb; g.b_bytes = (size_t)K_ * ldb * es;
        g.C = C; g.ldc = ldc; g.M = M_; g.N = N_; g.K = K_;
        g.splitk_ws = b.skws; g.splitk_ws_bytes = b.skws_bytes;
        return g;
    };
#define RUN(expr) do { if (int _e = (expr)) return _e; } while (0)
    RUN(sq_k_cast_pad(grad_out, G, b.dout_lp, dtype, Gp, B, G, st));
    {   // every layer's "dX = dY . W" below is an NT product on W^T: transpose those weights with one launch
        sq_transpose_jobs jobs;
        auto add = [&](const void* src, int lds_, void* dst, int ldd, int R, int C, int batch, long long ss, long long ds) {
            if (jobs.n == SQ_MAX_TRANSPOSE_JOBS) {               // deeper models than one argument block holds
                if (int e = sq_k_transpose_multi(jobs, es, st)) return e;
                jobs = sq_transpose_jobs();
            }
            return sq_transpose_jobs_add(&jobs, src, lds_, dst, ldd, R, C, batch, ss, ds);
        };
        const long long wcs = (long long)SQ_HEAD_DIM * 2 * SQ_HEAD_DIM, wcd = (long long)SQ_HEAD_DIM * SQ_HEAD_DIM;
        // (the head weight [G, D] -- 42 MB at G = 20 820 -- is NOT transposed: its dX product runs on the TN kernel from a
        // gene-major copy of the 64 x G output gradient instead; the transpose was 60 of this launch's 85 us)
        RUN(add(b.dout_lp, Gp, b.doutT, Bp, B, Gp, 1, 0, 0));
        for (int l = 0; l < c->depth; ++l) {
            const sq_vis_layer_offsets& L = lay.layer[l];
            RUN(add(W(L.ff2_w), D, b.wt[l].ff2, D, D, D, 1, 0, 0));
            RUN(add(W(L.ff1_w), D, b.wt[l].ff1, D, D, D, 1, 0, 0));
            RUN(add(W(L.proj_w), HD, b.wt[l].proj, D, D, HD, 1, 0, 0));          // Wp [D, HD] -> [HD, D]
            RUN(add(W(L.c_w), 2 * SQ_HEAD_DIM, b.wt[l].wc_lf, SQ_HEAD_DIM, SQ_HEAD_DIM, SQ_HEAD_DIM, H, wcs, wcd));
            RUN(add((const char*)W(L.c_w) + (size_t)SQ_HEAD_DIM * es, 2 * SQ_HEAD_DIM, b.wt[l].wc_ts,
                                      SQ_HEAD_DIM, SQ_HEAD_DIM, SQ_HEAD_DIM, H, wcs, wcd));
            RUN(add(W(L.s_w), D, b.wt[l].s, HD, HD, D, 1, 0, 0));               // Ws [HD, D] -> [D, HD]
            RUN(add(W(L.f_w), D, b.wt[l].f, HD, HD, D, 1, 0, 0));               // Wf [HD, D] -> [D, HD]
        }
        RUN(sq_k_transpose_multi(jobs, es, st));
    }

    // ---------------- head: out = LN(mean_n X) Wh^T + bh ----------------
    { GemmArgs g = gemm_tn(b.dout_lp, Gp, w.xn, D, Gp_(lay.head_w), D, G, D, B); g.colsum_a = Gp_(lay.head_b); RUN(sq_launch_gemm_tn(g, dtype, st)); }
    // dxn[b, :] = sum_g dout[b, g] Wh[g, :]: both operands gene-major (doutT from the launch above, Wh as stored), K slices of G
    { GemmArgs g = gemm_tn(b.doutT, Bp, W(lay.head_w), D, b.dxn, D, B, D, G); RUN(sq_launch_gemm_tn(g, dtype, st)); }
    RUN(sq_k_ln_rows_bwd(b.dxn, w.xm, Pf(lay.head_ln_g), nullptr, b.dxm, nullptr, Gp_(lay.head_ln_g), Gp_(lay.head_ln_b),
                         b.red_ws, B, D, st));
    RUN(bucket_done(0));
    const int top = c->depth - 1;
    RUN(sq_k_bcast_rows(b.dxm, 1.0f / (float)N, lean ? nullptr : b.dXa, lp ? (bf16_t*)b.lg[top].dXin_lp : nullptr, B, N, D, st));

    // ---- second stream for the weight-gradient products (bf16 mode).  They are off the critical path (nothing in
    // the dX chain reads a dW), and a GEMM of this size cannot overlap its own memory-bound epilogue with its MFMA
    // phase (all of its blocks are resident at once and move in lock-step): a dW product running beside a dX
    // product fills each other's idle phase.  The main stream records an event when an operand is final; the side
    // stream waits for it.  Operands the side stream reads exist per layer (LayerG), so the dX chain never waits.
    SqSideStream* side = nullptr;      // weight gradients
    SqSideStream* side2 = nullptr;     // summary branch (small per-slide kernels)
    if (lp && !sq_env_flag("SQ_BWD_ONE_STREAM")) {
        side = sq_side_stream(0, 7 * SQ_MAX_DEPTH + 1);
        side2 = sq_side_stream(1, 4 * SQ_MAX_DEPTH);
        SQ_REQUIRE(side != nullptr && side2 != nullptr, "vis_backward: could not create the helper streams");
    }
    hipStream_t sst = side ? side->stream : st;
    int ev_next = 0;
    // "operand ready": later side-stream launches may read what the main stream has produced so far
    auto ready = [&]() {
        if (!side) return (int)SQ_OK;
        hipEvent_t ev = side->events[ev_next++];
        SQ_HIP_CHECK(hipEventRecord(ev, st));
        SQ_HIP_CHECK(hipStreamWaitEvent(sst, ev, 0));
        return (int)SQ_OK;
    };
    hipStream_t s2 = side2 ? side2->stream : st;
    float* red2 = side2 ? b.red_ws2 : b.red_ws;
    int ev2_next = 0;
    auto handoff = [&](hipStream_t from, hipStream_t to) {        // `to` may use what `from` has produced so far
        if (!side2 || from == to) return (int)SQ_OK;
        hipEvent_t ev = side2->events[ev2_next++];
        SQ_HIP_CHECK(hipEventRecord(ev, from));
        SQ_HIP_CHECK(hipStreamWaitEvent(to, ev, 0));
        return (int)SQ_OK;
    };
    auto run_tn = [&](GemmArgs& g) {
        if (side) g.splitk_ws = b.skws_side;
        return sq_launch_gemm_tn(g, dtype, sst);
    };
    // The four large weight gradients of a layer (ff2, ff1, projection, f: [1024 x 1024] each over K = B*N token rows) are
    // independent of each other and of the dX chain.  Issued one by one each is 64 tiles of 128 x 128 -- a quarter of the
    // chip -- so the launcher sliced K and a second launch summed the slices (37 reductions per step).  They are queued
    // instead and leave as ONE grouped launch per layer (same-shape members share a launch: 4 x 64 = 256 tiles, whole K per
    // tile, no partials); their operands are per-layer buffers, so nothing is overwritten in between.
    const bool group_dw = !sq_env_flag("SQ_BWD_NO_GROUP");
    GemmArgs pend[4];
    int npend = 0;
    auto queue_tn = [&](GemmArgs& g) -> int {
        if (!group_dw) return run_tn(g);
        pend[npend++] = g;
        return SQ_OK;
    };
    auto flush_tn = [&]() -> int {
        bool done[4] = {false, false, false, false};
        for (int i = 0; i < npend; ++i) {
            if (done[i]) continue;
            GemmArgs g = pend[i];
            g.ngroup = 0;
            for (int j = i; j < npend; ++j) {
                const GemmArgs& q = pend[j];
                if (done[j] || q.M != g.M || q.N != g.N || q.K != g.K || q.lda != g.lda || q.ldb != g.ldb || q.ldc != g.ldc ||
                    (q.colsum_a == nullptr) != (g.colsum_a == nullptr)) continue;
                g.gA[g.ngroup] = q.A; g.gB[g.ngroup] = q.B; g.gC[g.ngroup] = q.C; g.gcs[g.ngroup] = q.colsum_a;
                ++g.ngroup;
                done[j] = true;
            }
            if (g.ngroup == 1) { g.ngroup = 0; }
            if (int e = run_tn(g)) return e;
        }
        npend = 0;
        return SQ_OK;
    };
    auto bucket_done_side = [&](int i) {                          // bucket i final once the side stream gets here
        if (n_bucket_events == 0) return (int)SQ_OK;
        SQ_HIP_CHECK(hipEventRecord((hipEvent_t)bucket_events[i], sst));
        return (int)SQ_OK;
    };

    float* dXcur = b.dXa;
    float* dXoth = b.dXb;

    for (int l = top; l >= 0; --l) {
        const sq_vis_layer_offsets& L = lay.layer[l];
        LayerG lg = b.lg[l];
        sq_colsum_jobs csj;              // this layer's small column sums (3 LayerNorms + the combiner bias): one launch
        sq_colsum_jobs* defer = &csj;
        if (!lp) { lg.dXin_lp = dXcur; lg.dX1_lp = dXoth; }
        // ---------------- FeedForward: X2 = GELU(LN(X1) W1^T + b1) W2^T + b2 + X1 ----------------
        RUN(ready());
        { GemmArgs g = gemm_tn(lg.dXin_lp, D, w.H1[l], D, Gp_(L.ff2_w), D, D, D, M); g.colsum_a = Gp_(L.ff2_b); RUN(queue_tn(g)); }
        {   // dU = (dX2 . W2) * GELU'(U)
            GemmArgs g = gemm(lg.dXin_lp, D, b.wt[l].ff2, D, nullptr, D, M, D, D);
            g.C = lg.dU; g.out_dtype = dtype; g.gelu_grad_of = w.U[l]; g.gg_dtype = sq_vis_preact_dtype(dtype); g.ldgg = D;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        RUN(ready());
        { GemmArgs g = gemm_tn(lg.dU, D, w.Y[l], D, Gp_(L.ff1_w), D, D, D, M); g.colsum_a = Gp_(L.ff1_b); RUN(queue_tn(g)); }
        { GemmArgs g = gemm(lg.dU, D, b.wt[l].ff1, D, b.dY, D, M, D, D); if (lean) g.out_dtype = SQ_BF16; RUN(sq_launch_gemm(g, dtype, st)); }
        // dX1 = dX2 + dLN(dY)
        RUN(sq_k_ln_rows_bwd_any(b.dY, sdt, w.X1[l], sdt, Pf(L.ffln_g), lean ? (const void*)lg.dXin_lp : (const void*)dXcur, sdt,
                                 lean ? nullptr : dXoth, lp ? (bf16_t*)lg.dX1_lp : nullptr, Gp_(L.ffln_g),
                                 Gp_(L.ffln_b), b.part_ws[0], M, D, st, defer));
        float* dX1 = dXoth;

        // ---------------- projection: X1 = O Wp^T + bp + X ----------------
        RUN(ready());
        { GemmArgs g = gemm_tn(lg.dX1_lp, D, w.O[l], HD, Gp_(L.proj_w), HD, D, HD, M); g.colsum_a = Gp_(L.proj_b); RUN(queue_tn(g)); }
        {   // dP = (dX1 . Wp) * GELU'(P)
            GemmArgs g = gemm(lg.dX1_lp, D, b.wt[l].proj, D, nullptr, HD, M, HD, D);
            g.C = lg.dP; g.out_dtype = dtype; g.gelu_grad_of = w.P[l]; g.gg_dtype = sq_vis_preact_dtype(dtype); g.ldgg = HD;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        // ---------------- combiner: P_h = Lf_h Wc_h[:, :64]^T + Ts_h Wc_h[:, 64:]^T + bc_h ----------------
        RUN(ready());
        {   // dWc_h[:, :64] = dP_h^T . Lf_h
            GemmArgs g = gemm_tn(lg.dP, HD, w.Lf[l], HD, Gp_(L.c_w), 2 * SQ_HEAD_DIM, SQ_HEAD_DIM, SQ_HEAD_DIM, M);
            g.batch = H; g.sA = SQ_HEAD_DIM; g.sB = SQ_HEAD_DIM; g.sC = SQ_HEAD_DIM * 2 * SQ_HEAD_DIM;
            RUN(run_tn(g));
        }
        // ---------------- summary branch: per-slide tensors, small launches -> third stream, beside the local branch
        RUN(handoff(st, s2));
        RUN(sq_k_group_sum(lg.dP, dtype, B, N, HD, 1.0f, b.dCs, s2));       // dCs[b] = sum_n dP[b, n]
        if (B <= 512 && defer) RUN(sq_colsum_jobs_add(&csj, b.dCs, B, HD, HD, Gp_(L.c_b), nullptr, 0));
        else RUN(sq_k_colsum(b.dCs, SQ_F32, B, HD, HD, red2, Gp_(L.c_b), s2));
        if (lp) RUN(sq_k_cast_pad(b.dCs, HD, lg.dCs_lp, dtype, HD, B, HD, s2));
        RUN(handoff(s2, sst));
        {   // dWc_h[:, 64:] = dCs_h^T . Ts_h
            GemmArgs g = gemm_tn(lg.dCs_lp, HD, w.Ts[l], HD, Gp_(L.c_w) + SQ_HEAD_DIM, 2 * SQ_HEAD_DIM, SQ_HEAD_DIM, SQ_HEAD_DIM, B);
            g.batch = H; g.sA = SQ_HEAD_DIM; g.sB = SQ_HEAD_DIM; g.sC = SQ_HEAD_DIM * 2 * SQ_HEAD_DIM;
            RUN(run_tn(g));
        }
        {   // dTs_h = dCs_h . Wc_h[:, 64:]
            GemmArgs g = gemm(lg.dCs_lp, HD, b.wt[l].wc_ts, SQ_HEAD_DIM, b.dTs, HD, B, SQ_HEAD_DIM, SQ_HEAD_DIM);
            g.a_bytes = (size_t)B * HD * es; g.b_bytes = (size_t)H * 64 * 64 * es;
            g.batch = H; g.sA = SQ_HEAD_DIM; g.sB = SQ_HEAD_DIM * SQ_HEAD_DIM; g.sC = SQ_HEAD_DIM;
            g.splitk_ws = b.skws2; g.splitk_ws_bytes = b.skws2_bytes;
            RUN(sq_launch_gemm(g, dtype, s2));
        }
        RUN(sq_k_ln64_gelu_bwd(b.dTs, w.Sm[l], Pf(L.lns_g), Pf(L.lns_b), lg.dSm, dtype, Gp_(L.lns_g), Gp_(L.lns_b), b.part_ws[1], B, HD, s2, defer));
        RUN(handoff(s2, sst));
        { GemmArgs g = gemm_tn(lg.dSm, HD, w.Xbar[l], D, Gp_(L.s_w), D, HD, D, B); g.colsum_a = Gp_(L.s_b); RUN(run_tn(g)); }
        {   // dXbar / N   (Xbar = mean_n X: every token of the slide receives dXbar / N)
            GemmArgs g = gemm(lg.dSm, HD, b.wt[l].s, HD, b.dXbar, D, B, D, HD);
            g.alpha = 1.0f / (float)N;
            g.splitk_ws = b.skws2; g.splitk_ws_bytes = b.skws2_bytes;
            RUN(sq_launch_gemm(g, dtype, s2));
        }
        hipEvent_t ev_xbar = nullptr;
        if (side2) { ev_xbar = side2->events[ev2_next++]; SQ_HIP_CHECK(hipEventRecord(ev_xbar, s2)); }
        // ---------------- local branch (main stream) ----------------
        {   // dLf_h = dP_h . Wc_h[:, :64]
            GemmArgs g = gemm(lg.dP, HD, b.wt[l].wc_lf, SQ_HEAD_DIM, b.dLf, HD, M, SQ_HEAD_DIM, SQ_HEAD_DIM);
            g.a_bytes = (size_t)M * HD * es; g.b_bytes = (size_t)H * 64 * 64 * es;
            g.batch = H; g.sA = SQ_HEAD_DIM; g.sB = SQ_HEAD_DIM * SQ_HEAD_DIM; g.sC = SQ_HEAD_DIM;
            if (lean) g.out_dtype = SQ_BF16;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        RUN(sq_k_ln64_gelu_bwd_any(b.dLf, sdt, w.F[l], sdt, Pf(L.lnf_g), Pf(L.lnf_b), lg.dF, dtype, Gp_(L.lnf_g), Gp_(L.lnf_b), b.part_ws[2], M, HD, st, defer));
        RUN(ready());       // (the side stream has by now also waited for every summary-branch gradient of this layer)
        { GemmArgs g = gemm_tn(lg.dF, HD, w.Xin_lp[l], D, Gp_(L.f_w), D, HD, D, M); g.colsum_a = Gp_(L.f_b); RUN(queue_tn(g)); RUN(flush_tn()); }
        if (ev_xbar) SQ_HIP_CHECK(hipStreamWaitEvent(st, ev_xbar, 0));
        RUN(sq_k_colsum_multi(csj, st));              // needs the summary-branch partials: after the wait above
        RUN(ready());                                 // the side stream sees the layer's last gradients ...
        if (l > 0) RUN(bucket_done_side(c->depth - l));     // ... before it declares the bucket final

        {   // dXin = dF . Wf + dX1 (residual) + dXbar/N (per-slide row bias)
            GemmArgs g = gemm(lg.dF, HD, b.wt[l].f, HD, dXcur, D, M, D, HD);
            g.res = dX1; g.ldres = D; g.rowbias = b.dXbar; g.ldrb = D; g.rows_per_group = N;
            g.C2 = (lp && l > 0) ? (bf16_t*)b.lg[l - 1].dXin_lp : nullptr; g.ldc2 = D;
            if (lean) {         // the residual is the bf16 dX1; the result goes straight into the next layer's bf16 dXin (layer 0: fp32, for the position-embedding sum)
                g.res = lg.dX1_lp; g.res_dtype = SQ_BF16; g.C2 = nullptr;
                if (l > 0) { g.C = b.lg[l - 1].dXin_lp; g.out_dtype = SQ_BF16; }
            }
            RUN(sq_launch_gemm(g, dtype, st));
        }
        // fp32 keeps the gradient in dXcur for the next layer (read as its dXin); dX1 lived in dXoth: no swap needed
    }
    RUN(sq_k_batch_sum(dXcur, gp + lay.pos, B, N * D, st));       // pos_emb1D is added to every slide
    if (side) {                                                    // every dW is final before anything after this call
        hipEvent_t ev = side->events[ev_next++];
        SQ_HIP_CHECK(hipEventRecord(ev, sst));
        SQ_HIP_CHECK(hipStreamWaitEvent(st, ev, 0));
    }
    RUN(bucket_done(c->depth));
    if (grad_x) SQ_HIP_CHECK(hipMemcpyAsync(grad_x, dXcur, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
#undef RUN
    (void)Gp;
    return SQ_OK;
}
