// Per-tile vote of the sliding-window predictions -- the accumulation half of
// spatial_vis/visualize.py:35-102 (`sliding_window_method`): every kept 10x10 window writes its prediction to all
// of its tiles; with stride < 10 a tile ends up with the MEAN over the windows that contain it (:97-101), with
// stride 10 the last writer wins (:90-92).
//
// The reference appends window_prediction[gene] to a Python list per (gene, tile).  Here the window predictions
// stay on the device as [W, G] and each tile gathers its <= 100 rows: block = (tile, 1024 genes), the window ids
// of the tile are block-uniform (scalar loads), rows are read as float4 -- neighbouring tiles share 90 % of their
// windows, so most of the gather is served by L2 / Infinity Cache; the [n_tiles, G] result is written once.
#include "../../include/sequoia_hip.h"
#include "sq_common.h"

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void window_vote_kernel(const float* __restrict__ win_pred, int G, const int32_t* __restrict__ tile_windows,
                                                          int V, int mode, float fill, float* __restrict__ out) {
    const int t = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (c >= G) return;
    const int32_t* wl = tile_windows + (size_t)t * V;
    double acc[VEC];
    float last[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[e] = 0.0; last[e] = fill; }
    int cnt = 0;
    for (int v = 0; v < V; ++v) {
        const int w = wl[v];
        if (w < 0) break;                                   // lists are packed, -1 padded
        const float* row = win_pred + (size_t)w * G + c;
        if constexpr (VEC == 4) {
            const float4 x = *reinterpret_cast<const float4*>(row);
            acc[0] += x.x; acc[1] += x.y; acc[2] += x.z; acc[3] += x.w;
            last[0] = x.x; last[1] = x.y; last[2] = x.z; last[3] = x.w;
        } else {
            acc[0] += row[0];
            last[0] = row[0];
        }
        ++cnt;
    }
    float* o = out + (size_t)t * G + c;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = cnt == 0 ? fill : (mode == 1 ? last[e] : (float)(acc[e] / cnt));
}

}  // namespace

extern "C" int sq_window_vote(const float* win_pred, int n_windows, int num_outputs, const int32_t* tile_windows, int n_tiles,
                              int max_votes, int mode, float fill, float* out, sq_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(win_pred && tile_windows && out, "window_vote: null pointer");
    SQ_REQUIRE(n_windows >= 1 && num_outputs >= 1 && n_tiles >= 1 && max_votes >= 1 && n_tiles <= 65535 * 16,
               "window_vote: n_windows=%d num_outputs=%d n_tiles=%d max_votes=%d", n_windows, num_outputs, n_tiles, max_votes);
    SQ_REQUIRE(mode == 0 || mode == 1, "window_vote: mode %d (0 mean, 1 last writer)", mode);
    // gridDim.y is limited to 65535: tiles go through in slabs
    for (int t0 = 0; t0 < n_tiles; t0 += 65535) {
        const int nt = n_tiles - t0 < 65535 ? n_tiles - t0 : 65535;
        const int32_t* tw = tile_windows + (size_t)t0 * max_votes;
        float* o = out + (size_t)t0 * num_outputs;
        if (num_outputs % 4 == 0 && ((uintptr_t)win_pred & 15) == 0 && ((uintptr_t)out & 15) == 0)
            hipLaunchKernelGGL(window_vote_kernel<4>, dim3((num_outputs / 4 + 255) / 256, nt), dim3(256), 0, st, win_pred, num_outputs, tw,
                               max_votes, mode, fill, o);
        else
            hipLaunchKernelGGL(window_vote_kernel<1>, dim3((num_outputs + 255) / 256, nt), dim3(256), 0, st, win_pred, num_outputs, tw,
                               max_votes, mode, fill, o);
        SQ_LAUNCH_CHECK();
    }
    return SQ_OK;
}
