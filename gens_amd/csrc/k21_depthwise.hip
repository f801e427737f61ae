// K21: depth-wise 2-D convolutions of the MnasNet trunk (reference: torchvision's MNASNet layers inside
// models/modules/feature_network_mnasnet.py:53-103 -- `nn.Conv2d(c, c, k, padding=k//2, stride=s, groups=c, bias=False)` with k in {3, 5},
// s in {1, 2}), forward, data gradient and weight gradient.
//
// Why it exists: MIOpen has no tuned solver for these shapes on gfx950 and falls back to `naive_conv_ab_nonpacked_{fwd,bwd}_nchw_float_double_float`
// (one thread per output, double accumulation): 3.9 ms of the 41 ms training step of BASELINE config[2] as GenS.forward runs it
// (profiles/r03_bench_kernel_stats.csv), for 0.25 GB of tensors per forward pass.  A depth-wise convolution is a stencil: k^2 multiply-adds per
// 8 bytes moved, bound by HBM, not by arithmetic.
//
// Mapping.  NCHW float32, contiguous.  A workgroup = 256 threads = a TY x TX tile of OUTPUT pixels of one (image, channel) plane, x fastest
// (coalesced rows); the k^2 weights of the channel are wave-uniform (scalar loads); the input taps of neighbouring lanes overlap and come from
// L1 / L2 (a plane is read once from HBM).  Taps that fall into the zero padding are skipped by bounds tests on the row / column (uniform per
// tap for most of a tile).  float32 accumulation in the order (ky, kx) of ATen's direct convolution loops.
//   forward   out[n][c][oy][ox] = sum_{ky,kx} w[c][ky][kx] in[n][c][s oy + ky - p][s ox + kx - p]
//   dgrad     din[n][c][iy][ix] = sum_{ky,kx : s | iy + p - ky, s | ix + p - kx} w[c][ky][kx] dout[n][c][(iy + p - ky) / s][(ix + p - kx) / s]
//   wgrad     dw[c][ky][kx]     = sum_{n,oy,ox} dout[n][c][oy][ox] in[n][c][s oy + ky - p][s ox + kx - p]
// wgrad: a workgroup owns a channel and a slice of its n * OH * OW outputs, every thread keeps the k^2 partial sums in registers, the wave and
// then the workgroup reduce them, and the slices' partials (parts, C, k^2) are added by the caller in slice order (deterministic: no atomics).
#include <stdlib.h>

#include "common.h"

struct DwGeom {
    int n, c, h, w;          // input
    int oh, ow;              // output
    int stride, pad;
};

// thread -> pixel (px, py) of a (tw x th) plane.  Wide planes: a 64 x 4 tile per workgroup (whole 256-byte row pieces per wave); planes
// narrower than a wave (the 15 x 20 ... 30 x 40 maps of the deep stages) are walked linearly, 256 consecutive pixels per workgroup.
__device__ __forceinline__ bool dw_pixel(int tw, int th, int& px, int& py) {
    if (tw >= 64) {
        px = blockIdx.x * 64 + (threadIdx.x & 63);
        py = blockIdx.y * 4 + (threadIdx.x >> 6);
    } else {
        const int i = blockIdx.x * 256 + threadIdx.x;
        py = i / tw;
        px = i - py * tw;
    }
    return px < tw && py < th;
}
static dim3 dw_grid(int tw, int th, unsigned planes) {
    if (tw >= 64) return dim3(gens_blocks(tw, 64), gens_blocks(th, 4), planes);
    return dim3(gens_blocks((int64_t)tw * th, 256), 1, planes);
}

template <int K>
__global__ __launch_bounds__(256) void dw_fwd_k(DwGeom g, const float* __restrict__ in, const float* __restrict__ wt, float* __restrict__ out) {
    const int plane = blockIdx.z;                              // n * C + c
    const int c = plane % g.c;
    int ox, oy;
    if (!dw_pixel(g.ow, g.oh, ox, oy)) return;
    const float* ip = in + (int64_t)plane * g.h * g.w;
    const float* wp = wt + c * K * K;
    const int iy0 = oy * g.stride - g.pad, ix0 = ox * g.stride - g.pad;
    // branch-free taps: every load is issued (from an index clamped into the plane) before the first multiply-add needs one -- K^2 independent
    // loads in flight instead of K^2 load-use pairs under their own exec masks; a tap in the zero padding contributes an exact 0
    float v[K * K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = iy0 + ky;
        const bool rok = (unsigned)iy < (unsigned)g.h;
        const float* row = ip + (int64_t)min(max(iy, 0), g.h - 1) * g.w;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ix0 + kx;
            const float t = row[min(max(ix, 0), g.w - 1)];
            v[ky * K + kx] = (rok && (unsigned)ix < (unsigned)g.w) ? t : 0.0f;
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc = __builtin_fmaf(wp[t], v[t], acc);
    out[((int64_t)plane * g.oh + oy) * g.ow + ox] = acc;
}

// The production forward (and the stride-1 data gradient, which is the same correlation with the taps reversed): a thread owns RY = 4
// vertically consecutive outputs of one column, so an input row is loaded ONCE for the up to K outputs it feeds -- (3 S + K) K loads per four
// outputs instead of 4 K^2 -- and the column offsets / bounds are computed once per thread.  A wave = G = 64 / TX row groups of TX columns
// (TX = 16, 32 or 64, the smallest that covers the plane's width: the 15 x 20 ... 60 x 80 planes of the deep stages keep >= 62 % of their
// lanes busy); no LDS, no barriers.  The simple kernels above (one output per thread) measured 20 - 27 us even on 1.7 M outputs: ~300
// instructions per output of address arithmetic and clamps around 25 loads -- instruction-bound at a tenth of the HBM rate.
template <int K, int S, int TXLOG, bool FLIP>
__global__ __launch_bounds__(256) void dw_tile_k(DwGeom g, const float* __restrict__ in, const float* __restrict__ wt, float* __restrict__ out) {
    constexpr int TX = 1 << TXLOG, G = 64 >> TXLOG, RY = 4, IR = (RY - 1) * S + K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ox = blockIdx.x * TX + (lane & (TX - 1));
    const int oy0 = ((int)blockIdx.y * 4 * G + wave * G + (lane >> TXLOG)) * RY;
    const int plane = blockIdx.z, c = plane % g.c;
    if (ox >= g.ow || oy0 >= g.oh) return;
    const float* ip = in + (int64_t)plane * g.h * g.w;
    const float* wp = wt + c * K * K;
    const int ix0 = ox * S - g.pad, iy0 = oy0 * S - g.pad;
    int ixc[K];
    bool cok[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        const int ix = ix0 + kx;
        cok[kx] = (unsigned)ix < (unsigned)g.w;
        ixc[kx] = min(max(ix, 0), g.w - 1);
    }
    float acc[RY];
#pragma unroll
    for (int j = 0; j < RY; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int r = 0; r < IR; ++r) {
        const int iy = iy0 + r;
        const bool rok = (unsigned)iy < (unsigned)g.h;
        const float* row = ip + (int64_t)min(max(iy, 0), g.h - 1) * g.w;
        float v[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float t = row[ixc[kx]];
            v[kx] = (rok && cok[kx]) ? t : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < RY; ++j) {
            const int ky = r - j * S;                       // (compile time: the loops are unrolled)
            if (ky < 0 || ky >= K) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[j] = __builtin_fmaf(FLIP ? wp[(K - 1 - ky) * K + (K - 1 - kx)] : wp[ky * K + kx], v[kx], acc[j]);
        }
    }
    float* op = out + ((int64_t)plane * g.oh + oy0) * g.ow + ox;
#pragma unroll
    for (int j = 0; j < RY; ++j)
        if (oy0 + j < g.oh) op[(int64_t)j * g.ow] = acc[j];
}

template <int K, int S, bool FLIP>
static void dw_tile_launch(const DwGeom& g, int tw, int th, unsigned planes, const float* in, const float* wt, float* out, hipStream_t s) {
    // tw x th: the OUTPUT plane of the launch (forward: oh x ow; stride-1 data gradient: the input plane, with the roles of in / out swapped)
    if (tw <= 16) dw_tile_k<K, S, 4, FLIP><<<dim3(gens_blocks(tw, 16), gens_blocks(th, 64), planes), 256, 0, s>>>(g, in, wt, out);
    else if (tw <= 32) dw_tile_k<K, S, 5, FLIP><<<dim3(gens_blocks(tw, 32), gens_blocks(th, 32), planes), 256, 0, s>>>(g, in, wt, out);
    else dw_tile_k<K, S, 6, FLIP><<<dim3(gens_blocks(tw, 64), gens_blocks(th, 16), planes), 256, 0, s>>>(g, in, wt, out);
}

template <int K, int S>
__global__ __launch_bounds__(256) void dw_dgrad_k(DwGeom g, const float* __restrict__ dout, const float* __restrict__ wt, float* __restrict__ din) {
    const int plane = blockIdx.z;
    const int c = plane % g.c;
    int ix, iy;
    if (!dw_pixel(g.w, g.h, ix, iy)) return;
    const float* dp = dout + (int64_t)plane * g.oh * g.ow;
    const float* wp = wt + c * K * K;
    float v[K * K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int ty = iy + g.pad - ky;
        const int oy = S == 2 ? ty >> 1 : ty;
        const bool rok = ty >= 0 && !(S == 2 && (ty & 1)) && oy < g.oh;
        const float* row = dp + (int64_t)min(max(oy, 0), g.oh - 1) * g.ow;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int tx = ix + g.pad - kx;
            const int ox = S == 2 ? tx >> 1 : tx;
            const float t = row[min(max(ox, 0), g.ow - 1)];
            v[ky * K + kx] = (rok && tx >= 0 && !(S == 2 && (tx & 1)) && ox < g.ow) ? t : 0.0f;
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc = __builtin_fmaf(wp[t], v[t], acc);
    din[((int64_t)plane * g.h + iy) * g.w + ix] = acc;
}

// partial[part][c][ky][kx]; grid (parts, C); a part = a contiguous range of the n * oh output ROWS of the channel
template <int K>
__global__ __launch_bounds__(256) void dw_wgrad_k(DwGeom g, const float* __restrict__ in, const float* __restrict__ dout, int rows_per_part,
                                                  float* __restrict__ partial) {
    const int part = blockIdx.x, c = blockIdx.y;
    const int total_rows = g.n * g.oh;
    const int r0 = part * rows_per_part, r1 = min(r0 + rows_per_part, total_rows);
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.0f;
    const int lane_x = threadIdx.x & 63, sub = threadIdx.x >> 6;
    // the part's outputs, 256 consecutive ones per trip whatever the row length (the 15 x 20 planes of the deepest stage would leave two
    // thirds of a wave idle in a walk along x)
    const int p_end = r1 * g.ow;
    for (int p = r0 * g.ow + (int)threadIdx.x; p < p_end; p += 256) {
        const int r = p / g.ow, ox = p - r * g.ow;
        const int n = r / g.oh, oy = r - n * g.oh;
        const int64_t plane = (int64_t)n * g.c + c;
        const float d = dout[(plane * g.oh + oy) * g.ow + ox];
        const float* ip = in + plane * g.h * g.w;
        const int iy0 = oy * g.stride - g.pad, ix0 = ox * g.stride - g.pad;
        float v[K * K];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = iy0 + ky;
            const bool rok = (unsigned)iy < (unsigned)g.h;
            const float* row = ip + (int64_t)min(max(iy, 0), g.h - 1) * g.w;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ix0 + kx;
                const float t = row[min(max(ix, 0), g.w - 1)];
                v[ky * K + kx] = (rok && (unsigned)ix < (unsigned)g.w) ? t : 0.0f;
            }
        }
#pragma unroll
        for (int t = 0; t < K * K; ++t) acc[t] = __builtin_fmaf(d, v[t], acc[t]);
    }
    __shared__ float red[4][K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const float s = wave_sum(acc[t]);
        if (lane_x == 0) red[sub][t] = s;
    }
    __syncthreads();
    if (threadIdx.x < K * K) {
        const int t = threadIdx.x;
        partial[((int64_t)part * g.c + c) * (K * K) + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}

// The production weight gradient: the same register tile as dw_tile_k -- a thread takes four vertically consecutive outputs of a column per
// trip, loads the (3 S + K) input rows they touch once, and adds d[j] * in[j S + ky][kx] into its K^2 sums.  A work unit = (image, group of
// four output rows); a part = a contiguous range of units of one channel.
template <int K, int S, int TXLOG>
__global__ __launch_bounds__(256) void dw_wgrad_tile_k(DwGeom g, const float* __restrict__ in, const float* __restrict__ dout, int units_per_part,
                                                       float* __restrict__ partial) {
    constexpr int TX = 1 << TXLOG, G = 64 >> TXLOG, RY = 4, IR = (RY - 1) * S + K;
    const int part = blockIdx.x, c = blockIdx.y;
    const int groups = (g.oh + RY - 1) / RY, units = g.n * groups;
    const int u0 = part * units_per_part, u1 = min(u0 + units_per_part, units);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = lane & (TX - 1), lu = wave * G + (lane >> TXLOG);
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.0f;
    for (int ub = u0; ub < u1; ub += 4 * G) {
        const int u = ub + lu;
        if (u >= u1) continue;
        const int n = u / groups, oy0 = (u - n * groups) * RY;
        const int64_t plane = (int64_t)n * g.c + c;
        const float* ip = in + plane * g.h * g.w;
        const float* dp = dout + (plane * g.oh + oy0) * g.ow;
        const int iy0 = oy0 * S - g.pad;
        for (int ox = lx; ox < g.ow; ox += TX) {
            const int ix0 = ox * S - g.pad;
            float d[RY];
#pragma unroll
            for (int j = 0; j < RY; ++j) d[j] = oy0 + j < g.oh ? dp[(int64_t)j * g.ow + ox] : 0.0f;
#pragma unroll
            for (int r = 0; r < IR; ++r) {
                const int iy = iy0 + r;
                const bool rok = (unsigned)iy < (unsigned)g.h;
                const float* row = ip + (int64_t)min(max(iy, 0), g.h - 1) * g.w;
                float v[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int ix = ix0 + kx;
                    const float t = row[min(max(ix, 0), g.w - 1)];
                    v[kx] = (rok && (unsigned)ix < (unsigned)g.w) ? t : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < RY; ++j) {
                    const int ky = r - j * S;
                    if (ky < 0 || ky >= K) continue;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) acc[ky * K + kx] = __builtin_fmaf(d[j], v[kx], acc[ky * K + kx]);
                }
            }
        }
    }
    __shared__ float red[4][K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const float s = wave_sum(acc[t]);
        if (lane == 0) red[wave][t] = s;
    }
    __syncthreads();
    if (threadIdx.x < K * K) {
        const int t = threadIdx.x;
        partial[((int64_t)part * g.c + c) * (K * K) + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}

static int dw_geom(const char* who, int n, int c, int h, int w, int k, int stride, DwGeom& g) {
    GENS_CHECK_ARG(n >= 0 && c > 0 && h > 0 && w > 0, GENS_EINVAL, "%s: bad shape (%d, %d, %d, %d)", who, n, c, h, w);
    GENS_CHECK_ARG(k == 3 || k == 5, GENS_ELIMIT, "%s: kernel size %d (3 or 5: the MnasNet trunk's)", who, k);
    GENS_CHECK_ARG(stride == 1 || stride == 2, GENS_ELIMIT, "%s: stride %d (1 or 2)", who, stride);
    g.n = n; g.c = c; g.h = h; g.w = w;
    g.stride = stride;
    g.pad = k / 2;
    g.oh = (h + 2 * g.pad - k) / stride + 1;
    g.ow = (w + 2 * g.pad - k) / stride + 1;
    GENS_CHECK_ARG(c <= 65535, GENS_ELIMIT, "%s: %d channels", who, c);
    return 0;
}

// (planes ride on grid.z, at most 65 535 per launch: larger batches go in slices of whole images -- the trunk has 5 views x 1 152 channels)
extern "C" int gens_depthwise_conv2d_fwd(const float* in, const float* weight, int n, int c, int h, int w, int k, int stride, float* out, void* stream) {
    DwGeom g;
    if (int e = dw_geom("gens_depthwise_conv2d_fwd", n, c, h, w, k, stride, g)) return e;
    if (n == 0) return 0;
    GENS_CHECK_ARG(in && weight && out, GENS_EINVAL, "gens_depthwise_conv2d_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t in_plane = (int64_t)h * w, out_plane = (int64_t)g.oh * g.ow;
    const int per = max(1, 65535 / c);                          // images per launch
    for (int n0 = 0; n0 < n; n0 += per) {
        const int nn = min(per, n - n0);
        const dim3 grid = dw_grid(g.ow, g.oh, (unsigned)(nn * c));
        const float* ip = in + (int64_t)n0 * c * in_plane;
        float* op = out + (int64_t)n0 * c * out_plane;
        if (getenv("GENS_K21_SIMPLE")) {                        // (the one-output-per-thread kernels: A/B switch and cross-check)
            if (k == 3) dw_fwd_k<3><<<grid, 256, 0, s>>>(g, ip, weight, op);
            else dw_fwd_k<5><<<grid, 256, 0, s>>>(g, ip, weight, op);
        } else if (k == 3 && stride == 1) dw_tile_launch<3, 1, false>(g, g.ow, g.oh, (unsigned)(nn * c), ip, weight, op, s);
        else if (k == 3) dw_tile_launch<3, 2, false>(g, g.ow, g.oh, (unsigned)(nn * c), ip, weight, op, s);
        else if (stride == 1) dw_tile_launch<5, 1, false>(g, g.ow, g.oh, (unsigned)(nn * c), ip, weight, op, s);
        else dw_tile_launch<5, 2, false>(g, g.ow, g.oh, (unsigned)(nn * c), ip, weight, op, s);
    }
    return gens_launch_status("gens_depthwise_conv2d_fwd");
}

extern "C" int gens_depthwise_conv2d_dgrad(const float* grad_out, const float* weight, int n, int c, int h, int w, int k, int stride, float* grad_in,
                                           void* stream) {
    DwGeom g;
    if (int e = dw_geom("gens_depthwise_conv2d_dgrad", n, c, h, w, k, stride, g)) return e;
    if (n == 0) return 0;
    GENS_CHECK_ARG(grad_out && weight && grad_in, GENS_EINVAL, "gens_depthwise_conv2d_dgrad: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t in_plane = (int64_t)h * w, out_plane = (int64_t)g.oh * g.ow;
    const int per = max(1, 65535 / c);
    for (int n0 = 0; n0 < n; n0 += per) {
        const int nn = min(per, n - n0);
        const dim3 grid = dw_grid(w, h, (unsigned)(nn * c));
        const float* dp = grad_out + (int64_t)n0 * c * out_plane;
        float* ip = grad_in + (int64_t)n0 * c * in_plane;
        if (stride == 1 && !getenv("GENS_K21_SIMPLE")) {        // stride 1: din = the forward correlation of dout with the taps reversed (oh = h, ow = w)
            if (k == 3) dw_tile_launch<3, 1, true>(g, w, h, (unsigned)(nn * c), dp, weight, ip, s);
            else dw_tile_launch<5, 1, true>(g, w, h, (unsigned)(nn * c), dp, weight, ip, s);
        } else if (k == 3 && stride == 1) dw_dgrad_k<3, 1><<<grid, 256, 0, s>>>(g, dp, weight, ip);
        else if (k == 3) dw_dgrad_k<3, 2><<<grid, 256, 0, s>>>(g, dp, weight, ip);
        else if (stride == 1) dw_dgrad_k<5, 1><<<grid, 256, 0, s>>>(g, dp, weight, ip);
        else dw_dgrad_k<5, 2><<<grid, 256, 0, s>>>(g, dp, weight, ip);
    }
    return gens_launch_status("gens_depthwise_conv2d_dgrad");
}

// number of partial sums gens_depthwise_conv2d_wgrad writes: partial is (parts, c, k, k) floats.  A part = a contiguous range of work units
// (image, group of four output rows) of one channel.
static int dw_wgrad_units(const DwGeom& g) { return g.n * ((g.oh + 3) / 4); }
static int dw_wgrad_units_per_trip(const DwGeom& g) { return g.ow <= 16 ? 16 : g.ow <= 32 ? 8 : 4; }      // 4 waves x (64 / TX) units
extern "C" int gens_depthwise_conv2d_wgrad_parts(int n, int c, int h, int w, int k, int stride) {
    DwGeom g;
    if (dw_geom("gens_depthwise_conv2d_wgrad_parts", n, c, h, w, k, stride, g)) return 0;
    const int units = dw_wgrad_units(g), trip = dw_wgrad_units_per_trip(g);
    int parts = (2048 + c - 1) / c;                              // ~2 048 workgroups per launch
    if (parts > (units + trip - 1) / trip) parts = (units + trip - 1) / trip;          // at least one trip of the workgroup per part
    return parts < 1 ? 1 : parts;
}

extern "C" int gens_depthwise_conv2d_wgrad(const float* in, const float* grad_out, int n, int c, int h, int w, int k, int stride, float* partial,
                                           void* stream) {
    DwGeom g;
    if (int e = dw_geom("gens_depthwise_conv2d_wgrad", n, c, h, w, k, stride, g)) return e;
    GENS_CHECK_ARG(partial && (n == 0 || (in && grad_out)), GENS_EINVAL, "gens_depthwise_conv2d_wgrad: null pointer");
    const int parts = gens_depthwise_conv2d_wgrad_parts(n, c, h, w, k, stride);
    const int units = dw_wgrad_units(g), upp = (units + parts - 1) / parts;
    const dim3 grid((unsigned)parts, (unsigned)c);
    hipStream_t s = (hipStream_t)stream;
#define DW_WGRAD(K_, S_)                                                                                        \
    do {                                                                                                        \
        if (g.ow <= 16) dw_wgrad_tile_k<K_, S_, 4><<<grid, 256, 0, s>>>(g, in, grad_out, upp, partial);         \
        else if (g.ow <= 32) dw_wgrad_tile_k<K_, S_, 5><<<grid, 256, 0, s>>>(g, in, grad_out, upp, partial);    \
        else dw_wgrad_tile_k<K_, S_, 6><<<grid, 256, 0, s>>>(g, in, grad_out, upp, partial);                    \
    } while (0)
    if (getenv("GENS_K21_SIMPLE")) {
        // (the one-output-per-thread kernel walks output ROWS: the same parts, read as row ranges of four rows per unit)
        const int rpp = upp * 4;
        if (k == 3) dw_wgrad_k<3><<<grid, 256, 0, s>>>(g, in, grad_out, rpp, partial);
        else dw_wgrad_k<5><<<grid, 256, 0, s>>>(g, in, grad_out, rpp, partial);
    } else if (k == 3 && stride == 1) DW_WGRAD(3, 1);
    else if (k == 3) DW_WGRAD(3, 2);
    else if (stride == 1) DW_WGRAD(5, 1);
    else DW_WGRAD(5, 2);
#undef DW_WGRAD
    return gens_launch_status("gens_depthwise_conv2d_wgrad");
}
