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
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = iy0 + ky;
        if ((unsigned)iy >= (unsigned)g.h) continue;
        const float* row = ip + (int64_t)iy * g.w;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ix0 + kx;
            if ((unsigned)ix < (unsigned)g.w) acc = __builtin_fmaf(wp[ky * K + kx], row[ix], acc);
        }
    }
    out[((int64_t)plane * g.oh + oy) * g.ow + ox] = acc;
}

template <int K, int S>
__global__ __launch_bounds__(256) void dw_dgrad_k(DwGeom g, const float* __restrict__ dout, const float* __restrict__ wt, float* __restrict__ din) {
    const int plane = blockIdx.z;
    const int c = plane % g.c;
    int ix, iy;
    if (!dw_pixel(g.w, g.h, ix, iy)) return;
    const float* dp = dout + (int64_t)plane * g.oh * g.ow;
    const float* wp = wt + c * K * K;
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int ty = iy + g.pad - ky;
        if (ty < 0 || (S == 2 && (ty & 1))) continue;
        const int oy = S == 2 ? ty >> 1 : ty;
        if (oy >= g.oh) continue;
        const float* row = dp + (int64_t)oy * g.ow;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int tx = ix + g.pad - kx;
            if (tx < 0 || (S == 2 && (tx & 1))) continue;
            const int ox = S == 2 ? tx >> 1 : tx;
            if (ox < g.ow) acc = __builtin_fmaf(wp[ky * K + kx], row[ox], acc);
        }
    }
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
    const int lane_x = threadIdx.x & 63, sub = threadIdx.x >> 6;          // a wave walks along x, the four waves take rows r0 + sub, + 4, ...
    for (int r = r0 + sub; r < r1; r += 4) {
        const int n = r / g.oh, oy = r - n * g.oh;
        const int64_t plane = (int64_t)n * g.c + c;
        const float* dp = dout + (plane * g.oh + oy) * g.ow;
        const float* ip = in + plane * g.h * g.w;
        const int iy0 = oy * g.stride - g.pad;
        for (int ox = lane_x; ox < g.ow; ox += 64) {
            const float d = dp[ox];
            const int ix0 = ox * g.stride - g.pad;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = iy0 + ky;
                if ((unsigned)iy >= (unsigned)g.h) continue;
                const float* row = ip + (int64_t)iy * g.w;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int ix = ix0 + kx;
                    if ((unsigned)ix < (unsigned)g.w) acc[ky * K + kx] = __builtin_fmaf(d, row[ix], acc[ky * K + kx]);
                }
            }
        }
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
        if (k == 3) dw_fwd_k<3><<<grid, 256, 0, s>>>(g, ip, weight, op);
        else dw_fwd_k<5><<<grid, 256, 0, s>>>(g, ip, weight, op);
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
        if (k == 3 && stride == 1) dw_dgrad_k<3, 1><<<grid, 256, 0, s>>>(g, dp, weight, ip);
        else if (k == 3) dw_dgrad_k<3, 2><<<grid, 256, 0, s>>>(g, dp, weight, ip);
        else if (stride == 1) dw_dgrad_k<5, 1><<<grid, 256, 0, s>>>(g, dp, weight, ip);
        else dw_dgrad_k<5, 2><<<grid, 256, 0, s>>>(g, dp, weight, ip);
    }
    return gens_launch_status("gens_depthwise_conv2d_dgrad");
}

// number of partial sums (slices of the output rows) gens_depthwise_conv2d_wgrad writes: partial is (parts, c, k, k) floats
extern "C" int gens_depthwise_conv2d_wgrad_parts(int n, int c, int h, int w, int k, int stride) {
    DwGeom g;
    if (dw_geom("gens_depthwise_conv2d_wgrad_parts", n, c, h, w, k, stride, g)) return 0;
    const int rows = n * g.oh;
    int parts = (2048 + c - 1) / c;                              // ~2 048 workgroups per launch
    if (parts > (rows + 3) / 4) parts = (rows + 3) / 4;          // at least four rows (one per wave) per part
    return parts < 1 ? 1 : parts;
}

extern "C" int gens_depthwise_conv2d_wgrad(const float* in, const float* grad_out, int n, int c, int h, int w, int k, int stride, float* partial,
                                           void* stream) {
    DwGeom g;
    if (int e = dw_geom("gens_depthwise_conv2d_wgrad", n, c, h, w, k, stride, g)) return e;
    GENS_CHECK_ARG(c <= 65535, GENS_ELIMIT, "gens_depthwise_conv2d_wgrad: %d channels", c);
    GENS_CHECK_ARG(partial && (n == 0 || (in && grad_out)), GENS_EINVAL, "gens_depthwise_conv2d_wgrad: null pointer");
    const int parts = gens_depthwise_conv2d_wgrad_parts(n, c, h, w, k, stride);
    const int rows = n * g.oh, rpp = (rows + parts - 1) / parts;
    const dim3 grid((unsigned)parts, (unsigned)c);
    hipStream_t s = (hipStream_t)stream;
    if (k == 3) dw_wgrad_k<3><<<grid, 256, 0, s>>>(g, in, grad_out, rpp, partial);
    else dw_wgrad_k<5><<<grid, 256, 0, s>>>(g, in, grad_out, rpp, partial);
    return gens_launch_status("gens_depthwise_conv2d_wgrad");
}
