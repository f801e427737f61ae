// K22: BatchNorm2d in TRAINING mode (batch statistics) fused with the ReLU that follows it in the MnasNet trunk (reference: the
// `nn.BatchNorm2d(c, momentum = 1 - 0.9997)` [+ `nn.ReLU(inplace=True)`] pairs of torchvision's MNASNet inside
// models/modules/feature_network_mnasnet.py:53-103), forward and backward, NCHW float32.
//
// Why it exists: a training step of BASELINE config[2] runs 102 of these forward (the trainable network and its frozen matching copy,
// gens.py:124-140) and 51 backward, on tensors of 0.2 - 50 MB.  MIOpen's spatial batch-norm kernels take 26 us each whatever the size, every
// BatchNorm adds a one-element `num_batches_tracked += 1` launch and every ReLU two element-wise launches: 5 ms of a 35 ms step for 0.9 GB of
// traffic.
//
//   forward    pass 1: per (channel, part) sums of x and x^2 in float64;  pass 2: every workgroup adds its channel's partials (<= 256 of
//              them), forms mean and the biased variance, y = [max](((x - mean) * rstd) * gamma + beta[, 0]) -- the float32 expression of
//              ATen's batch_norm -- and workgroup 0 of the channel leaves (mean, rstd) for the backward pass and moves the running
//              statistics (running_var with the unbiased variance, momentum as nn.BatchNorm2d); one thread bumps num_batches_tracked.
//   backward   g = gy [y > 0] (y recomputed with the forward's expression: the same decision);  pass 1: sums of g and g xhat;
//              pass 2: gx = (g - mean(g) - xhat mean(g xhat)) rstd gamma;  d gamma = sum g xhat, d beta = sum g.
// A part = (image, chunk of the plane): the loops walk contiguous memory (16-byte accesses when the plane size is a multiple of four).
#include <stdlib.h>

#include "common.h"

struct BnGeom {
    int n, c, hw;
    int chunks;              // parts per channel = n * chunks
};

__device__ __forceinline__ double bn_wave_sum(double v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
    return v;                                                                     // lane 0
}
// two sums of the workgroup -> (a, b) in thread 0
__device__ __forceinline__ void bn_block_sum2(double& a, double& b) {
    __shared__ double red[2][4];
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    __syncthreads();
}
// this workgroup's range [r0, r1) of plane (image i, channel c) and the plane's base offset
__device__ __forceinline__ void bn_range(const BnGeom& g, int part, int c, int64_t& base, int& r0, int& r1) {
    const int img = part / g.chunks, chunk = part - img * g.chunks;
    int per = (g.hw + g.chunks - 1) / g.chunks;
    per = (per + 3) & ~3;                                                        // chunk boundaries on multiples of four: vector accesses stay aligned
    r0 = min(chunk * per, g.hw);
    r1 = min(r0 + per, g.hw);
    base = ((int64_t)img * g.c + c) * g.hw;
}
// the channel's totals from its partials (every thread gets them)
__device__ __forceinline__ void bn_totals(const double* __restrict__ partial, int parts, int c, double& s0, double& s1) {
    double a = 0.0, b = 0.0;
    for (int p = threadIdx.x; p < parts; p += 256) {
        a += partial[((int64_t)c * parts + p) * 2];
        b += partial[((int64_t)c * parts + p) * 2 + 1];
    }
    bn_block_sum2(a, b);
    s0 = a;
    s1 = b;
}

__global__ __launch_bounds__(256) void bn_stats_k(BnGeom g, const float* __restrict__ x, double* __restrict__ partial) {
    const int part = blockIdx.x, c = blockIdx.y, parts = g.n * g.chunks;
    int64_t base;
    int r0, r1;
    bn_range(g, part, c, base, r0, r1);
    const float* xp = x + base;
    double s = 0.0, ss = 0.0;
    if ((g.hw & 3) == 0) {
        for (int r = r0 + 4 * (int)threadIdx.x; r < r1; r += 1024) {
            const float4 v = *(const float4*)(xp + r);
            const double a = v.x, b = v.y, cc = v.z, d = v.w;
            s += (a + b) + (cc + d);
            ss += (a * a + b * b) + (cc * cc + d * d);
        }
    } else {
        for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
            const double a = xp[r];
            s += a;
            ss += a * a;
        }
    }
    bn_block_sum2(s, ss);
    if (threadIdx.x == 0) {
        partial[((int64_t)c * parts + part) * 2] = s;
        partial[((int64_t)c * parts + part) * 2 + 1] = ss;
    }
}

__device__ __forceinline__ float bn_value(float x, float mean, float rstd, float gamma, float beta) { return ((x - mean) * rstd) * gamma + beta; }

template <bool RELU>
__global__ __launch_bounds__(256) void bn_apply_k(BnGeom g, const float* __restrict__ x, const double* __restrict__ partial, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float eps, float momentum, float* __restrict__ y,
                                                  float* __restrict__ mean_rstd, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                  int64_t* __restrict__ num_batches) {
    const int part = blockIdx.x, c = blockIdx.y, parts = g.n * g.chunks;
    double s, ss;
    bn_totals(partial, parts, c, s, ss);
    const double m = (double)g.n * g.hw;
    const double mean_d = s / m;
    double var_d = ss / m - mean_d * mean_d;                                      // biased (what normalises the batch)
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var_d + (double)eps));
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    if (part == 0 && threadIdx.x == 0) {
        mean_rstd[2 * c] = mean;
        mean_rstd[2 * c + 1] = rstd;
        if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(m > 1.0 ? var_d * m / (m - 1.0) : var_d);
        if (c == 0 && num_batches) num_batches[0] += 1;
    }
    int64_t base;
    int r0, r1;
    bn_range(g, part, c, base, r0, r1);
    const float* xp = x + base;
    float* yp = y + base;
    if ((g.hw & 3) == 0) {
        for (int r = r0 + 4 * (int)threadIdx.x; r < r1; r += 1024) {
            const float4 v = *(const float4*)(xp + r);
            float4 o = make_float4(bn_value(v.x, mean, rstd, ga, be), bn_value(v.y, mean, rstd, ga, be), bn_value(v.z, mean, rstd, ga, be),
                                   bn_value(v.w, mean, rstd, ga, be));
            if (RELU) o = make_float4(fmaxf(o.x, 0.0f), fmaxf(o.y, 0.0f), fmaxf(o.z, 0.0f), fmaxf(o.w, 0.0f));
            *(float4*)(yp + r) = o;
        }
    } else {
        for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
            const float o = bn_value(xp[r], mean, rstd, ga, be);
            yp[r] = RELU ? fmaxf(o, 0.0f) : o;
        }
    }
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_stats_k(BnGeom g, const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ mean_rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ partial) {
    const int part = blockIdx.x, c = blockIdx.y, parts = g.n * g.chunks;
    const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1];
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    int64_t base;
    int r0, r1;
    bn_range(g, part, c, base, r0, r1);
    const float* xp = x + base;
    const float* gp = gy + base;
    double s = 0.0, sx = 0.0;
    for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
        const float xv = xp[r], xhat = (xv - mean) * rstd;
        float gv = gp[r];
        if (RELU && !(xhat * ga + be > 0.0f)) gv = 0.0f;
        s += (double)gv;
        sx += (double)gv * (double)xhat;
    }
    bn_block_sum2(s, sx);
    if (threadIdx.x == 0) {
        partial[((int64_t)c * parts + part) * 2] = s;
        partial[((int64_t)c * parts + part) * 2 + 1] = sx;
    }
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(BnGeom g, const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ mean_rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ partial,
                                                      float* __restrict__ gx, float* __restrict__ g_gamma, float* __restrict__ g_beta) {
    const int part = blockIdx.x, c = blockIdx.y, parts = g.n * g.chunks;
    double s, sx;
    bn_totals(partial, parts, c, s, sx);
    const double m = (double)g.n * g.hw;
    const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1];
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    const float mg = (float)(s / m), mgx = (float)(sx / m);
    if (part == 0 && threadIdx.x == 0) {
        if (g_gamma) g_gamma[c] = (float)sx;
        if (g_beta) g_beta[c] = (float)s;
    }
    if (!gx) return;
    int64_t base;
    int r0, r1;
    bn_range(g, part, c, base, r0, r1);
    const float* xp = x + base;
    const float* gp = gy + base;
    float* op = gx + base;
    const float k = rstd * ga;
    for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
        const float xv = xp[r], xhat = (xv - mean) * rstd;
        float gv = gp[r];
        if (RELU && !(xhat * ga + be > 0.0f)) gv = 0.0f;
        op[r] = ((gv - mg) - xhat * mgx) * k;
    }
}

// Small maps (the deep stages of the trunk: at most BN_ONE_MAX elements per channel; with more, a channel per workgroup leaves most of the chip idle): ONE launch each way, a workgroup per channel reads its n
// planes once -- sums, then values from the registers.  Two thirds of the trunk's normalisations are of this kind and
// cost a launch's latency, not traffic: half the launches.
#define BN_ONE_MAX 8192
// (element e of the channel's n * hw values: image e / hw, offset e % hw; a thread keeps its <= BN_ONE_MAX / 256 values in registers between the passes)
#define BN_ONE_PER (BN_ONE_MAX / 256)
__device__ __forceinline__ int64_t bn_one_offset(const BnGeom& g, int c, int e, float inv_hw) {
    int img = (int)((float)e * inv_hw);                       // e / hw for e < 2^13 via one multiply, corrected by one step either way
    int r = e - img * g.hw;
    if (r < 0) { --img; r += g.hw; }
    if (r >= g.hw) { ++img; r -= g.hw; }
    return ((int64_t)img * g.c + c) * g.hw + r;
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_fwd_one_k(BnGeom g, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                    float momentum, float* __restrict__ y, float* __restrict__ mean_rstd, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, int64_t* __restrict__ num_batches) {
    const int c = blockIdx.x, total = g.n * g.hw;
    const float inv_hw = 1.0f / (float)g.hw;
    float v[BN_ONE_PER];
    int64_t off[BN_ONE_PER];
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int j = 0; j < BN_ONE_PER; ++j) {
        const int e = (int)threadIdx.x + 256 * j;
        v[j] = 0.0f;
        off[j] = 0;
        if (e < total) {
            off[j] = bn_one_offset(g, c, e, inv_hw);
            v[j] = x[off[j]];
            const double a = v[j];
            s += a;
            ss += a * a;
        }
    }
    bn_block_sum2(s, ss);
    const double m = (double)total;
    const double mean_d = s / m;
    double var_d = ss / m - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var_d + (double)eps));
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    if (threadIdx.x == 0) {
        mean_rstd[2 * c] = mean;
        mean_rstd[2 * c + 1] = rstd;
        if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(m > 1.0 ? var_d * m / (m - 1.0) : var_d);
        if (c == 0 && num_batches) num_batches[0] += 1;
    }
#pragma unroll
    for (int j = 0; j < BN_ONE_PER; ++j) {
        const int e = (int)threadIdx.x + 256 * j;
        if (e < total) {
            const float o = bn_value(v[j], mean, rstd, ga, be);
            y[off[j]] = RELU ? fmaxf(o, 0.0f) : o;
        }
    }
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_one_k(BnGeom g, const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ mean_rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ gx,
                                                    float* __restrict__ g_gamma, float* __restrict__ g_beta) {
    const int c = blockIdx.x, total = g.n * g.hw;
    const float inv_hw = 1.0f / (float)g.hw;
    const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1];
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    float xh[BN_ONE_PER], gv[BN_ONE_PER];
    int64_t off[BN_ONE_PER];
    double s = 0.0, sx = 0.0;
#pragma unroll
    for (int j = 0; j < BN_ONE_PER; ++j) {
        const int e = (int)threadIdx.x + 256 * j;
        xh[j] = 0.0f;
        gv[j] = 0.0f;
        off[j] = 0;
        if (e < total) {
            off[j] = bn_one_offset(g, c, e, inv_hw);
            xh[j] = (x[off[j]] - mean) * rstd;
            gv[j] = gy[off[j]];
            if (RELU && !(xh[j] * ga + be > 0.0f)) gv[j] = 0.0f;
            s += (double)gv[j];
            sx += (double)gv[j] * (double)xh[j];
        }
    }
    bn_block_sum2(s, sx);
    const double m = (double)total;
    const float mg = (float)(s / m), mgx = (float)(sx / m);
    if (threadIdx.x == 0) {
        if (g_gamma) g_gamma[c] = (float)sx;
        if (g_beta) g_beta[c] = (float)s;
    }
    if (!gx) return;
    const float k = rstd * ga;
#pragma unroll
    for (int j = 0; j < BN_ONE_PER; ++j) {
        const int e = (int)threadIdx.x + 256 * j;
        if (e < total) gx[off[j]] = ((gv[j] - mg) - xh[j] * mgx) * k;
    }
}
static bool bn_one_launch(int n, int hw) { return (int64_t)n * hw <= BN_ONE_MAX && getenv("GENS_K22_TWO_LAUNCHES") == nullptr; }

static int bn_geom(const char* who, int n, int c, int hw, BnGeom& g) {
    GENS_CHECK_ARG(n > 0 && c > 0 && hw > 0, GENS_EINVAL, "%s: bad shape (%d, %d, %d)", who, n, c, hw);
    GENS_CHECK_ARG(c <= 65535, GENS_ELIMIT, "%s: %d channels", who, c);
    GENS_CHECK_ARG((int64_t)n * c * hw < ((int64_t)1 << 40), GENS_ELIMIT, "%s: tensor too large", who);
    g.n = n; g.c = c; g.hw = hw;
    int chunks = (hw + 4095) / 4096;                           // ~4 096 elements (four per thread and trip, four trips) per part ...
    const int want = (1024 + n * c - 1) / (n * c);             // ... unless the launch would have fewer than ~1 000 workgroups
    if (chunks < want) chunks = want;
    if (chunks > (hw + 255) / 256) chunks = (hw + 255) / 256;  // at least 256 elements per part
    if (chunks > 256 / n) chunks = 256 / n;                    // at most 256 parts per channel (one trip of bn_totals)
    if (chunks < 1) chunks = 1;
    g.chunks = chunks;
    GENS_CHECK_ARG(n * chunks <= 4096, GENS_ELIMIT, "%s: batch of %d", who, n);
    return 0;
}

// doubles of scratch a forward or backward call needs: 2 per (channel, part)
extern "C" int64_t gens_batchnorm2d_scratch_doubles(int n, int c, int hw) {
    BnGeom g;
    if (bn_geom("gens_batchnorm2d_scratch_doubles", n, c, hw, g)) return 0;
    return (int64_t)2 * c * n * g.chunks;
}

extern "C" int gens_batchnorm2d_train_fwd(const float* x, const float* gamma, const float* beta, int n, int c, int hw, float eps, float momentum, int relu,
                                          float* y, float* mean_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                          double* scratch, void* stream) {
    BnGeom g;
    if (int e = bn_geom("gens_batchnorm2d_train_fwd", n, c, hw, g)) return e;
    GENS_CHECK_ARG(x && y && mean_rstd && scratch, GENS_EINVAL, "gens_batchnorm2d_train_fwd: null pointer");
    const dim3 grid((unsigned)(n * g.chunks), (unsigned)c);
    hipStream_t s = (hipStream_t)stream;
    if (bn_one_launch(n, hw)) {
        if (relu) bn_fwd_one_k<true><<<c, 256, 0, s>>>(g, x, gamma, beta, eps, momentum, y, mean_rstd, running_mean, running_var, num_batches_tracked);
        else bn_fwd_one_k<false><<<c, 256, 0, s>>>(g, x, gamma, beta, eps, momentum, y, mean_rstd, running_mean, running_var, num_batches_tracked);
        return gens_launch_status("gens_batchnorm2d_train_fwd");
    }
    bn_stats_k<<<grid, 256, 0, s>>>(g, x, scratch);
    if (relu) bn_apply_k<true><<<grid, 256, 0, s>>>(g, x, scratch, gamma, beta, eps, momentum, y, mean_rstd, running_mean, running_var, num_batches_tracked);
    else bn_apply_k<false><<<grid, 256, 0, s>>>(g, x, scratch, gamma, beta, eps, momentum, y, mean_rstd, running_mean, running_var, num_batches_tracked);
    return gens_launch_status("gens_batchnorm2d_train_fwd");
}

extern "C" int gens_batchnorm2d_train_bwd(const float* x, const float* grad_y, const float* mean_rstd, const float* gamma, const float* beta, int n, int c,
                                          int hw, int relu, float* grad_x, float* grad_gamma, float* grad_beta, double* scratch, void* stream) {
    BnGeom g;
    if (int e = bn_geom("gens_batchnorm2d_train_bwd", n, c, hw, g)) return e;
    GENS_CHECK_ARG(x && grad_y && mean_rstd && scratch, GENS_EINVAL, "gens_batchnorm2d_train_bwd: null pointer");
    GENS_CHECK_ARG(grad_x || grad_gamma || grad_beta, GENS_EINVAL, "gens_batchnorm2d_train_bwd: no output requested");
    const dim3 grid((unsigned)(n * g.chunks), (unsigned)c);
    hipStream_t s = (hipStream_t)stream;
    if (bn_one_launch(n, hw)) {
        if (relu) bn_bwd_one_k<true><<<c, 256, 0, s>>>(g, x, grad_y, mean_rstd, gamma, beta, grad_x, grad_gamma, grad_beta);
        else bn_bwd_one_k<false><<<c, 256, 0, s>>>(g, x, grad_y, mean_rstd, gamma, beta, grad_x, grad_gamma, grad_beta);
        return gens_launch_status("gens_batchnorm2d_train_bwd");
    }
    if (relu) {
        bn_bwd_stats_k<true><<<grid, 256, 0, s>>>(g, x, grad_y, mean_rstd, gamma, beta, scratch);
        bn_bwd_apply_k<true><<<grid, 256, 0, s>>>(g, x, grad_y, mean_rstd, gamma, beta, scratch, grad_x, grad_gamma, grad_beta);
    } else {
        bn_bwd_stats_k<false><<<grid, 256, 0, s>>>(g, x, grad_y, mean_rstd, gamma, beta, scratch);
        bn_bwd_apply_k<false><<<grid, 256, 0, s>>>(g, x, grad_y, mean_rstd, gamma, beta, scratch, grad_x, grad_gamma, grad_beta);
    }
    return gens_launch_status("gens_batchnorm2d_train_bwd");
}
