// K16: InstanceNorm3d (no affine, biased variance) + ReLU of the cost-volume U-Net's blocks (reference: nn.InstanceNorm3d followed by
// nn.ReLU after every convolution, models/modules/reg_network.py:16-17,39-40), forward and backward, batch 1.  torch runs the pair as
// batch_norm + relu in 11.4 ms forward / 23 ms forward + backward per 8-channel 256^3 tensor (scripts/probe/cnn_probe.py); it is four
// streaming passes over 537 MB planes: HBM-bound.
//
//   forward    mean_c, var_c over the plane (pass 1: per-workgroup sums of x and x^2, accumulated in float64, added up by the host side
//              in float64);  y = max((x - mean) * rstd, 0)  (pass 2)                                      A = 4 B + 8 B per element
//   backward   g = gy [xhat > 0];  gx = rstd (g - mean(g) - xhat mean(g xhat))     (pass 1: sums of g and g xhat; pass 2: gx)
//              xhat is recomputed from x with the float32 expression of the forward pass, so the ReLU decision is the same one
//                                                                                                        A = 8 B + 12 B per element
// Layout: x (c, n) channel planes; grid (workgroups per plane, c); 16-byte accesses when n is a multiple of 4.
#include "common.h"

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
    return v;                                                                     // lane 0
}

// partial sums of a workgroup -> out[(channel * gridDim.x + block) * 2 + {0, 1}]
__device__ __forceinline__ void block_pair_out(double a, double b, double* __restrict__ out) {
    __shared__ double red[2][4];
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = out + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void instnorm_stats_k(const float* __restrict__ x, int64_t n, double* __restrict__ out) {
    const float* xc = x + (int64_t)blockIdx.y * n;
    double s = 0.0, ss = 0.0;
    if (VEC) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const float4 v = ((const float4*)xc)[i];
            const double a = v.x, b = v.y, c = v.z, d = v.w;
            s += (a + b) + (c + d);
            ss += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const double a = xc[i];
            s += a;
            ss += a * a;
        }
    }
    block_pair_out(s, ss, out);
}

__device__ __forceinline__ float norm_relu(float x, float m, float r) { return fmaxf((x - m) * r, 0.0f); }

// y = max((x - mean) * rstd, 0) [+ skip]: the decoder blocks of the U-Net add the encoder's tensor right after the block (reg_network.py:158)
template <bool VEC>
__global__ __launch_bounds__(256) void instnorm_relu_fwd_k(const float* __restrict__ x, const float* __restrict__ mr, const float* __restrict__ skip,
                                                           int64_t n, float* __restrict__ y) {
    const float m = mr[2 * blockIdx.y], r = mr[2 * blockIdx.y + 1];
    const float* xc = x + (int64_t)blockIdx.y * n;
    const float* sc = skip ? skip + (int64_t)blockIdx.y * n : nullptr;
    float* yc = y + (int64_t)blockIdx.y * n;
    if (VEC) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const float4 v = ((const float4*)xc)[i];
            float4 o = make_float4(norm_relu(v.x, m, r), norm_relu(v.y, m, r), norm_relu(v.z, m, r), norm_relu(v.w, m, r));
            if (sc) {
                const float4 k = ((const float4*)sc)[i];
                o = make_float4(o.x + k.x, o.y + k.y, o.z + k.z, o.w + k.w);
            }
            ((float4*)yc)[i] = o;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) yc[i] = norm_relu(xc[i], m, r) + (sc ? sc[i] : 0.0f);
    }
}

__device__ __forceinline__ void bwd_terms(float x, float gy, float m, float r, double& s1, double& s2) {
    const float xh = (x - m) * r;
    if (xh > 0.0f) {
        s1 += (double)gy;
        s2 += (double)gy * (double)xh;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void instnorm_relu_bwd_stats_k(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ mr,
                                                                 int64_t n, double* __restrict__ out) {
    const float m = mr[2 * blockIdx.y], r = mr[2 * blockIdx.y + 1];
    const float* xc = x + (int64_t)blockIdx.y * n;
    const float* gc = gy + (int64_t)blockIdx.y * n;
    double s1 = 0.0, s2 = 0.0;
    if (VEC) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const float4 v = ((const float4*)xc)[i], g = ((const float4*)gc)[i];
            bwd_terms(v.x, g.x, m, r, s1, s2);
            bwd_terms(v.y, g.y, m, r, s1, s2);
            bwd_terms(v.z, g.z, m, r, s1, s2);
            bwd_terms(v.w, g.w, m, r, s1, s2);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bwd_terms(xc[i], gc[i], m, r, s1, s2);
    }
    block_pair_out(s1, s2, out);
}

// gx = r (g - m1 - xhat m2), g = gy [xhat > 0], m1 = mean(g), m2 = mean(g xhat)
__device__ __forceinline__ float bwd_value(float x, float gy, float m, float r, float m1, float m2) {
    const float xh = (x - m) * r;
    const float g = xh > 0.0f ? gy : 0.0f;
    return r * ((g - m1) - xh * m2);
}

template <bool VEC>
__global__ __launch_bounds__(256) void instnorm_relu_bwd_k(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ mr,
                                                           const float* __restrict__ m12, int64_t n, float* __restrict__ gx) {
    const float m = mr[2 * blockIdx.y], r = mr[2 * blockIdx.y + 1], m1 = m12[2 * blockIdx.y], m2 = m12[2 * blockIdx.y + 1];
    const float* xc = x + (int64_t)blockIdx.y * n;
    const float* gc = gy + (int64_t)blockIdx.y * n;
    float* oc = gx + (int64_t)blockIdx.y * n;
    if (VEC) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const float4 v = ((const float4*)xc)[i], g = ((const float4*)gc)[i];
            ((float4*)oc)[i] = make_float4(bwd_value(v.x, g.x, m, r, m1, m2), bwd_value(v.y, g.y, m, r, m1, m2), bwd_value(v.z, g.z, m, r, m1, m2),
                                           bwd_value(v.w, g.w, m, r, m1, m2));
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) oc[i] = bwd_value(xc[i], gc[i], m, r, m1, m2);
    }
}

extern "C" int gens_instnorm_blocks(int c, int64_t n) {
    if (c <= 0 || n <= 0) return 0;
    int64_t b = (n + 8191) / 8192;                        // >= 8 float4 per thread where the plane is that large
    const int64_t cap = (8192 + c - 1) / c;               // ~8 000 workgroups per launch
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

#define INSTNORM_LAUNCH(kernel, ...)                                                                              \
    do {                                                                                                          \
        const dim3 grid(gens_instnorm_blocks(c, n), c);                                                           \
        if ((n & 3) == 0) hipLaunchKernelGGL((kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);           \
    } while (0)

extern "C" int gens_instnorm_stats(const float* x, int c, int64_t n, double* partials, void* stream) {
    GENS_CHECK_ARG(x && partials && c > 0 && n > 0 && c <= 65535, GENS_EINVAL, "gens_instnorm_stats: bad argument");
    INSTNORM_LAUNCH(instnorm_stats_k, x, n, partials);
    return gens_launch_status("gens_instnorm_stats");
}

// partials (c, blocks, 2) of either statistics kernel -> out (c, 2) floats in ONE launch (the PyTorch glue this replaces was nine tiny launches per
// layer forward and three backward): mode 0 = (mean, 1 / sqrt(max(E[x^2] - mean^2, 0) + eps)), mode 1 = (sum_0 / n, sum_1 / n); float64 inside.
__global__ __launch_bounds__(64) void instnorm_finish_k(const double* __restrict__ part, int c, int blocks, double inv_n, double eps, int mode,
                                                        float* __restrict__ out) {
    const int ch = blockIdx.x, lane = threadIdx.x;          // a wavefront per channel: lane l adds partials l, l + 64, ... (a fixed order), then the lanes meet
    double s0 = 0.0, s1 = 0.0;
    for (int b = lane; b < blocks; b += 64) {
        s0 += part[((int64_t)ch * blocks + b) * 2];
        s1 += part[((int64_t)ch * blocks + b) * 2 + 1];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        s0 += __shfl_down(s0, s, 64);
        s1 += __shfl_down(s1, s, 64);
    }
    if (lane != 0) return;
    s0 *= inv_n;
    s1 *= inv_n;
    if (mode == 0) {
        const double var = fmax(s1 - s0 * s0, 0.0) + eps;
        out[2 * ch] = (float)s0;
        out[2 * ch + 1] = (float)(1.0 / sqrt(var));
    } else {
        out[2 * ch] = (float)s0;
        out[2 * ch + 1] = (float)s1;
    }
}

extern "C" int gens_instnorm_finish(const double* partials, int c, int64_t n, double eps, int mode, float* out, void* stream) {
    GENS_CHECK_ARG(partials && out && c > 0 && n > 0 && c <= 65535 && (mode == 0 || mode == 1), GENS_EINVAL, "gens_instnorm_finish: bad argument");
    hipLaunchKernelGGL(instnorm_finish_k, dim3(c), dim3(64), 0, (hipStream_t)stream, partials, c, gens_instnorm_blocks(c, n), 1.0 / (double)n,
                       eps, mode, out);
    return gens_launch_status("gens_instnorm_finish");
}

extern "C" int gens_instnorm_relu_fwd(const float* x, const float* mean_rstd, int c, int64_t n, float* y, void* stream) {
    GENS_CHECK_ARG(x && mean_rstd && y && c > 0 && n > 0 && c <= 65535, GENS_EINVAL, "gens_instnorm_relu_fwd: bad argument");
    INSTNORM_LAUNCH(instnorm_relu_fwd_k, x, mean_rstd, (const float*)nullptr, n, y);
    return gens_launch_status("gens_instnorm_relu_fwd");
}

extern "C" int gens_instnorm_relu_add_fwd(const float* x, const float* mean_rstd, const float* skip, int c, int64_t n, float* y, void* stream) {
    GENS_CHECK_ARG(x && mean_rstd && skip && y && c > 0 && n > 0 && c <= 65535, GENS_EINVAL, "gens_instnorm_relu_add_fwd: bad argument");
    INSTNORM_LAUNCH(instnorm_relu_fwd_k, x, mean_rstd, skip, n, y);
    return gens_launch_status("gens_instnorm_relu_add_fwd");
}

extern "C" int gens_instnorm_relu_bwd_stats(const float* x, const float* gy, const float* mean_rstd, int c, int64_t n, double* partials, void* stream) {
    GENS_CHECK_ARG(x && gy && mean_rstd && partials && c > 0 && n > 0 && c <= 65535, GENS_EINVAL, "gens_instnorm_relu_bwd_stats: bad argument");
    INSTNORM_LAUNCH(instnorm_relu_bwd_stats_k, x, gy, mean_rstd, n, partials);
    return gens_launch_status("gens_instnorm_relu_bwd_stats");
}

extern "C" int gens_instnorm_relu_bwd(const float* x, const float* gy, const float* mean_rstd, const float* g_means, int c, int64_t n, float* gx,
                                      void* stream) {
    GENS_CHECK_ARG(x && gy && mean_rstd && g_means && gx && c > 0 && n > 0 && c <= 65535, GENS_EINVAL, "gens_instnorm_relu_bwd: bad argument");
    INSTNORM_LAUNCH(instnorm_relu_bwd_k, x, gy, mean_rstd, g_means, n, gx);
    return gens_launch_status("gens_instnorm_relu_bwd");
}
