// K17: the SDF network of a TRAINING step in four launches -- value, gradient and `smooth` vector forward; the loss backward
// into every layer's matrix / bias and into the volume pyramid -- instead of the ~2 400 launches of the PyTorch autograd graph.
//
// Replaces, in train / fine-tune mode, SDFNetwork.sdf + SDFNetwork.gradient (create_graph twice) and the backward of both
// (/root/reference/models/modules/sdf_network.py:98-154, driven from implicit_surface.py:179-191,257,305,490 and loss.backward()),
// including the multi-level trilinear look-up with the reference's truncation: what grad2_3d returns is constant in the loss
// backward (cuda_gridsample.py:110-123), so no third derivative of the sampler appears.
//
// Mathematics (oracle/sdf_train_oracle.py `by_sweeps` is the same text in torch; tests/test_sdf_train_oracle.py proves it equal to
// autograd).  Per point, with a_l = W_l z_l + b_l, h_l = softplus(a_l), v = (1,1,1):
//   forward launch (sdf_train_fwd_k):   value sweep a, tangent sweep a' (along v)           -> y
//                                        reverse sweep lambda (adjoint of y), its tangent mu -> g = dy/dx, s = d(sum g)/dx
//   backward launch (sdf_train_bwd_k):  a, a' again, tangent nu (along s_bar), second tangent kappa (along (v, s_bar) and g_bar)
//                                        reverse sweeps lambda, mu, rho (tangent of lambda along s_bar), omega (adjoint of the loss)
//     dL/dW_l = sum_points  omega_a z^T + rho_a z'^T + lambda_a kappa_z^T + mu_a nu_z^T      (gens_gemm_tn over operand rows written here)
//     dL/dvolumes: three trilinear scatters (sdf_train_scatter_k)
//
// MI355X mapping (K6's, widened).  A workgroup = 4 waves = 32 points; wave w owns output columns [32w, 32w+32) of every layer.  The
// sweeps that share a weight matrix run TOGETHER: 2 (forward launch) or 4 (backward launch) activation tiles in LDS are multiplied by
// ONE stream of B fragments, so a 16-byte weight load feeds 8 / 16 v_mfma_f32_32x32x2_f32 instead of 4 and the independent accumulator
// chains keep the matrix pipe busy with a single wave per SIMD (the four 32 x 236 tiles of the L = 5 backward fill 139 KB of the CU's
// 160 KB LDS).  Per-element layer state the reverse sweeps need (a, a', nu_a, kappa_a) is parked in an HBM scratch in accumulator
// layout (1-KB coalesced stores, read back by the same lane), not in registers: 6 layers x 4 values x 16 would be 384 VGPRs.
// Exact fp32 arithmetic throughout (fp32 MFMA is an fmaf chain); activations use the hardware exp / log / rcp like K6.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TR_M 32
#define TR_H 128
#define TR_PE 27
#define TR_PE_K 32
#define TR_PE_STRIDE 36
#define TR_SKIP_H 101
#define TR_NLAYER 6
#define TR_SQ2 0.70710678118654752440f
#ifndef TR_PF_B
#define TR_PF_B 8            // weight groups in flight per wave in the backward launch's products (one wave per SIMD: nothing else hides a load)
#endif
#define TR_DEAD (-1.0e30f)   // pre-activation stored for the 27 pass-through columns of layer 2: softplus' = '' = ''' = 0

struct SdfTrainWeights {
    const float4* wf[TR_NLAYER];   // forward B groups [4][G_l][64] float4, bias in reduction row K_l (layout of gens_sdf_mlp)
    const float4* wb[TR_NLAYER];   // transposed B groups [n_tile][16][64] float4
    const float* w_last;           // row 0 of layer 6 (128 + FE)
    const float* b_last;           // DEVICE scalar: bias 0 of layer 6 (weights change every step; no host read-back)
};

// softplus(beta = 100, threshold 20) with its first three derivatives, branch-free (hardware exp2 / log2 / rcp).  Above the
// threshold torch's softplus is the identity: derivatives 1, 0, 0 (v_cndmask discards the inf / NaN of the other branch).
__device__ __forceinline__ void softplus_d3(float a, float& h, float& d1, float& d2, float& d3) {
#pragma clang fp contract(fast)      // (no index or mask decision hangs on these values: fused multiply-adds are as good as the separate roundings)
    const float e = __builtin_amdgcn_exp2f(a * 144.269504088896340736f);   // e^{100 a}
    const float u = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(u);                              // 1 - sigmoid(100 a)
    const float t = e * r;                                                 // sigmoid(100 a)
    const float q = t * r;
    const bool lin = a > 0.2f;
    h = lin ? a : __builtin_amdgcn_logf(u) * 0.0069314718055994530942f;
    d1 = lin ? 1.0f : t;
    d2 = lin ? 0.0f : 100.0f * q;
    d3 = lin ? 0.0f : 1.0e4f * q * (r - t);
}

// acc[t] += A_t (32 x 8G, rows of stride given by the caller's pointer arithmetic; tile t starts `tstride` floats after tile 0)
// * B (G packed groups).  `a` points at this lane's row + 4 * half of tile 0, `b` at this lane's float4 of group 0.
template <int G, int T, int PF = 4>
__device__ __forceinline__ void mfma_tiles(const float* __restrict__ a, const int tstride, const float4* __restrict__ b, f32x16 (&acc)[T]) {
    float4 pre[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (j < G) pre[j] = b[64 * j];
    float4 av[T];
#pragma unroll
    for (int t = 0; t < T; ++t) av[t] = *(const float4*)(a + t * tstride);
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const float4 bv = pre[j % PF];
        float4 ac[T];
#pragma unroll
        for (int t = 0; t < T; ++t) ac[t] = av[t];
        if (j + 1 < G) {
#pragma unroll
            for (int t = 0; t < T; ++t) av[t] = *(const float4*)(a + t * tstride + 8 * (j + 1));
        }
        if (j + PF < G) pre[j % PF] = b[64 * (j + PF)];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t].x, bv.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t].y, bv.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t].z, bv.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t].w, bv.w, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ int tr_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int T>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

// The 8 corners of one level around x: value and the derivatives of the interpolant along up to three directions.
// dir[k] (k < ND) are direction vectors in point space; out[0] = f, out[1 + k] = J dir[k]; when MIXED, mix[a] = sum_b d2f/dx_a dx_b
// (direction v = 1) and jac[a] = df/dx_a.
template <int ND, bool MIXED>
__device__ __forceinline__ void corner_sums(const LevelSet& vols, int l, const float x[3], bool live, const float (*dir)[3], float4 (&out)[1 + ND],
                                            float4 (&jac)[3], float4 (&mix)[3]) {
    const int Xd = vols.dx[l], Yd = vols.dy[l], Zd = vols.dz[l];
    const float4* v = (const float4*)vols.data[l];
    const int sz[3] = {Xd, Yd, Zd};
    float w0[3], w1[3], k[3];
    int i0[3];
    bool in0[3], in1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pos = (x[a] + 1.0f) / 2.0f * (float)(sz[a] - 1);
        const float f = fminf(fmaxf(floorf(pos), -2.0f), (float)sz[a] + 1.0f);
        i0[a] = (int)f;
        w0[a] = (f + 1.0f) - pos;
        w1[a] = pos - f;
        in0[a] = i0[a] >= 0 && i0[a] < sz[a];
        in1[a] = i0[a] + 1 >= 0 && i0[a] + 1 < sz[a];
        k[a] = (float)(sz[a] - 1) / 2.0f;
    }
#pragma unroll
    for (int q = 0; q < 1 + ND; ++q) out[q] = f4_zero();
#pragma unroll
    for (int a = 0; a < 3; ++a) { jac[a] = f4_zero(); mix[a] = f4_zero(); }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int a = c >> 2, b = (c >> 1) & 1, d = c & 1;
        const bool ok = live && (a ? in1[0] : in0[0]) && (b ? in1[1] : in0[1]) && (d ? in1[2] : in0[2]);
        const int cx = min(max(i0[0] + a, 0), Xd - 1), cy = min(max(i0[1] + b, 0), Yd - 1), cz = min(max(i0[2] + d, 0), Zd - 1);
        float4 t = v[((int64_t)cx * Yd + cy) * Zd + cz];
        if (!ok) t = f4_zero();
        const float wx = a ? w1[0] : w0[0], wy = b ? w1[1] : w0[1], wz = d ? w1[2] : w0[2];
        const float sx = (a ? k[0] : -k[0]), sy = (b ? k[1] : -k[1]), sz_ = (d ? k[2] : -k[2]);
        const float dwx = sx * wy * wz, dwy = wx * sy * wz, dwz = wx * wy * sz_;
        out[0] = f4_madd(out[0], t, wx * wy * wz);
#pragma unroll
        for (int q = 0; q < ND; ++q) out[1 + q] = f4_madd(out[1 + q], t, dwx * dir[q][0] + dwy * dir[q][1] + dwz * dir[q][2]);
        if constexpr (MIXED) {
            jac[0] = f4_madd(jac[0], t, dwx);
            jac[1] = f4_madd(jac[1], t, dwy);
            jac[2] = f4_madd(jac[2], t, dwz);
            const float mxy = sx * sy * wz, mxz = sx * wy * sz_, myz = wx * sy * sz_;
            mix[0] = f4_madd(mix[0], t, mxy + mxz);
            mix[1] = f4_madd(mix[1], t, mxy + myz);
            mix[2] = f4_madd(mix[2], t, mxz + myz);
        }
    }
}

__device__ __forceinline__ float f4_at(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// ====================================================================================================================
// forward launch: y, g = dy/dx, s = d(sum_k g_k)/dx
// ====================================================================================================================
// (round 5) SIXTEEN points per workgroup, value and tangent sweep STACKED in one 32-row tile (rows 0-15 / 16-31), as the backward launch below:
// 40 KB of LDS (50 KB at five levels) and < 168 registers, three workgroups per CU instead of two.
#define TR_MB 16
template <int FE>
__global__ __launch_bounds__(256, 3) void sdf_train_fwd_k(SdfTrainWeights W, LevelSet vols, const float* __restrict__ pts,
                                                          const int64_t* __restrict__ index, int64_t n_max, const int32_t* __restrict__ n_dev,
                                                          float2* __restrict__ stash, float* __restrict__ y_out, float* __restrict__ g_out,
                                                          float* __restrict__ s_out) {
    constexpr int CF = FE / 5, KIN = TR_H + FE, KP = (KIN + 8) / 8 * 8, GIN = KP / 8, RS = KP + 4;
    static_assert(KP > KIN, "a pad column carries the bias");
    constexpr int NT_B = 4 + ((FE + 31) / 32 <= 2 ? 2 : 4);      // 4 hidden tiles + the conditioning tiles, padded to 2 or 4 (zero columns: gens_sdf_train_pack)
    constexpr int XS = TR_MB * RS, PS = TR_MB * TR_PE_STRIDE;    // from a point's value row to its tangent row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                        // [32][RS]: value rows [h | e | 1 | 0], tangent rows [h' | e' | 0]; h part reused by the reverse sweeps
    float* PE = X + 32 * RS;                // [32][TR_PE_STRIDE]: point encoding and its tangent
    float* GPE = PE + 32 * TR_PE_STRIDE;    // [32][TR_PE_STRIDE]: lambda / mu with respect to the point encoding
    float* JAC = GPE + 32 * TR_PE_STRIDE;   // [16][CF][3] df/dx ; then [16][CF][3] sum_b d2f/dx dx_b
    float* LF = JAC + 2 * TR_MB * CF * 3;   // [2][16][CF]: lambda_f, mu_f
    float* RED = GPE;                       // [16][8]: the output row's partial sums, dead before the reverse sweeps write GPE

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * TR_MB;
    // point i of the launch is pts[index[i]] (NULL: i) and its results go to row index[i] of the outputs; only the first min(n_max, *n_dev)
    // points exist (the masked evaluation of implicit_surface.py:174-191 without a host-side count): later workgroups leave at once
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    if (m0 >= n) return;
    const int a_lane = lane & 31, a_half = 4 * (lane >> 5);
    const int col = 32 * wave + (lane & 31);

    // ------------------------------------------------------------------ prologue: threads 0..127 gather the levels, 128..255 encode the point
    {
        const int p = (tid & 127) >> 3, sub = tid & 7;
        const bool enc = tid >= 128;
        const int64_t row = m0 + p;
        const bool live = row < n;
        float x[3] = {0.f, 0.f, 0.f};
        if (live) {
            const int64_t src = index ? index[row] : row;
            x[0] = pts[3 * src]; x[1] = pts[3 * src + 1]; x[2] = pts[3 * src + 2];
        }
        if (enc && sub < 3) {
            const int a = sub;
            float* pe = PE + p * TR_PE_STRIDE;
            float* pd = pe + PS;
            pe[a] = x[a];
            pd[a] = 1.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = (float)(1 << k);
                float s, c;
                hw_sincos(x[a] * f, s, c);
                pe[3 + 6 * k + a] = s;
                pe[6 + 6 * k + a] = c;
                pd[3 + 6 * k + a] = f * c;
                pd[6 + 6 * k + a] = -f * s;
            }
            if (a == 0) {
                pe[TR_PE] = 1.0f;
                pd[TR_PE] = 0.0f;
#pragma unroll
                for (int k = TR_PE + 1; k < TR_PE_K; ++k) { pe[k] = 0.0f; pd[k] = 0.0f; }
            }
        }
        if (enc && sub == 7) {
            X[p * RS + KIN] = 1.0f;
            X[XS + p * RS + KIN] = 0.0f;
#pragma unroll
            for (int k = KIN + 1; k < KP; ++k) { X[p * RS + k] = 0.0f; X[XS + p * RS + k] = 0.0f; }
        }
        if (!enc && sub < vols.n) {
            const int l = sub;
            const float dir[1][3] = {{1.0f, 1.0f, 1.0f}};
            float4 o[2], jac[3], mix[3];
            corner_sums<1, true>(vols, l, x, live, dir, o, jac, mix);
            float* xr = X + p * RS + TR_H;
            float* xd = xr + XS;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * l + c;
                const float f = f4_at(o[0], c), fd = f4_at(o[1], c);
                float s1, c1, s2, c2;
                hw_sincos(f, s1, c1);
                hw_sincos(2.0f * f, s2, c2);
                xr[ch] = f; xr[CF + ch] = s1; xr[2 * CF + ch] = c1; xr[3 * CF + ch] = s2; xr[4 * CF + ch] = c2;
                xd[ch] = fd; xd[CF + ch] = c1 * fd; xd[2 * CF + ch] = -s1 * fd; xd[3 * CF + ch] = 2.0f * c2 * fd; xd[4 * CF + ch] = -2.0f * s2 * fd;
                float* j = JAC + (p * CF + ch) * 3;
                float* jd = j + TR_MB * CF * 3;
#pragma unroll
                for (int a = 0; a < 3; ++a) { j[a] = f4_at(jac[a], c); jd[a] = f4_at(mix[a], c); }
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ forward sweeps: value + tangent
    const int64_t sbase = (int64_t)blockIdx.x * TR_NLAYER;
    for (int l = 0; l < TR_NLAYER; ++l) {
        f32x16 acc[1];
        zero_acc(acc);
        if (l == 0)
            mfma_tiles<TR_PE_K / 8, 1>(PE + a_lane * TR_PE_STRIDE + a_half, 0, W.wf[0] + (size_t)wave * (TR_PE_K / 8) * 64 + lane, acc);
        else
            mfma_tiles<GIN, 1>(X + a_lane * RS + a_half, 0, W.wf[l] + (size_t)wave * GIN * 64 + lane, acc);
        __syncthreads();
        float2* st = stash + (((sbase + l) * 4 + wave) * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = tr_acc_row(i, lane);                 // 0..15: the point; its tangent sits 16 rows further (accumulator register i + 8)
            float h, d1, d2, d3;
            softplus_d3(acc[0][i], h, d1, d2, d3);
            (void)d3;
            float hd = d1 * acc[0][i + 8];
            float2 keep = make_float2(d1, d2 * acc[0][i + 8]);
            if (l == 2) {   // z_3 = [h_2 | pe] / sqrt(2)   (sdf_network.py:111-112)
                if (col < TR_SKIP_H) {
                    h *= TR_SQ2;
                    hd *= TR_SQ2;
                } else {
                    h = PE[row * TR_PE_STRIDE + (col - TR_SKIP_H)] * TR_SQ2;
                    hd = PE[PS + row * TR_PE_STRIDE + (col - TR_SKIP_H)] * TR_SQ2;
                    keep = make_float2(0.0f, 0.0f);
                }
            }
            X[row * RS + col] = h;
            X[XS + row * RS + col] = hd;
            st[i * 64] = keep;
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ layer 6, sdf row only
    if (tid < 128) {
        const int p = tid >> 3, sub = tid & 7;
        const float* xr = X + p * RS;
        float s = 0.0f;
        for (int k = sub; k < KIN; k += 8) s += xr[k] * W.w_last[k];
        RED[p * 8 + sub] = s;
    }
    __syncthreads();
    if (tid < TR_MB) {
        const int64_t row = m0 + tid;
        if (row < n) {
            float s = W.b_last[0];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += RED[tid * 8 + k];
            y_out[index ? index[row] : row] = s;
        }
    }

    // ------------------------------------------------------------------ reverse sweeps: lambda + mu
    constexpr bool SPLIT_K = (NT_B - 4) == 2;
    const int fe_tile = SPLIT_K ? 4 + (wave & 1) : 4 + wave;
    const int fe_g0 = SPLIT_K ? 8 * (wave >> 1) : 0;
    constexpr int FE_G = SPLIT_K ? 8 : 16;
    f32x16 gfe[1];
    zero_acc(gfe);
    __syncthreads();
    {
        const float wl = W.w_last[col];
        const float2* st = stash + (((sbase + 5) * 4 + wave) * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = tr_acc_row(i, lane);
            const float2 k = st[i * 64];
            X[row * RS + col] = k.x * wl;            // lambda_a5 = softplus' w6
            X[XS + row * RS + col] = k.y * wl;       // mu_a5 = softplus'' a' w6
        }
    }
    __syncthreads();
    for (int l = 5; l >= 1; --l) {
        f32x16 gh[1];
        zero_acc(gh);
        mfma_tiles<16, 1>(X + a_lane * RS + a_half, 0, W.wb[l] + (size_t)wave * 16 * 64 + lane, gh);
        mfma_tiles<FE_G, 1>(X + a_lane * RS + a_half + 8 * fe_g0, 0, W.wb[l] + ((size_t)fe_tile * 16 + fe_g0) * 64 + lane, gfe);
        __syncthreads();
        const float2* st = stash + (((sbase + (l - 1)) * 4 + wave) * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = tr_acc_row(i, lane);
            float lh = gh[0][i], mh = gh[0][i + 8];
            if (l == 3) {
                lh *= TR_SQ2;
                mh *= TR_SQ2;
                if (col >= TR_SKIP_H) {
                    GPE[row * TR_PE_STRIDE + (col - TR_SKIP_H)] = lh;
                    GPE[PS + row * TR_PE_STRIDE + (col - TR_SKIP_H)] = mh;
                }
            }
            const float2 k = st[i * 64];
            X[row * RS + col] = k.x * lh;
            X[XS + row * RS + col] = k.y * lh + k.x * mh;
        }
        __syncthreads();
    }
    // layer 0: (32 x 128) x (128 x 27); the four waves split the reduction, partial tiles summed in a fixed order
    {
        f32x16 gp[1];
        zero_acc(gp);
        mfma_tiles<4, 1>(X + a_lane * RS + a_half + 32 * wave, 0, W.wb[0] + (size_t)(4 * wave) * 64 + lane, gp);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            X[tr_acc_row(i, lane) * RS + col] = gp[0][i];
            X[XS + tr_acc_row(i, lane) * RS + col] = gp[0][i + 8];
        }
        __syncthreads();
        for (int i = tid; i < 2 * TR_MB * 32; i += 256) {
            const int t = i >> 9, row = (i >> 5) & 15, c = i & 31;
            if (c < TR_PE) {
                const float* xr = X + t * XS + row * RS + c;
                GPE[t * PS + row * TR_PE_STRIDE + c] += ((xr[0] + xr[32]) + xr[64]) + xr[96];
            }
        }
    }
    __syncthreads();
    // conditioning adjoints -> the (dead) h part of the two sweeps
    {
        const int c = 32 * (fe_tile - 4) + (lane & 31);
        if (c < FE && (!SPLIT_K || wave < 2)) {
            const float wl = W.w_last[TR_H + c];      // layer 6 adds the same vector for every point; its tangent is zero
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                X[tr_acc_row(i, lane) * RS + c] = gfe[0][i] + wl;
                X[XS + tr_acc_row(i, lane) * RS + c] = gfe[0][i + 8];
            }
        }
        if (SPLIT_K) {
            __syncthreads();
            if (c < FE && wave >= 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    X[tr_acc_row(i, lane) * RS + c] += gfe[0][i];
                    X[XS + tr_acc_row(i, lane) * RS + c] += gfe[0][i + 8];
                }
            }
        }
    }
    __syncthreads();
    if (tid < 128) {   // lambda_f = E'^T lambda_e ; mu_f = E'^T mu_e + E''[f'] lambda_e
        const int p = tid >> 3, sub = tid & 7;
        if (sub < vols.n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * sub + c;
                const float* e = X + p * RS + TR_H;           // [f, sin f, cos f, sin 2f, cos 2f]
                const float fd = X[XS + p * RS + TR_H + ch];
                const float s1 = e[CF + ch], c1 = e[2 * CF + ch], s2 = e[3 * CF + ch], c2 = e[4 * CF + ch];
                const float* le = X + p * RS;
                const float* me = le + XS;
                const float l0 = le[ch], l1 = le[CF + ch], l2 = le[2 * CF + ch], l3 = le[3 * CF + ch], l4 = le[4 * CF + ch];
                const float lam_f = l0 + c1 * l1 - s1 * l2 + 2.0f * (c2 * l3 - s2 * l4);
                const float mu_f = me[ch] + c1 * me[CF + ch] - s1 * me[2 * CF + ch] + 2.0f * (c2 * me[3 * CF + ch] - s2 * me[4 * CF + ch]) -
                                   fd * (s1 * l1 + c1 * l2 + 4.0f * (s2 * l3 + c2 * l4));
                LF[p * CF + ch] = lam_f;
                LF[TR_MB * CF + p * CF + ch] = mu_f;
            }
        }
    }
    __syncthreads();
    if (tid < TR_MB * 3) {
        const int p = tid / 3, a = tid % 3;
        const int64_t row = m0 + p;
        if (row < n) {
            const float* pe = PE + p * TR_PE_STRIDE;
            const float* lp = GPE + p * TR_PE_STRIDE;
            const float* mp = lp + PS;
            float g = lp[a], s = mp[a];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = (float)(1 << k);
                const float S = pe[3 + 6 * k + a], C = pe[6 + 6 * k + a];
                const float ls = lp[3 + 6 * k + a], lc = lp[6 + 6 * k + a];
                g += f * (ls * C - lc * S);
                s += f * (mp[3 + 6 * k + a] * C - mp[6 + 6 * k + a] * S) - f * f * (ls * S + lc * C);
            }
            const float* j = JAC + p * CF * 3;
            const float* jd = j + TR_MB * CF * 3;
            const float* lf = LF + p * CF;
            const float* mf = lf + TR_MB * CF;
            for (int c = 0; c < CF; ++c) {
                g += j[c * 3 + a] * lf[c];
                s += jd[c * 3 + a] * lf[c] + j[c * 3 + a] * mf[c];
            }
            const int64_t dst = index ? index[row] : row;
            g_out[3 * dst + a] = g;
            s_out[3 * dst + a] = s;
        }
    }
}

// ====================================================================================================================
// backward launch: operand rows of the weight-gradient products, and the three per-point vectors of the volume scatter
// tiles (LDS) / operand pairs:  0: z  with omega_a   1: z' with rho_a   2: kappa_z with lambda_a   3: nu_z with mu_a
// ====================================================================================================================
// (operand rows are POINT-major, the four sweeps of a point adjacent: the rows of the live points are then one contiguous range
// [0, 4 * 32 ceil(n / 32)) whatever the launch's upper bound was, and the weight-gradient products stop there -- gens_gemm_tn_batch's k_live)
// Cycle stamps of one workgroup's phases (build with -DGENS_K17_STAMPS; scripts/probe/k17_stamps_probe.py reads them): never in the shipped library.
#ifdef GENS_K17_STAMPS
__device__ unsigned long long k17_stamps[4][128];
#define TR_STAMP()                                                                                            \
    do {                                                                                                      \
        if (blockIdx.x == 300 && lane == 0 && n_stamp < 128) k17_stamps[wave][n_stamp++] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define TR_STAMP() do { } while (0)
#endif

struct SdfTrainBwdOut {
    float* lop;     // [npad][4][6][128]   omega_a, rho_a, lambda_a, mu_a of layers 0..5
    float* rh;      // [5][npad][4][128]   h parts of the inputs of layers 1..5: z, z', kappa_z, nu_z
    float* re;      // [npad][4][KP-128]   conditioning parts (+ the bias column) of the same four
    float* r0;      // [npad][4][32]       inputs of layer 0
    float* f_hat;   // [npad][CF]          cotangent of the looked-up features
    float* mu_f;    // [npad][CF]
    float* lam_f;   // [npad][CF]
    float* w6_part; // [npad / 16][KP]     this workgroup's share of d loss / d w_last (column K = its bias)
    int64_t npad;
};

// Layout of the backward launch (round 5): a workgroup owns SIXTEEN points and its four sweeps are STACKED two to a 32-row tile -- tile t, row R
// holds sweep 2 t + (R >> 4) of point R & 15 -- so every v_mfma_f32_32x32x2_f32 still multiplies 32 full rows, the four sweeps of a (point, column)
// still meet in ONE lane (accumulator registers i and i + 8 of the two tiles), and the workgroup needs 59 KB of LDS (three levels; 70 KB at five)
// and < 256 registers instead of 118 KB and 360: TWO workgroups per CU, one wave of each per SIMD.  With one workgroup per CU nothing overlapped a
// layer's element-wise pass, its operand-row stores or the prologue's gathers with the matrix pipe: 439 k cycles per 32 points of which 250 k were
// MFMA issue (scripts/probe/k17_stamps_probe.py); now the other workgroup's products run under them.
template <int FE>
__global__ __launch_bounds__(256, 2) void sdf_train_bwd_k(SdfTrainWeights W, LevelSet vols, const float* __restrict__ pts,
                                                          const int64_t* __restrict__ index, int64_t n_max, const int32_t* __restrict__ n_dev,
                                                          const float* __restrict__ y_bar, const float* __restrict__ g_bar,
                                                          const float* __restrict__ s_bar, float4* __restrict__ stash, SdfTrainBwdOut O) {
    constexpr int CF = FE / 5, KIN = TR_H + FE, KP = (KIN + 8) / 8 * 8, GIN = KP / 8, RS = KP + 4, FEP = KP - TR_H;
    constexpr int NT_B = 4 + ((FE + 31) / 32 <= 2 ? 2 : 4);      // 4 hidden tiles + the conditioning tiles, padded to 2 or 4 (zero columns: gens_sdf_train_pack)
    constexpr int XT = 32 * RS, PT = 32 * TR_PE_STRIDE;          // one stacked tile
    constexpr int XS = TR_MB * RS, PS = TR_MB * TR_PE_STRIDE;    // from a sweep to the other sweep of its tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;              // [2][32][RS]
    float* PE = X + 2 * XT;       // [2][32][TR_PE_STRIDE]
    float* YB = PE + 2 * PT;      // [16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * TR_MB;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;      // (as the forward launch; operand rows past the count are never read)
    // the weight-gradient products read whole groups of 32 points (128 operand rows): the second half of the last group is computed (dead points:
    // zero rows on one side of every product) even when none of its points exists
    if (m0 >= (n + 31) / 32 * 32) {
        if (threadIdx.x < KP) O.w6_part[(int64_t)blockIdx.x * KP + threadIdx.x] = 0.0f;
        return;
    }
    const int a_lane = lane & 31, a_half = 4 * (lane >> 5);
    const int col = 32 * wave + (lane & 31);
    const int64_t npad = O.npad;
#ifdef GENS_K17_STAMPS
    int n_stamp = 0;
#endif
    TR_STAMP();

    // ------------------------------------------------------------------ prologue: threads 0..127 gather the levels, 128..255 encode the point
    {
        const int p = (tid & 127) >> 3, sub = tid & 7;
        const bool enc = tid >= 128;
        const int64_t row = m0 + p;
        const bool live = row < n;
        float x[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f};
        const int64_t src = live ? (index ? index[row] : row) : 0;
        if (live) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                x[a] = pts[3 * src + a];
                sb[a] = s_bar ? s_bar[3 * src + a] : 0.0f;
                gb[a] = g_bar ? g_bar[3 * src + a] : 0.0f;
            }
        }
        if (enc && sub == 6) YB[p] = (live && y_bar) ? y_bar[src] : 0.0f;
        if (enc && sub < 3) {
            const int a = sub;
            float* p0 = PE + p * TR_PE_STRIDE;
            float *p1 = p0 + PS, *p2 = p0 + PT, *p3 = p0 + PT + PS;
            p0[a] = x[a];
            p1[a] = 1.0f;
            p2[a] = gb[a];
            p3[a] = sb[a];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = (float)(1 << k);
                float s, c;
                hw_sincos(x[a] * f, s, c);
                const int is = 3 + 6 * k + a, ic = 6 + 6 * k + a;
                p0[is] = s;
                p0[ic] = c;
                p1[is] = f * c;
                p1[ic] = -f * s;
                p3[is] = f * c * sb[a];
                p3[ic] = -f * s * sb[a];
                p2[is] = -f * f * s * sb[a] + f * c * gb[a];
                p2[ic] = -f * f * c * sb[a] - f * s * gb[a];
            }
            if (a == 0) {
                p0[TR_PE] = 1.0f;
                p1[TR_PE] = p2[TR_PE] = p3[TR_PE] = 0.0f;
#pragma unroll
                for (int k = TR_PE + 1; k < TR_PE_K; ++k) p0[k] = p1[k] = p2[k] = p3[k] = 0.0f;
            }
        }
        if (enc && sub == 7) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float* xr = X + (t >> 1) * XT + (t & 1) * XS + p * RS;
                xr[KIN] = t == 0 ? 1.0f : 0.0f;
#pragma unroll
                for (int k = KIN + 1; k < KP; ++k) xr[k] = 0.0f;
            }
        }
        if (!enc && sub < vols.n) {
            const int l = sub;
            const float dir[3][3] = {{1.0f, 1.0f, 1.0f}, {gb[0], gb[1], gb[2]}, {sb[0], sb[1], sb[2]}};
            float4 o[4], jac[3], mix[3];
            corner_sums<3, false>(vols, l, x, live, dir, o, jac, mix);
            float* x0 = X + p * RS + TR_H;
            float *x1 = x0 + XS, *x2 = x0 + XT, *x3 = x0 + XT + XS;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * l + c;
                const float f = f4_at(o[0], c), fd = f4_at(o[1], c), kf = f4_at(o[2], c), nf = f4_at(o[3], c);
                float s1, c1, s2, c2;
                hw_sincos(f, s1, c1);
                hw_sincos(2.0f * f, s2, c2);
                const float fn = fd * nf;
                x0[ch] = f; x0[CF + ch] = s1; x0[2 * CF + ch] = c1; x0[3 * CF + ch] = s2; x0[4 * CF + ch] = c2;
                x1[ch] = fd; x1[CF + ch] = c1 * fd; x1[2 * CF + ch] = -s1 * fd; x1[3 * CF + ch] = 2.0f * c2 * fd; x1[4 * CF + ch] = -2.0f * s2 * fd;
                x3[ch] = nf; x3[CF + ch] = c1 * nf; x3[2 * CF + ch] = -s1 * nf; x3[3 * CF + ch] = 2.0f * c2 * nf; x3[4 * CF + ch] = -2.0f * s2 * nf;
                x2[ch] = kf;
                x2[CF + ch] = c1 * kf - s1 * fn;
                x2[2 * CF + ch] = -s1 * kf - c1 * fn;
                x2[3 * CF + ch] = 2.0f * c2 * kf - 4.0f * s2 * fn;
                x2[4 * CF + ch] = -2.0f * s2 * kf - 4.0f * c2 * fn;
            }
        }
    }
    __syncthreads();
    // operand rows that are already complete: the conditioning parts and the inputs of layer 0  (sweep t of point `row` = tile t >> 1, row 16 (t & 1) + row)
    for (int i = tid; i < 4 * TR_MB * FEP; i += 256) {
        const int t = i / (TR_MB * FEP), rem = i % (TR_MB * FEP), row = rem / FEP, c = rem % FEP;
        O.re[((m0 + row) * 4 + t) * FEP + c] = X[(t >> 1) * XT + (t & 1) * XS + row * RS + TR_H + c];
    }
    for (int i = tid; i < 4 * TR_MB * 32; i += 256) {
        const int t = i >> 9, row = (i >> 5) & 15, c = i & 31;
        O.r0[((m0 + row) * 4 + t) * 32 + c] = PE[(t >> 1) * PT + (t & 1) * PS + row * TR_PE_STRIDE + c];
    }

    // ------------------------------------------------------------------ forward sweeps: a, a', kappa_a, nu_a
    const int64_t sbase = (int64_t)blockIdx.x * TR_NLAYER;
    for (int l = 0; l < TR_NLAYER; ++l) {
        f32x16 acc[2];
        zero_acc(acc);
        TR_STAMP();
        if (l == 0)
            mfma_tiles<TR_PE_K / 8, 2>(PE + a_lane * TR_PE_STRIDE + a_half, PT, W.wf[0] + (size_t)wave * (TR_PE_K / 8) * 64 + lane, acc);
        else
            mfma_tiles<GIN, 2, TR_PF_B>(X + a_lane * RS + a_half, XT, W.wf[l] + (size_t)wave * GIN * 64 + lane, acc);
        TR_STAMP();
        __syncthreads();
        TR_STAMP();
        float4* st = stash + (((sbase + l) * 4 + wave) * 8) * 64 + lane;
        float* rh = O.rh + (((int64_t)l * npad + m0) * 4) * TR_H + col;       // inputs of layer l + 1
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma clang fp contract(fast)
            const int row = tr_acc_row(i, lane);                              // 0..15: the point; its other sweeps sit 16 rows / one tile further
            const float a = acc[0][i], ad = acc[0][i + 8], ak = acc[1][i], an = acc[1][i + 8];
            float h, d1, d2, d3;
            softplus_d3(a, h, d1, d2, d3);
            (void)d3;
            float hd = d1 * ad, hn = d1 * an, hk = d1 * ak + d2 * ad * an;
            float4 keep = make_float4(a, ad, an, ak);
            if (l == 2) {
                if (col < TR_SKIP_H) {
                    h *= TR_SQ2; hd *= TR_SQ2; hk *= TR_SQ2; hn *= TR_SQ2;
                } else {
                    const float* pe = PE + row * TR_PE_STRIDE + (col - TR_SKIP_H);
                    h = pe[0] * TR_SQ2; hd = pe[PS] * TR_SQ2; hk = pe[PT] * TR_SQ2; hn = pe[PT + PS] * TR_SQ2;
                    keep = make_float4(TR_DEAD, 0.0f, 0.0f, 0.0f);
                }
            }
            float* xr = X + row * RS + col;
            xr[0] = h; xr[XS] = hd; xr[XT] = hk; xr[XT + XS] = hn;
#ifndef GENS_K17_NO_RH      // (timing probe scripts/probe/k17_rh_probe.py: what do these 0.64 GB of stores cost the launch?  WRONG weight gradients without them)
            if (l < 5) {   // (the inputs of the output row stay in LDS: its weight gradient is summed below)
                float* gr = rh + (int64_t)row * 4 * TR_H;
                gr[0] = h; gr[TR_H] = hd; gr[2 * TR_H] = hk; gr[3 * TR_H] = hn;
            }
#endif
            st[i * 64] = keep;
        }
        TR_STAMP();
        __syncthreads();
    }
    TR_STAMP();

    // d loss / d w_last = sum_points y_bar z_6 + kappa_z6 (the constant-1 column gives the bias): this workgroup's 16 points in row order
    if (tid < KP) {
        float s = 0.0f;
        for (int row = 0; row < TR_MB; ++row) s += YB[row] * X[row * RS + tid] + X[XT + row * RS + tid];
        O.w6_part[(int64_t)blockIdx.x * KP + tid] = s;
    }

    // ------------------------------------------------------------------ reverse sweeps: omega, rho, lambda, mu
    constexpr bool SPLIT_K = (NT_B - 4) == 2;
    const int fe_tile = SPLIT_K ? 4 + (wave & 1) : 4 + wave;
    const int fe_g0 = SPLIT_K ? 8 * (wave >> 1) : 0;
    constexpr int FE_G = SPLIT_K ? 8 : 16;
    f32x16 gfe[2];
    zero_acc(gfe);
    const float wl_col = W.w_last[col];
    for (int l = 5; l >= 0; --l) {
        // cotangents of h_l: from the output row (l = 5) or from the reverse products of layer l + 1
        f32x16 gh[2];
        zero_acc(gh);
        // this layer's parked state (a, a', nu_a, kappa_a), asked for BEFORE the products so that it arrives under them
        const float4* st = stash + (((sbase + l) * 4 + wave) * 8) * 64 + lane;
        float4 parked[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) parked[i] = st[i * 64];
        TR_STAMP();
        if (l < 5) {
            mfma_tiles<16, 2, TR_PF_B>(X + a_lane * RS + a_half, XT, W.wb[l + 1] + (size_t)wave * 16 * 64 + lane, gh);
            mfma_tiles<FE_G, 2, TR_PF_B>(X + a_lane * RS + a_half + 8 * fe_g0, XT, W.wb[l + 1] + ((size_t)fe_tile * 16 + fe_g0) * 64 + lane, gfe);
        }
        TR_STAMP();
        __syncthreads();
        TR_STAMP();
        float* lop = O.lop + ((int64_t)m0 * 4 * TR_NLAYER + l) * TR_H + col;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma clang fp contract(fast)
            const int row = tr_acc_row(i, lane);
            float oh, rho_h, lh, mh;
            if (l == 5) {
                oh = YB[row] * wl_col; rho_h = 0.0f; lh = wl_col; mh = 0.0f;
            } else {
                oh = gh[0][i]; rho_h = gh[0][i + 8]; lh = gh[1][i]; mh = gh[1][i + 8];
                if (l == 2) { oh *= TR_SQ2; rho_h *= TR_SQ2; lh *= TR_SQ2; mh *= TR_SQ2; }
            }
            const float4 k = parked[i];                        // a, a', nu_a, kappa_a
            float h, d1, d2, d3;
            softplus_d3(k.x, h, d1, d2, d3);
            (void)h;
            const float lam_a = d1 * lh;
            const float mu_a = d2 * k.y * lh + d1 * mh;
            const float rho_a = d2 * k.z * lh + d1 * rho_h;
            const float om_a = d1 * oh + d3 * k.y * k.z * lh + d2 * (mh * k.z + k.y * rho_h + lh * k.w);
            float* xr = X + row * RS + col;
            xr[0] = om_a; xr[XS] = rho_a; xr[XT] = lam_a; xr[XT + XS] = mu_a;
            float* gr = lop + (int64_t)row * 4 * TR_NLAYER * TR_H;
            constexpr int qs = TR_NLAYER * TR_H;
            gr[0] = om_a; gr[qs] = rho_a; gr[2 * qs] = lam_a; gr[3 * qs] = mu_a;
        }
        TR_STAMP();
        __syncthreads();
    }
    TR_STAMP();
    // conditioning cotangents -> the (dead) h part of the four sweeps
    {
        const int c = 32 * (fe_tile - 4) + (lane & 31);
        if (c < FE && (!SPLIT_K || wave < 2)) {
            const float wl = W.w_last[TR_H + c];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = tr_acc_row(i, lane);
                float* xr = X + row * RS + c;
                xr[0] = gfe[0][i] + YB[row] * wl;
                xr[XS] = gfe[0][i + 8];
                xr[XT] = gfe[1][i] + wl;
                xr[XT + XS] = gfe[1][i + 8];
            }
        }
        if (SPLIT_K) {
            __syncthreads();
            if (c < FE && wave >= 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float* xr = X + tr_acc_row(i, lane) * RS + c;
                    xr[0] += gfe[0][i]; xr[XS] += gfe[0][i + 8]; xr[XT] += gfe[1][i]; xr[XT + XS] += gfe[1][i + 8];
                }
            }
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int p = tid >> 3, sub = tid & 7;
        if (sub < vols.n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * sub + c;
                const float* x0 = X + p * RS;
                const float *x1 = x0 + XS, *x2 = x0 + XT, *x3 = x0 + XT + XS;
                const float s1 = x0[TR_H + CF + ch], c1 = x0[TR_H + 2 * CF + ch], s2 = x0[TR_H + 3 * CF + ch], c2 = x0[TR_H + 4 * CF + ch];
                const float fd = x1[TR_H + ch], kf = x2[TR_H + ch], nf = x3[TR_H + ch];
                // E' = [1, c1, -s1, 2 c2, -2 s2], E'' = [0, -s1, -c1, -4 s2, -4 c2], E''' = [0, -c1, s1, -8 c2, 8 s2]
#define TR_E1(v) ((v)[ch] + c1 * (v)[CF + ch] - s1 * (v)[2 * CF + ch] + 2.0f * (c2 * (v)[3 * CF + ch] - s2 * (v)[4 * CF + ch]))
#define TR_E2(v) (-(s1 * (v)[CF + ch] + c1 * (v)[2 * CF + ch] + 4.0f * (s2 * (v)[3 * CF + ch] + c2 * (v)[4 * CF + ch])))
#define TR_E3(v) (-c1 * (v)[CF + ch] + s1 * (v)[2 * CF + ch] + 8.0f * (s2 * (v)[4 * CF + ch] - c2 * (v)[3 * CF + ch]))
                const float lam_f = TR_E1(x2);
                const float e2l = TR_E2(x2);
                const float mu_f = TR_E1(x3) + fd * e2l;
                const float f_hat = TR_E1(x0) + fd * TR_E2(x1) + nf * TR_E2(x3) + nf * fd * TR_E3(x2) + kf * e2l;
#undef TR_E1
#undef TR_E2
#undef TR_E3
                const int64_t o = (m0 + p) * CF + ch;
                O.f_hat[o] = f_hat;
                O.mu_f[o] = mu_f;
                O.lam_f[o] = lam_f;
            }
        }
    }
}

// ====================================================================================================================
// volume scatter: dV += w f_hat + (grad w . s_bar) mu_f + (grad w . g_bar) lambda_f     (planar (4, X, Y, Z) gradients)
// ====================================================================================================================
// THIRTY-TWO lanes share a (point, level) pair -- lane = (channel, corner), the two z-neighbours of a corner pair adjacent: L2 serves float atomics
// per request, and the lanes of an instruction that hit consecutive floats share one (scripts/probe/atomic_scope_probe.py); the planes of a planar
// gradient only have z-neighbours contiguous.
__global__ __launch_bounds__(256) void sdf_train_scatter_k(LevelSet vs, const float* __restrict__ pts, const float* __restrict__ g_bar,
                                                           const float* __restrict__ s_bar, const float4* __restrict__ f_hat,
                                                           const float4* __restrict__ mu_f, const float4* __restrict__ lam_f,
                                                           const int64_t* __restrict__ index, int64_t n_max, const int32_t* __restrict__ n_dev) {
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    const int L = vs.n;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    if (gid >= n * L) return;
    const int sub = threadIdx.x & 31, ch = sub >> 3, a = (sub >> 2) & 1, b = (sub >> 1) & 1, d = sub & 1;
    const int l = (int)(gid % L);
    const int64_t i = gid / L;                         // compact row: f_hat / mu_f / lam_f
    const int64_t src = index ? index[i] : i;          // dense row: the point and its cotangents
    float* gv = vs.grad[l];
    if (!gv) return;
    const int sz[3] = {vs.dx[l], vs.dy[l], vs.dz[l]};
    const int64_t nvox = (int64_t)sz[0] * sz[1] * sz[2];
    float w0[3], w1[3], k[3], sb[3], gb[3];
    int i0[3];
    bool in0[3], in1[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float x = pts[3 * src + ax];
        const float pos = (x + 1.0f) / 2.0f * (float)(sz[ax] - 1);
        const float f = fminf(fmaxf(floorf(pos), -2.0f), (float)sz[ax] + 1.0f);
        i0[ax] = (int)f;
        w0[ax] = (f + 1.0f) - pos;
        w1[ax] = pos - f;
        in0[ax] = i0[ax] >= 0 && i0[ax] < sz[ax];
        in1[ax] = i0[ax] + 1 >= 0 && i0[ax] + 1 < sz[ax];
        if (!(pos == pos)) in0[ax] = in1[ax] = false;
        k[ax] = (float)(sz[ax] - 1) / 2.0f;
        sb[ax] = s_bar ? s_bar[3 * src + ax] : 0.0f;
        gb[ax] = g_bar ? g_bar[3 * src + ax] : 0.0f;
    }
    const bool ok = (a ? in1[0] : in0[0]) && (b ? in1[1] : in0[1]) && (d ? in1[2] : in0[2]);
    if (!ok) return;
    const float* fp = (const float*)(f_hat + i * L + l);
    const float* mp = (const float*)(mu_f + i * L + l);
    const float* lp = (const float*)(lam_f + i * L + l);
    const float fh = fp[ch], mf = mp[ch], lf = lp[ch];
    const float wx = a ? w1[0] : w0[0], wy = b ? w1[1] : w0[1], wz = d ? w1[2] : w0[2];
    const float dwx = (a ? k[0] : -k[0]) * wy * wz, dwy = wx * (b ? k[1] : -k[1]) * wz, dwz = wx * wy * (d ? k[2] : -k[2]);
    const float w = wx * wy * wz;
    const float ts = dwx * sb[0] + dwy * sb[1] + dwz * sb[2];
    const float tg = dwx * gb[0] + dwy * gb[1] + dwz * gb[2];
    const int64_t lin = ((int64_t)(i0[0] + a) * sz[1] + (i0[1] + b)) * sz[2] + (i0[2] + d);
    atomicAdd(gv + ch * nvox + lin, w * fh + ts * mf + tg * lf);
}

// ====================================================================================================================
// weight packing: effective matrices (row major, weight norm already applied) -> the two B streams of every layer
// ====================================================================================================================
struct SdfPackArgs {
    const float* w[TR_NLAYER];
    const float* b[TR_NLAYER];
    const float* scale[TR_NLAYER];          // per output row g / |v| (weight norm: w = the raw direction matrix v), or NULL: w is effective
    float4* wf[TR_NLAYER];
    float4* wb[TR_NLAYER];
    int rows[TR_NLAYER], cols[TR_NLAYER];   // J_l (128 / 101), K_l (27 / 128 + FE)
    int gf[TR_NLAYER];                      // forward groups per n-tile: ceil((K_l + 1) / 8)
    int ntb[TR_NLAYER];                     // backward n-tiles: ceil(K_l / 32)
};

__global__ __launch_bounds__(256) void sdf_train_pack_k(SdfPackArgs A) {
    const int l = blockIdx.y;
    const int J = A.rows[l], K = A.cols[l], G = A.gf[l];
    const int nf = 4 * G * 64, nb = A.ntb[l] * 16 * 64;
    const float* w = A.w[l];
    const float* sc = A.scale[l];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nf + nb; i += gridDim.x * 256) {
        float v[4];
        if (i < nf) {
            const int lane = i & 63, g = (i >> 6) % G, nt = (i >> 6) / G;
            const int j = 32 * nt + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 8 * g + 4 * (lane >> 5) + q;
                v[q] = (j < J && k < K) ? w[(size_t)j * K + k] * (sc ? sc[j] : 1.0f) : (j < J && k == K) ? A.b[l][j] : 0.0f;
            }
            A.wf[l][i] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            const int ii = i - nf;
            const int lane = ii & 63, g = (ii >> 6) & 15, nt = ii >> 10;
            const int jj = 32 * nt + (lane & 31);              // input column of W
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = 8 * g + 4 * (lane >> 5) + q;    // output row of W
                v[q] = (jj < K && kk < J) ? w[(size_t)kk * K + jj] * (sc ? sc[kk] : 1.0f) : 0.0f;
            }
            A.wb[l][ii] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------
// weight norm (nn.utils.weight_norm, sdf_network.py:90-91: W = g v / |v| per output row) around the packed streams
// --------------------------------------------------------------------------------------------------------------------
struct SdfNormArgs {
    const float* v[TR_NLAYER + 1];          // weight_v of lin0..lin6, (rows_l, cols_l) row major
    const float* g[TR_NLAYER + 1];          // weight_g (rows_l)
    const float* b6;                        // bias of lin6 (its first entry is the SDF's)
    int rows[TR_NLAYER + 1], cols[TR_NLAYER + 1];
    float* scale[TR_NLAYER + 1];            // out: g / |v| per row
    float *w_last, *b_last;                 // out: row 0 of the effective lin6 (cols_6), its bias (1)
};

// one wave per (layer, row): scale = g / |v|; row 0 of lin6 also leaves as the effective w_last
__global__ __launch_bounds__(64) void sdf_train_norm_k(SdfNormArgs A) {
    const int l = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    if (r >= A.rows[l]) return;
    const int K = A.cols[l];
    const float* v = A.v[l] + (size_t)r * K;
    // |v| from a float64 sum, rounded once: within half an ulp of the exact norm whatever the summation order (torch's float32 tree sum
    // is one of many orders; the hierarchical sampling downstream amplifies last-bit differences of the effective matrices)
    double s = 0.0;
    for (int k = lane; k < K; k += 64) s += (double)v[k] * (double)v[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float sc = A.g[l][r] / (float)sqrt(s);
    if (lane == 0) A.scale[l][r] = sc;
    if (l == TR_NLAYER && r == 0) {
        for (int k = lane; k < K; k += 64) A.w_last[k] = v[k] * sc;
        if (lane == 0) A.b_last[0] = A.b6[0];
    }
}

// d loss / d (weight_v, weight_g, bias) of every layer from the effective-matrix gradients the batched products left in `cc`
// (gens_sdf_train_bwd's header: per layer l = 1..5 the tiles [128 x 128 | 128 x FEP], then layer 0's [128 x 32]) and the column sums of
// w6_part.  One wave per (layer, row):  s = dW . v,  dg = s / |v|,  dv = (g / |v|) (dW - v s / |v|^2),  db = the bias column.
struct SdfWgradArgs {
    const float* v[TR_NLAYER + 1];
    const float* g[TR_NLAYER + 1];
    int rows[TR_NLAYER + 1], cols[TR_NLAYER + 1];
    const float* cc;                        // K14's concatenated tiles
    const float* w6_sum;                    // (KP) column sums of w6_part
    int fe, fep;
    float* dv[TR_NLAYER + 1];
    float* dg[TR_NLAYER + 1];
    float* db[TR_NLAYER + 1];
};

__global__ __launch_bounds__(64) void sdf_train_wgrad_k(SdfWgradArgs A) {
    const int l = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    if (r >= A.rows[l]) return;
    const int K = A.cols[l];
    const float* v = A.v[l] + (size_t)r * K;
    const int64_t per_layer = (int64_t)TR_H * TR_H + (int64_t)TR_H * A.fep;
    float dw[4], vv[4];                      // K <= 4 * 64
    float s = 0.0f, n2 = 0.0f, bias = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        dw[q] = 0.0f;
        vv[q] = 0.0f;
        if (k < K) {
            vv[q] = v[k];
            if (l == 0) dw[q] = A.cc[5 * per_layer + (int64_t)r * 32 + k];
            else if (l < TR_NLAYER) {
                const float* h = A.cc + (int64_t)(l - 1) * per_layer;
                dw[q] = k < TR_H ? h[(int64_t)r * TR_H + k] : h[(int64_t)TR_H * TR_H + (int64_t)r * A.fep + (k - TR_H)];
            } else dw[q] = r == 0 ? A.w6_sum[k] : 0.0f;
        }
        s += dw[q] * vv[q];
        n2 += vv[q] * vv[q];
    }
    s = wave_sum(s);
    n2 = wave_sum(n2);
    if (l == 0) bias = A.cc[5 * per_layer + (int64_t)r * 32 + TR_PE];
    else if (l < TR_NLAYER) bias = A.cc[(int64_t)(l - 1) * per_layer + (int64_t)TR_H * TR_H + (int64_t)r * A.fep + A.fe];
    else bias = r == 0 ? A.w6_sum[K] : 0.0f;
    const float nrm = sqrtf(n2), sc = A.g[l][r] / nrm;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        if (k < K) A.dv[l][(size_t)r * K + k] = sc * (dw[q] - vv[q] * (s / n2));
    }
    if (lane == 0) {
        A.dg[l][r] = s / nrm;
        A.db[l][r] = bias;
    }
}

// ====================================================================================================================
// C ABI
// ====================================================================================================================
int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

template <int FE>
static constexpr size_t fwd_lds_bytes() {
    constexpr int CF = FE / 5, KP = (TR_H + FE + 8) / 8 * 8, RS = KP + 4;
    return sizeof(float) * (32 * RS + 2 * 32 * TR_PE_STRIDE + 2 * TR_MB * CF * 3 + 2 * TR_MB * CF);
}
template <int FE>
static constexpr size_t bwd_lds_bytes() {
    constexpr int KP = (TR_H + FE + 8) / 8 * 8, RS = KP + 4;
    return sizeof(float) * (2 * 32 * RS + 2 * 32 * TR_PE_STRIDE + TR_MB);
}

static int fill_train_weights(const char* who, SdfTrainWeights* W, const float* const* wf, const float* const* wb, const float* w_last,
                              const float* b_last) {
    GENS_CHECK_ARG(wf && wb && w_last, GENS_EINVAL, "%s: null weight table", who);
    for (int l = 0; l < TR_NLAYER; ++l) {
        GENS_CHECK_ARG(wf[l] && wb[l], GENS_EINVAL, "%s: layer %d weights are null", who, l);
        W->wf[l] = (const float4*)wf[l];
        W->wb[l] = (const float4*)wb[l];
    }
    W->w_last = w_last;
    W->b_last = b_last;
    return 0;
}

extern "C" int64_t gens_sdf_train_stash_bytes(int64_t n, int backward) {       // (the same bytes either way: 32 points x 6 layers x 128 columns)
    return gens_blocks(n, TR_M) * (int64_t)TR_NLAYER * 4 * 16 * 64 * (backward ? 16 : 8);
}

extern "C" int gens_sdf_train_pack(const float* const* w, const float* const* b, int n_levels, float* const* wf, float* const* wb,
                                   void* stream) {
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_train_pack: built for 1 to 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(w && b && wf && wb, GENS_EINVAL, "gens_sdf_train_pack: null table");
    SdfPackArgs A;
    const int kin = TR_H + 20 * n_levels;
    int most = 0;
    for (int l = 0; l < TR_NLAYER; ++l) {
        GENS_CHECK_ARG(w[l] && b[l] && wf[l] && wb[l], GENS_EINVAL, "gens_sdf_train_pack: layer %d is null", l);
        A.w[l] = w[l];
        A.b[l] = b[l];
        A.scale[l] = nullptr;
        A.wf[l] = (float4*)wf[l];
        A.wb[l] = (float4*)wb[l];
        A.rows[l] = l == 2 ? TR_SKIP_H : TR_H;
        A.cols[l] = l == 0 ? TR_PE : kin;
        A.gf[l] = (A.cols[l] + 1 + 7) / 8;
        A.ntb[l] = l == 0 ? 1 : 4 + ((20 * n_levels + 31) / 32 <= 2 ? 2 : 4);      // (the kernels' NT_B: conditioning tiles padded to two or four)
        most = max(most, 4 * A.gf[l] * 64 + A.ntb[l] * 16 * 64);
    }
    sdf_train_pack_k<<<dim3(gens_blocks(most, 256), TR_NLAYER), 256, 0, (hipStream_t)stream>>>(A);
    return gens_launch_status("gens_sdf_train_pack");
}

extern "C" int gens_sdf_train_fwd(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                                  const float* const* wb, const float* w_last, const float* b_last, const float* pts, const int64_t* index,
                                  int64_t n, const int32_t* n_device, void* stash, float* y_out, float* g_out, float* s_out, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_train_fwd", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_train_fwd: built for 1 to 5 volume levels, got %d", n_levels);
    SdfTrainWeights W;
    if (int e = fill_train_weights("gens_sdf_train_fwd", &W, wf, wb, w_last, b_last)) return e;
    GENS_CHECK_ARG(b_last, GENS_EINVAL, "gens_sdf_train_fwd: null b_last");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && stash && y_out && g_out && s_out)), GENS_EINVAL, "gens_sdf_train_fwd: null pts / stash / output");
    if (n == 0) return 0;
    const unsigned grid = gens_blocks(n, TR_MB);
    hipStream_t s = (hipStream_t)stream;
#define TR_FWD(FE_)                                                                                                                         \
    {                                                                                                                                      \
        static GensLdsOptIn lds;                                                                                                           \
        if (int e = gens_lds_opt_in(lds, (const void*)sdf_train_fwd_k<FE_>, (int)fwd_lds_bytes<FE_>(), "gens_sdf_train_fwd")) return e;    \
        sdf_train_fwd_k<FE_><<<grid, 256, fwd_lds_bytes<FE_>(), s>>>(W, vs, pts, index, n, n_device, (float2*)stash, y_out, g_out, s_out); \
    }
    switch (n_levels) {
        case 1: TR_FWD(20) break;
        case 2: TR_FWD(40) break;
        case 3: TR_FWD(60) break;
        case 4: TR_FWD(80) break;
        default: TR_FWD(100) break;
    }
#undef TR_FWD
    return gens_launch_status("gens_sdf_train_fwd");
}

extern "C" int gens_sdf_train_bwd(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                                  const float* const* wb, const float* w_last, const float* pts, const int64_t* index, int64_t n,
                                  const int32_t* n_device, const float* y_bar, const float* g_bar, const float* s_bar, void* stash, float* lop,
                                  float* rh, float* re, float* r0, float* f_hat, float* mu_f, float* lam_f, float* w6_part, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_train_bwd", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_train_bwd: built for 1 to 5 volume levels, got %d", n_levels);
    SdfTrainWeights W;
    if (int e = fill_train_weights("gens_sdf_train_bwd", &W, wf, wb, w_last, nullptr)) return e;
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && stash && lop && rh && re && r0 && f_hat && mu_f && lam_f && w6_part)), GENS_EINVAL,
                   "gens_sdf_train_bwd: null pts / stash / output");
    if (n == 0) return 0;
    const unsigned grid = gens_blocks(n, TR_M) * (TR_M / TR_MB);        // workgroups of 16 points, in whole groups of 32 (the operand rows' granularity)
    SdfTrainBwdOut O = {lop, rh, re, r0, f_hat, mu_f, lam_f, w6_part, (int64_t)grid * TR_MB};
    hipStream_t s = (hipStream_t)stream;
#define TR_BWD(FE_)                                                                                                                                  \
    {                                                                                                                                               \
        static GensLdsOptIn lds;                                                                                                                    \
        if (int e = gens_lds_opt_in(lds, (const void*)sdf_train_bwd_k<FE_>, (int)bwd_lds_bytes<FE_>(), "gens_sdf_train_bwd")) return e;             \
        sdf_train_bwd_k<FE_><<<grid, 256, bwd_lds_bytes<FE_>(), s>>>(W, vs, pts, index, n, n_device, y_bar, g_bar, s_bar, (float4*)stash, O);       \
    }
    switch (n_levels) {
        case 1: TR_BWD(20) break;
        case 2: TR_BWD(40) break;
        case 3: TR_BWD(60) break;
        case 4: TR_BWD(80) break;
        default: TR_BWD(100) break;
    }
#undef TR_BWD
    return gens_launch_status("gens_sdf_train_bwd");
}

extern "C" int gens_sdf_train_scatter(const int* dims, int n_levels, const float* pts, const float* g_bar, const float* s_bar,
                                      const float* f_hat, const float* mu_f, const float* lam_f, const int64_t* index, int64_t n,
                                      const int32_t* n_device, float* const* g_vols, void* stream) {
    GENS_CHECK_ARG(dims && g_vols, GENS_EINVAL, "gens_sdf_train_scatter: null table");
    GENS_CHECK_ARG(n_levels > 0 && n_levels <= GENS_MAX_LEVELS, GENS_ELIMIT, "gens_sdf_train_scatter: n_levels=%d not in 1..%d", n_levels,
                   GENS_MAX_LEVELS);
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && f_hat && mu_f && lam_f)), GENS_EINVAL, "gens_sdf_train_scatter: null pointer");
    if (n == 0) return 0;
    LevelSet vs;
    vs.n = n_levels;
    vs.bits = 0;
    for (int l = 0; l < GENS_MAX_LEVELS; ++l) {
        vs.data[l] = nullptr;
        vs.aux[l] = nullptr;
        vs.grad[l] = l < n_levels ? g_vols[l] : nullptr;
        vs.dx[l] = l < n_levels ? dims[3 * l] : 1;
        vs.dy[l] = l < n_levels ? dims[3 * l + 1] : 1;
        vs.dz[l] = l < n_levels ? dims[3 * l + 2] : 1;
    }
    sdf_train_scatter_k<<<gens_blocks(n * n_levels * 32, 256), 256, 0, (hipStream_t)stream>>>(vs, pts, g_bar, s_bar, (const float4*)f_hat,
                                                                                         (const float4*)mu_f, (const float4*)lam_f, index, n, n_device);
    return gens_launch_status("gens_sdf_train_scatter");
}

static void sdf_layer_shapes(int n_levels, int* rows, int* cols) {
    const int kin = TR_H + 20 * n_levels;
    for (int l = 0; l <= TR_NLAYER; ++l) {
        rows[l] = l == 2 ? TR_SKIP_H : (l == TR_NLAYER ? TR_H + 1 : TR_H);
        cols[l] = l == 0 ? TR_PE : kin;
    }
}

// gens_sdf_train_pack from the RAW weight-normed parameters (weight_v, weight_g, bias of lin0..lin6) in two launches: the row scales
// g / |v| (+ the effective output row w_last (K floats) and its bias b_last (1)), then the streams.  scale: 7 caller-allocated DEVICE
// arrays of rows_l floats, kept for gens_sdf_train_wgrad.
extern "C" int gens_sdf_train_pack_wn(const float* const* v, const float* const* g, const float* const* b, int n_levels, float* const* scale,
                                      float* const* wf, float* const* wb, float* w_last, float* b_last, void* stream) {
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_train_pack_wn: built for 1 to 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(v && g && b && scale && wf && wb && w_last && b_last, GENS_EINVAL, "gens_sdf_train_pack_wn: null table");
    SdfNormArgs N;
    sdf_layer_shapes(n_levels, N.rows, N.cols);
    for (int l = 0; l <= TR_NLAYER; ++l) {
        GENS_CHECK_ARG(v[l] && g[l] && b[l] && scale[l], GENS_EINVAL, "gens_sdf_train_pack_wn: layer %d is null", l);
        N.v[l] = v[l];
        N.g[l] = g[l];
        N.scale[l] = scale[l];
    }
    N.b6 = b[TR_NLAYER];
    N.w_last = w_last;
    N.b_last = b_last;
    hipStream_t s = (hipStream_t)stream;
    sdf_train_norm_k<<<dim3(TR_H + 1, TR_NLAYER + 1), 64, 0, s>>>(N);
    SdfPackArgs A;
    int most = 0;
    for (int l = 0; l < TR_NLAYER; ++l) {
        GENS_CHECK_ARG(wf[l] && wb[l], GENS_EINVAL, "gens_sdf_train_pack_wn: stream %d is null", l);
        A.w[l] = v[l];
        A.b[l] = b[l];
        A.scale[l] = scale[l];
        A.wf[l] = (float4*)wf[l];
        A.wb[l] = (float4*)wb[l];
        A.rows[l] = N.rows[l];
        A.cols[l] = N.cols[l];
        A.gf[l] = (A.cols[l] + 1 + 7) / 8;
        A.ntb[l] = l == 0 ? 1 : 4 + ((20 * n_levels + 31) / 32 <= 2 ? 2 : 4);      // (the kernels' NT_B: conditioning tiles padded to two or four)
        most = max(most, 4 * A.gf[l] * 64 + A.ntb[l] * 16 * 64);
    }
    sdf_train_pack_k<<<dim3(gens_blocks(most, 256), TR_NLAYER), 256, 0, s>>>(A);
    return gens_launch_status("gens_sdf_train_pack_wn");
}

// The parameter gradients of a step from the products of gens_sdf_train_bwd's operand rows: cc = gens_gemm_tn_batch's output for the
// eleven products in the order (l = 1..5: [lop_l^T rh_l (128 x 128) | lop_l^T re (128 x FEP)], then lop_0^T r0 (128 x 32)), w6_sum (KP)
// = the column sums of w6_part.  dv[l] (rows_l, cols_l), dg[l] (rows_l), db[l] (rows_l) for lin0..lin6: weight norm's backward included.
extern "C" int gens_sdf_train_wgrad(const float* const* v, const float* const* g, int n_levels, const float* cc, const float* w6_sum,
                                    float* const* dv, float* const* dg, float* const* db, void* stream) {
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_train_wgrad: built for 1 to 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(v && g && cc && w6_sum && dv && dg && db, GENS_EINVAL, "gens_sdf_train_wgrad: null pointer");
    SdfWgradArgs A;
    sdf_layer_shapes(n_levels, A.rows, A.cols);
    for (int l = 0; l <= TR_NLAYER; ++l) {
        GENS_CHECK_ARG(v[l] && g[l] && dv[l] && dg[l] && db[l], GENS_EINVAL, "gens_sdf_train_wgrad: layer %d is null", l);
        A.v[l] = v[l]; A.g[l] = g[l]; A.dv[l] = dv[l]; A.dg[l] = dg[l]; A.db[l] = db[l];
    }
    A.cc = cc;
    A.w6_sum = w6_sum;
    A.fe = 20 * n_levels;
    A.fep = (TR_H + A.fe + 1 + 7) / 8 * 8 - TR_H;                    // KP - 128, KP = 8 ceil((K + 1) / 8)
    sdf_train_wgrad_k<<<dim3(TR_H + 1, TR_NLAYER + 1), 64, 0, (hipStream_t)stream>>>(A);
    return gens_launch_status("gens_sdf_train_wgrad");
}

#ifdef GENS_K17_STAMPS
extern "C" int gens_debug_k17_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k17_stamps), sizeof(unsigned long long) * 4 * 128) == hipSuccess ? 0 : -1;
}
#endif
