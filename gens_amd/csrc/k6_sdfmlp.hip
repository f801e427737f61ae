// K6: the SDF network evaluated in ONE kernel -- multi-level volume look-up (K2) as
// prologue, positional encodings, the 7 weight-normed layers on the fp32 matrix cores, and (optionally) the exact
// first derivative d sdf / d x by reverse mode, without any activation ever leaving the CU.
//
// Replaces, for inference, SDFNetwork.forward / .sdf / .gradient(first order)
// (/root/reference/models/modules/sdf_network.py:98-146) as driven by implicit_surface.py:125,179-191,375.
// Architecture is the shipped one (confs/gens.conf:69-86): d_hidden 128, 6 hidden layers, skip at layer 3,
// multires 4 (27-wide point encoding), feat_multires 2 (5x volume channels), Softplus(beta=100), scale folded by the host.
//
// MI355X mapping.  v_mfma_f32_32x32x2_f32 is exact float32 (an fmaf chain) at the full fp32 rate, so parity with the
// rocBLAS path is float32 round-off.  A workgroup = 4 wavefronts = 32 points; wave w owns output columns
// [32w, 32w+32) of every layer, so its pre-activations -- and therefore softplus' = sigmoid(100 pre), all that the
// reverse pass needs -- stay in ITS registers (16 VGPRs per layer): no activation stash in LDS or HBM.  LDS holds one
// 32 x (128+FE) row-padded tile (A operand, odd stride => conflict-free ds_read_b32 for the MFMA A layout); weights
// are pre-packed by the host in MFMA B-fragment order, so every B operand is one coalesced 256-B load from L2
// (the whole network is < 1 MB).  HBM traffic per point: 12 B in, 4 or 16 B out (+ 8 texel gathers per level).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_M 32            // points per workgroup
#define MLP_H 128           // hidden width
#define MLP_PE 27           // 3 * (1 + 2*4)
#define MLP_PE_K 32         // layer-0 reduction length, padded to whole 8-wide groups (zero columns / zero weights)
#define MLP_PE_STRIDE 36    // 16-B aligned rows, stride = 4 (mod 8) words => conflict-free ds_read_b128 of the A operand
#define MLP_SKIP_H 101      // hidden columns produced by layer 2 (128 - 27)
#define MLP_NLAYER 6        // GEMM layers 0..5; layer 6 is a dot product (only the sdf row is needed)

// Operand streams.  The reduction index of every GEMM is consumed in groups of 8: within group j the four MFMAs i = 0..3
// take k = 8j + 4h + i from lane half h (any fixed permutation of k is a valid order of the fp32 sum), so a lane's
// A values for a group are 4 consecutive floats of its LDS row (ONE ds_read_b128) and its B values are 4 consecutive
// floats of the host-packed weight stream (ONE global_load_dwordx4, 1 KB per wave, fully coalesced): 2 memory
// instructions per 4 MFMAs instead of 8.  B groups are software-prefetched 4 groups (1024 MFMA cycles) ahead, across
// layer boundaries and barriers (weights do not depend on activations).
struct SdfMlpWeights {
    const float4* wf[MLP_NLAYER];  // forward B groups  [n_tile(4)][group][64]   (float4 per lane)
    const float4* wb[MLP_NLAYER];  // backward B groups [n_tile][16][64]; wb[0]: 1 tile (27 -> 32 columns)
    const float* w_last;           // (128 + FE) row 0 of layer 6
    const float* w_fwd;            // the row the forward dot product reads: w_last, or its pre-scaled form (see softplus_t)
    float b_last;
    const float* b_last_dev;       // when non-null: the bias is read from the device (training: weights change every step, no host read-back)
    float inv_scale;               // 1 / scale  (sdf_network.py:123)
    float scale;
};

// Softplus(beta = 100) and its derivative, branch-free: 5 VALU + 2 transcendental issues (+3 with the derivative).
// fp32 MFMA shares the SIMD's FMA datapath with the vector ALU (the two peaks are the same 157.3 TFLOP/s and the PMC
// busy cycles add up), so every VALU cycle of the epilogue is a matrix cycle lost.  Overflow needs no clamp: above the
// torch threshold (100 x > 20) both results are replaced by the linear branch through v_cndmask, which discards inf/NaN.
template <bool DERIV>
__device__ __forceinline__ float softplus100(float x, float& dsig) {
    const float e = __builtin_amdgcn_exp2f(x * 144.269504088896340736f);     // e^{100 x}
    const float u = 1.0f + e;
    const bool lin = x > 0.2f;
    if constexpr (DERIV) dsig = lin ? 1.0f : e * __builtin_amdgcn_rcpf(u);
    return lin ? x : __builtin_amdgcn_logf(u) * 0.0069314718055994530942f;   // log2(u) * ln2 / 100
}

// The same activation on a PRE-SCALED pre-activation t = (100 / ln 2) a, returning the scaled hidden value h~ = (100 / ln 2) softplus(a)
// = log2(1 + 2^t) (or t itself above the threshold): the two multiplications of softplus100 disappear from the epilogue (5 issue slots
// instead of 7, 8 instead of 10 with the derivative -- and every VALU cycle of the epilogue is a matrix cycle lost).  The host folds the
// factor into the weight streams instead (gens_amd.ops.SdfMlpPlan): a GEMM whose hidden inputs are h~ yields t directly when the columns
// fed by h~ keep the plain weights and the columns fed by unscaled inputs (point encoding, volume features, the bias column) carry
// 100 / ln 2 times theirs; the output row divides its hidden part by the same factor.  The derivative sigmoid(100 a) = 2^t / (1 + 2^t)
// is unchanged, so the reverse pass runs on the plain transposed streams.
template <bool DERIV>
__device__ __forceinline__ float softplus_t(float t, float& dsig) {
    const float e = __builtin_amdgcn_exp2f(t);
    const float u = 1.0f + e;
    const bool lin = t > 28.853900817779268f;                      // 0.2 * 100 / ln 2: torch's threshold (100 a > 20)
    if constexpr (DERIV) dsig = lin ? 1.0f : e * __builtin_amdgcn_rcpf(u);
    return lin ? t : __builtin_amdgcn_logf(u);
}

// acc += A(32 x 8G, LDS rows of stride rs; `a` already points at this lane's row + 4 * half) * B(packed groups; `b`
// already points at this lane's float4 of group 0).  pre[] holds groups 0..3 of THIS product (loads in flight); on
// return it holds groups 0..3 of the NEXT product `bn` (NEXT = false: nothing more to fetch).
template <int G, int MLP_PF, bool NEXT = true>
__device__ __forceinline__ f32x16 mfma_groups(const float* __restrict__ a, const float4* __restrict__ b, f32x16 acc, float4 (&pre)[MLP_PF],
                                              const float4* __restrict__ bn) {
    float4 av = *(const float4*)a;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const float4 bv = pre[j % MLP_PF];
        const float4 ac = av;
        if (j + 1 < G) av = *(const float4*)(a + 8 * (j + 1));
        if (j + MLP_PF < G) pre[j % MLP_PF] = b[64 * (j + MLP_PF)];
        else if (NEXT) pre[j % MLP_PF] = bn[64 * (j + MLP_PF - G)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac.w, bv.w, acc, 0, 0, 0);
#ifndef K6_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);   // keep the 4-group prefetch distance: hipcc otherwise sinks the loads next to their use
#endif
    }
    if constexpr (NEXT && G % MLP_PF != 0) {   // (L = 5: 29 groups) bring the next product's group 0 back to slot 0
        float4 t[MLP_PF];
#pragma unroll
        for (int i = 0; i < MLP_PF; ++i) t[i] = pre[(i + G) % MLP_PF];
#pragma unroll
        for (int i = 0; i < MLP_PF; ++i) pre[i] = t[i];
    }
    return acc;
}

template <int MLP_PF>
__device__ __forceinline__ void prefetch_groups(float4 (&pre)[MLP_PF], const float4* __restrict__ b) {
#pragma unroll
    for (int j = 0; j < MLP_PF; ++j) pre[j] = b[64 * j];
}

// row of accumulator register r for this lane (C/D layout of 32x32 MFMA)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int FE, bool GRAD, bool PRE>
__global__ __launch_bounds__(256, GRAD ? 2 : 4) void sdf_mlp_k(SdfMlpWeights W, LevelSet vols, const float* __restrict__ pts,
                                                     const int64_t* __restrict__ index, int64_t n_max, const int32_t* __restrict__ n_dev,
                                                     float* __restrict__ sdf_out, float* __restrict__ grad_out) {
    constexpr int CF = FE / 5;            // raw volume channels (4 per level)
    constexpr int MLP_PF = 4;             // B prefetch depth in groups (6 and 8 measured no faster)
    constexpr int KIN = MLP_H + FE;       // input width of layers 1..6
    constexpr int KP = (KIN + 8) / 8 * 8; // ... + the bias column, padded to whole groups (the other pad columns are zero)
    constexpr int GIN = KP / 8;           // groups per forward product of layers 1..5
    constexpr int RS = KP + 4;            // row stride: 16-B aligned, = 4 (mod 8) words
    constexpr int NT_B = 4 + ((FE + 31) / 32 <= 2 ? 2 : 4);   // backward n-tiles: 4 of the h part + the conditioning tiles, padded to 2 or 4 (zero columns)
    __shared__ __attribute__((aligned(16))) float X[MLP_M * RS];              // [h | fe | 0] tile; reused as the G buffer in the reverse pass
    __shared__ __attribute__((aligned(16))) float PE[MLP_M * MLP_PE_STRIDE];  // point encoding (27, cols 27..31 = 0)
    __shared__ float GPE[GRAD ? MLP_M * MLP_PE_STRIDE : 1];   // d/d(point encoding) from the skip connection
    __shared__ float JAC[GRAD ? MLP_M * CF * 3 : 1];     // d feat_c / d x_a
    __shared__ float XYZ[MLP_M * 3];
    __shared__ float RED[MLP_M * 8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * MLP_M;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;   // device-side point count (no host sync after compaction)
    if (m0 >= n) return;
    const int a_lane = (lane & 31), a_half = 4 * (lane >> 5);
    float4 pre[MLP_PF];
    prefetch_groups(pre, W.wf[0] + (size_t)wave * (MLP_PE_K / 8) * 64 + lane);   // layer 0 weights fly during the prologue

    // ------------------------------------------------------------------ prologue: look-up, encodings
    {
        const int p = tid >> 3, sub = tid & 7;
        const int64_t row = m0 + p;
        const bool live = row < n;
        const int64_t src = live ? (index ? index[row] : row) : 0;
        float x[3] = {0.f, 0.f, 0.f};
        if (live) { x[0] = pts[3 * src]; x[1] = pts[3 * src + 1]; x[2] = pts[3 * src + 2]; }
        if (sub < 3) {
            const int a = sub;
            float v = x[a] * W.scale;
            float* pe = PE + p * MLP_PE_STRIDE;
            XYZ[p * 3 + a] = v;
            pe[a] = v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float f = (float)(1 << k);
                hw_sincos(v * f, pe[3 + 6 * k + a], pe[6 + 6 * k + a]);
            }
            if (a == 0) {
                pe[MLP_PE] = 1.0f;                                   // bias column of layer 0
#pragma unroll
                for (int k = MLP_PE + 1; k < MLP_PE_K; ++k) pe[k] = 0.0f;
            }
        }
        if (sub == 7) {
            static_assert(KP > KIN, "a pad column carries the bias");
            X[p * RS + KIN] = 1.0f;                                  // bias column of layers 1..5
#pragma unroll
            for (int k = KIN + 1; k < KP; ++k) X[p * RS + k] = 0.0f;
        }
        if (sub < vols.n) {   // one thread per (point, level): 8 texel gathers
            const int l = sub;
            const int Xd = vols.dx[l], Yd = vols.dy[l], Zd = vols.dz[l];
            const float4* v = (const float4*)vols.data[l];
            float pos[3], w0[3], w1[3];
            int i0[3];
            bool in0[3], in1[3];
            const int sz[3] = {Xd, Yd, Zd};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                pos[a] = (x[a] + 1.0f) / 2.0f * (float)(sz[a] - 1);
                float f = fminf(fmaxf(floorf(pos[a]), -2.0f), (float)sz[a] + 1.0f);
                i0[a] = (int)f;
                w0[a] = (f + 1.0f) - pos[a];
                w1[a] = pos[a] - f;
                in0[a] = i0[a] >= 0 && i0[a] < sz[a];
                in1[a] = i0[a] + 1 >= 0 && i0[a] + 1 < sz[a];
            }
            float4 acc = f4_zero(), jx = f4_zero(), jy = f4_zero(), jz = f4_zero();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int a = c >> 2, b = (c >> 1) & 1, d = c & 1;
                // branch-free zero padding: the texel is read from clamped indices (8 loads in flight, one wait) and dropped
                // by a select when the corner lies outside the volume
                const bool ok = live && (a ? in1[0] : in0[0]) && (b ? in1[1] : in0[1]) && (d ? in1[2] : in0[2]);
                const int cx = min(max(i0[0] + a, 0), Xd - 1), cy = min(max(i0[1] + b, 0), Yd - 1), cz = min(max(i0[2] + d, 0), Zd - 1);
                float4 t = v[((int64_t)cx * Yd + cy) * Zd + cz];
                if (!ok) t = f4_zero();
                float wx = a ? w1[0] : w0[0], wy = b ? w1[1] : w0[1], wz = d ? w1[2] : w0[2];
                acc = f4_madd(acc, t, wx * wy * wz);
                if constexpr (GRAD) {
                    jx = f4_madd(jx, t, (a ? 1.0f : -1.0f) * wy * wz);
                    jy = f4_madd(jy, t, wx * (b ? 1.0f : -1.0f) * wz);
                    jz = f4_madd(jz, t, wx * wy * (d ? 1.0f : -1.0f));
                }
            }
            const float fv[4] = {acc.x, acc.y, acc.z, acc.w};
            float* xr = X + p * RS + MLP_H;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * l + c;
                xr[ch] = fv[c];
                hw_sincos(fv[c], xr[CF + ch], xr[2 * CF + ch]);
                hw_sincos(2.0f * fv[c], xr[3 * CF + ch], xr[4 * CF + ch]);
            }
            if constexpr (GRAD) {
                const float sx = (float)(Xd - 1) / 2.0f, sy = (float)(Yd - 1) / 2.0f, sz_ = (float)(Zd - 1) / 2.0f;
                const float gx[4] = {jx.x, jx.y, jx.z, jx.w}, gy[4] = {jy.x, jy.y, jy.z, jy.w}, gz[4] = {jz.x, jz.y, jz.z, jz.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float* j = JAC + (p * CF + 4 * l + c) * 3;
                    j[0] = gx[c] * sx;
                    j[1] = gy[c] * sy;
                    j[2] = gz[c] * sz_;
                }
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ forward: layers 0..5
    f32x16 dsig[GRAD ? MLP_NLAYER : 1];
    const int col = 32 * wave + (lane & 31);
#pragma unroll
    for (int l = 0; l < MLP_NLAYER; ++l) {
        f32x16 acc;   // the bias rides in the GEMM: input column K (first pad column) is 1, weight row K holds the bias
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // what to prefetch while this product runs: the next layer's first groups, or the reverse pass' first tile
        const float4* nxt = (l + 1 < MLP_NLAYER) ? W.wf[l + 1] + (size_t)wave * GIN * 64 + lane
                                                 : (GRAD ? W.wb[5] + (size_t)wave * 16 * 64 + lane : nullptr);
        if (l == 0)
            acc = mfma_groups<MLP_PE_K / 8, MLP_PF>(PE + a_lane * MLP_PE_STRIDE + a_half, W.wf[0] + (size_t)wave * (MLP_PE_K / 8) * 64 + lane, acc, pre, nxt);
        else if (l + 1 < MLP_NLAYER || GRAD)
            acc = mfma_groups<GIN, MLP_PF>(X + a_lane * RS + a_half, W.wf[l] + (size_t)wave * GIN * 64 + lane, acc, pre, nxt);
        else
            acc = mfma_groups<GIN, MLP_PF, false>(X + a_lane * RS + a_half, W.wf[l] + (size_t)wave * GIN * 64 + lane, acc, pre, nxt);
        __syncthreads();   // every wave has finished reading this layer's input tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            float ds = 0.0f;
            float h = PRE ? softplus_t<GRAD>(acc[r], ds) : softplus100<GRAD>(acc[r], ds);
            if (l == 2) {   // skip connection feeding layer 3: x = cat([h, pe]) / sqrt(2)   (sdf_network.py:111-112)
                if (col < MLP_SKIP_H) {
                    h *= 0.70710678118654752440f;
                } else {
                    h = PE[row * MLP_PE_STRIDE + (col - MLP_SKIP_H)] * 0.70710678118654752440f;
                    ds = 0.0f;
                }
            }
            X[row * RS + col] = h;
            if constexpr (GRAD) dsig[l][r] = ds;
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ layer 6, sdf row only: a dot product per point
    {
        const int p = tid >> 3, sub = tid & 7;
        const float* xr = X + p * RS;
        float s = 0.0f;
        for (int k = sub; k < KIN; k += 8) s += xr[k] * W.w_fwd[k];
        RED[p * 8 + sub] = s;
    }
    __syncthreads();
    if (tid < MLP_M) {
        const int64_t row = m0 + tid;
        if (row < n) {
            float s = W.b_last_dev ? W.b_last_dev[0] : W.b_last;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += RED[tid * 8 + k];
            sdf_out[index ? index[row] : row] = s * W.inv_scale;
        }
    }
    if constexpr (!GRAD) return;

    // ------------------------------------------------------------------ reverse pass
    // G_5[m][j] = w_last[j] * softplus'(pre_5[m][j]) ; conditioning part of d/d input accumulates in registers
    f32x16 gfe;     // this wave's 32-column tile of d sdf / d fe  (tiles 4.. of the backward GEMM output)
#pragma unroll
    for (int r = 0; r < 16; ++r) gfe[r] = 0.0f;
    // FE = 100: four conditioning tiles, one per wave.  FE = 60: two tiles; waves w and w+2 share tile 4 + (w & 1) and
    // each reduces half of the 128-long K range (partials are summed in the epilogue), so all four waves stay busy.
    constexpr bool SPLIT_K = (NT_B - 4) == 2;
    const int fe_tile = SPLIT_K ? 4 + (wave & 1) : 4 + wave;
    const int fe_g0 = SPLIT_K ? 8 * (wave >> 1) : 0;          // first group of this wave's share of the 16-group reduction
    constexpr int FE_G = SPLIT_K ? 8 : 16;
    __syncthreads();
    {
        const float wl = W.w_last[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) X[acc_row(r, lane) * RS + col] = wl * dsig[5][r];
    }
    __syncthreads();
#pragma unroll
    for (int l = 5; l >= 1; --l) {
        f32x16 gh;
#pragma unroll
        for (int r = 0; r < 16; ++r) gh[r] = 0.0f;
        const float4* b_fe = W.wb[l] + ((size_t)fe_tile * 16 + fe_g0) * 64 + lane;
        const float4* nxt = (l > 1) ? W.wb[l - 1] + (size_t)wave * 16 * 64 + lane : W.wb[0] + (size_t)(4 * wave) * 64 + lane;
        gh = mfma_groups<16, MLP_PF>(X + a_lane * RS + a_half, W.wb[l] + (size_t)wave * 16 * 64 + lane, gh, pre, b_fe);
        gfe = mfma_groups<FE_G, MLP_PF>(X + a_lane * RS + a_half + 8 * fe_g0, b_fe, gfe, pre, nxt);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            float g = gh[r];
            if (l == 3) {
                g *= 0.70710678118654752440f;
                if (col >= MLP_SKIP_H) GPE[row * MLP_PE_STRIDE + (col - MLP_SKIP_H)] = g;
            }
            X[row * RS + col] = g * dsig[l - 1][r];
        }
        __syncthreads();
    }
    // layer 0: d/d(point encoding) = G_0 (32x128) * W_0 (128 x 27): one n-tile; the four waves split the reduction (4 groups
    // each) and the partial tiles are summed in a fixed order through the (now dead) h part of X -- deterministic
    {
        f32x16 gp;
#pragma unroll
        for (int r = 0; r < 16; ++r) gp[r] = 0.0f;
        gp = mfma_groups<4, MLP_PF, false>(X + a_lane * RS + a_half + 32 * wave, W.wb[0] + (size_t)(4 * wave) * 64 + lane, gp, pre, nullptr);
        __syncthreads();                                   // every wave has read G_0
#pragma unroll
        for (int r = 0; r < 16; ++r) X[acc_row(r, lane) * RS + col] = gp[r];
        __syncthreads();
        for (int i = tid; i < MLP_M * 32; i += 256) {
            const int row = i >> 5, c = i & 31;
            if (c < MLP_PE) {
                const float* xr = X + row * RS + c;
                GPE[row * MLP_PE_STRIDE + c] += ((xr[0] + xr[32]) + xr[64]) + xr[96];   // columns written by the l == 3 step are the same 27
            }
        }
    }
    __syncthreads();
    // conditioning gradient tiles -> LDS (reuse the h part of X: it is dead now)
    {
        const int c = 32 * (fe_tile - 4) + (lane & 31); // column inside the fe block
        if (c < FE && (!SPLIT_K || wave < 2)) {
            const float wl = W.w_last[MLP_H + c];       // layer 6 contributes the same vector for every point
#pragma unroll
            for (int r = 0; r < 16; ++r) X[acc_row(r, lane) * RS + c] = gfe[r] + wl;
        }
        if (SPLIT_K) {
            __syncthreads();
            if (c < FE && wave >= 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) X[acc_row(r, lane) * RS + c] += gfe[r];
            }
        }
    }
    __syncthreads();
    if (tid < MLP_M * 3) {
        const int p = tid / 3, a = tid % 3;
        const int64_t row = m0 + p;
        if (row < n) {
            const float* gpe = GPE + p * MLP_PE_STRIDE;
            const float* pe = PE + p * MLP_PE_STRIDE;
            float g = gpe[a];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = (float)(1 << k);
                g += f * (gpe[3 + 6 * k + a] * pe[6 + 6 * k + a] - gpe[6 + 6 * k + a] * pe[3 + 6 * k + a]);
            }
            g *= W.scale;
            const float* gf = X + p * RS;               // d/d fe  (FE values)
            const float* fe = X + p * RS + MLP_H;       // fe = [f, sin f, cos f, sin 2f, cos 2f]
            for (int c = 0; c < CF; ++c) {
                float df = gf[c] + gf[CF + c] * fe[2 * CF + c] - gf[2 * CF + c] * fe[CF + c] +
                           2.0f * (gf[3 * CF + c] * fe[4 * CF + c] - gf[4 * CF + c] * fe[3 * CF + c]);
                g += df * JAC[(p * CF + c) * 3 + a];
            }
            grad_out[3 * (index ? index[row] : row) + a] = g * W.inv_scale;
        }
    }
}

int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

static int sdf_mlp_launch(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                          const float* const* wb, const float* w_last, const float* w_last_scaled, float b_last, const float* b_last_dev, float scale,
                          const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* sdf_out, float* grad_out,
                          void* stream);

extern "C" int gens_sdf_mlp(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                            const float* const* wb, const float* w_last, const float* w_last_scaled, float b_last, float scale,
                            const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* sdf_out, float* grad_out,
                            void* stream) {
    return sdf_mlp_launch(vols_packed, dims, n_levels, wf, wb, w_last, w_last_scaled, b_last, nullptr, scale, pts, index, n, n_device, sdf_out, grad_out,
                          stream);
}

extern "C" int gens_sdf_mlp_dev(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                                const float* const* wb, const float* w_last, const float* b_last_dev, float scale,
                                const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* sdf_out, float* grad_out,
                                void* stream) {
    GENS_CHECK_ARG(b_last_dev, GENS_EINVAL, "gens_sdf_mlp_dev: null bias pointer");
    return sdf_mlp_launch(vols_packed, dims, n_levels, wf, wb, w_last, nullptr, 0.0f, b_last_dev, scale, pts, index, n, n_device, sdf_out, grad_out, stream);
}

static int sdf_mlp_launch(const float* const* vols_packed, const int* dims, int n_levels, const float* const* wf,
                          const float* const* wb, const float* w_last, const float* w_last_scaled, float b_last, const float* b_last_dev, float scale,
                          const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* sdf_out, float* grad_out,
                          void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_mlp", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_mlp: built for 1 to 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(!grad_out || n_levels == 3 || n_levels == 5, GENS_ELIMIT,
                   "gens_sdf_mlp: the row-major value + gradient kernel is built for 3 or 5 levels (gens_sdf_grad serves 1 to 5), got %d", n_levels);
    GENS_CHECK_ARG(wf && w_last && (wb || !grad_out), GENS_EINVAL, "gens_sdf_mlp: null weight table");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && sdf_out)), GENS_EINVAL, "gens_sdf_mlp: null pts / output");
    GENS_CHECK_ARG(scale != 0.0f, GENS_EINVAL, "gens_sdf_mlp: scale must be non-zero");
    if (n == 0) return 0;
    SdfMlpWeights W;
    for (int l = 0; l < MLP_NLAYER; ++l) {
        GENS_CHECK_ARG(wf[l] && (!grad_out || wb[l]), GENS_EINVAL, "gens_sdf_mlp: layer %d weights are null", l);
        W.wf[l] = (const float4*)wf[l];
        W.wb[l] = grad_out ? (const float4*)wb[l] : nullptr;
    }
    W.w_last = w_last;
    W.w_fwd = w_last_scaled ? w_last_scaled : w_last;
    W.b_last = b_last;
    W.b_last_dev = b_last_dev;
    W.scale = scale;
    W.inv_scale = 1.0f / scale;
    unsigned grid = gens_blocks(n, MLP_M);
    hipStream_t s = (hipStream_t)stream;
#define SDF_LAUNCH(FE_, GRAD_)                                                                                              \
    {                                                                                                                       \
        if (w_last_scaled) sdf_mlp_k<FE_, GRAD_, true><<<grid, 256, 0, s>>>(W, vs, pts, index, n, n_device, sdf_out, grad_out); \
        else sdf_mlp_k<FE_, GRAD_, false><<<grid, 256, 0, s>>>(W, vs, pts, index, n, n_device, sdf_out, grad_out);          \
    }
    switch (n_levels) {
        case 1: SDF_LAUNCH(20, false) break;
        case 2: SDF_LAUNCH(40, false) break;
        case 3: if (grad_out) SDF_LAUNCH(60, true) else SDF_LAUNCH(60, false) break;
        case 4: SDF_LAUNCH(80, false) break;
        default: if (grad_out) SDF_LAUNCH(100, true) else SDF_LAUNCH(100, false) break;
    }
#undef SDF_LAUNCH
    return gens_launch_status("gens_sdf_mlp");
}
