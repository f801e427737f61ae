// K6t: the VALUE of the SDF network (no gradient) in exact float32 on the matrix cores, TRANSPOSED: the kernel of the hierarchical
// sampling passes (implicit_surface.py:125, 329-352: 112 of the 240 network evaluations per ray) and of the 512^3 lattice of
// extract_geometry (:407-427).  Same arithmetic as k6_sdfmlp.hip (v_mfma_f32_32x32x2_f32, pre-scaled hidden units), other dataflow:
//
//   * the weights are the A operand (32 output features x 2 K), the activations of 32 points the B operand; ONE wavefront owns 32
//     points and all 128 hidden units (four accumulator tiles).  Register r of accumulator tile t holds, in lane (point n, half h),
//     feature 32 t + 8 (r >> 2) + 4 h + (r & 3) -- which is exactly what the B operand of an MFMA over the feature pair
//     {.. + 0 h, .. + 4 h} wants: the activated accumulator register IS the next layer's operand.  No LDS, no barrier, no address
//     arithmetic between the layers; waves are independent (k6_sdfmlp.hip: 12 barriers and 2 x 128 LDS stores per 32 points).
//   * the host packs the weights in that pair order (gens_amd.ops._pack_value_stream): per group of four pairs and output tile one
//     float4 per lane = four consecutive columns of a weight row, streamed from L2 one group ahead of its use.
//   * point encoding, volume features and the bias are further pairs (constant-one slot); layer 3's skip columns are the
//     point-encoding registers again (1 / sqrt(2) folded into the weights) and the 27 hidden pairs it does not read are not issued;
//     a trailing odd pair of a block is not issued either: 1972 MFMAs per 32 points instead of 1984 + padding.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The same mask-free formulation as k6g_sdf_grad.hip's epilogue, so that the value this kernel returns and the value the gradient kernel returns
// are the SAME float32 number for the same point (tests/test_hip_properties.py): log2(1 + 2^t) >= t always, and where torch switches to the
// linear branch (100 a > 20, t > 28.85) it already equals t to the last bit or two, so max(t, .) is the threshold; the clamp keeps 2^t finite.
__device__ __forceinline__ float softplus_ct(float t) {   // c * softplus_100(a) for t = c a, c = 100 / ln 2  (k6_sdfmlp.hip::softplus_t)
    const float u = 1.0f + __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(t, 126.0f, -3.0e38f));
    return __builtin_amdgcn_fmed3f(t, __builtin_amdgcn_logf(u), 3.0e38f);
}

// value of one packed (X, Y, Z, 4) volume at x (zero padding, align_corners=True): the taps of k6_sdfmlp.hip's prologue
__device__ __forceinline__ float4 sample_volume4t(const float4* __restrict__ v, int Xd, int Yd, int Zd, const float x[3], bool live) {
    float w0[3], w1[3];
    int i0[3];
    bool in0[3], in1[3];
    const int sz[3] = {Xd, Yd, Zd};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pos = (x[a] + 1.0f) / 2.0f * (float)(sz[a] - 1);
        const float f = fminf(fmaxf(floorf(pos), -2.0f), (float)sz[a] + 1.0f);
        i0[a] = (int)f;
        w0[a] = (f + 1.0f) - pos;
        w1[a] = pos - f;
        in0[a] = i0[a] >= 0 && i0[a] < sz[a];
        in1[a] = i0[a] + 1 >= 0 && i0[a] + 1 < sz[a];
    }
    float4 acc = f4_zero();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int a = c >> 2, b = (c >> 1) & 1, d = c & 1;
        const bool ok = live && (a ? in1[0] : in0[0]) && (b ? in1[1] : in0[1]) && (d ? in1[2] : in0[2]);
        const int cx = min(max(i0[0] + a, 0), Xd - 1), cy = min(max(i0[1] + b, 0), Yd - 1), cz = min(max(i0[2] + d, 0), Zd - 1);
        float4 t = v[((int64_t)cx * Yd + cy) * Zd + cz];
        if (!ok) t = f4_zero();
        acc = f4_madd(acc, t, (a ? w1[0] : w0[0]) * (b ? w1[1] : w0[1]) * (d ? w1[2] : w0[2]));
    }
    return acc;
}

template <int NLEV>
struct ValueShapeT {
    static constexpr int CF = 4 * NLEV;
    static constexpr int NCH = CF / 2;                 // channels per lane half
    static constexpr int NCS = 5 * NCH + 1;            // conditioning slots per half: 5 encodings per channel + the constant one
    static constexpr int GC = (NCS + 3) / 4;           // ... in groups of four pairs
    static constexpr int GP = 4;                       // point encoding: 15 slots per half
    static constexpr int NG = GP + 4 * (16 + GC) + (13 + GP + GC);     // groups of the six layers
};

#define TV_WAVES 4     // independent waves per workgroup

template <int NLEV>
__global__ __launch_bounds__(64 * TV_WAVES, 2) void sdf_value_t_k(LevelSet vols, const float4* __restrict__ wstream, const float* w_out, float b_last,
                                                                  float scale, float inv_scale, const float* __restrict__ pts,
                                                                  const int64_t* __restrict__ index, int64_t n_max,
                                                                  const int32_t* __restrict__ n_dev, float* __restrict__ sdf_out) {
    typedef ValueShapeT<NLEV> S;
    // the channels of a point are split between the two lane halves: half 0 takes the first NLEV / 2 levels, half 1 the last NLEV / 2, and with an
    // odd level count the middle level is shared, two channels each (gens_amd.ops._value_pairs packs the weights in the same order)
    constexpr int NCH = S::NCH, NCS = S::NCS, GC = S::GC, GP = S::GP, MID = NLEV / 2, ODD = NLEV & 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_pt = lane & 31, half = lane >> 5;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    const int64_t m0 = ((int64_t)blockIdx.x * TV_WAVES + wave) * 32;
    if (m0 >= n) return;

    // weights of group 0 are in flight during the prologue
    const float4* wp = wstream;          // wave-uniform: the address of a load is this scalar base + 16 lane + an immediate
    float4 wbuf[2][4];                   // this group's weights / the next group's (in flight): the two sets swap roles every group
    int par = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) wbuf[0][t] = wp[lane + 64 * t];

    // ------------------------------------------------------------------ prologue: this lane's B-operand slots
    const int64_t row = m0 + n_pt;
    const bool live = row < n;
    const int64_t src = live ? (index ? index[row] : row) : 0;
    float x[3] = {0.f, 0.f, 0.f};
    if (live) { x[0] = pts[3 * src]; x[1] = pts[3 * src + 1]; x[2] = pts[3 * src + 2]; }

    float pe[4 * GP];      // half 0: x, octaves 0 and 1 (pe[0:15]); half 1: octaves 2 and 3 (pe[15:27]), ONE, zeros
#pragma unroll
    for (int k = 0; k < 4 * GP; ++k) pe[k] = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = x[a] * scale;
        float s0, c0, s1, c1;
        hw_sincos(v * (half ? 4.0f : 1.0f), s0, c0);
        hw_sincos(v * (half ? 8.0f : 2.0f), s1, c1);
        if (half == 0) {
            pe[a] = v; pe[3 + a] = s0; pe[6 + a] = c0; pe[9 + a] = s1; pe[12 + a] = c1;
        } else {
            pe[a] = s0; pe[3 + a] = c0; pe[6 + a] = s1; pe[9 + a] = c1;
        }
    }
    if (half) pe[12] = 1.0f;

    float cnd[4 * GC];     // volume features: 5 encodings of this half's NCH channels, then ONE (half 0), then zeros
    const float* wo = w_out + half * (64 + 4 * GC);
    float s_cond = 0.0f;
    {
        float f[NCH];
#pragma unroll
        for (int j = 0; j < MID; ++j) {     // whole levels of this half; with an odd count level MID is shared, two channels each
            const int l = half ? MID + ODD + j : j;
            const float4 t = sample_volume4t((const float4*)vols.data[l], vols.dx[l], vols.dy[l], vols.dz[l], x, live);
            f[4 * j] = t.x; f[4 * j + 1] = t.y; f[4 * j + 2] = t.z; f[4 * j + 3] = t.w;
        }
        if constexpr (ODD) {
            const float4 t = sample_volume4t((const float4*)vols.data[MID], vols.dx[MID], vols.dy[MID], vols.dz[MID], x, live);
            f[4 * MID] = half ? t.z : t.x;
            f[4 * MID + 1] = half ? t.w : t.y;
        }
        // not-a-number inputs must come out as not-a-number (the reference's layers propagate them; the max / median forms of the
        // activation below would drop them): a poison term 0 * (sum of the inputs), NaN iff one of them is NaN or infinite
        {
            float acc_in = x[0] + x[1] + x[2];
#pragma unroll
            for (int j = 0; j < NCH; ++j) acc_in += f[j];
            s_cond = 0.0f * acc_in;
        }
#pragma unroll
        for (int k = NCS - 1; k < 4 * GC; ++k) cnd[k] = (k == NCS - 1 && half == 0) ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float e[5];
            e[0] = f[j];
            hw_sincos(f[j], e[1], e[2]);
            hw_sincos(2.0f * f[j], e[3], e[4]);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                cnd[5 * j + q] = e[q];
                s_cond = __builtin_fmaf(e[q], wo[64 + 5 * j + q], s_cond);       // layer 6 reads the conditioning features too
            }
        }
    }

    // ------------------------------------------------------------------ the six layers
    f32x16 acc[4], H[4];

    // one group: four (CNT: three) MFMAs per output tile over the pairs whose B operands are b0..b3; the next group's weights are
    // requested first, into the other register set, so that they travel while this group's 16 MFMAs (1024 cycles) run
#define TV_GROUP(CNT, b0, b1, b2, b3)                                                                         \
    {                                                                                                        \
        float4(&a_)[4] = wbuf[par];                                                                          \
        par ^= 1;                                                                                            \
        wp += 256;                                                                                           \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) wbuf[par][t_] = wp[lane + 64 * t_];                 \
        __builtin_amdgcn_sched_barrier(0);         /* the requests stay AHEAD of this group's MFMAs */        \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t_].x, (b0), acc[t_], 0, 0, 0); \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t_].y, (b1), acc[t_], 0, 0, 0); \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t_].z, (b2), acc[t_], 0, 0, 0); \
        if ((CNT) == 4) {                                                                                    \
            _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[t_].w, (b3), acc[t_], 0, 0, 0); \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
#define TV_HIDDEN(NT)                                                                        \
    _Pragma("unroll") for (int t2_ = 0; t2_ < (NT); ++t2_)                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                     \
            TV_GROUP(4, H[t2_][4 * g_], H[t2_][4 * g_ + 1], H[t2_][4 * g_ + 2], H[t2_][4 * g_ + 3])
#define TV_COND()                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < GC; ++g_)                                        \
        TV_GROUP((4 * g_ + 4 <= NCS) ? 4 : 3, cnd[4 * g_], cnd[4 * g_ + 1], cnd[4 * g_ + 2], cnd[4 * g_ + 3])
#define TV_PE()                                                                              \
    _Pragma("unroll") for (int g_ = 0; g_ < GP; ++g_)                                        \
        TV_GROUP(g_ < GP - 1 ? 4 : 3, pe[4 * g_], pe[4 * g_ + 1], pe[4 * g_ + 2], pe[4 * g_ + 3])
#define TV_ZERO()                                             \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)          \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[t_][r_] = 0.0f;
#define TV_ACTIVATE()                                         \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)          \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) H[t_][r_] = softplus_ct(acc[t_][r_]);

    static_assert(4 * GC >= NCS && 4 * GC - NCS <= 3, "the conditioning block ends with a partly filled group (its empty pairs carry zeros on both sides)");
    TV_ZERO();
    TV_PE();
    TV_ACTIVATE();
#pragma unroll
    for (int l = 1; l < 6; ++l) {
        TV_ZERO();
        if (l == 3) {     // x = cat([h[:101], pe]) / sqrt(2): features 104.. of the hidden state are not read (group (3, 0) holds 96..103)
            TV_HIDDEN(3);
            TV_GROUP(4, H[3][0], H[3][1], H[3][2], H[3][3]);
            TV_PE();
        } else {
            TV_HIDDEN(4);
        }
        TV_COND();
        TV_ACTIVATE();
    }
#undef TV_GROUP
#undef TV_HIDDEN
#undef TV_COND
#undef TV_PE
#undef TV_ZERO
#undef TV_ACTIVATE

    // ------------------------------------------------------------------ layer 6, sdf row only: this lane's half of the dot product
    {
        asm volatile("" ::: "memory");             // keep the 64 output weights from being loaded (and spilled) ahead of the layers
        float s = s_cond;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s = __builtin_fmaf(H[t][r], wo[16 * t + r], s);
        s += __shfl_xor(s, 32, 64);
        if (half == 0 && live) sdf_out[src] = (s + b_last) * inv_scale;
    }
}

int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

extern "C" int gens_sdf_value_groups(int n_levels) {
    switch (n_levels) {
        case 1: return ValueShapeT<1>::NG;
        case 2: return ValueShapeT<2>::NG;
        case 3: return ValueShapeT<3>::NG;
        case 4: return ValueShapeT<4>::NG;
        case 5: return ValueShapeT<5>::NG;
        default: return 0;
    }
}

extern "C" int gens_sdf_value(const float* const* vols_packed, const int* dims, int n_levels, const float* wstream, const float* w_out,
                              float b_last, float scale, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device,
                              float* sdf_out, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_value", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_value: built for 1 to 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(wstream && w_out, GENS_EINVAL, "gens_sdf_value: null weight stream");
    GENS_CHECK_ARG(((uintptr_t)wstream & 15) == 0, GENS_EINVAL, "gens_sdf_value: the weight stream must be 16-byte aligned");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && sdf_out)), GENS_EINVAL, "gens_sdf_value: null pts / output");
    GENS_CHECK_ARG(scale != 0.0f, GENS_EINVAL, "gens_sdf_value: scale must be non-zero");
    if (n == 0) return 0;
    const unsigned grid = gens_blocks(n, 32 * TV_WAVES);
    hipStream_t s = (hipStream_t)stream;
#define TV_LAUNCH(NL) sdf_value_t_k<NL><<<grid, 64 * TV_WAVES, 0, s>>>(vs, (const float4*)wstream, w_out, b_last, scale, 1.0f / scale, pts, index, n, n_device, sdf_out)
    switch (n_levels) {
        case 1: TV_LAUNCH(1); break;
        case 2: TV_LAUNCH(2); break;
        case 3: TV_LAUNCH(3); break;
        case 4: TV_LAUNCH(4); break;
        default: TV_LAUNCH(5); break;
    }
#undef TV_LAUNCH
    return gens_launch_status("gens_sdf_value");
}
