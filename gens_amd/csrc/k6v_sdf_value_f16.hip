// K6v: the VALUE of the SDF network (no gradient) on the f16 matrix cores with split operands -- the kernel behind the 512^3
// lattice of extract_geometry (/root/reference/models/modules/implicit_surface.py:407-427 -> sdf_network.py:98-129) and behind
// the opt-in "f16x2" arithmetic of the up-sampling passes.  Same arithmetic contract as k6h_sdfmlp_f16.hip (every operand an
// (hi, lo) pair of halfs, a*b = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, float32 accumulation, overflow flag + float32
// re-run by the caller), different dataflow:
//
//   * TRANSPOSED products: the weights are the A operand (32 output features x 16 K), the activations of 32 points the B operand.
//     A wave owns 32 points and ALL 128 output features (four accumulator tiles).  The accumulator layout of an output tile -- lane
//     (point n, half h) holds features 8 j + 4 h + i -- is exactly the B layout of two K blocks of the next layer once the
//     reduction index is PERMUTED accordingly, and the host packs the weights in that permuted order: the activations never leave
//     the registers.  No LDS traffic, no address arithmetic and no barrier between the layers of a wave.
//   * The weight stream (one 8 KB "unit" per 16-deep K block: 4 feature tiles x {hi, lo} x 64 lanes x 16 B, 64 units for three
//     volume levels) goes global -> LDS by LDS-direct loads in chunks of four units, double buffered, ONE stream per workgroup of
//     128 points: a quarter of the L2 traffic per point of the row-major kernel, whose 32-point workgroups were bound by it.
//   * Point encoding, volume features and the bias are extra K blocks (constant-one slot), the skip connection of layer 3
//     re-uses the point-encoding registers with layer 3's skip columns as their weights, 1 / sqrt(2) folded into the weights.
//   * Hidden units travel pre-scaled by c = 100 / ln 2 (see k6_sdfmlp.hip::softplus_t): softplus is exp2, add, log2, select.
//   * Two workgroups per CU run out of phase, so one's activation epilogue (VALU) overlaps the other's products (matrix pipe).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifndef HV_WAVES
#define HV_WAVES 4                    // waves per workgroup = 128 points per pass of the weight stream (8 waves sharing one pass, one workgroup per CU: 2.71 against 2.61 ms per 3.4 M points -- the two workgroups of a CU run out of phase, one pass in lock step does not)
#endif
#define HV_UNIT 8192                  // bytes of one K block of weights: 4 tiles x 2 parts x 64 lanes x 16 B
#define HV_CHUNK_UNITS 4
#define HV_CHUNK (HV_UNIT * HV_CHUNK_UNITS)

__device__ __forceinline__ float softplus_c(float t) {   // c * softplus_100(a) for t = c a, c = 100 / ln 2
    const float u = 1.0f + __builtin_amdgcn_exp2f(t);
    return t > 28.853900817779268f ? t : __builtin_amdgcn_logf(u);
}

struct HalfPair {
    f16x8 h, l;
};
// element e of the pair <- split(x);  E is a compile-time constant after unrolling
#define HV_PUT(P, E, X)                                   \
    do {                                                  \
        const float x_ = (X);                             \
        const _Float16 h_ = (_Float16)x_;                 \
        (P).h[E] = h_;                                    \
        (P).l[E] = (_Float16)(x_ - (float)h_);            \
    } while (0)

// value of one packed (X, Y, Z, 4) volume at x (zero padding, align_corners=True): the same taps as k6_sdfmlp.hip's prologue
__device__ __forceinline__ float4 sample_volume4(const float4* __restrict__ v, int Xd, int Yd, int Zd, const float x[3], bool live) {
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
struct ValueShape {
    static constexpr int CF = 4 * NLEV;                 // volume channels
    static constexpr int NCH = CF / 2;                  // channels per lane half
    static constexpr int NC = (5 * NCH + 1 + 7) / 8;    // conditioning K blocks: 5 encodings per channel + the constant-one slot
    static constexpr int NU = 2 + 4 * (8 + NC) + (10 + NC);       // K blocks of the six layers
    static constexpr int NCHUNK = (NU + HV_CHUNK_UNITS - 1) / HV_CHUNK_UNITS;
};

template <int NLEV>
__global__ __launch_bounds__(64 * HV_WAVES, 8 / HV_WAVES) void sdf_value_h_k(LevelSet vols, const char* __restrict__ units, const float* w_out,
                                                                  float b_last, float scale, float inv_scale, const float* __restrict__ pts,
                                                                  const int64_t* __restrict__ index, int64_t n_max,
                                                                  const int32_t* __restrict__ n_dev, float* __restrict__ sdf_out,
                                                                  int* __restrict__ overflow) {
    typedef ValueShape<NLEV> S;
    constexpr int NC = S::NC, NCH = S::NCH, MID = NLEV / 2;
    __shared__ __attribute__((aligned(16))) char WBUF[2 * HV_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pt = lane & 31, half = lane >> 5;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    const int64_t m0 = (int64_t)blockIdx.x * (32 * HV_WAVES);
    if (m0 >= n) return;

    // chunk c of the weight stream -> WBUF[c & 1]: 32 pieces of 1 KB, wave w takes pieces w, w + 4, ... (LDS-direct: the
    // destination of lane i is the wave-uniform piece base + 16 i)
    auto stage = [&](int c) {
        const char* src = units + (size_t)c * HV_CHUNK + lane * 16;
        char* dst = WBUF + (c & 1) * HV_CHUNK;
#pragma unroll
        for (int p = 0; p < HV_CHUNK / 1024 / HV_WAVES; ++p) {
            const int piece = wave + HV_WAVES * p;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
        }
    };
    stage(0);

    // ------------------------------------------------------------------ prologue: this lane's B-operand slots
    const int64_t row = m0 + 32 * wave + n_pt;
    const bool live = row < n;
    const int64_t src = live ? (index ? index[row] : row) : 0;
    float x[3] = {0.f, 0.f, 0.f};
    if (live) { x[0] = pts[3 * src]; x[1] = pts[3 * src + 1]; x[2] = pts[3 * src + 2]; }
    float vmax = 0.0f;        // largest magnitude handed to the half-precision operands (NaN-propagating: see the end of the kernel)

    HalfPair P[2];     // point encoding: half 0 = x, octaves 0 and 1, ONE; half 1 = octaves 2 and 3, zeros
    {
        float q[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) q[k] = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = x[a] * scale;
            float s0, c0, s1, c1;
            hw_sincos(v * (half ? 4.0f : 1.0f), s0, c0);
            hw_sincos(v * (half ? 8.0f : 2.0f), s1, c1);
            if (half == 0) {
                q[a] = v; q[3 + a] = s0; q[6 + a] = c0; q[9 + a] = s1; q[12 + a] = c1;
            } else {
                q[a] = s0; q[3 + a] = c0; q[6 + a] = s1; q[9 + a] = c1;
            }
        }
        if (half == 0) q[15] = 1.0f;
        vmax = fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fabsf(q[2]));
#pragma unroll
        for (int k = 0; k < 16; ++k) HV_PUT(P[k >> 3], k & 7, q[k]);
    }

    HalfPair C[NC];    // volume features: 5 encodings of this half's NCH channels, then ONE (half 0), then zeros
    const float* wo = w_out + half * (64 + 8 * NC);
    float s_cond = 0.0f, nan_sum = 0.0f;
    {
        float f[NCH];
        // whole levels of this half: 0 .. MID-1 (half 0) or MID+1 .. NLEV-1 (half 1); level MID is shared, two channels each
#pragma unroll
        for (int j = 0; j < MID; ++j) {
            const int l = half ? MID + 1 + j : j;
            const float4 t = sample_volume4((const float4*)vols.data[l], vols.dx[l], vols.dy[l], vols.dz[l], x, live);
            f[4 * j] = t.x; f[4 * j + 1] = t.y; f[4 * j + 2] = t.z; f[4 * j + 3] = t.w;
        }
        {
            const float4 t = sample_volume4((const float4*)vols.data[MID], vols.dx[MID], vols.dy[MID], vols.dz[MID], x, live);
            f[4 * MID] = half ? t.z : t.x;
            f[4 * MID + 1] = half ? t.w : t.y;
        }
#pragma unroll
        for (int k = 0; k < 8 * NC; ++k) {
            if (k >= 5 * NCH) {
                C[k >> 3].h[k & 7] = (k == 5 * NCH && half == 0) ? (_Float16)1.0f : (_Float16)0.0f;
                C[k >> 3].l[k & 7] = (_Float16)0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float e[5];
            e[0] = f[j];
            hw_sincos(f[j], e[1], e[2]);
            hw_sincos(2.0f * f[j], e[3], e[4]);
            vmax = fmaxf(vmax, fabsf(f[j]));
            nan_sum += f[j];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                HV_PUT(C[(5 * j + q) >> 3], (5 * j + q) & 7, e[q]);
                s_cond = __builtin_fmaf(e[q], wo[64 + 5 * j + q], s_cond);       // layer 6 reads the conditioning features too
            }
        }
    }

    // ------------------------------------------------------------------ the six layers
    f32x16 acc[4];
    HalfPair H[8];
    float hmax = 0.0f;

    // products of CNT consecutive K blocks (global unit numbers U0 ..) with the B operands b[0 .. CNT).  The A operands of a K block come
    // out of LDS in two groups of two feature tiles (hi and lo), ONE GROUP AHEAD of the MFMAs that read them (two register sets of 16:
    // what one K block took before) -- also across calls: with NXT the last group asks for the first group of the unit that follows in the
    // stream, and that call says PRE.  Within a group the two accumulators alternate.
    f16x8 abuf[2][4];
    int par = 0;
    // group G_ (tiles 2 G_, 2 G_ + 1) of unit U_ -> register set SET_; the first group of a chunk waits for the chunk and refills the other buffer
#define HV_LOADG(SET_, U_, G_)                                                                                      \
    {                                                                                                               \
        const int u__ = (U_);                                                                                       \
        if ((G_) == 0 && (u__ & (HV_CHUNK_UNITS - 1)) == 0) {     /* chunk boundary: this chunk has landed, the other buffer is free */ \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
            __syncthreads();                                                                                        \
            if (u__ / HV_CHUNK_UNITS + 1 < S::NCHUNK) stage(u__ / HV_CHUNK_UNITS + 1);                              \
        }                                                                                                           \
        const f16x8* A__ = (const f16x8*)(WBUF + ((u__ / HV_CHUNK_UNITS) & 1) * HV_CHUNK + (u__ & (HV_CHUNK_UNITS - 1)) * HV_UNIT) + lane; \
        _Pragma("unroll") for (int j__ = 0; j__ < 4; ++j__) abuf[SET_][j__] = A__[(4 * (G_) + j__) * 64];           \
    }
#define HV_GEMM(U0, CNT, B, PRE, NXT)                                                                               \
    {                                                                                                               \
        if (!(PRE)) HV_LOADG(par, (U0), 0)                                                                          \
        _Pragma("unroll") for (int q_ = 0; q_ < 2 * (CNT); ++q_) {                                                  \
            const int i_ = q_ >> 1, g_ = q_ & 1;                                                                    \
            if (q_ + 1 < 2 * (CNT)) HV_LOADG(par ^ 1, (U0) + ((q_ + 1) >> 1), (q_ + 1) & 1)                         \
            else if (NXT) HV_LOADG(par ^ 1, (U0) + (CNT), 0)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) acc[2 * g_ + t_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abuf[par][2 * t_], (B)[i_].h, acc[2 * g_ + t_], 0, 0, 0); \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) acc[2 * g_ + t_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abuf[par][2 * t_], (B)[i_].l, acc[2 * g_ + t_], 0, 0, 0); \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) acc[2 * g_ + t_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abuf[par][2 * t_ + 1], (B)[i_].h, acc[2 * g_ + t_], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            par ^= 1;                                                                                               \
        }                                                                                                           \
    }
#define HV_ZERO()                                             \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)          \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[t_][r_] = 0.0f;
    // activation; tile t, register r becomes element r & 7 of hidden K block 2 t + (r >> 3) of the next layer
#define HV_ACTIVATE(SPLIT)                                                         \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)                               \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                        \
            const float h_v = softplus_c(acc[t_][r_]);                             \
            acc[t_][r_] = h_v;                                                     \
            if (SPLIT) {                                                           \
                hmax = fmaxf(hmax, h_v);                                           \
                HV_PUT(H[2 * t_ + (r_ >> 3)], r_ & 7, h_v);                        \
            }                                                                      \
        }

    constexpr int LW = 8 + NC;                 // K blocks of a plain hidden layer
    constexpr int U1 = 2, U2 = U1 + LW, U3 = U2 + LW, U4 = U3 + LW + 2, U5 = U4 + LW;
    static_assert(U5 + LW == S::NU, "unit count");
    HV_ZERO();
    HV_GEMM(0, 2, P, 0, 1);
    HV_ACTIVATE(true);
    HV_ZERO();
    HV_GEMM(U1, 8, H, 1, 1);
    HV_GEMM(U1 + 8, NC, C, 1, 1);
    HV_ACTIVATE(true);
    HV_ZERO();
    HV_GEMM(U2, 8, H, 1, 1);
    HV_GEMM(U2 + 8, NC, C, 1, 1);
    HV_ACTIVATE(true);
    HV_ZERO();                                 // layer 3: hidden (skip rows 101.. have zero weights), point encoding, conditioning
    HV_GEMM(U3, 8, H, 1, 1);
    HV_GEMM(U3 + 8, 2, P, 1, 1);
    HV_GEMM(U3 + 10, NC, C, 1, 1);
    HV_ACTIVATE(true);
    HV_ZERO();
    HV_GEMM(U4, 8, H, 1, 1);
    HV_GEMM(U4 + 8, NC, C, 1, 1);
    HV_ACTIVATE(true);
    HV_ZERO();
    HV_GEMM(U5, 8, H, 1, 1);
    HV_GEMM(U5 + 8, NC, C, 1, 0);
    HV_ACTIVATE(false);
#undef HV_GEMM
#undef HV_LOADG
#undef HV_ZERO
#undef HV_ACTIVATE

    // ------------------------------------------------------------------ layer 6, sdf row only: this lane's half of the dot product
    {
        asm volatile("" ::: "memory");             // keep the 64 output weights from being loaded (and spilled) ahead of the layers
        float s = s_cond;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s = __builtin_fmaf(acc[t][r], wo[16 * t + r], s);
        s += __shfl_xor(s, 32, 64);
        if (half == 0 && live) sdf_out[src] = (s + b_last) * inv_scale;
    }
    // out of the half range, or not a number (fmaxf drops NaNs: the sum of the volume features keeps them; the point encoding and the
    // hidden units can only be NaN if those are)
    const bool big = !(fmaxf(vmax, hmax) < 3.0e4f) || nan_sum != nan_sum;
    if (__any(big) && lane == 0) atomicOr(overflow, 1);
}

int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

extern "C" int gens_sdf_value_f16_units(int n_levels) {
    return n_levels == 3 ? ValueShape<3>::NCHUNK * HV_CHUNK_UNITS : n_levels == 5 ? ValueShape<5>::NCHUNK * HV_CHUNK_UNITS : 0;
}

extern "C" int gens_sdf_value_f16(const float* const* vols_packed, const int* dims, int n_levels, const void* units, const float* w_out,
                                  float b_last, float scale, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device,
                                  float* sdf_out, int* overflow_flag, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_value_f16", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels == 3 || n_levels == 5, GENS_ELIMIT, "gens_sdf_value_f16: built for 3 or 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(units && w_out && overflow_flag, GENS_EINVAL, "gens_sdf_value_f16: null weight stream / flag");
    GENS_CHECK_ARG(((uintptr_t)units & 15) == 0, GENS_EINVAL, "gens_sdf_value_f16: the weight stream must be 16-byte aligned");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && sdf_out)), GENS_EINVAL, "gens_sdf_value_f16: null pts / output");
    GENS_CHECK_ARG(scale != 0.0f, GENS_EINVAL, "gens_sdf_value_f16: scale must be non-zero");
    if (n == 0) return 0;
    const unsigned grid = gens_blocks(n, 32 * HV_WAVES);
    hipStream_t s = (hipStream_t)stream;
    if (n_levels == 3)
        sdf_value_h_k<3><<<grid, 64 * HV_WAVES, 0, s>>>(vs, (const char*)units, w_out, b_last, scale, 1.0f / scale, pts, index, n, n_device, sdf_out,
                                                        overflow_flag);
    else
        sdf_value_h_k<5><<<grid, 64 * HV_WAVES, 0, s>>>(vs, (const char*)units, w_out, b_last, scale, 1.0f / scale, pts, index, n, n_device, sdf_out,
                                                        overflow_flag);
    return gens_launch_status("gens_sdf_value_f16");
}
