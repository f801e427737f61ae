// K6gh: value AND gradient d sdf / dx of the SDF network with SPLIT operands on the f16 matrix cores -- the opt-in "f16x2" arithmetic of
// render_core's 128 samples per ray (/root/reference/models/modules/implicit_surface.py:179-191 -> sdf_network.py:98-154).  The arithmetic
// contract of k6v_sdf_value_f16.hip (every operand an (hi, lo) pair of halfs, a * b = hi * hi + hi * lo + lo * hi on
// v_mfma_f32_32x32x16_f16, float32 accumulation, overflow flag + float32 re-run by the caller), the dataflow of k6g_sdf_grad.hip:
//
//   * one wavefront owns 32 points and all 128 hidden units; weights are the A operand, activations the B operand; the activated
//     accumulators of a layer, split into halfs, ARE the B operands of the next one (the host packs the reduction index in that order).
//     The reverse pass is the same chain on the transposed matrices: G_{l-1} = (W_l^T G_l) * softplus'(a_{l-1}); the gradients with respect
//     to the volume-feature slots and the point-encoding slots of a lane accumulate in their own tiles whose rows the host orders so
//     that the chain rule to x is lane-local (k6g_sdf_grad.hip's header).
//   * the f16 pipe runs 16 x the float32 rate, three products per operand pair: 5.3 x less matrix time than K6g -- and five times the
//     weight bytes per unit time.  The weight stream (1 KB "pieces": the A operand of one output tile and one 16-deep K block, hi or
//     lo) therefore goes global -> LDS ONCE per workgroup of four waves (128 points), by LDS-direct loads into a ring of four 8 KB
//     chunks filled three chunks ahead, and each wave reads its A operands from there in groups of two output tiles, one group ahead of the six
//     MFMAs that use them (also across layers: the stream is consumed strictly in order).
//   * softplus' of the six layers (float32: the gradient's precision is theirs): layer 5 is consumed on the spot, layers 3 and 4 stay
//     in registers, layers 0 and 1 wait in LDS (32 KB per wave: with the ring exactly the CU's 160 KB, so ONE workgroup per CU, one
//     wave per SIMD, 512 registers), layer 2 and the trilinear Jacobians in a slot of global memory that belongs to the wave's
//     (CU, wave) pair and stays in the XCD's L2 (k6g_sdf_grad.hip).
//   * gradients travel multiplied by a power of two (g_scale, chosen by the host from |w_last|) so that their lo parts stay normal
//     halfs; the largest |G| is tracked and raises the overflow flag like the largest hidden unit does.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define GH_WAVES 4
#define GH_PIECE 1024                       // bytes of one piece: 64 lanes x 8 halfs
#ifndef GH_CH
#define GH_CH 8                             // pieces per chunk of the ring
#define GH_RING 4
#endif
#ifndef GH_GT
#define GH_GT 2                             // output tiles per group of A operands (measured: 2 5.90 ms, 3 6.07, 4 6.09 per 3.4 M points)
#endif
#define GH_LPW (GH_CH / GH_WAVES)           // LDS-direct loads per wave and chunk
#define GH_W_BYTES (GH_RING * GH_CH * GH_PIECE)
#define GH_D_BYTES (2 * 16 * 64 * 16)       // softplus' of layers 0 and 1 of one wave
#define GH_LDS_BYTES (GH_W_BYTES + GH_WAVES * GH_D_BYTES)
#define GH_JL_OFF 16384                     // stash slot: [0, 16 KB) softplus' of layer 2, then 32 rows x 256 B of Jacobians, then the lock
#define GH_LOCK_OFF (GH_JL_OFF + 32 * 256)
#define GH_SLOT_BYTES (GH_LOCK_OFF + 256)
#define GH_SLOTS 2048                       // (XCC 3 bits, SE 2, CU 4, wave 2)
#define GH_C 144.26950408889634f            // 100 / ln 2: hidden units travel as c * softplus (k6_sdfmlp.hip::softplus_t)

struct SplitBlock {       // the B operand of one 16-deep K block: this lane's 8 values as halfs, hi and lo parts
    u32x4 h, l;
};

// (a, b) -> packed hi halfs and packed lo halfs; a - hi is exact in float32, so hi + lo carries 22 bits of a
struct SplitWord {
    uint32_t hi, lo;
};
__device__ __forceinline__ SplitWord split_pair(float a, float b) {
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
    float ra, rb;      // a - (float)hi in ONE instruction: the mixed-precision fma reads the half straight out of the packed word
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
    return {h, __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(ra, rb))};
}
#define GH_PUT(BLK_, W_, A_, B_)                     \
    {                                                \
        const SplitWord sw__ = split_pair((A_), (B_)); \
        (BLK_).h[(W_)] = sw__.hi;                    \
        (BLK_).l[(W_)] = sw__.lo;                    \
    }

// value of one packed (X, Y, Z, 4) volume at x and its derivative with respect to x (zero padding, align_corners=True)
__device__ __forceinline__ float4 sample_volume4j(const float4* __restrict__ v, int Xd, int Yd, int Zd, const float x[3], bool live, float4& jx,
                                                  float4& jy, float4& jz) {
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
    jx = f4_zero(); jy = f4_zero(); jz = f4_zero();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int a = c >> 2, b = (c >> 1) & 1, d = c & 1;
        const bool ok = live && (a ? in1[0] : in0[0]) && (b ? in1[1] : in0[1]) && (d ? in1[2] : in0[2]);
        const int cx = min(max(i0[0] + a, 0), Xd - 1), cy = min(max(i0[1] + b, 0), Yd - 1), cz = min(max(i0[2] + d, 0), Zd - 1);
        float4 t = v[((int64_t)cx * Yd + cy) * Zd + cz];
        if (!ok) t = f4_zero();
        const float wx = a ? w1[0] : w0[0], wy = b ? w1[1] : w0[1], wz = d ? w1[2] : w0[2];
        acc = f4_madd(acc, t, wx * wy * wz);
        jx = f4_madd(jx, t, (a ? 1.0f : -1.0f) * wy * wz);
        jy = f4_madd(jy, t, wx * (b ? 1.0f : -1.0f) * wz);
        jz = f4_madd(jz, t, wx * wy * (d ? 1.0f : -1.0f));
    }
    const float sx = (float)(Xd - 1) / 2.0f, sy = (float)(Yd - 1) / 2.0f, sz_ = (float)(Zd - 1) / 2.0f;
    jx.x *= sx; jx.y *= sx; jx.z *= sx; jx.w *= sx;
    jy.x *= sy; jy.y *= sy; jy.z *= sy; jy.w *= sy;
    jz.x *= sz_; jz.y *= sz_; jz.z *= sz_; jz.w *= sz_;
    return acc;
}

// The piece stream, in the order the kernel consumes it (gens_amd.ops._pack_grad_pieces):
//   forward   layer 0: 2 point-encoding K blocks; layers 1..5: NC conditioning K blocks, (layer 3: 2 point-encoding K blocks,) 8 hidden
//             K blocks; every K block = 4 output tiles x {hi, lo}
//   reverse   layers 5..1: 8 K blocks of G_l (layer 2: 7 -- 101 units), each = (4 hidden tiles, TC conditioning tiles, at layer 3 the
//             point-encoding tile) x {hi, lo}; then 8 K blocks of G_0 x the point-encoding tile x {hi, lo}
template <int NLEV>
struct GradShapeH {
    static constexpr int CF = 4 * NLEV;
    static constexpr int NCH = CF / 2;                     // channels per lane half
    static constexpr int NC = (5 * NCH + 1 + 7) / 8;       // conditioning K blocks: 5 encodings per channel + the constant-one slot
    static constexpr int TC = (5 * NCH + 15) / 16;         // accumulator tiles of the conditioning gradient (16 slots of a half per tile)
    static constexpr int fwd(int l) { return l == 0 ? 0 : 8 * (2 + (l - 1) * (NC + 8) + (l > 3 ? 2 : 0)); }      // first piece of forward layer l
    static constexpr int FWD_PIECES = fwd(5) + 8 * (NC + 8);
    static constexpr int rev_tiles(int l) { return 4 + TC + (l == 3 ? 1 : 0); }
    static constexpr int rev_blocks(int l) { return l == 2 ? 7 : 8; }
    static constexpr int rev(int l) {      // first piece of reverse layer l = 5 .. 1; rev(0): the G_0 blocks
        int p = FWD_PIECES;
        for (int k = 5; k > l; --k) p += 2 * rev_tiles(k) * rev_blocks(k);
        return p;
    }
    static constexpr int PIECES = rev(0) + 16;
    static constexpr int NCHUNK = (PIECES + GH_CH - 1) / GH_CH;
};

template <int NLEV>
__global__ __launch_bounds__(64 * GH_WAVES, 1) void sdf_grad_h_k(LevelSet vols, const char* __restrict__ pieces, const float* w_out, float b_last,
                                                                 float scale, float inv_scale, float g_scale, float inv_g_scale,
                                                                 const float* __restrict__ pts, const int64_t* __restrict__ index, int64_t n_max,
                                                                 const int32_t* __restrict__ n_dev, float* __restrict__ sdf_out,
                                                                 float* __restrict__ grad_out, char* __restrict__ stash_all,
                                                                 int* __restrict__ overflow) {
    typedef GradShapeH<NLEV> S;
    constexpr int NCH = S::NCH, NC = S::NC, TC = S::TC, MID = NLEV / 2, NCHUNK = S::NCHUNK;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_pt = lane & 31, half = lane >> 5;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    const int64_t m0 = (int64_t)blockIdx.x * (32 * GH_WAVES);
    if (m0 >= n) return;
    float4* const DS = (float4*)(lds + GH_W_BYTES + wave * GH_D_BYTES);       // [layer][accumulator register / 4][lane]

    // chunk c of the stream -> ring slot c % GH_RING: wave w brings pieces w, w + 4, .. (LDS-direct buffer loads: the piece's offset in
    // the stream is a scalar, 16 lane the one vector offset, lane i lands at the piece's base in LDS + 16 i)
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)pieces, 0, NCHUNK * GH_CH * GH_PIECE, 0x00020000);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto stage = [&](int c) {
        char* dst = lds + (c % GH_RING) * (GH_CH * GH_PIECE);
#pragma unroll
        for (int p = 0; p < GH_LPW; ++p) {
            const int piece = wave + GH_WAVES * p;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(dst + piece * GH_PIECE), 16, lane16,
                                                     (uint32_t)(c * (GH_CH * GH_PIECE) + piece * GH_PIECE), 0, 0);
        }
    };
#pragma unroll
    for (int c = 0; c < GH_RING - 1; ++c) stage(c);

    // this wave's slot of the stash (one workgroup per CU -- the LDS allocation sees to it -- so (CU, wave) is unique; the lock word
    // makes that a performance assumption instead of a correctness one)
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));          // [11:8] CU, [15:13] SE (scripts/probe/hwid_probe.py)
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const uint32_t slot = ((((xcc_id & 7u) << 2 | ((hw_id >> 13) & 3u)) << 4 | ((hw_id >> 8) & 15u)) << 2) | (uint32_t)wave;
    char* const stash_slot = stash_all + (size_t)slot * GH_SLOT_BYTES;
    uint32_t* const lock = (uint32_t*)(stash_slot + GH_LOCK_OFF);
    uint32_t lock_seen = lane == 0 ? atomicCAS(lock, 0u, 1u) : 0u;      // (asked for here, looked at before the first store into the slot)
    const __amdgpu_buffer_rsrc_t stash = __builtin_amdgcn_make_buffer_rsrc((void*)stash_slot, 0, GH_LOCK_OFF, 0x00020000);
    const uint32_t stash_lane = (uint32_t)lane * 16u, jl_lane = GH_JL_OFF + (uint32_t)lane * 4u;

    // ------------------------------------------------------------------ prologue: this lane's B-operand slots
    const int64_t row = m0 + 32 * wave + n_pt;
    const bool live = row < n;
    const int64_t src = live ? (index ? index[row] : row) : 0;
    float x[3] = {0.f, 0.f, 0.f};
    if (live) { x[0] = pts[3 * src]; x[1] = pts[3 * src + 1]; x[2] = pts[3 * src + 2]; }
    float vmax = 0.0f, hmax = 0.0f, gmax = 0.0f, nan_sum = 0.0f;      // largest magnitudes handed to half precision (NaN-propagating sum beside them)

    SplitBlock P[2];     // point encoding: half 0 = x, octaves 0 and 1, ONE; half 1 = octaves 2 and 3, zeros
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
            nan_sum += x[a];
        }
        if (half == 0) q[15] = 1.0f;
        vmax = fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fabsf(q[2]));
#pragma unroll
        for (int k = 0; k < 16; k += 2) GH_PUT(P[k >> 3], (k & 7) >> 1, q[k], q[k + 1])
    }

    SplitBlock C[NC];    // volume features: 5 encodings of this half's NCH channels, then ONE (half 0), then zeros
    const float* wo = w_out + half * (64 + 16 * TC);
    float s_cond = 0.0f;
    float f[NCH];        // the raw features: the chain rule at the end re-derives the encodings from them
    {
        float jl[3 * NCH];
#pragma unroll
        for (int j = 0; j <= MID; ++j) {     // whole levels of this half (j < MID); level MID is shared, two channels each
            const int l = j < MID ? (half ? MID + 1 + j : j) : MID;
            float4 jx, jy, jz;
            const float4 t = sample_volume4j((const float4*)vols.data[l], vols.dx[l], vols.dy[l], vols.dz[l], x, live, jx, jy, jz);
            float tv[4] = {t.x, t.y, t.z, t.w}, ax[4] = {jx.x, jx.y, jx.z, jx.w}, ay[4] = {jy.x, jy.y, jy.z, jy.w}, az[4] = {jz.x, jz.y, jz.z, jz.w};
            if (j == MID && half) {
                tv[0] = tv[2]; tv[1] = tv[3]; ax[0] = ax[2]; ax[1] = ax[3]; ay[0] = ay[2]; ay[1] = ay[3]; az[0] = az[2]; az[1] = az[3];
            }
#pragma unroll
            for (int c = 0; c < (j < MID ? 4 : 2); ++c) {
                const int ch = 4 * j + c;
                f[ch] = tv[c];
                jl[3 * ch] = ax[c]; jl[3 * ch + 1] = ay[c]; jl[3 * ch + 2] = az[c];
            }
        }
        if (lane == 0) gens_lock_slot(lock, lock_seen);
#pragma unroll
        for (int k = 0; k < 3 * NCH; ++k) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, jl[k]), stash, jl_lane + (uint32_t)k * 256u, 0, 0);
        float e[8 * NC];
#pragma unroll
        for (int k = 5 * NCH; k < 8 * NC; ++k) e[k] = (k == 5 * NCH && half == 0) ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            e[5 * j] = f[j];
            hw_sincos(f[j], e[5 * j + 1], e[5 * j + 2]);
            hw_sincos(2.0f * f[j], e[5 * j + 3], e[5 * j + 4]);
            vmax = fmaxf(vmax, fabsf(f[j]));
            nan_sum += f[j];
#pragma unroll
            for (int q = 0; q < 5; ++q) s_cond = __builtin_fmaf(e[5 * j + q], wo[64 + 5 * j + q], s_cond);       // layer 6 reads the conditioning features too
        }
#pragma unroll
        for (int k = 0; k < 8 * NC; k += 2) GH_PUT(C[k >> 3], (k & 7) >> 1, e[k], e[k + 1])
    }

    f32x16 acc[4];                 // the product being accumulated
    SplitBlock H[8];               // the operand of the running product: activations, then G_l (block 2 t + (r >> 3), slot r & 7 <- tile t, register r)
    f32x16 D3[4], D4[4];           // softplus' of layers 3, 4
    f32x16 gc[TC], gp;             // d sdf / d (this lane's conditioning slots), d sdf / d (its point-encoding slots), times g_scale
    u32x4 abuf[2][2 * GH_GT];      // A operands: this group's and the next one's
    int par = 0;                   // (compile-time after unrolling, like every index below)
    int dirty = 1;                 // stores are outstanding: the next chunk boundary waits for everything

    // chunk boundary: this wave's share of chunk C_ has landed (its later chunks may still be in flight), every wave says so, and the
    // ring slot everyone has just finished reading is refilled three chunks ahead
#define GH_BOUNDARY(C_)                                                                                  \
    {                                                                                                    \
        const int c__ = (C_);                                                                            \
        const int later__ = (c__ + GH_RING - 2 < NCHUNK - 1 ? c__ + GH_RING - 2 : NCHUNK - 1) - c__;     \
        if (dirty || later__ <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      \
        else if (later__ == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GH_LPW) : "memory");             \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GH_LPW) : "memory");                           \
        dirty = 0;                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                    \
        asm volatile("" ::: "memory");                                                                   \
        if (c__ + GH_RING - 1 < NCHUNK) stage(c__ + GH_RING - 1);                                        \
    }
    // pieces P0_ .. P0_ + CNT_ -> A register set SET_
#define GH_LOAD(SET_, P0_, CNT_)                                                                         \
    _Pragma("unroll") for (int j_ = 0; j_ < 2 * GH_GT; ++j_) if (j_ < (CNT_)) {                           \
        const int p_ = (P0_) + j_;                                                                       \
        if (p_ % GH_CH == 0) GH_BOUNDARY(p_ / GH_CH)                                                     \
        abuf[SET_][j_] = *((const u32x4*)(lds + ((p_ / GH_CH) % GH_RING) * (GH_CH * GH_PIECE) + (p_ % GH_CH) * GH_PIECE) + lane); \
    }
#define GH_MFMA(ACC_, A_, B_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (A_)), __builtin_bit_cast(f16x8, (B_)), ACC_, 0, 0, 0)
    // accumulator of output tile T_ of a reverse K block: hidden tiles, conditioning tiles, the point-encoding tile
#define GH_RACC(T_) (*((T_) < 4 ? &acc[(T_) < 4 ? (T_) : 0] : (T_) < 4 + TC ? &gc[(T_) < 4 + TC && (T_) >= 4 ? (T_) - 4 : 0] : &gp))
    // CNT_ K blocks of NT_ output tiles from piece P0_ on, B operands B_[0 .. CNT_); REV_: the accumulators of a reverse block.  The A
    // operands arrive in groups of GH_GT tiles (hi and lo pieces), one group ahead of the MFMAs that read them: two register sets of 24
    // instead of two whole K blocks of up to 56.  Within a group the accumulators alternate, so an MFMA never waits for the one before it.
#define GH_SEGMENT(P0_, CNT_, NT_, B_, REV_)                                                             \
    {                                                                                                    \
        constexpr int ng__ = ((NT_) + GH_GT - 1) / GH_GT;      /* (CNT_ and NT_ are constant expressions at every call) */ \
        GH_LOAD(par, (P0_), 2 * ((NT_) < GH_GT ? (NT_) : GH_GT))                                         \
        _Pragma("unroll") for (int q_ = 0; q_ < (CNT_) * ng__; ++q_) {                                   \
            const int i_ = q_ / ng__, t0_ = (q_ % ng__) * GH_GT, nt_ = (NT_) - t0_ < GH_GT ? (NT_) - t0_ : GH_GT; \
            if (q_ + 1 < (CNT_) * ng__) {                                                                \
                const int i2_ = (q_ + 1) / ng__, t2_ = ((q_ + 1) % ng__) * GH_GT, n2_ = (NT_) - t2_ < GH_GT ? (NT_) - t2_ : GH_GT; \
                GH_LOAD(par ^ 1, (P0_) + 2 * (NT_) * i2_ + 2 * t2_, 2 * n2_)                              \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            _Pragma("unroll") for (int t_ = 0; t_ < GH_GT; ++t_) if (t_ < nt_) { if (REV_) GH_MFMA(GH_RACC(t0_ + t_), abuf[par][2 * t_], (B_)[i_].h); else GH_MFMA(acc[(t0_ + t_) & 3], abuf[par][2 * t_], (B_)[i_].h); } \
            _Pragma("unroll") for (int t_ = 0; t_ < GH_GT; ++t_) if (t_ < nt_) { if (REV_) GH_MFMA(GH_RACC(t0_ + t_), abuf[par][2 * t_], (B_)[i_].l); else GH_MFMA(acc[(t0_ + t_) & 3], abuf[par][2 * t_], (B_)[i_].l); } \
            _Pragma("unroll") for (int t_ = 0; t_ < GH_GT; ++t_) if (t_ < nt_) { if (REV_) GH_MFMA(GH_RACC(t0_ + t_), abuf[par][2 * t_ + 1], (B_)[i_].h); else GH_MFMA(acc[(t0_ + t_) & 3], abuf[par][2 * t_ + 1], (B_)[i_].h); } \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            par ^= 1;                                                                                    \
        }                                                                                                \
    }
#define GH_ZERO()                                             \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)          \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[t_][r_] = 0.0f;
    // h~ = c softplus(a) = log2(1 + 2^t) and softplus' = 2^t / (1 + 2^t) of one accumulator tile (k6g_sdf_grad.hip: no compare masks, 16
    // independent instructions of one kind back to back)
#define GH_SOFTPLUS_TILE(T_, HT_, DT_)                                                                     \
    {                                                                                                      \
        f32x16 e__, u__, r__;                                                                              \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) e__[r_] = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f((T_)[r_], 126.0f, -3.0e38f)); \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) u__[r_] = 1.0f + e__[r_];                        \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) r__[r_] = __builtin_amdgcn_rcpf(u__[r_]);        \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) u__[r_] = __builtin_amdgcn_logf(u__[r_]);        \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) (DT_)[r_] = e__[r_] * r__[r_];                   \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) (HT_)[r_] = __builtin_amdgcn_fmed3f((T_)[r_], u__[r_], 3.0e38f); \
    }
    // the 16 values of tile T_ -> K blocks 2 T_, 2 T_ + 1 of H, the largest magnitude into MAX_
#define GH_SPLIT_TILE(T_, V_, MAX_)                                                                        \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; r_ += 2) {                                                 \
        MAX_ = fmaxf(MAX_, fmaxf(fabsf((V_)[r_]), fabsf((V_)[r_ + 1])));                                   \
        GH_PUT(H[2 * (T_) + (r_ >> 3)], (r_ & 7) >> 1, (V_)[r_], (V_)[r_ + 1])                              \
    }                                                                                                      \
    asm volatile("" : "+v"(MAX_));        /* (or the running maximum is re-associated into one pass at the end over 700 spilled values) */

#define GH_SPLIT_MAX(V_, MAX_)                                                                              \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; r_ += 2) MAX_ = fmaxf(MAX_, fmaxf(fabsf((V_)[r_]), fabsf((V_)[r_ + 1]))); \
    asm volatile("" : "+v"(MAX_));

    // ------------------------------------------------------------------ forward
    float s_val = s_cond;
    GH_ZERO();
    GH_SEGMENT(S::fwd(0), 2, 4, P, false);
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        if (l > 0) {
            GH_ZERO();
            GH_SEGMENT(S::fwd(l), NC, 4, C, false);
            if (l == 3) GH_SEGMENT(S::fwd(3) + 8 * NC, 2, 4, P, false);
            GH_SEGMENT(S::fwd(l) + 8 * (NC + (l == 3 ? 2 : 0)), 8, 4, H, false);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 h, d;
            if (l == 3) {
                GH_SOFTPLUS_TILE(acc[t], h, D3[t]);
            } else if (l == 4) {
                GH_SOFTPLUS_TILE(acc[t], h, D4[t]);
            } else {
                GH_SOFTPLUS_TILE(acc[t], h, d);
            }
            if (l < 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) DS[(l * 16 + 4 * t + q) * 64 + lane] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
            }
            if (l == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q)      // (the register's offset in the VECTOR offset: see k6g_sdf_grad.hip)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4v){d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]}), stash,
                                                           stash_lane + (uint32_t)(4 * t + q) * 1024u, 0, 0);
                dirty = 1;
            }
            if (l < 5) {
                GH_SPLIT_TILE(t, h, hmax);
            } else {    // layer 6 is one row: the value is a dot product, and G_5 = w_last * softplus' (times g_scale) starts the reverse pass
                f32x16 g;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float w = wo[16 * t + r];                      // w_last / c in this lane's accumulator order
                    s_val = __builtin_fmaf(h[r], w, s_val);
                    g[r] = (w * (GH_C * g_scale)) * d[r];
                }
                GH_SPLIT_TILE(t, g, gmax);
                GH_SPLIT_MAX(h, hmax);
            }
        }
    }
    s_val += __shfl_xor(s_val, 32, 64);

    // ------------------------------------------------------------------ reverse pass
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) gc[c][r] = (16 * c + r < 5 * NCH) ? wo[64 + 16 * c + r] * g_scale : 0.0f;      // layer 6 reads the features directly
#pragma unroll
    for (int r = 0; r < 16; ++r) gp[r] = 0.0f;
#pragma unroll
    for (int l = 5; l >= 1; --l) {
        GH_ZERO();
        if (l == 3) {      // softplus' of layer 2 comes back into the registers layer 4's has left (requested before the products that hide the trip)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(stash, stash_lane + (uint32_t)(4 * t + q) * 1024u, 0, 0));
                    D4[t][4 * q] = v[0]; D4[t][4 * q + 1] = v[1]; D4[t][4 * q + 2] = v[2]; D4[t][4 * q + 3] = v[3];
                }
        }
        if (l == 3) {
            GH_SEGMENT(S::rev(3), 8, 4 + TC + 1, H, true);
        } else if (l == 2) {
            GH_SEGMENT(S::rev(2), 7, 4 + TC, H, true);
        } else {
            GH_SEGMENT(S::rev(l), 8, 4 + TC, H, true);
        }
        // G_{l-1} = (W_l^T G_l) * softplus'(a_{l-1})
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 g;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 d;
                if (l == 5) d = make_float4(D4[t][4 * q], D4[t][4 * q + 1], D4[t][4 * q + 2], D4[t][4 * q + 3]);
                if (l == 4) d = make_float4(D3[t][4 * q], D3[t][4 * q + 1], D3[t][4 * q + 2], D3[t][4 * q + 3]);
                if (l == 3) d = make_float4(D4[t][4 * q], D4[t][4 * q + 1], D4[t][4 * q + 2], D4[t][4 * q + 3]);
                if (l == 2) d = DS[(16 + 4 * t + q) * 64 + lane];
                if (l == 1) d = DS[(4 * t + q) * 64 + lane];
                g[4 * q] = acc[t][4 * q] * d.x;
                g[4 * q + 1] = acc[t][4 * q + 1] * d.y;
                g[4 * q + 2] = acc[t][4 * q + 2] * d.z;
                g[4 * q + 3] = acc[t][4 * q + 3] * d.w;
            }
            GH_SPLIT_TILE(t, g, gmax);
        }
    }
    {   // layer 0 reads the point encoding only: three independent chains (hi hi, hi lo, lo hi), added at the end
        GH_ZERO();
        GH_LOAD(par, S::rev(0), 2)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i + 1 < 8) { GH_LOAD(par ^ 1, S::rev(0) + 2 * (i + 1), 2) }
            __builtin_amdgcn_sched_barrier(0);
            GH_MFMA(gp, abuf[par][0], H[i].h);
            GH_MFMA(acc[0], abuf[par][0], H[i].l);
            GH_MFMA(acc[1], abuf[par][1], H[i].h);
            __builtin_amdgcn_sched_barrier(0);
            par ^= 1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) gp[r] += acc[0][r] + acc[1][r];
    }
#undef GH_BOUNDARY
#undef GH_LOAD
#undef GH_MFMA
#undef GH_RACC
#undef GH_SEGMENT
#undef GH_ZERO
#undef GH_SOFTPLUS_TILE
#undef GH_SPLIT_TILE
#undef GH_SPLIT_MAX

    // ------------------------------------------------------------------ chain rule to x, lane-local (sdf_network.py:131-154)
    {
        float g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) asm volatile("" : "+v"(x[a]));      // (opaque: or the compiler keeps the prologue's encodings alive instead)
#pragma unroll
        for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(f[j]));
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = x[a] * scale;
            float s0, c0, s1, c1;
            hw_sincos(v * (half ? 4.0f : 1.0f), s0, c0);
            hw_sincos(v * (half ? 8.0f : 2.0f), s1, c1);
            // half 0 slots: x, sin / cos of octaves 0 and 1; half 1: sin / cos of octaves 2 and 3
            g[a] = half ? 4.0f * (gp[a] * c0 - gp[3 + a] * s0) + 8.0f * (gp[6 + a] * c1 - gp[9 + a] * s1)
                        : gp[a] + (gp[3 + a] * c0 - gp[6 + a] * s0) + 2.0f * (gp[9 + a] * c1 - gp[12 + a] * s1);
            g[a] *= scale;
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int q = 5 * j;
            float s1, c1, s2, c2;
            hw_sincos(f[j], s1, c1);
            hw_sincos(2.0f * f[j], s2, c2);
            const float df = gc[q >> 4][q & 15] + gc[(q + 1) >> 4][(q + 1) & 15] * c1 - gc[(q + 2) >> 4][(q + 2) & 15] * s1 +
                             2.0f * (gc[(q + 3) >> 4][(q + 3) & 15] * c2 - gc[(q + 4) >> 4][(q + 4) & 15] * s2);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float jl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(stash, jl_lane + (uint32_t)(3 * j + a) * 256u, 0, 0));
                g[a] = __builtin_fmaf(df, jl, g[a]);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] += __shfl_xor(g[a], 32, 64);
        if (half == 0 && live) {
            sdf_out[src] = (s_val + b_last) * inv_scale;
            grad_out[3 * src] = g[0] * (inv_scale * inv_g_scale);
            grad_out[3 * src + 1] = g[1] * (inv_scale * inv_g_scale);
            grad_out[3 * src + 2] = g[2] * (inv_scale * inv_g_scale);
        }
    }
    // out of the half range, or not a number (fmaxf drops NaNs: the sum of the inputs keeps them; the hidden units and the gradients can
    // only be NaN if those are): the caller re-runs the image in float32
    const bool big = !(fmaxf(fmaxf(vmax, hmax), gmax) < 3.0e4f) || nan_sum != nan_sum;
    if (__any(big) && lane == 0) atomicOr(overflow, 1);
    if (lane == 0) atomicExch(lock, 0u);
}

int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

extern "C" int gens_sdf_grad_f16_pieces(int n_levels) {
    return n_levels == 3 ? GradShapeH<3>::NCHUNK * GH_CH : n_levels == 5 ? GradShapeH<5>::NCHUNK * GH_CH : 0;
}

extern "C" int64_t gens_sdf_grad_f16_stash_bytes(void) { return (int64_t)GH_SLOTS * GH_SLOT_BYTES; }

extern "C" int gens_sdf_grad_f16_stash_reset(void* stash, void* stream) {
    GENS_CHECK_ARG(stash, GENS_EINVAL, "gens_sdf_grad_f16_stash_reset: null stash");
    if (hipMemsetAsync(stash, 0, (size_t)gens_sdf_grad_f16_stash_bytes(), (hipStream_t)stream) != hipSuccess)
        return gens_launch_status("gens_sdf_grad_f16_stash_reset");
    return 0;
}

extern "C" int gens_sdf_grad_f16(const float* const* vols_packed, const int* dims, int n_levels, const void* pieces, const float* w_out,
                                 float b_last, float scale, float g_scale, const float* pts, const int64_t* index, int64_t n,
                                 const int32_t* n_device, float* sdf_out, float* grad_out, void* stash, int* overflow_flag, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_grad_f16", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels == 3 || n_levels == 5, GENS_ELIMIT, "gens_sdf_grad_f16: built for 3 or 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(pieces && w_out && overflow_flag, GENS_EINVAL, "gens_sdf_grad_f16: null weight stream / flag");
    GENS_CHECK_ARG(((uintptr_t)pieces & 15) == 0, GENS_EINVAL, "gens_sdf_grad_f16: the weight stream must be 16-byte aligned");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && sdf_out && grad_out)), GENS_EINVAL, "gens_sdf_grad_f16: null pts / output");
    GENS_CHECK_ARG(scale != 0.0f && g_scale > 0.0f, GENS_EINVAL, "gens_sdf_grad_f16: scale must be non-zero, g_scale positive");
    GENS_CHECK_ARG(stash && ((uintptr_t)stash & 15) == 0, GENS_EINVAL,
                   "gens_sdf_grad_f16: null or misaligned stash (gens_sdf_grad_f16_stash_bytes() bytes, zeroed once)");
    if (n == 0) return 0;
    static GensLdsOptIn lds3, lds5;
    if (int e = n_levels == 3 ? gens_lds_opt_in(lds3, (const void*)sdf_grad_h_k<3>, GH_LDS_BYTES, "gens_sdf_grad_f16")
                              : gens_lds_opt_in(lds5, (const void*)sdf_grad_h_k<5>, GH_LDS_BYTES, "gens_sdf_grad_f16"))
        return e;
    const unsigned grid = gens_blocks(n, 32 * GH_WAVES);
    if (n_levels == 3)
        sdf_grad_h_k<3><<<grid, 64 * GH_WAVES, GH_LDS_BYTES, (hipStream_t)stream>>>(vs, (const char*)pieces, w_out, b_last, scale, 1.0f / scale, g_scale,
                                                                                   1.0f / g_scale, pts, index, n, n_device, sdf_out, grad_out,
                                                                                   (char*)stash, overflow_flag);
    else
        sdf_grad_h_k<5><<<grid, 64 * GH_WAVES, GH_LDS_BYTES, (hipStream_t)stream>>>(vs, (const char*)pieces, w_out, b_last, scale, 1.0f / scale, g_scale,
                                                                                   1.0f / g_scale, pts, index, n, n_device, sdf_out, grad_out,
                                                                                   (char*)stash, overflow_flag);
    return gens_launch_status("gens_sdf_grad_f16");
}
