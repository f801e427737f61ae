// K6g: value AND gradient d sdf / dx of the SDF network in exact float32 on the matrix cores, TRANSPOSED -- the dominant kernel of
// validation rendering (render_core's 128 samples per ray: /root/reference/models/modules/implicit_surface.py:179-191 ->
// sdf_network.py:98-154).  Same arithmetic as k6_sdfmlp.hip's gradient variant, the dataflow of k6t_sdf_value.hip:
//
//   * ONE wavefront owns 32 points and all 128 hidden units; the weights are the A operand, the activations the B operand, and the
//     activated accumulator registers of a layer ARE the B operands of the next one (the host packs the weights in that order).  The
//     reverse pass is the same chain on the transposed matrices: G_{l-1} = (W_l^T G_l) * softplus'(a_{l-1}), the accumulators of one
//     product multiplied by the stored derivative are the operands of the next.  No activation goes through LDS, there are no
//     barriers, waves are independent (k6_sdfmlp.hip: 24 barriers, 4 x 128 LDS stores and every A operand read from LDS per 32 points).
//   * softplus' of the six layers is 6 x 64 registers per lane: layer 5's is consumed on the spot (G_5 = w_last * softplus'),
//     layers 2..4 stay in registers, layers 0 and 1 wait in LDS (32 KB per wave) -- one wave per SIMD, 512 registers.
//   * the gradient with respect to the volume features (61 slots per lane half) and to the point encoding (15 slots) accumulate over
//     the layers in their own accumulator tiles, whose ROWS the host orders so that a lane ends up with the gradients of exactly the
//     slots it computed in the prologue (and whose trilinear Jacobians it keeps): the chain rule to x is lane-local, two lane halves
//     are added with one cross-lane move.
//   * every group of the weight stream is 4 float4 per lane and feeds 16 MFMAs (1024 cycles), requested one group ahead: 4 tiles x 4
//     feature pairs (products of hidden units), 2 tiles x 8 pairs (volume-feature gradients) or 1 tile x 16 pairs (point-encoding
//     gradients).  Pairs and tiles that carry nothing (layer 2 has 101 units) are not issued.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TG_SLOT_F4 (16 * 64 + 16)       // float4 per stash slot: 16 KB of softplus' + the lock word's 256 bytes
#ifndef TG_STASH_LOAD_POLICY
#define TG_STASH_LOAD_POLICY 0
#endif
#define TG_SLOTS 2048                   // (XCC 3 bits, SE 2, CU 4, SIMD 2)
#define TG_C 144.26950408889634f        // 100 / ln 2: hidden units travel as c * softplus (k6_sdfmlp.hip::softplus_t)

// value of one packed (X, Y, Z, 4) volume at x and its derivative with respect to x (zero padding, align_corners=True)
__device__ __forceinline__ float4 sample_volume4g(const float4* __restrict__ v, int Xd, int Yd, int Zd, const float x[3], bool live, float4& jx,
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

template <int NLEV>
struct GradShapeT {
    static constexpr int CF = 4 * NLEV;
    static constexpr int NCH = CF / 2;                 // channels per lane half
    static constexpr int NCS = 5 * NCH + 1;            // conditioning slots per half (5 encodings per channel + the constant one)
    static constexpr int GC = (NCS + 3) / 4;           // ... in groups of four pairs (forward)
    static constexpr int TC = ((5 * NCH + 15) / 16 + 1) / 2 * 2;   // accumulator tiles of the conditioning gradient (16 slots of a half per tile), in PAIRS (an odd count is padded with a tile of zero rows)
    static constexpr int GP = 4;
    static constexpr int NG_FWD = GP + 4 * (16 + GC) + (13 + GP + GC);
    // reverse pass: layers 5, 4, 3, 1: 16 groups of hidden products + 8 per pair of conditioning tiles; layer 2 (101 units): 13 + 7;
    // point encoding: 4 groups at layer 3 and at layer 0
    static constexpr int NG_BWD = 4 * (16 + 8 * (TC / 2)) + (13 + 7 * (TC / 2)) + 4 + 4;
};

template <int NLEV>
__global__ __launch_bounds__(64, 1) void sdf_grad_t_k(LevelSet vols, const float4* __restrict__ wstream, const float* w_out, float b_last, float scale,
                                                      float inv_scale, const float* __restrict__ pts, const int64_t* __restrict__ index,
                                                      int64_t n_max, const int32_t* __restrict__ n_dev, float* __restrict__ sdf_out,
                                                      float* __restrict__ grad_out, float4* __restrict__ stash_all) {
    typedef GradShapeT<NLEV> S;
    constexpr int NCH = S::NCH, NCS = S::NCS, GC = S::GC, GP = S::GP, TC = S::TC, MID = NLEV / 2, ODD = NLEV & 1;      // (the split of the levels between the lane halves: k6t_sdf_value.hip)
    static_assert(TC % 2 == 0, "the conditioning gradient is accumulated two tiles at a time");
    __shared__ float4 DS[2][16][64];        // softplus' of layers 0 and 1: [layer][accumulator register / 4][lane]
    __shared__ float JL[3 * NCH][64];       // trilinear Jacobians of this lane's channels
    const int lane = threadIdx.x;
    const int n_pt = lane & 31, half = lane >> 5;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    if (m0 >= n) return;
    // softplus' of layer 2 (16 KB per wave) waits in GLOBAL memory, in a slot that belongs to the SIMD this wave runs on: with 512 registers a
    // SIMD holds one wave of this kernel at a time, so 1 024 slots (16 MB, 2 MB per XCD: resident in its L2) serve the whole launch.  As the
    // compiler's private memory the same 16 KB went to whichever of a SIMD's wave slots the wave landed in, the footprint outgrew the L2, and
    // 4.3 GB per launch crossed the fabric for 0.15 GB of algorithmic bytes (PMC: FETCH_SIZE / WRITE_SIZE).  The slot's lock word makes the
    // one-wave-per-SIMD argument a performance assumption instead of a correctness one (it is always found free).
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));          // [5:4] SIMD, [11:8] CU, [15:13] SE (scripts/probe/hwid_probe.py)
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const uint32_t slot = ((((xcc_id & 7u) << 2 | ((hw_id >> 13) & 3u)) << 4 | ((hw_id >> 8) & 15u)) << 2) | ((hw_id >> 4) & 3u);
    float4* const stash_slot = stash_all + (size_t)slot * TG_SLOT_F4;
    uint32_t* const lock = (uint32_t*)(stash_slot + 16 * 64);
    if (lane == 0) gens_lock_slot(lock, atomicCAS(lock, 0u, 1u));
    // (a buffer descriptor: the slot's base in SGPRs, 16 lane in ONE VGPR, the register's offset as a scalar -- no vector address arithmetic)
    const __amdgpu_buffer_rsrc_t stash = __builtin_amdgcn_make_buffer_rsrc((void*)stash_slot, 0, 16 * 64 * 16, 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const uint32_t stash_lane = (uint32_t)lane * 16u;

    // the weight stream: a wave-uniform (scalar) base that advances by one group + 16 lane + an immediate per tile; three register
    // sets rotate: this group's weights and the next TWO groups' (in flight): with one wave per SIMD nothing else hides an L2 miss
    // Three levels: buffer loads -- descriptor + 16 lane in a VGPR + the group's byte offset in an SGPR + an immediate per tile: no vector
    // address arithmetic (with flat global loads hipcc spent ~540 VALU instructions and 300 wait states per tile on 64-bit addresses).
    // Five levels keep the scalar pointer + 16 lane form: with buffer loads the allocator spills the weight registers there.
    constexpr bool WBUF = NLEV <= 3;
    typedef float f32x4b __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wstream, 0, (int)((S::NG_FWD + S::NG_BWD + 2) * 4096), 0x00020000);
    const uint32_t wlane = (uint32_t)lane * 16u;
    uint32_t woff = 0;
    const float4* wp = wstream;
#define TG_WLOAD(DST, T_)                                                                                                     \
    if constexpr (WBUF) {                                                                                                     \
        const f32x4b v_ = __builtin_bit_cast(f32x4b, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + 1024u * (T_), woff, 0)); \
        DST = make_float4(v_[0], v_[1], v_[2], v_[3]);                                                                        \
    } else {                                                                                                                  \
        DST = wp[lane + 64 * (T_)];                                                                                           \
    }
#define TG_WNEXT()      \
    woff += 4096u;      \
    wp += 256;
    constexpr int NB = NLEV <= 3 ? 3 : 2;      // (five levels: the conditioning operands take the third set's registers; one group ahead. Three sets there: 11 instead of 36 spilled registers and TWICE the time, 626 against 301 ms per step)
    float4 wbuf[NB][4];
    int par = 0;
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { TG_WLOAD(wbuf[b][t], t) }
        TG_WNEXT()
    }

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
    const float* wo = w_out + half * (64 + 16 * TC);
    float s_cond = 0.0f;
    float f[NCH];          // the raw features: the chain rule at the end re-derives the encodings from them (46 registers less to carry)
    {
#pragma unroll
        for (int j = 0; j < MID + ODD; ++j) {     // whole levels of this half (j < MID); with an odd count level MID is shared, two channels each
            const int l = j < MID ? (half ? MID + ODD + j : j) : MID;
            float4 jx, jy, jz;
            const float4 t = sample_volume4g((const float4*)vols.data[l], vols.dx[l], vols.dy[l], vols.dz[l], x, live, jx, jy, jz);
            if (j < MID) {
                const float tv[4] = {t.x, t.y, t.z, t.w}, ax[4] = {jx.x, jx.y, jx.z, jx.w}, ay[4] = {jy.x, jy.y, jy.z, jy.w},
                            az[4] = {jz.x, jz.y, jz.z, jz.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f[4 * j + c] = tv[c];
                    JL[3 * (4 * j + c)][lane] = ax[c]; JL[3 * (4 * j + c) + 1][lane] = ay[c]; JL[3 * (4 * j + c) + 2][lane] = az[c];
                }
            } else if constexpr (ODD != 0) {
                f[4 * MID] = half ? t.z : t.x;
                f[4 * MID + 1] = half ? t.w : t.y;
                JL[3 * (4 * MID)][lane] = half ? jx.z : jx.x; JL[3 * (4 * MID) + 1][lane] = half ? jy.z : jy.x; JL[3 * (4 * MID) + 2][lane] = half ? jz.z : jz.x;
                JL[3 * (4 * MID + 1)][lane] = half ? jx.w : jx.y; JL[3 * (4 * MID + 1) + 1][lane] = half ? jy.w : jy.y;
                JL[3 * (4 * MID + 1) + 2][lane] = half ? jz.w : jz.y;
            }
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

    f32x16 acc[4], H[4];          // the product being accumulated / the operand of the running product (activations, then G_l)
    f32x16 D3[4], D4[4];          // softplus' of layers 3, 4
    // (softplus' of layer 2: the stash above -- with it in registers the allocator spilled ~200 values at places of its own choosing, and
    // every reload drained the weight prefetch, vmcnt counting in order)
    f32x16 gc[TC], gp;            // d sdf / d (this lane's conditioning slots), d sdf / d (its point-encoding slots)

    // one group = 4 float4 of weights per lane (requested NB - 1 groups ahead, into the register set just freed) and up to 16 MFMAs
#define TG_FETCH()                                                                                    \
    float4(&a_)[4] = wbuf[par];                                                                       \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { TG_WLOAD(wbuf[(par + NB - 1) % NB][t_], t_) }  \
    TG_WNEXT()                                                                                        \
    par = (par + 1) % NB;                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define TG_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), ACC, 0, 0, 0)
    // 4 tiles x 4 pairs (CNT: 3): acc[T] += A_T(pair i) * b_i
#define TG_GROUP4(CNT, b0, b1, b2, b3)                                                                \
    {                                                                                                 \
        TG_FETCH();                                                                                   \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) TG_MFMA(acc[t_], a_[t_].x, (b0));            \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) TG_MFMA(acc[t_], a_[t_].y, (b1));            \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) TG_MFMA(acc[t_], a_[t_].z, (b2));            \
        if ((CNT) == 4) {                                                                             \
            _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) TG_MFMA(acc[t_], a_[t_].w, (b3));        \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
    // 2 tiles x 8 pairs (HALF: the first 4 only): X0 += A_{0,j}(i) * B[4 j + i], X1 likewise; the stream holds [tile][j]
#define TG_GROUP2(HALF, X0, X1, B, R0)                                                                \
    {                                                                                                 \
        TG_FETCH();                                                                                   \
        TG_MFMA(X0, a_[0].x, (B)[(R0)]); TG_MFMA(X1, a_[2].x, (B)[(R0)]);                             \
        TG_MFMA(X0, a_[0].y, (B)[(R0) + 1]); TG_MFMA(X1, a_[2].y, (B)[(R0) + 1]);                     \
        TG_MFMA(X0, a_[0].z, (B)[(R0) + 2]); TG_MFMA(X1, a_[2].z, (B)[(R0) + 2]);                     \
        TG_MFMA(X0, a_[0].w, (B)[(R0) + 3]); TG_MFMA(X1, a_[2].w, (B)[(R0) + 3]);                     \
        if (!(HALF)) {                                                                                \
            TG_MFMA(X0, a_[1].x, (B)[(R0) + 4]); TG_MFMA(X1, a_[3].x, (B)[(R0) + 4]);                 \
            TG_MFMA(X0, a_[1].y, (B)[(R0) + 5]); TG_MFMA(X1, a_[3].y, (B)[(R0) + 5]);                 \
            TG_MFMA(X0, a_[1].z, (B)[(R0) + 6]); TG_MFMA(X1, a_[3].z, (B)[(R0) + 6]);                 \
            TG_MFMA(X0, a_[1].w, (B)[(R0) + 7]); TG_MFMA(X1, a_[3].w, (B)[(R0) + 7]);                 \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
    // 1 tile x 16 pairs: X += A_j(i) * B[4 j + i]
#define TG_GROUP1(X, B)                                                                               \
    {                                                                                                 \
        TG_FETCH();                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                            \
            TG_MFMA(X, a_[j_].x, (B)[4 * j_]); TG_MFMA(X, a_[j_].y, (B)[4 * j_ + 1]);                 \
            TG_MFMA(X, a_[j_].z, (B)[4 * j_ + 2]); TG_MFMA(X, a_[j_].w, (B)[4 * j_ + 3]);             \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define TG_HIDDEN(NT)                                                                        \
    _Pragma("unroll") for (int t2_ = 0; t2_ < (NT); ++t2_)                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)                                     \
            TG_GROUP4(4, H[t2_][4 * g_], H[t2_][4 * g_ + 1], H[t2_][4 * g_ + 2], H[t2_][4 * g_ + 3])
#define TG_COND()                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < GC; ++g_)                                        \
        TG_GROUP4((4 * g_ + 4 <= NCS) ? 4 : 3, cnd[4 * g_], cnd[4 * g_ + 1], cnd[4 * g_ + 2], cnd[4 * g_ + 3])
#define TG_PE()                                                                              \
    _Pragma("unroll") for (int g_ = 0; g_ < GP; ++g_)                                        \
        TG_GROUP4(g_ < GP - 1 ? 4 : 3, pe[4 * g_], pe[4 * g_ + 1], pe[4 * g_ + 2], pe[4 * g_ + 3])
#define TG_ZERO()                                             \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)          \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[t_][r_] = 0.0f;
    // h~ = c softplus(a) = log2(1 + 2^t) and softplus' = 2^t / (1 + 2^t) of one accumulator tile WITHOUT compare masks (16 scalar
    // register pairs per tile otherwise: the kernel spilled scalars): log2(1 + 2^t) >= t always, and where torch switches to the linear
    // branch (100 a > 20, t > 28.85) it already equals t to the last bit or two (2^-t < 2e-9), so max(t, .) IS the threshold; the clamp
    // keeps 2^t finite, so the derivative needs no select either (2^126 / (1 + 2^126) = 1).  Written PHASE-WISE over the 16 registers: with one wave per SIMD nothing hides the latency of a dependent exp -> add -> rcp / log chain,
    // so 16 independent instructions of one kind are issued back to back (hipcc keeps the source order of such a stream)
#define TG_SOFTPLUS_TILE(T_, HT_, DT_)                                                                     \
    {                                                                                                      \
        f32x16 e__, u__, r__;                                                                              \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) e__[r_] = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f((T_)[r_], 126.0f, -3.0e38f)); \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) u__[r_] = 1.0f + e__[r_];                        \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) r__[r_] = __builtin_amdgcn_rcpf(u__[r_]);        \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) u__[r_] = __builtin_amdgcn_logf(u__[r_]);        \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) (DT_)[r_] = e__[r_] * r__[r_];                   \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) (HT_)[r_] = __builtin_amdgcn_fmed3f((T_)[r_], u__[r_], 3.0e38f); \
    }

    static_assert(4 * GC >= NCS && 4 * GC - NCS <= 3, "the conditioning block ends with a partly filled group (its empty pairs carry zeros on both sides)");
    // ------------------------------------------------------------------ forward
    TG_ZERO();
    TG_PE();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 d;
        TG_SOFTPLUS_TILE(acc[t], H[t], d);
#pragma unroll
        for (int q = 0; q < 4; ++q) DS[0][4 * t + q][lane] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
    }
#pragma unroll
    for (int l = 1; l < 6; ++l) {
        TG_ZERO();
        if (l == 3) {     // x = cat([h[:101], pe]) / sqrt(2): features 104.. of the hidden state are not read (group (3, 0) holds 96..103)
            TG_HIDDEN(3);
            TG_GROUP4(4, H[3][0], H[3][1], H[3][2], H[3][3]);
            TG_PE();
        } else {
            TG_HIDDEN(4);
        }
        TG_COND();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (l == 1) {
                f32x16 d;
                TG_SOFTPLUS_TILE(acc[t], H[t], d);
#pragma unroll
                for (int q = 0; q < 4; ++q) DS[1][4 * t + q][lane] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
            }
            if (l == 2) {
                f32x16 d;
                TG_SOFTPLUS_TILE(acc[t], H[t], d);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    // (the register's offset in the VECTOR offset: a 16-byte buffer store with an SGPR offset whose data registers the next VALU
                    // instruction overwrites lost values here -- hipcc 7.0 leaves no wait state between the two on gfx950)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4s){d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]}), stash,
                                                           stash_lane + (uint32_t)(4 * t + q) * 1024u, 0, 0);
            }
            if (l == 3) TG_SOFTPLUS_TILE(acc[t], H[t], D3[t]);
            if (l == 4) TG_SOFTPLUS_TILE(acc[t], H[t], D4[t]);
            if (l == 5) {   // layer 6 is one row: the value is a dot product, and G_5 = w_last * softplus' starts the reverse pass
                f32x16 d;
                TG_SOFTPLUS_TILE(acc[t], H[t], d);
                acc[t] = d;
            }
        }
    }
    {
        asm volatile("" ::: "memory");             // keep the 64 output weights from being loaded (and spilled) ahead of the layers
        float s = s_cond;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float w = wo[16 * t + r];                      // w_last / c in this lane's accumulator order
                s = __builtin_fmaf(H[t][r], w, s);
                H[t][r] = (w * TG_C) * acc[t][r];
            }
            asm volatile("" ::: "memory");                           // one tile's 16 weights at a time
        }
        // not-a-number inputs must come out as not-a-number (the reference's layers propagate them; the max / median forms of the
        // activation drop them): a poison term 0 * (sum of the inputs), NaN iff one of them is NaN or infinite -- computed from values that
        // are live here anyway, here and again before the gradient is written
        {
            float acc_in = x[0] + x[1] + x[2];
#pragma unroll
            for (int j = 0; j < NCH; ++j) acc_in += f[j];
            s += 0.0f * acc_in;
        }
        s += __shfl_xor(s, 32, 64);
        if (half == 0 && live) sdf_out[src] = (s + b_last) * inv_scale;
    }

    // ------------------------------------------------------------------ reverse pass
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) gc[c][r] = (16 * c + r < 5 * NCH) ? wo[64 + 16 * c + r] : 0.0f;      // layer 6 reads the features directly
#pragma unroll
    for (int r = 0; r < 16; ++r) gp[r] = 0.0f;
#pragma unroll
    for (int l = 5; l >= 1; --l) {
        // hidden units of layer l feed the products: layer 2 has 101 of them (accumulator registers 4.. of tile 3 carry nothing)
        TG_ZERO();
        if (l == 2) {
            TG_HIDDEN(3);
            TG_GROUP4(4, H[3][0], H[3][1], H[3][2], H[3][3]);
        } else {
            TG_HIDDEN(4);
        }
#pragma unroll
        for (int c = 0; c < TC; c += 2) {
#pragma unroll
            for (int t = 0; t < (l == 2 ? 3 : 4); ++t) {
                TG_GROUP2(false, gc[c], gc[c + 1], H[t], 0);
                TG_GROUP2(false, gc[c], gc[c + 1], H[t], 8);
            }
            if (l == 2) TG_GROUP2(true, gc[c], gc[c + 1], H[3], 0);
        }
        if (l == 3) {
#pragma unroll
            for (int t = 0; t < 4; ++t) TG_GROUP1(gp, H[t]);
        }
        // G_{l-1} = (W_l^T G_l) * softplus'(a_{l-1})
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 d;
                if (l == 5) d = make_float4(D4[t][4 * q], D4[t][4 * q + 1], D4[t][4 * q + 2], D4[t][4 * q + 3]);
                if (l == 4) d = make_float4(D3[t][4 * q], D3[t][4 * q + 1], D3[t][4 * q + 2], D3[t][4 * q + 3]);
                if (l == 3) {
                    const f32x4s v = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(stash, stash_lane, (uint32_t)(4 * t + q) * 1024u, TG_STASH_LOAD_POLICY));
                    d = make_float4(v[0], v[1], v[2], v[3]);
                }
                if (l == 2) d = DS[1][4 * t + q][lane];
                if (l == 1) d = DS[0][4 * t + q][lane];
                H[t][4 * q] = acc[t][4 * q] * d.x;
                H[t][4 * q + 1] = acc[t][4 * q + 1] * d.y;
                H[t][4 * q + 2] = acc[t][4 * q + 2] * d.z;
                H[t][4 * q + 3] = acc[t][4 * q + 3] * d.w;
            }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) TG_GROUP1(gp, H[t]);        // layer 0 reads the point encoding only
#undef TG_FETCH
#undef TG_WLOAD
#undef TG_WNEXT
#undef TG_MFMA
#undef TG_GROUP4
#undef TG_GROUP2
#undef TG_GROUP1
#undef TG_HIDDEN
#undef TG_COND
#undef TG_PE
#undef TG_ZERO
#undef TG_SOFTPLUS_TILE

    // ------------------------------------------------------------------ chain rule to x, lane-local (sdf_network.py:131-154)
    {
        // the encodings are re-derived HERE, not carried through the reverse pass: the empty asm makes their inputs opaque, or the
        // compiler would merge these evaluations with the prologue's and keep 36 values alive (spilled) instead
        float g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) asm volatile("" : "+v"(x[a]));
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
        {
            float acc_in = x[0] + x[1] + x[2];
#pragma unroll
            for (int j = 0; j < NCH; ++j) acc_in += f[j];
#pragma unroll
            for (int a = 0; a < 3; ++a) g[a] += 0.0f * acc_in;
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
            for (int a = 0; a < 3; ++a) g[a] = __builtin_fmaf(df, JL[3 * j + a][lane], g[a]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] += __shfl_xor(g[a], 32, 64);
        if (half == 0 && live) {
            grad_out[3 * src] = g[0] * inv_scale;
            grad_out[3 * src + 1] = g[1] * inv_scale;
            grad_out[3 * src + 2] = g[2] * inv_scale;
        }
    }
    if (lane == 0) atomicExch(lock, 0u);                                            // (the stash was read long before: its values fed the reverse pass)
}

int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

extern "C" int gens_sdf_grad_groups(int n_levels) {
    switch (n_levels) {
        case 1: return GradShapeT<1>::NG_FWD + GradShapeT<1>::NG_BWD;
        case 2: return GradShapeT<2>::NG_FWD + GradShapeT<2>::NG_BWD;
        case 3: return GradShapeT<3>::NG_FWD + GradShapeT<3>::NG_BWD;
        case 4: return GradShapeT<4>::NG_FWD + GradShapeT<4>::NG_BWD;
        case 5: return GradShapeT<5>::NG_FWD + GradShapeT<5>::NG_BWD;
        default: return 0;
    }
}

extern "C" int64_t gens_sdf_grad_stash_bytes(void) { return (int64_t)TG_SLOTS * TG_SLOT_F4 * 16; }

extern "C" int gens_sdf_grad_stash_reset(void* stash, void* stream) {
    GENS_CHECK_ARG(stash, GENS_EINVAL, "gens_sdf_grad_stash_reset: null stash");
    if (hipMemsetAsync(stash, 0, (size_t)gens_sdf_grad_stash_bytes(), (hipStream_t)stream) != hipSuccess) return gens_launch_status("gens_sdf_grad_stash_reset");
    return 0;
}

extern "C" int gens_sdf_grad(const float* const* vols_packed, const int* dims, int n_levels, const float* wstream, const float* w_out,
                             float b_last, float scale, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device,
                             float* sdf_out, float* grad_out, void* stash, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_grad", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_sdf_grad: built for 1 to 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(wstream && w_out, GENS_EINVAL, "gens_sdf_grad: null weight stream");
    GENS_CHECK_ARG(((uintptr_t)wstream & 15) == 0, GENS_EINVAL, "gens_sdf_grad: the weight stream must be 16-byte aligned");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && sdf_out && grad_out)), GENS_EINVAL, "gens_sdf_grad: null pts / output");
    GENS_CHECK_ARG(scale != 0.0f, GENS_EINVAL, "gens_sdf_grad: scale must be non-zero");
    GENS_CHECK_ARG(stash && ((uintptr_t)stash & 15) == 0, GENS_EINVAL, "gens_sdf_grad: null or misaligned stash (gens_sdf_grad_stash_bytes() bytes, zeroed once)");
    if (n == 0) return 0;
    const unsigned grid = gens_blocks(n, 32);
#define TG_LAUNCH(NL) sdf_grad_t_k<NL><<<grid, 64, 0, (hipStream_t)stream>>>(vs, (const float4*)wstream, w_out, b_last, scale, 1.0f / scale, pts, index, n, n_device, sdf_out, grad_out, (float4*)stash)
    switch (n_levels) {
        case 1: TG_LAUNCH(1); break;
        case 2: TG_LAUNCH(2); break;
        case 3: TG_LAUNCH(3); break;
        case 4: TG_LAUNCH(4); break;
        default: TG_LAUNCH(5); break;
    }
#undef TG_LAUNCH
    return gens_launch_status("gens_sdf_grad");
}
