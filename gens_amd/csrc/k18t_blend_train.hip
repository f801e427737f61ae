// K18t: the BACKWARD launch of the colour branch of a training step, transposed (round 6).
//
// Replaces, like k18_blend_train.hip's backward (which stays for other view counts and as the cross-check), the autograd backward of
// lookup_feature + compute_angle (/root/reference/models/modules/projector.py:278-349) followed by BlendingNetwork.forward
// (models/modules/blending_network.py:69-118) as implicit_surface.py:196-199 calls them -- for TWO, THREE or FOUR source views, the counts the
// reference ships (confs/gens.conf:9,24, confs/gens_finetune.conf:15).
//
// k18_blend_train.hip walks a 32-row tile through eleven layers with a workgroup barrier around every product (36 barrier segments, two
// workgroups per CU: 21 % of the fp32-MFMA peak, bound by the latency chain of one tile).  Here NOTHING is shared between waves except the weights:
//
//   * one wave = 16 (point, view) rows and ALL of their forward, reverse and weight-gradient work; four waves per workgroup, one workgroup per CU,
//     persistent.  No barrier after the weights are in LDS.
//   * the eleven layers' raw weights and biases live in LDS for the whole launch (49 KB, loaded once per workgroup; pitches = 4 mod 8 floats, so
//     a forward operand -- 4 consecutive k of one output row -- is ONE conflict-free ds_read_b128).
//   * TRANSPOSED products on v_mfma_f32_16x16x4_f32: weights are the A operand, the 16 rows the N axis.  An activation vector lives in "x-layout":
//     register r of tile t of lane (kq = lane / 16, n = lane % 16) is channel 16 t + 4 kq + r of row n -- which is both what an accumulator tile
//     delivers and what the B operand of reduction slot r wants, so a layer's output IS the next layer's operand: no LDS round trip, no barrier
//     in the chain.  The reverse products read the same LDS weights column-wise.
//   * every activation is also parked in a wave-private LDS "store" in [channel][row] order (pitch 16 floats, XOR-swizzled): the weight-gradient
//     product dW_l += L_l^T [R_l | 1] reduces over ROWS, and in that order both operands -- 4 consecutive rows of one channel -- are again one
//     ds_read_b128 each.  A layer's cotangent L_l overwrites the activation it belongs to (dead by then).  The store also serves as the
//     re-layout device ([mean | var | x] packing, the gathered texels, H0's cotangent back to x-layout).
//   * the eleven [dW | db] blocks are 48 (three levels) ... 58 (five) accumulator tiles = 192 ... 232 registers of the wave, for the whole launch:
//     one wave per SIMD owns the SIMD's 512 registers.  The single-output rows (vis_fc.2's 33rd, vis_fc2.2, rgb_fc.4) are per-lane dot products
//     and per-lane partial sums, reduced over the 16 row lanes once, at the end.
//   * at the end the four waves' accumulators are added through LDS in wave order and the workgroup leaves ONE block of sums (layout of
//     gens_gemm_tn_batch's result, as gens_blend_train_bwd_acc does) and one partial of d loss / d |s|; blend_train_t_reduce_k adds the blocks
//     in a fixed order (deterministic).
//
// Per 16 rows at three levels: 152 forward + 148 reverse + 192 weight-gradient MFMAs (15.7 k matrix-pipe cycles) for 0.88 MFLOP of useful work.
#include "k4_common.h"
#include <type_traits>

// (a constexpr FUNCTION in an expression is a run-time call -- a recursive one a real s_swappc with the calling convention's spills around it --
// unless the context demands a constant)
#define KT_C(expr) (std::integral_constant<int, (expr)>::value)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define KT_WAVES 4
#define KT_THREADS (64 * KT_WAVES)
#define KT_NLAYER 11

namespace {

__host__ __device__ constexpr int kt_ev(int x) { return (x + 1) & ~1; }
__host__ __device__ constexpr int kt_in(int l, int F) { return l == 0 ? 4 : l == 1 ? 16 : l == 2 ? 3 * F : l == 3 ? 64 : l == 8 ? 37 : l == 9 ? 16 : l == 10 ? 8 : 32; }
__host__ __device__ constexpr int kt_out(int l, int F) { return l == 0 ? 16 : l == 1 ? F : l == 2 ? 64 : l == 5 ? 33 : l == 7 ? 1 : l == 8 ? 16 : l == 9 ? 8 : l == 10 ? 1 : 32; }
__host__ __device__ constexpr int kt_up(int x, int m) { return (x + m - 1) / m * m; }
// LDS pitch of a weight matrix with `in` columns: a multiple of 4 (16-byte rows) that is 4 mod 8 (eight consecutive rows of one k quad: eight bank groups)
__host__ __device__ constexpr int kt_pitch(int in) { return (kt_up(in, 4) % 8 == 4) ? kt_up(in, 4) : kt_up(in, 4) + 4; }
__host__ __device__ constexpr int kt_w_off(int l, int F) { return l == 0 ? 0 : kt_w_off(l - 1, F) + kt_out(l - 1, F) * kt_pitch(kt_in(l - 1, F)); }
__host__ __device__ constexpr int kt_b_off(int l, int F) { return l == 0 ? 0 : kt_b_off(l - 1, F) + kt_up(kt_out(l - 1, F), 16); }
#define KT_W_TAIL 256                         // zeros behind the last matrix: padded rows of the last tiles read on
__host__ __device__ constexpr int kt_w_total(int F) { return kt_w_off(KT_NLAYER, F) + KT_W_TAIL; }
__host__ __device__ constexpr int kt_b_total(int F) { return kt_b_off(KT_NLAYER, F) + 16; }
// result block of layer l (gens_gemm_tn_batch's layout: rows even(out), leading dimension even(in + 1))
__host__ __device__ constexpr int kt_cc_off(int l, int F) { return l == 0 ? 0 : kt_cc_off(l - 1, F) + kt_ev(kt_out(l - 1, F)) * kt_ev(kt_in(l - 1, F) + 1); }

// the wave-private store: first channel of every activation ([channel][16 rows]); multiples of 8
template <int F>
struct St {
    // H0 = [mean | var | x], every block padded to FP channels (the columns of base_fc.0's LDS copy are padded the same way): every base a multiple of 8
    static constexpr int FP = kt_up(F, 8), H0P = 3 * FP;
    static constexpr int T2 = 0, T1 = 8, D1 = 24, DFE = 40, H0 = DFE + FP, TB = H0 + H0P, H = TB + 64, TV = H + 32, HV = TV + 32, TU = HV + 32, H2 = TU + 32,
                         HX = H2 + 32, RD = HX + 8, CH = RD + 8;   // HX: [vis2, rd0, rd1, rd2, rd3, -, -, -] -- [H2 | HX] is rgb_fc.0's input; RD: [rd0 .. rd3, -] ray_dir_fc.0's
    static constexpr int RS = CH * 16;                       // per-row scalars behind the store: w, vis, rgb0, rgb1, rgb2 (16 floats each)
    static constexpr int FLOATS = RS + 160;                  // (+ slack: ray_dir_fc.0's operand tile starts 8 channels before the store's end)
    static_assert(H0P <= 96, "H0's cotangent is re-laid out through [TB | H]");
};
// accumulator tiles of the weight gradients: layer l has MT x NT tiles (vis_fc.2: rows 0..31 only; the single-output layers none)
__host__ __device__ constexpr int kt_mt(int l, int F) { return (l == 7 || l == 10) ? 0 : l == 5 ? 2 : kt_up(kt_out(l, F), 16) / 16; }
// (base_fc.0's input columns are the padded [mean | var | x] blocks, 3 FP of them; its bias column sits in the first pad slot, column F: F is odd, FP > F)
__host__ __device__ constexpr int kt_nt(int l, int F) { return (l == 7 || l == 10) ? 0 : l == 2 ? kt_up(3 * kt_up(F, 8), 16) / 16 : kt_up(kt_in(l, F) + 1, 16) / 16; }
__host__ __device__ constexpr int kt_acc_off(int l, int F) { return l == 0 ? 0 : kt_acc_off(l - 1, F) + kt_mt(l - 1, F) * kt_nt(l - 1, F); }

struct KtArgs {
    const float* w[23];           // raw nn.Linear parameters in gens_blend_train_*'s order
    MapSet fs;
    const float4* imgs;
    const float *w2c, *intr, *c2w;
    int nv;
    const float* pts;
    const int64_t* index;         // point i of the launch is pts[index[i]]; g_rgb lives at row index[i] (NULL: i)
    const int32_t* n_dev;         // only min(n, *n_dev) points exist (NULL: n)
    int64_t n;
    const float* g_rgb;           // (N, 3) cotangent of the colours
    float* g_feat;                // (n, S, F) cotangent of the looked-up [rgb | features] rows, or NULL
    float* s_part;                // (workgroups) partial sums of d loss / d |s|
    float* parts;                 // (workgroups, csz) the workgroups' sums of L^T [R | 1]
    int csz;
    float* dbg_r[KT_NLAYER];      // DUMP launches: the operand rows of every layer as gens_blend_train_bwd leaves them, in THIS kernel's row order
    float* dbg_l[KT_NLAYER];
};

// elu(x) = x for x > 0, exp(x) - 1 below: the MEDIAN of (x, exp(x) - 1, 0) -- exp(x) - 1 > x > 0 above, x < exp(x) - 1 < 0 below (k7_blend.hip::elu1).
// (A NaN input comes out as the median of two NaNs and 0, which is not NaN: the forward launch of the step carries the poison in the row mask; this
// launch's inputs are the step's own forward values -- finite, or the loss is NaN already.)
__device__ __forceinline__ float kt_elu(float x) { return __builtin_amdgcn_fmed3f(x, hw_exp(x) - 1.0f, 0.0f); }
// y * elu'(a) from the OUTPUT out = elu(a): elu' = 1 above 0, out + 1 below -> y + y * min(out, 0)
// (min(out, 0) as the median of (out, 0, -inf): one instruction, where fminf costs a canonicalising max besides)
__device__ __forceinline__ float kt_min0(float out) { return __builtin_amdgcn_fmed3f(out, 0.0f, -__builtin_inff()); }
__device__ __forceinline__ float kt_elu_dy(float y, float out) { return __builtin_fmaf(y, kt_min0(out), y); }
__device__ __forceinline__ float kt_elu_d(float out) { return kt_min0(out) + 1.0f; }                  // d elu / d a from the OUTPUT
__device__ __forceinline__ f32x4 kt_splat(float v) { return (f32x4){v, v, v, v}; }

// The store: channel rows of 16 floats.  Inside every aligned group of 8 channels bits 0 and 2 of the channel index trade places (so that the lanes
// of the two k quads that write together fall into different bank halves) and the row quad is XOR-ed with the channel's low two bits (so that eight
// consecutive channels read by eight lanes as 16-byte operands cover all banks): address(c, n) = 16 row(c) + 4 ((n >> 2) ^ (c & 3)) + (n & 3).
// Every access splits into a LANE part (a handful of registers, below) and a compile-time part that rides in the instruction's offset field.
__host__ __device__ constexpr int kt_row(int c) { return (c & ~5) | ((c >> 2) & 1) | ((c & 1) << 2); }
__host__ __device__ constexpr int kt_perm(int b) { return (b & 1) * 4 + (b & 2); }      // kt_row of the low bits b of an x-layout channel (without its k-quad bit)
struct KtLane {
    int xo[4];      // x-layout, register r: 128 (kq >> 1) + 16 (kq & 1) + 4 ((n >> 2) ^ r) + (n & 3)
    int co[4];      // a fixed channel with c & 3 = b, row n: 4 ((n >> 2) ^ b) + (n & 3)
    int op;         // operand (rows 4 kq .. 4 kq + 3 of channel base + i): 16 row(i) + 4 (kq ^ (i & 3))
};
__device__ __forceinline__ KtLane kt_lane(int n, int kq) {
    KtLane L;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        L.co[r] = ((((n >> 2) ^ r) & 3) << 2) + (n & 3);
        L.xo[r] = 128 * (kq >> 1) + 16 * (kq & 1) + L.co[r];
    }
    L.op = 16 * ((n & ~5) | ((n >> 2) & 1) | ((n & 1) << 2)) + (((kq ^ n) & 3) << 2);
    return L;
}
// a fixed channel c of row n
#define KT_AT(c) S[L.co[(c) & 3] + 16 * kt_row(c)]
// rows 4 kq .. 4 kq + 3 of channel base + i (base a multiple of 8): an operand of the weight-gradient product
__device__ __forceinline__ f32x4 kt_op(const float* S, const KtLane& L, int base) { return *(const f32x4*)(S + L.op + 16 * base); }
// x-layout tile t of an activation with `limit` channels at store channel `base` (a multiple of 8)
__device__ __forceinline__ void kt_put(float* S, const KtLane& L, int base, int t, f32x4 v, int kq, int limit) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (16 * t + 4 * kq + r < limit) S[L.xo[r] + 16 * (base + 16 * t + kt_perm(r))] = v[r];
}
__device__ __forceinline__ f32x4 kt_get(const float* S, const KtLane& L, int base, int t, int kq, int limit) {
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (16 * t + 4 * kq + r < limit) ? S[L.xo[r] + 16 * (base + 16 * t + kt_perm(r))] : 0.0f;
    return v;
}
// channels >= limit of tile t -> 0
__device__ __forceinline__ f32x4 kt_mask(f32x4 v, int t, int kq, int limit) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (16 * t + 4 * kq + r < limit) ? v[r] : 0.0f;
    return v;
}

// sums / extrema over the G adjacent lanes of a point (= its views), in every lane.  (update_dpp with bound_ctrl and no `old` value: the compiler
// folds the lane permutation into the add / min / max as its DPP operand -- with `old` = the value itself it emitted a copy, the hazard nop and a
// separate v_mov_dpp per butterfly step.)
template <int CTRL>
__device__ __forceinline__ float kt_dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
template <int G>
__device__ __forceinline__ float kt_gsum(float v) {
    v += kt_dpp<0xB1>(v);                                 // quad_perm:[1,0,3,2]
    if (G == 4) v += kt_dpp<0x4E>(v);                     // quad_perm:[2,3,0,1]
    return v;
}
template <int G>
__device__ __forceinline__ float kt_gmin(float v) {
    v = fminf(v, kt_dpp<0xB1>(v));
    if (G == 4) v = fminf(v, kt_dpp<0x4E>(v));
    return v;
}
template <int G>
__device__ __forceinline__ float kt_gmax(float v) {
    v = fmaxf(v, kt_dpp<0xB1>(v));
    if (G == 4) v = fmaxf(v, kt_dpp<0x4E>(v));
    return v;
}
// Across the four lane groups kq of a row (lanes n, n + 16, n + 32, n + 48): gfx950's row / half swaps on the vector ALU -- a ds_bpermute pair
// costs two LDS round trips, and one wave per SIMD has nothing to run under them.  v_permlane16_swap exchanges the odd 16-lane rows of its first
// operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second.  (Inline assembly:
// the builtin with the same value in both operands was compiled to x + x on this toolchain, as if its two results were one; the s_nop is the
// VALU-write -> permlane-read hazard the compiler pads for its own instructions.)
__device__ __forceinline__ void kt_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void kt_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float kt_qsum(float v) {       // sum over the four lane groups kq of a row, in every lane
    float a = v, b = v;
    kt_swap16(a, b);                                      // a = [v0 v0 v2 v2], b = [v1 v1 v3 v3] (rows of 16 lanes)
    float s = a + b;
    a = s; b = s;
    kt_swap32(a, b);                                      // a = [s01 s01 s01 s01], b = [s23 s23 s23 s23]
    return a + b;
}
__device__ __forceinline__ float kt_from_q0(float v) {    // the value lane group 0 holds for this row, in every lane
    float a = v, b = v;
    kt_swap16(a, b);                                      // a = [v0 v0 v2 v2]
    b = a;
    kt_swap32(a, b);                                      // a = [v0 v0 v0 v0]
    return a;
}
__device__ __forceinline__ float kt_rowsum16(float v) {   // sum over the 16 lanes of a lane group (a DPP row), valid in its LAST lane
    v += dpp_move<0x111, 0xF>(0.0f, v);                   // row_shr:1
    v += dpp_move<0x112, 0xF>(0.0f, v);                   // row_shr:2
    v += dpp_move<0x114, 0xF>(0.0f, v);                   // row_shr:4
    v += dpp_move<0x118, 0xF>(0.0f, v);                   // row_shr:8
    return v;
}
__device__ __forceinline__ float kt_dot(const f32x4& a, const f32x4& b) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }

// The operands of the NEXT group of products are asked for before the current group is multiplied (hipcc otherwise sinks every ds_read next to its
// use -- one exposed LDS round trip per pair of products, and with one wave per SIMD nothing else runs under it; `sched_barrier(0)` pins the order,
// as k6_sdfmlp.hip does for its weight stream).
#define KT_PIN() __builtin_amdgcn_sched_barrier(0)

// Y (MT tiles of the layer's outputs) = bias + W X: the weights' rows are the A operand (one ds_read_b128 = the four reduction slots of a tile)
template <int KT, int MT>
__device__ __forceinline__ void kt_fwd(const float* W, int P, const float* bias, const f32x4 (&X)[KT], f32x4 (&Y)[MT], int i, int kq) {
    f32x4 a[2][MT];
#pragma unroll
    for (int to = 0; to < MT; ++to) a[0][to] = *(const f32x4*)(W + (16 * to + i) * P + 4 * kq);
#pragma unroll
    for (int to = 0; to < MT; ++to) Y[to] = *(const f32x4*)(bias + 16 * to + 4 * kq);
#pragma unroll
    for (int ti = 0; ti < KT; ++ti) {
        if (ti + 1 < KT) {
#pragma unroll
            for (int to = 0; to < MT; ++to) a[(ti + 1) & 1][to] = *(const f32x4*)(W + (16 * to + i) * P + 16 * (ti + 1) + 4 * kq);
        }
        KT_PIN();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int to = 0; to < MT; ++to) Y[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti & 1][to][r], X[ti][r], Y[to], 0, 0, 0);
        KT_PIN();
    }
}
// Y (MT tiles of the layer's INPUTS) = W^T G over KT tiles of its outputs: column reads of the same LDS matrix.  Channels of Y beyond the layer's
// input count are garbage (the caller masks them); channels of G beyond its output count must be zero.
template <int KT, int MT>
__device__ __forceinline__ void kt_rev(const float* W, int P, const f32x4 (&G)[KT], f32x4 (&Y)[MT], int i, int kq) {
    float a[2][4][MT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ti = 0; ti < MT; ++ti) a[0][r][ti] = W[(4 * kq + r) * P + i + 16 * ti];
#pragma unroll
    for (int ti = 0; ti < MT; ++ti) Y[ti] = kt_splat(0.0f);
#pragma unroll
    for (int to = 0; to < KT; ++to) {
        if (to + 1 < KT) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ti = 0; ti < MT; ++ti) a[(to + 1) & 1][r][ti] = W[(16 * (to + 1) + 4 * kq + r) * P + i + 16 * ti];
        }
        KT_PIN();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ti = 0; ti < MT; ++ti) Y[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[to & 1][r][ti], G[to][r], Y[ti], 0, 0, 0);
        KT_PIN();
    }
}
// block += L^T [R | 1] over this wave's 16 rows: L = MT tiles of channels from store channel cL, R = `in` channels from cR (times the per-row
// scalars `scale` when given: the two layers whose input is h * weight / x * visibility); D[m = out][n = in], reduction slot j of lane group kq = row 4 kq + j
// `one`: the column that carries the constant 1 (the bias gradient): `in` -- behind the inputs -- except for base_fc.0
template <int OFF, int MT, int NT, int NTILES, int CH>
__device__ __forceinline__ void kt_dw(f32x4 (&wacc)[NTILES], const float* S, const KtLane& L, int cL, int cR, int in, const float* scale, int i, int kq, int one = -1) {
    if (one < 0) one = in;
    f32x4 a[MT], b[2];
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) a[mo] = kt_op(S, L, cL + 16 * mo);
    f32x4 sc = kt_splat(1.0f);
    if (scale) sc = *(const f32x4*)(scale + 4 * kq);
    // (a tile that lies behind the inputs altogether -- the bias column of a 32- or 64-wide layer -- is not read at all)
    b[0] = kt_op(S, L, cR);
#pragma unroll
    for (int no = 0; no < NT; ++no) {
        if (no + 1 < NT && 16 * (no + 1) < in) b[(no + 1) & 1] = kt_op(S, L, cR + 16 * (no + 1));
        KT_PIN();
        const int col = 16 * no + i;
        f32x4 bb = 16 * no < in ? b[no & 1] : kt_splat(0.0f);
        if (scale) bb *= sc;
        bb = col == one ? kt_splat(1.0f) : (col < in ? bb : kt_splat(0.0f));
#pragma unroll
        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
            for (int j = 0; j < 4; ++j) wacc[OFF + mo * NT + no] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mo][j], bb[j], wacc[OFF + mo * NT + no], 0, 0, 0);
        KT_PIN();
    }
}

// DUMP launches: the operand rows [R | 1 | 0] / L of layer l as k18_blend_train.hip's backward stores them, rows in this kernel's tile order
template <int l, int F, int MT, int NT, int CH>
__device__ __forceinline__ void kt_dump(const __attribute__((address_space(4))) KtArgs& A, const float* S, const KtLane& L, int cL, int cR, int in, const float* scale, int64_t row0, int i, int kq) {
    constexpr int out = l == 5 ? 32 : kt_out(l, F);
    constexpr int lw = kt_ev(kt_out(l, F)), rw = kt_ev(kt_in(l, F) + 1);
    for (int mo = 0; mo < MT; ++mo) {
        const int c = 16 * mo + i;
        const f32x4 a = kt_op(S, L, cL + 16 * mo);
        if (c < out)
            for (int j = 0; j < 4; ++j) A.dbg_l[l][(row0 + 4 * kq + j) * lw + c] = a[j];
    }
    f32x4 sc = kt_splat(1.0f);
    if (scale) sc = *(const f32x4*)(scale + 4 * kq);
    for (int no = 0; no < NT; ++no) {
        const int col = 16 * no + i;
        f32x4 b = kt_op(S, L, cR + 16 * no);
        if (scale) b *= sc;
        b = col < in ? b : kt_splat(col == in ? 1.0f : 0.0f);
        int dst = col;
        if (l == 2) {                                       // padded [mean | var | x] blocks -> the reference's 3 F columns; the 1 sits in column F
            constexpr int FP = kt_up(F, 8);
            const int blk = col / FP, c = col - blk * FP;
            dst = col == F ? 3 * F : ((blk < 3 && c < F) ? blk * F + c : rw);
            if (col == F) b = kt_splat(1.0f);
            if (col == 0) for (int j = 0; j < 4; ++j) if (3 * F + 1 < rw) A.dbg_r[l][(row0 + 4 * kq + j) * rw + 3 * F + 1] = 0.0f;
        }
        if (dst < rw)
            for (int j = 0; j < 4; ++j) A.dbg_r[l][(row0 + 4 * kq + j) * rw + dst] = b[j];
    }
}

// the LDS copy of a weight matrix: rows of pitch kt_pitch(IN), zeros behind column IN
template <int OUT, int IN>
__device__ __forceinline__ void kt_fill(float* dst, const float* __restrict__ src, int tid) {
    constexpr int P = kt_pitch(IN);
    for (int e = tid; e < OUT * P; e += KT_THREADS) {
        const int o = e / P, k = e - o * P;
        dst[e] = k < IN ? src[o * IN + k] : 0.0f;
    }
}
// base_fc.0: its 3 F input columns [mean | var | x] spread to three blocks of FP (the store's H0 layout)
template <int F>
__device__ __forceinline__ void kt_fill_b1(float* dst, const float* __restrict__ src, int tid) {
    constexpr int FP = kt_up(F, 8), P = kt_pitch(3 * FP);
    for (int e = tid; e < 64 * P; e += KT_THREADS) {
        const int o = e / P, k = e - o * P, blk = k / FP, c = k - blk * FP;
        dst[e] = (blk < 3 && c < F) ? src[o * 3 * F + blk * F + c] : 0.0f;
    }
}
// LDS offsets of the weight matrices (base_fc.0 with its padded pitch)
__host__ __device__ constexpr int kt_lp(int l, int F) { return l == 2 ? kt_pitch(3 * kt_up(F, 8)) : kt_pitch(kt_in(l, F)); }
__host__ __device__ constexpr int kt_lw(int l, int F) { return l == 0 ? 0 : kt_lw(l - 1, F) + kt_out(l - 1, F) * kt_lp(l - 1, F); }
__host__ __device__ constexpr int kt_lw_total(int F) { return kt_lw(KT_NLAYER, F) + KT_W_TAIL; }

typedef const __attribute__((address_space(4))) KtArgs* KtArgsPtr;

// Cycle stamps of one wave's phases (build with -DGENS_K18T_STAMPS: make -C gens_amd/csrc stamps; scripts/probe/k18t_stamps_probe.py reads them):
// never in the shipped library.  A stamp waits for everything outstanding first, so that a phase is charged with its own latencies.
#ifdef GENS_K18T_STAMPS
__device__ unsigned long long k18t_stamps[4][64];
#define KT_STAMP()                                                                                           \
    do {                                                                                                     \
        if (blockIdx.x == 7 && tile_no == 3) {                                                               \
            if (GENS_K18T_STAMPS == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           \
            if (lane == 0 && n_stamp < 64) k18t_stamps[wave][n_stamp++] = __builtin_readcyclecounter();      \
        }                                                                                                    \
    } while (0)
#else
#define KT_STAMP() do { } while (0)
#endif

// S source views as G lanes per point: G = 4 for three (one dead lane) or four views, G = 2 for two
template <int NLEV, int G, bool DUMP>
__global__ __launch_bounds__(KT_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void blend_train_t_k(KtArgs A_) {
    constexpr int F = 3 + 4 * NLEV, FP = kt_up(F, 8), XT = (F + 15) / 16, HT = (3 * FP + 15) / 16;
    typedef St<F> ST;
    constexpr int NTILES = kt_acc_off(KT_NLAYER, F);
    constexpr int PPW = 16 / G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const WL = smem;                                   // weights
    float* const BL = WL + KT_C(kt_lw_total(F));              // biases (padded to 16 with zeros)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const S = BL + KT_C(kt_b_total(F)) + wave * ST::FLOATS;  // this wave's store
    float* const RSW = S + ST::RS, * const RSV = RSW + 16, * const RC0 = RSV + 16, * const RC1 = RC0 + 16, * const RC2 = RC1 + 16;
    const int n = lane & 15, kq = lane >> 4, i = n;
    const KtLane L = kt_lane(n, kq);
    // The arguments are read IN the kernarg segment (constant address space, scalar loads where they are needed): as by-value parameters the 23
    // weight pointers and the map geometry stay live in ~100 scalar registers across the tile loop and spill to vector lanes (k18_blend_train.hip).
    KtArgsPtr kp = (KtArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();

    // ---------------------------------------------------------------- once per workgroup: weights, biases, a clean store
#define KT_FILL(l) kt_fill<kt_out(l, F), kt_in(l, F)>(WL + KT_C(kt_lw(l, F)), kp->w[2 * (l)], tid);
    KT_FILL(0) KT_FILL(1)
    kt_fill_b1<F>(WL + KT_C(kt_lw(2, F)), kp->w[4], tid);
    KT_FILL(3) KT_FILL(4) KT_FILL(5) KT_FILL(6) KT_FILL(7) KT_FILL(8) KT_FILL(9) KT_FILL(10)
#undef KT_FILL
    for (int e = tid; e < KT_W_TAIL; e += KT_THREADS) WL[KT_C(kt_lw(KT_NLAYER, F)) + e] = 0.0f;
    for (int e = tid; e < KT_C(kt_b_total(F)); e += KT_THREADS) BL[e] = 0.0f;
    for (int e = lane; e < ST::FLOATS; e += 64) S[e] = 0.0f;
    __syncthreads();
#define KT_BIAS(l) for (int e = tid; e < KT_C(kt_out(l, F)); e += KT_THREADS) BL[KT_C(kt_b_off(l, F)) + e] = kp->w[2 * (l) + 1][e];
    KT_BIAS(0) KT_BIAS(1) KT_BIAS(2) KT_BIAS(3) KT_BIAS(4) KT_BIAS(5) KT_BIAS(6) KT_BIAS(7) KT_BIAS(8) KT_BIAS(9) KT_BIAS(10)
#undef KT_BIAS
    __syncthreads();
#define KT_W(l) (WL + KT_C(kt_lw(l, F)))
#define KT_P(l) KT_C(kt_lp(l, F))
#define KT_B(l) (BL + KT_C(kt_b_off(l, F)))

    const int Sv = kp->nv - 1;
    const int64_t npts = kp->n_dev ? min(kp->n, (int64_t)kp->n_dev[0]) : kp->n;
    const int64_t n_tiles = (npts + PPW - 1) / PPW;
    const float s_abs = fabsf(kp->w[22][0]);

    f32x4 wacc[NTILES];
#pragma unroll
    for (int k = 0; k < NTILES; ++k) wacc[k] = kt_splat(0.0f);
    f32x4 sp_v2[2] = {kt_splat(0.0f), kt_splat(0.0f)}, sp_u2[2] = {kt_splat(0.0f), kt_splat(0.0f)}, sp_r3 = kt_splat(0.0f);
    float sp_v2b = 0.0f, sp_u2b = 0.0f, sp_r3b = 0.0f, s_acc = 0.0f;

#ifdef GENS_K18T_STAMPS
    int n_stamp = 0, tile_no = -1;
#endif
    // ---------------------------------------------------------------- the rows of a tile; the NEXT tile's points are loaded while this one is worked on
    const int pl = n / G, v = n % G;
    const bool dead = v >= Sv;                                        // (three views: the fourth lane of a point carries no view)
    const int sv = dead ? Sv : v + 1;
    const int64_t stride = (int64_t)gridDim.x * KT_WAVES;
    // dense row of this lane's point in tile t, -1 if it has none (index -> coordinates is a chain of two loads: the index runs two tiles ahead)
#define KT_SRC(t) ((!dead && (t) < n_tiles && (t) * PPW + pl < npts) ? (kp->index ? kp->index[(t) * PPW + pl] : (t) * PPW + pl) : (int64_t)-1)
    struct KtPoint { float x, y, z, g0, g1, g2; };
    auto load_point = [&](int64_t src) {
        KtPoint P = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (src >= 0) {
            const float* q = kp->pts + 3 * src;
            const float* g = kp->g_rgb + 3 * src;
            P.x = q[0]; P.y = q[1]; P.z = q[2];
            P.g0 = g[0]; P.g1 = g[1]; P.g2 = g[2];
        }
        return P;
    };
    // ---------------------------------------------------------------- look-up (K4), ISSUE half: every lane projects its row; lane group l & 3 reads level l,
    // lane group 1 the image besides.  The texel loads of tile t + 1 are issued in the middle of tile t's reverse pass and used one phase into tile
    // t + 1: a global load takes 2 - 3 us here, 5 000 - 7 000 cycles of a 45 000-cycle tile when it is waited for.
    // slot A: the lane group's own level 0..3; slot B: level 4 (group 0) / the image (group 1); weights zero where a tap is outside the map
// The look-up of one tile, WITHOUT a branch: lane (kq, n) reads feature level kq of row n into qa (levels 0 .. 3) and, kq = 0, level 4 / kq = 1, the image
// into qb.  Every lane projects into ITS level only -- the level's constants picked per lane out of the scalar table -- and the mask term "inside every
// level" is the four lanes' terms combined by one lane-group sum.  (The first form walked the levels one after the other with every lane projecting into
// all of them and the quarter of the lanes that owns a level reading under an exec mask: six divergent regions of ~60 instructions in the middle of
// the reverse pass, 42 of the launch's 347 us.)  Lanes without a level (kq >= NLEV; kq >= 2 for qb) read level 0 at weight 0.
#define KT_SEL4(a0, a1, a2, a3) (kqg_ == 0 ? (a0) : kqg_ == 1 ? (a1) : kqg_ == 2 ? (a2) : (a3))
#define KT_LV(field, l) (kp->fs.field[(l) < NLEV ? (l) : 0])
#define KT_TAPS(q_, w_out_, p_, h_, w_, base_, on_)                                                                                \
    {                                                                                                                              \
        const Taps2 t_ = bilinear_taps((p_).ix, (p_).iy, h_, w_);                                                                  \
        const int x0_ = min(max(t_.x0, 0), (w_) - 1), x1_ = min(max(t_.x0 + 1, 0), (w_) - 1);                                      \
        const int y0_ = min(max(t_.y0, 0), (h_) - 1), y1_ = min(max(t_.y0 + 1, 0), (h_) - 1);                                      \
        const f32x4* r0_ = (base_) + ((int64_t)sv * (h_) + y0_) * (w_), * r1_ = (base_) + ((int64_t)sv * (h_) + y1_) * (w_);       \
        q_[0] = r0_[x0_]; q_[1] = r0_[x1_]; q_[2] = r1_[x0_]; q_[3] = r1_[x1_];                                                    \
        w_out_[0] = (on_) && t_.ok00 ? t_.w00 : 0.0f; w_out_[1] = (on_) && t_.ok01 ? t_.w01 : 0.0f;                                \
        w_out_[2] = (on_) && t_.ok10 ? t_.w10 : 0.0f; w_out_[3] = (on_) && t_.ok11 ? t_.w11 : 0.0f;                                \
    }
#define KT_GATHER(P_, LIVE_, qa_, qb_, wa_, wb_, inside_)                                                                          \
    {                                                                                                                              \
        /* (the level constants a lane picks do not change from tile to tile: an opaque copy of kq keeps them from being hoisted out of the    \
            tile loop, where they would hold twenty of the wave's registers for the whole launch) */                                      \
        int kqg_ = kq;                                                                                                             \
        asm volatile("" : "+v"(kqg_));                                                                                             \
        const SrcBase pb_ = project_src_base(kp->w2c + 16 * sv, kp->intr + 16 * sv, (P_).x, (P_).y, (P_).z);                      \
        const int ha_ = KT_SEL4(KT_LV(h, 0), KT_LV(h, 1), KT_LV(h, 2), KT_LV(h, 3)), wd_a_ = KT_SEL4(KT_LV(w, 0), KT_LV(w, 1), KT_LV(w, 2), KT_LV(w, 3)); \
        const float cwa_ = KT_SEL4(KT_LV(cw, 0), KT_LV(cw, 1), KT_LV(cw, 2), KT_LV(cw, 3)), cha_ = KT_SEL4(KT_LV(ch, 0), KT_LV(ch, 1), KT_LV(ch, 2), KT_LV(ch, 3)); \
        const float rcwa_ = KT_SEL4(KT_LV(rcw, 0), KT_LV(rcw, 1), KT_LV(rcw, 2), KT_LV(rcw, 3)), rcha_ = KT_SEL4(KT_LV(rch, 0), KT_LV(rch, 1), KT_LV(rch, 2), KT_LV(rch, 3)); \
        const f32x4* da_ = (const f32x4*)KT_SEL4(KT_LV(data, 0), KT_LV(data, 1), KT_LV(data, 2), KT_LV(data, 3));                  \
        const bool has_a_ = kqg_ < NLEV;                                                                                           \
        const SrcProj pa_ = project_src_level(pb_, KT_SEL4(1.0f, 0.5f, 0.25f, 0.125f), ha_, wd_a_, cwa_, cha_, rcwa_, rcha_);      \
        KT_TAPS(qa_, wa_, pa_, ha_, wd_a_, da_, has_a_ && (LIVE_))                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        const bool b4_ = NLEV > 4 && kqg_ == 0;                                                                                    \
        const int hb_ = b4_ ? KT_LV(h, 4) : KT_LV(h, 0), wd_b_ = b4_ ? KT_LV(w, 4) : KT_LV(w, 0);                                   \
        const float cwb_ = b4_ ? KT_LV(cw, 4) : KT_LV(cw, 0), chb_ = b4_ ? KT_LV(ch, 4) : KT_LV(ch, 0);                             \
        const float rcwb_ = b4_ ? KT_LV(rcw, 4) : KT_LV(rcw, 0), rchb_ = b4_ ? KT_LV(rch, 4) : KT_LV(rch, 0);                       \
        const f32x4* db_ = b4_ ? (const f32x4*)KT_LV(data, 4) : (const f32x4*)kp->imgs;                                            \
        const SrcProj pq_ = project_src_level(pb_, b4_ ? 0.0625f : 1.0f, hb_, wd_b_, cwb_, chb_, rcwb_, rchb_);                    \
        KT_TAPS(qb_, wb_, pq_, hb_, wd_b_, db_, (b4_ || kqg_ == 1) && (LIVE_))                                                     \
        const float out_ = ((has_a_ && !pa_.inside) ? 1.0f : 0.0f) + ((b4_ && !pq_.inside) ? 1.0f : 0.0f);                         \
        inside_ = kt_qsum(out_) == 0.0f;                                                                                           \
    }
    int64_t tile = (int64_t)blockIdx.x * KT_WAVES + wave;
    int64_t src_cur = KT_SRC(tile), src_n1 = KT_SRC(tile + stride);
    KtPoint P_cur = load_point(src_cur);
    f32x4 qa[4], qb[4];
    float wa[4], wb[4];
    bool inside;
    KT_GATHER(P_cur, src_cur >= 0, qa, qb, wa, wb, inside)
    for (; tile < n_tiles; tile += stride) {
#ifdef GENS_K18T_STAMPS
        ++tile_no;
#endif
        KT_STAMP();                                                   // 0: tile start
        asm volatile("" : "+s"(kp));                                  // (the loads through it belong to this tile: not hoisted, not kept)
        const KtArgs __attribute__((address_space(4)))& A = *kp;
        const KtPoint P_n1 = load_point(src_n1);                      // in flight until the next trip
        const int64_t src_n2 = KT_SRC(tile + 2 * stride);
        const bool live = src_cur >= 0;
        const int64_t pt = tile * PPW + pl;
        const float x = P_cur.x, y = P_cur.y, z = P_cur.z;
        // (the look-up of THIS tile was issued one tile ago -- KT_GATHER in the reverse pass below, or ahead of the loop: qa / qb / wa / wb / inside)
        const float mask = (live && inside) ? 1.0f : 0.0f;
        // compute_angle (projector.py:278-291), IEEE square roots / divisions as the PyTorch path takes them: the view weights below are differences of
        // exponentials of rd[3] - 1 -- with two source views ONE difference, ~1e-6 where the viewing angles agree -- and 1 ulp here is per cent there
        // (hardware sqrt / rcp / exp in these few lines: operand rows 3e-5 .. 2e-4 away from the row-major kernel's instead of 1e-6, for 3 % of a tile)
        float rd[4];
        {
            float rx = A.c2w[3] - x, ry = A.c2w[7] - y, rz = A.c2w[11] - z;
            const float rn = sqrtf(rx * rx + ry * ry + rz * rz) + 1e-6f;
            rx /= rn; ry /= rn; rz /= rn;
            const float* cs = A.c2w + 16 * sv;
            float sx = cs[3] - x, sy = cs[7] - y, sz = cs[11] - z;
            const float sn = sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f;
            sx /= sn; sy /= sn; sz /= sn;
            const float dx = rx - sx, dy = ry - sy, dz = rz - sz;
            const float dn = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
            rd[0] = live ? dx / dn : 0.0f;
            rd[1] = live ? dy / dn : 0.0f;
            rd[2] = live ? dz / dn : 0.0f;
            rd[3] = live ? rx * sx + ry * sy + rz * sz : 0.0f;
        }
        if (kq == 0) {                                                // rgb_fc.0's (HX + 1 ..) and ray_dir_fc.0's (RD ..) copies of the ray difference
            KT_AT(ST::HX + 1) = rd[0]; KT_AT(ST::HX + 2) = rd[1]; KT_AT(ST::HX + 3) = rd[2]; KT_AT(ST::HX + 4) = rd[3];
            KT_AT(ST::RD) = rd[0]; KT_AT(ST::RD + 1) = rd[1]; KT_AT(ST::RD + 2) = rd[2]; KT_AT(ST::RD + 3) = rd[3];
        }
        KT_STAMP();                                                   // 1: projection, texel loads issued, compute_angle
        // ================================================================ forward
        // ---------------------------------------------------------------- ray_dir_fc (:87-88)
        f32x4 DFE[XT];
        {
            f32x4 RDt[1], D1[1];
            RDt[0] = kq == 0 ? (f32x4){rd[0], rd[1], rd[2], rd[3]} : kt_splat(0.0f);
            kt_fwd<1, 1>(KT_W(0), KT_P(0), KT_B(0), RDt, D1, i, kq);
#pragma unroll
            for (int r = 0; r < 4; ++r) D1[0][r] = kt_elu(D1[0][r]);
            kt_put(S, L, ST::D1, 0, D1[0], kq, 16);
            kt_fwd<1, XT>(KT_W(1), KT_P(1), KT_B(1), D1, DFE, i, kq);
#pragma unroll
            for (int t = 0; t < XT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) DFE[t][r] = kt_elu(DFE[t][r]);
                DFE[t] = kt_mask(DFE[t], t, kq, F);
                kt_put(S, L, ST::DFE, t, DFE[t], kq, F);
            }
        }
        KT_STAMP();                                                   // 2: ray_dir_fc
        // ---------------------------------------------------------------- the texels have arrived: [rgb | features] of the row into H0's x block
        {
            f32x4 fa = (f32x4){0.f, 0.f, 0.f, 0.f}, fb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                fa.x = __builtin_fmaf(qa[k].x, wa[k], fa.x); fa.y = __builtin_fmaf(qa[k].y, wa[k], fa.y);
                fa.z = __builtin_fmaf(qa[k].z, wa[k], fa.z); fa.w = __builtin_fmaf(qa[k].w, wa[k], fa.w);
                fb.x = __builtin_fmaf(qb[k].x, wb[k], fb.x); fb.y = __builtin_fmaf(qb[k].y, wb[k], fb.y);
                fb.z = __builtin_fmaf(qb[k].z, wb[k], fb.z); fb.w = __builtin_fmaf(qb[k].w, wb[k], fb.w);
            }
            // level kq's four channels sit at x-block channels 3 + 4 kq ..: channel c = base + 4 kq + j with base = H0 + 2 FP + 3 (+ j)
            if (kq < NLEV) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = ST::H0 + 2 * FP + 3 + j + 4 * kq;
                    S[kt_row(c) * 16 + L.co[(ST::H0 + 2 * FP + 3 + j) & 3]] = j == 0 ? fa.x : j == 1 ? fa.y : j == 2 ? fa.z : fa.w;
                }
            }
            if (NLEV == 5 && kq == 0) {
                KT_AT(ST::H0 + 2 * FP + 19) = fb.x; KT_AT(ST::H0 + 2 * FP + 20) = fb.y; KT_AT(ST::H0 + 2 * FP + 21) = fb.z; KT_AT(ST::H0 + 2 * FP + 22) = fb.w;
            }
            if (kq == 1) {
                KT_AT(ST::H0 + 2 * FP) = fb.x; KT_AT(ST::H0 + 2 * FP + 1) = fb.y; KT_AT(ST::H0 + 2 * FP + 2) = fb.z;
                RC0[n] = fb.x; RC1[n] = fb.y; RC2[n] = fb.z;
            }
        }
        // ---------------------------------------------------------------- x = rgb_feat + direction feature (:89)
        f32x4 xq[XT];
#pragma unroll
        for (int t = 0; t < XT; ++t) {
            xq[t] = kt_get(S, L, ST::H0 + 2 * FP, t, kq, F) + DFE[t];
            kt_put(S, L, ST::H0 + 2 * FP, t, xq[t], kq, F);
        }
        // ---------------------------------------------------------------- view weights, weighted mean / variance (:93-101)
        const float e = expf(s_abs * (rd[3] - 1.0f));
        const float mn = kt_gmin<G>(dead ? __builtin_inff() : e);
        const float arg = kt_gmin<G>((!dead && e == mn) ? (float)v : 99.0f);        // the first view that attains the minimum
        const float wr = dead ? 0.0f : (e - mn) * mask;
        const float wsum = kt_gsum<G>(wr);
        const float wn = wr / (wsum + 1e-8f);
        if (kq == 0) RSW[n] = wn;
#pragma unroll
        for (int t = 0; t < XT; ++t) {
            f32x4 mean, var;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mean[r] = kt_gsum<G>(wn * xq[t][r]);
                const float d = xq[t][r] - mean[r];
                var[r] = kt_gsum<G>(wn * (d * d));
            }
            kt_put(S, L, ST::H0, t, mean, kq, F);
            kt_put(S, L, ST::H0 + FP, t, var, kq, F);
        }
        KT_STAMP();                                                   // 3: texels interpolated, view weights, mean / variance
        // ---------------------------------------------------------------- base_fc (:103-104)
        f32x4 H[2];
        {
            f32x4 H0t[HT], TB[4];
#pragma unroll
            for (int t = 0; t < HT; ++t) H0t[t] = kt_get(S, L, ST::H0, t, kq, 3 * FP);      // (the pad channels of the three blocks hold zeros)
            kt_fwd<HT, 4>(KT_W(2), KT_P(2), KT_B(2), H0t, TB, i, kq);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) TB[t][r] = kt_elu(TB[t][r]);
                kt_put(S, L, ST::TB, t, TB[t], kq, 64);
            }
            kt_fwd<4, 2>(KT_W(3), KT_P(3), KT_B(3), TB, H, i, kq);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) H[t][r] = kt_elu(H[t][r]);
                kt_put(S, L, ST::H, t, H[t], kq, 32);
            }
        }
        KT_STAMP();                                                   // 4: base_fc
        // ---------------------------------------------------------------- vis_fc on h * w (:106-109)
        f32x4 H2[2];
        float hv32, vis;
        {
            f32x4 A0[2], TV[2], HV[2];
            A0[0] = H[0] * wn; A0[1] = H[1] * wn;
            kt_fwd<2, 2>(KT_W(4), KT_P(4), KT_B(4), A0, TV, i, kq);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) TV[t][r] = kt_elu(TV[t][r]);
                kt_put(S, L, ST::TV, t, TV[t], kq, 32);
            }
            kt_fwd<2, 2>(KT_W(5), KT_P(5), KT_B(5), TV, HV, i, kq);
            const f32x4 v2w0 = *(const f32x4*)(KT_W(5) + 32 * KT_P(5) + 4 * kq), v2w1 = *(const f32x4*)(KT_W(5) + 32 * KT_P(5) + 16 + 4 * kq);
            hv32 = kt_elu(kt_qsum(kt_dot(v2w0, TV[0]) + kt_dot(v2w1, TV[1])) + KT_B(5)[32]);          // the 33rd output reads the same hidden layer
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) HV[t][r] = kt_elu(HV[t][r]);
                kt_put(S, L, ST::HV, t, HV[t], kq, 32);
                H2[t] = H[t] + HV[t];                                                                 // x = x + x_res
                kt_put(S, L, ST::H2, t, H2[t], kq, 32);
            }
            vis = hw_sigmoid(hv32) * mask;
            if (kq == 0) RSV[n] = vis;
        }
        // ---------------------------------------------------------------- vis_fc2 on x * vis (:110), rgb_fc on cat([x, vis, ray_diff]) (:113-115)
        float vis2, score;
        {
            f32x4 A0[2], TU[2];
            A0[0] = H2[0] * vis; A0[1] = H2[1] * vis;
            kt_fwd<2, 2>(KT_W(6), KT_P(6), KT_B(6), A0, TU, i, kq);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) TU[t][r] = kt_elu(TU[t][r]);
                kt_put(S, L, ST::TU, t, TU[t], kq, 32);
            }
            const f32x4 u2w0 = *(const f32x4*)(KT_W(7) + 4 * kq), u2w1 = *(const f32x4*)(KT_W(7) + 16 + 4 * kq);
            vis2 = hw_sigmoid(kt_qsum(kt_dot(u2w0, TU[0]) + kt_dot(u2w1, TU[1])) + KT_B(7)[0]) * mask;
            if (kq == 0) KT_AT(ST::HX) = vis2;
            f32x4 HH[3], T1[1], T2[1];
            HH[0] = H2[0]; HH[1] = H2[1];
            HH[2] = kq == 0 ? (f32x4){vis2, rd[0], rd[1], rd[2]} : (kq == 1 ? (f32x4){rd[3], 0.0f, 0.0f, 0.0f} : kt_splat(0.0f));
            kt_fwd<3, 1>(KT_W(8), KT_P(8), KT_B(8), HH, T1, i, kq);
#pragma unroll
            for (int r = 0; r < 4; ++r) T1[0][r] = kt_elu(T1[0][r]);
            kt_put(S, L, ST::T1, 0, T1[0], kq, 16);
            kt_fwd<1, 1>(KT_W(9), KT_P(9), KT_B(9), T1, T2, i, kq);
#pragma unroll
            for (int r = 0; r < 4; ++r) T2[0][r] = kt_elu(T2[0][r]);
            T2[0] = kt_mask(T2[0], 0, kq, 8);
            kt_put(S, L, ST::T2, 0, T2[0], kq, 8);
            const f32x4 r3w = kt_mask(*(const f32x4*)(KT_W(10) + 4 * kq), 0, kq, 8);
            const float sc = kt_qsum(kt_dot(r3w, T2[0])) + KT_B(10)[0];
            score = dead ? -__builtin_inff() : (mask == 0.0f ? -1e9f : sc);                         // masked_fill(mask == 0, -1e9)  (:115)
        }
        // ---------------------------------------------------------------- softmax over views (:116-117)
        const float mx = kt_gmax<G>(score);
        const float ex = dead ? 0.0f : hw_exp(score - mx);
        const float p = ex * hw_rcp(kt_gsum<G>(ex));
        KT_STAMP();                                                   // 5: vis_fc, vis_fc2, rgb_fc, soft-max

        // ================================================================ reverse
        // (Nothing of the forward is kept in registers but the per-row scalars: every activation is read back from the store where its derivative or
        // a per-lane sum needs it -- four ds_read_b32 per tile against ~100 registers held through the whole reverse pass.)
        // colour = sum_v rgb_in p_v: score_bar = p (p_bar - sum_u p_u p_bar_u), rgb_in_bar = g p
        const float g0 = P_cur.g0, g1 = P_cur.g1, g2 = P_cur.g2;
        const float dotv = (g0 * RC0[n] + g1 * RC1[n]) + g2 * RC2[n];
        const float pdot = kt_gsum<G>(p * dotv);
        // masked_fill passes no gradient to the score it replaced: with a visible view beside it p is 0 there anyway, a point NO source view sees
        // has p = 1 / S on every (masked) view and its softmax gradient must not reach the network (k18_blend_train.hip)
        const float sb = (live && mask != 0.0f) ? p * (dotv - pdot) : 0.0f;
        const float gx0 = live ? g0 * p : 0.0f, gx1 = live ? g1 * p : 0.0f, gx2 = live ? g2 * p : 0.0f;
        const int64_t row0 = tile * 16;
        f32x4 GH[2];
        float vis2_bar;
        {
            // rgb_fc.4 (one output): per-lane sums of sb [T2 | 1]; its input's cotangent
            const f32x4 T2 = kt_get(S, L, ST::T2, 0, kq, 8);
            const f32x4 r3w = kt_mask(*(const f32x4*)(KT_W(10) + 4 * kq), 0, kq, 8);
            sp_r3 += T2 * sb;
            if (kq == 0) sp_r3b += sb;
            f32x4 L9[1], L8[1], HHb[3];
#pragma unroll
            for (int r = 0; r < 4; ++r) L9[0][r] = sb * r3w[r] * kt_elu_d(T2[r]);
            L9[0] = kt_mask(L9[0], 0, kq, 8);
            if (DUMP) {                                                  // rgb_fc.4: L = [sb | 0], R = [T2 | 1 | 0]
                if (kq == 0) { A.dbg_l[10][2 * (row0 + n)] = sb; A.dbg_l[10][2 * (row0 + n) + 1] = 0.0f; A.dbg_r[10][10 * (row0 + n) + 8] = 1.0f; A.dbg_r[10][10 * (row0 + n) + 9] = 0.0f; }
                for (int r = 0; r < 4; ++r)
                    if (4 * kq + r < 8) A.dbg_r[10][10 * (row0 + n) + 4 * kq + r] = T2[r];
            }
            kt_put(S, L, ST::T2, 0, L9[0], kq, 8);
            kt_rev<1, 1>(KT_W(9), KT_P(9), L9, L8, i, kq);
            kt_dw<kt_acc_off(9, F), 1, 2, NTILES, ST::CH>(wacc, S, L, ST::T2, ST::T1, 16, nullptr, i, kq);
            if (DUMP) kt_dump<9, F, 1, 2, ST::CH>(A, S, L, ST::T2, ST::T1, 16, nullptr, row0, i, kq);
            const f32x4 T1 = kt_get(S, L, ST::T1, 0, kq, 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) L8[0][r] = kt_elu_dy(L8[0][r], T1[r]);
            kt_put(S, L, ST::T1, 0, L8[0], kq, 16);
            // rgb_fc.0 input = [h2 (32) | vis2 | ray difference]
            kt_rev<1, 3>(KT_W(8), KT_P(8), L8, HHb, i, kq);
            kt_dw<kt_acc_off(8, F), 1, 3, NTILES, ST::CH>(wacc, S, L, ST::T1, ST::H2, 37, nullptr, i, kq);
            if (DUMP) kt_dump<8, F, 1, 3, ST::CH>(A, S, L, ST::T1, ST::H2, 37, nullptr, row0, i, kq);
            GH[0] = HHb[0]; GH[1] = HHb[1];
            vis2_bar = kt_from_q0(HHb[2][0]);
        }
        float hv32_bar;
        {
            // vis2 = sigmoid(q) mask ; q = u2 . tu + b (one output)
            const float q_bar = vis2_bar * mask * vis2 * (1.0f - vis2);
            const f32x4 TU0 = kt_get(S, L, ST::TU, 0, kq, 32), TU1 = kt_get(S, L, ST::TU, 1, kq, 32);
            const f32x4 u2w0 = *(const f32x4*)(KT_W(7) + 4 * kq), u2w1 = *(const f32x4*)(KT_W(7) + 16 + 4 * kq);
            sp_u2[0] += TU0 * q_bar; sp_u2[1] += TU1 * q_bar;
            if (kq == 0) sp_u2b += q_bar;
            f32x4 L6[2], M6[2];
#pragma unroll
            for (int r = 0; r < 4; ++r) { L6[0][r] = q_bar * u2w0[r] * kt_elu_d(TU0[r]); L6[1][r] = q_bar * u2w1[r] * kt_elu_d(TU1[r]); }
            if (DUMP) {                                                  // vis_fc2.2: L = [q_bar | 0], R = [TU | 1 | 0]
                if (kq == 0) {
                    A.dbg_l[7][2 * (row0 + n)] = q_bar; A.dbg_l[7][2 * (row0 + n) + 1] = 0.0f;
                    A.dbg_r[7][34 * (row0 + n) + 32] = 1.0f; A.dbg_r[7][34 * (row0 + n) + 33] = 0.0f;
                }
                for (int r = 0; r < 4; ++r) { A.dbg_r[7][34 * (row0 + n) + 4 * kq + r] = TU0[r]; A.dbg_r[7][34 * (row0 + n) + 16 + 4 * kq + r] = TU1[r]; }
            }
            kt_put(S, L, ST::TU, 0, L6[0], kq, 32); kt_put(S, L, ST::TU, 1, L6[1], kq, 32);
            // vis_fc2.0 input = h2 * vis: h2_bar += m vis ; vis_bar = sum_k m_k h2_k
            kt_rev<2, 2>(KT_W(6), KT_P(6), L6, M6, i, kq);
            kt_dw<kt_acc_off(6, F), 2, 3, NTILES, ST::CH>(wacc, S, L, ST::TU, ST::H2, 32, RSV, i, kq);
            if (DUMP) kt_dump<6, F, 2, 3, ST::CH>(A, S, L, ST::TU, ST::H2, 32, RSV, row0, i, kq);
            const f32x4 H20 = kt_get(S, L, ST::H2, 0, kq, 32), H21 = kt_get(S, L, ST::H2, 1, kq, 32);
            const float vis_bar = kt_qsum(kt_dot(M6[0], H20) + kt_dot(M6[1], H21));
            GH[0] += M6[0] * vis; GH[1] += M6[1] * vis;
            hv32_bar = vis_bar * mask * vis * (1.0f - vis) * kt_elu_d(hv32);                            // cotangent of vis_fc.2's 33rd pre-activation
        }
        KT_STAMP();                                                   // 6: reverse rgb_fc, vis_fc2 (+ their weight gradients)
        float w_bar;
        {
            // h2 = h + hv[:32]: hv_bar[:32] = h_bar = GH
            const f32x4 TV0 = kt_get(S, L, ST::TV, 0, kq, 32), TV1 = kt_get(S, L, ST::TV, 1, kq, 32);
            sp_v2[0] += TV0 * hv32_bar; sp_v2[1] += TV1 * hv32_bar;
            if (kq == 0) sp_v2b += hv32_bar;
            f32x4 L5[2], L4[2], M4[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 HV = kt_get(S, L, ST::HV, t, kq, 32);
#pragma unroll
                for (int r = 0; r < 4; ++r) L5[t][r] = kt_elu_dy(GH[t][r], HV[r]);
                kt_put(S, L, ST::HV, t, L5[t], kq, 32);
            }
            if (DUMP && kq == 0) { A.dbg_l[5][34 * (row0 + n) + 32] = hv32_bar; A.dbg_l[5][34 * (row0 + n) + 33] = 0.0f; }
            kt_rev<2, 2>(KT_W(5), KT_P(5), L5, L4, i, kq);
            kt_dw<kt_acc_off(5, F), 2, 3, NTILES, ST::CH>(wacc, S, L, ST::HV, ST::TV, 32, nullptr, i, kq);
            if (DUMP) kt_dump<5, F, 2, 3, ST::CH>(A, S, L, ST::HV, ST::TV, 32, nullptr, row0, i, kq);
            const f32x4 v2w0 = *(const f32x4*)(KT_W(5) + 32 * KT_P(5) + 4 * kq), v2w1 = *(const f32x4*)(KT_W(5) + 32 * KT_P(5) + 16 + 4 * kq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                L4[0][r] = kt_elu_dy(L4[0][r] + hv32_bar * v2w0[r], TV0[r]);
                L4[1][r] = kt_elu_dy(L4[1][r] + hv32_bar * v2w1[r], TV1[r]);
            }
            kt_put(S, L, ST::TV, 0, L4[0], kq, 32); kt_put(S, L, ST::TV, 1, L4[1], kq, 32);
            // vis_fc.0 input = h * w: h_bar += m w ; w_bar += sum_k m_k h_k
            kt_rev<2, 2>(KT_W(4), KT_P(4), L4, M4, i, kq);
            kt_dw<kt_acc_off(4, F), 2, 3, NTILES, ST::CH>(wacc, S, L, ST::TV, ST::H, 32, RSW, i, kq);
            if (DUMP) kt_dump<4, F, 2, 3, ST::CH>(A, S, L, ST::TV, ST::H, 32, RSW, row0, i, kq);
            const f32x4 H0_ = kt_get(S, L, ST::H, 0, kq, 32), H1_ = kt_get(S, L, ST::H, 1, kq, 32);
            w_bar = kt_qsum(kt_dot(M4[0], H0_) + kt_dot(M4[1], H1_));
            f32x4 L3[2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                                                  // base_fc.2 pre-activation
                L3[0][r] = kt_elu_dy(GH[0][r] + M4[0][r] * wn, H0_[r]);
                L3[1][r] = kt_elu_dy(GH[1][r] + M4[1][r] * wn, H1_[r]);
            }
            KT_STAMP();                                               // 7: reverse vis_fc
            // (H's store still fed the weight gradient of vis_fc.0 above: L3 goes in after it)
            kt_put(S, L, ST::H, 0, L3[0], kq, 32); kt_put(S, L, ST::H, 1, L3[1], kq, 32);
            f32x4 L2[4];
            kt_rev<2, 4>(KT_W(3), KT_P(3), L3, L2, i, kq);
            kt_dw<kt_acc_off(3, F), 2, 5, NTILES, ST::CH>(wacc, S, L, ST::H, ST::TB, 64, nullptr, i, kq);
            if (DUMP) kt_dump<3, F, 2, 5, ST::CH>(A, S, L, ST::H, ST::TB, 64, nullptr, row0, i, kq);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 TB = kt_get(S, L, ST::TB, t, kq, 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) L2[t][r] = kt_elu_dy(L2[t][r], TB[r]);
                kt_put(S, L, ST::TB, t, L2[t], kq, 64);
            }
            // cotangent of [mean | var | x] (padded blocks) -> through [TB | H] back to x-layout
            f32x4 H0b[HT];
            kt_rev<4, HT>(KT_W(2), KT_P(2), L2, H0b, i, kq);
            kt_dw<kt_acc_off(2, F), 4, kt_nt(2, F), NTILES, ST::CH>(wacc, S, L, ST::TB, ST::H0, 3 * FP, nullptr, i, kq, F);
            if (DUMP) kt_dump<2, F, 4, kt_nt(2, F), ST::CH>(A, S, L, ST::TB, ST::H0, 3 * FP, nullptr, row0, i, kq);
            KT_STAMP();                                               // 8: reverse base_fc, weight gradients of base_fc
#pragma unroll
            for (int t = 0; t < HT; ++t) kt_put(S, L, ST::TB, t, H0b[t], kq, 3 * FP);
        }
        // ---------------------------------------------------------------- the NEXT tile's look-up goes out now (its points arrived long ago)
        f32x4 qa_n[4], qb_n[4];
        float wa_n[4], wb_n[4];
        bool inside_n;
#ifdef GENS_K18T_NO_GATHER      // (timing probe, WRONG results: what does the next tile's look-up cost inside the reverse pass?)
        _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { qa_n[k_] = qa[k_]; qb_n[k_] = qb[k_]; wa_n[k_] = wa[k_]; wb_n[k_] = wb[k_]; }
        inside_n = inside;
#else
        KT_GATHER(P_n1, src_n1 >= 0, qa_n, qb_n, wa_n, wb_n, inside_n)
#endif
        // mean = sum_v w x, var = sum_v w (x - mean)^2 (shared by the views of a point)
        f32x4 GX[XT];
        {
            f32x4 L1[XT], L0[1];
            float wb_part = 0.0f;
#pragma unroll
            for (int t = 0; t < XT; ++t) {
                const f32x4 mean_b = kt_get(S, L, ST::TB, t, kq, F), var_b = kt_get(S, L, ST::TB + FP, t, kq, F), x_b = kt_get(S, L, ST::TB + 2 * FP, t, kq, F);
                const f32x4 mean = kt_get(S, L, ST::H0, t, kq, F), xv = kt_get(S, L, ST::H0 + 2 * FP, t, kq, F), dfe = kt_get(S, L, ST::DFE, t, kq, F);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mb = kt_gsum<G>(mean_b[r]);
                    const float vb = kt_gsum<G>(var_b[r]);
                    const float d = xv[r] - mean[r];
                    const float cross = kt_gsum<G>(wn * d);
                    mb -= 2.0f * vb * cross;                                                             // var depends on mean too
                    const float xb = x_b[r] + wn * mb + 2.0f * wn * d * vb;
                    GX[t][r] = xb;
                    L1[t][r] = kt_elu_dy(xb, dfe[r]);                                                    // ray_dir_fc.2 pre-activation
                    wb_part += (16 * t + 4 * kq + r < F) ? mb * xv[r] + vb * d * d : 0.0f;
                }
                L1[t] = kt_mask(L1[t], t, kq, F);
                kt_put(S, L, ST::DFE, t, L1[t], kq, F);
            }
            w_bar += kt_qsum(wb_part);
            // w = wr / (sum wr + 1e-8), wr = (e - min e) mask, e = exp(|s| (dot - 1))
            {
                const float sum = wsum + 1e-8f;
                const float ww = kt_gsum<G>(w_bar * wn);
                float wrb = (w_bar - ww) / sum * mask;
                const float tot = kt_gsum<G>(wrb);
                if (!dead && (float)v == arg) wrb -= tot;                                                // the minimum's share
                if (kq == 0 && live) s_acc += wrb * e * (rd[3] - 1.0f);
            }
            kt_rev<XT, 1>(KT_W(1), KT_P(1), L1, L0, i, kq);
            kt_dw<kt_acc_off(1, F), XT, 2, NTILES, ST::CH>(wacc, S, L, ST::DFE, ST::D1, 16, nullptr, i, kq);
            if (DUMP) kt_dump<1, F, XT, 2, ST::CH>(A, S, L, ST::DFE, ST::D1, 16, nullptr, row0, i, kq);
            const f32x4 D1 = kt_get(S, L, ST::D1, 0, kq, 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) L0[0][r] = kt_elu_dy(L0[0][r], D1[r]);
            kt_put(S, L, ST::D1, 0, L0[0], kq, 16);
            kt_dw<kt_acc_off(0, F), 1, 1, NTILES, ST::CH>(wacc, S, L, ST::D1, ST::RD, 4, nullptr, i, kq);
            if (DUMP) kt_dump<0, F, 1, 1, ST::CH>(A, S, L, ST::D1, ST::RD, 4, nullptr, row0, i, kq);
        }
        KT_STAMP();                                                   // 9: reverse base_fc.0, mean / variance, ray_dir_fc
        // ---------------------------------------------------------------- cotangent of the looked-up [rgb | features] rows (K4's backward reads it)
        if (A.g_feat && live) {
            float* gf = A.g_feat + (pt * Sv + v) * F;
#pragma unroll
            for (int t = 0; t < XT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * t + 4 * kq + r;
                    if (c < F) gf[c] = GX[t][r] + (c == 0 ? gx0 : c == 1 ? gx1 : c == 2 ? gx2 : 0.0f);
                }
        }
        KT_STAMP();                                                   // 10: g_feat stored
        P_cur = P_n1;
        src_cur = src_n1;
        src_n1 = src_n2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { qa[k] = qa_n[k]; qb[k] = qb_n[k]; wa[k] = wa_n[k]; wb[k] = wb_n[k]; }
        inside = inside_n;
    }
#undef KT_SRC
#undef KT_GATHER

    // ================================================================ the workgroup's block of sums
    // The four waves' accumulators meet in LDS (the stores and the weights are dead now): in rounds of CAP tiles every wave parks its registers,
    // then wave (t & 3) adds the four copies of tile t in wave order -- a fixed order -- and writes the sums into the workgroup's block of `parts`
    // (layout of gens_gemm_tn_batch's result).  The per-lane sums of the single-output rows and of d loss / d |s| ride along as six more tiles.
    {
        constexpr int NEXT = 6, NALL = NTILES + NEXT;
        constexpr int CAP = (int)((kt_lw_total(F) + kt_b_total(F) + KT_WAVES * ST::FLOATS) / (KT_WAVES * 256));
        constexpr int ROUNDS = (NALL + CAP - 1) / CAP;
        static_assert(CAP >= 8, "LDS too small for the final reduction");
        float* out = kp->parts + (size_t)blockIdx.x * kp->csz;
        const f32x4 ext[NEXT] = {sp_v2[0], sp_v2[1], sp_u2[0], sp_u2[1], sp_r3, (f32x4){sp_v2b, sp_u2b, sp_r3b, s_acc}};
        // the sum of tile t (compile-time t of the current round) over the four waves
#define KT_SUM(t_) (((*(const f32x4*)(smem + ((0 * CAP + (t_) % CAP) * 64 + lane) * 4) + *(const f32x4*)(smem + ((1 * CAP + (t_) % CAP) * 64 + lane) * 4)) + \
                     (*(const f32x4*)(smem + ((2 * CAP + (t_) % CAP) * 64 + lane) * 4) + *(const f32x4*)(smem + ((3 * CAP + (t_) % CAP) * 64 + lane) * 4))))
#define KT_MINE(t_) ((t_) / CAP == round && ((t_) & 3) == wave)
#pragma unroll
        for (int round = 0; round < ROUNDS; ++round) {
            __syncthreads();                                            // (round 0: every wave has left the tile loop; later: the previous round was read)
#pragma unroll
            for (int t = 0; t < NALL; ++t)
                if (t / CAP == round) *(f32x4*)(smem + ((wave * CAP + t % CAP) * 64 + lane) * 4) = t < NTILES ? wacc[t < NTILES ? t : 0] : ext[t < NTILES ? 0 : t - NTILES];
            __syncthreads();
            // accumulator register r of lane (kq, i) = D[m = 4 kq + r][n = i] of its tile
#define KT_FLUSH(l)                                                                                          \
            {                                                                                                \
                constexpr int ms_ = (l) == 5 ? 32 : kt_out(l, F), ns_ = kt_in(l, F) + 1, ld_ = kt_ev(kt_in(l, F) + 1); \
                constexpr int mt_ = kt_mt(l, F), nt_ = kt_nt(l, F), off_ = kt_cc_off(l, F), ab_ = kt_acc_off(l, F); \
                _Pragma("unroll") for (int mo_ = 0; mo_ < mt_; ++mo_)                                        \
                    _Pragma("unroll") for (int no_ = 0; no_ < nt_; ++no_)                                    \
                        if (KT_MINE(ab_ + mo_ * nt_ + no_)) {                                                \
                            const f32x4 v_ = KT_SUM(ab_ + mo_ * nt_ + no_);                                  \
                            _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                               \
                                const int m_ = 16 * mo_ + 4 * kq + r_, n_ = 16 * no_ + i;                    \
                                if (m_ < ms_ && n_ < ns_) out[off_ + m_ * ld_ + n_] = v_[r_];                \
                            }                                                                                \
                        }                                                                                    \
            }
            KT_FLUSH(0) KT_FLUSH(1) KT_FLUSH(3) KT_FLUSH(4) KT_FLUSH(5) KT_FLUSH(6) KT_FLUSH(8) KT_FLUSH(9)
#undef KT_FLUSH
            {   // base_fc.0: its columns were the padded [mean | var | x] blocks; column F (the first pad slot) = the bias
                constexpr int nt_ = kt_nt(2, F), off_ = kt_cc_off(2, F), ab_ = kt_acc_off(2, F), ld_ = kt_ev(3 * F + 1);
#pragma unroll
                for (int mo = 0; mo < 4; ++mo)
#pragma unroll
                    for (int no = 0; no < nt_; ++no)
                        if (KT_MINE(ab_ + mo * nt_ + no)) {
                            const f32x4 v_ = KT_SUM(ab_ + mo * nt_ + no);
                            const int q = 16 * no + i, blk = q / FP, c = q - blk * FP;
                            const int col = q == F ? 3 * F : ((blk < 3 && c < F) ? blk * F + c : -1);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (col >= 0) out[off_ + (16 * mo + 4 * kq + r) * ld_ + col] = v_[r];
                        }
            }
            // the single-output rows: per-lane sums over this lane's row -> over the 16 rows of the lane group (valid in lane 15 of the group)
#define KT_ROWS(e_, t_, l, row_)                                                                             \
            if (KT_MINE(NTILES + (e_))) {                                                                    \
                const f32x4 v_ = KT_SUM(NTILES + (e_));                                                      \
                _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                           \
                    const float w_ = kt_rowsum16(v_[r_]);                                                    \
                    const int c_ = 16 * (t_) + 4 * kq + r_;                                                  \
                    if (i == 15 && c_ < KT_C(kt_in(l, F))) out[KT_C(kt_cc_off(l, F)) + (row_) * KT_C(kt_ev(kt_in(l, F) + 1)) + c_] = w_; \
                }                                                                                            \
            }
            KT_ROWS(0, 0, 5, 32) KT_ROWS(1, 1, 5, 32) KT_ROWS(2, 0, 7, 0) KT_ROWS(3, 1, 7, 0) KT_ROWS(4, 0, 10, 0)
#undef KT_ROWS
            if (KT_MINE(NTILES + 5)) {
                const f32x4 v_ = KT_SUM(NTILES + 5);
                const float b5 = kt_rowsum16(v_[0]), b7 = kt_rowsum16(v_[1]), b10 = kt_rowsum16(v_[2]);
                if (lane == 15) {                                       // (the three bias sums were kept by lane group 0 only)
                    out[KT_C(kt_cc_off(5, F)) + 32 * 34 + 32] = b5;
                    out[KT_C(kt_cc_off(7, F)) + 32] = b7;
                    out[KT_C(kt_cc_off(10, F)) + 8] = b10;
                }
                const float ss = wave_sum(v_[3]);
                if (lane == 0) kp->s_part[blockIdx.x] = ss;
            }
        }
#undef KT_SUM
#undef KT_MINE
    }
#undef KT_W
#undef KT_P
#undef KT_B
}

template <int F>
constexpr size_t kt_lds_bytes() { return sizeof(float) * (kt_lw_total(F) + kt_b_total(F) + KT_WAVES * St<F>::FLOATS); }

}  // namespace

// out[e] = sum over the waves' blocks in a FIXED order (the result does not depend on scheduling): a workgroup = 32 elements x 8 groups of parts, a
// thread adds the parts p = g, g + 8, ... of its element, the eight sums are added in group order (as k18_blend_train.hip's)
__global__ __launch_bounds__(256) void blend_train_t_reduce_k(const float* __restrict__ parts, int n_parts, int csz, float* __restrict__ out) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5, e = blockIdx.x * 32 + el;
    float s = 0.0f;
    if (e < csz)
        for (int p = g; p < n_parts; p += 8) s += parts[(size_t)p * csz + e];
    red[g][el] = s;
    __syncthreads();
    if (g == 0 && e < csz) out[e] = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
}
int gens_fill_maps(const char* who, MapSet* ms, const float* const* feats, const int* hw, int n_levels);
extern "C" int gens_blend_train_acc_floats(int n_levels);

// Number of partial blocks (= workgroups) of a gens_blend_train_bwd_t launch over n points of nv views: four waves per workgroup, one persistent
// workgroup per CU at most (GENS_K18T_WGS overrides the cap: occupancy probes).  0: nothing to launch, or a view count this kernel is not built for.
extern "C" int gens_blend_train_t_parts(int64_t n, int nv) {
    if (n <= 0 || nv < 3 || nv > 5) return 0;
    const int ppw = nv == 3 ? 8 : 4;
    const int64_t tiles = (n + ppw - 1) / ppw;
    int64_t cap = getenv("GENS_K18T_WGS") ? atoi(getenv("GENS_K18T_WGS")) : 256;
    if (cap < 1) cap = 1;
    const int64_t wgs = (tiles + KT_WAVES - 1) / KT_WAVES;
    return (int)(wgs < cap ? wgs : cap);
}

static int kt_launch(const char* who, bool dump, const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                     const float* c2w, int nv, const float* const* weights, KtArgs& A, void* stream) {
    GENS_CHECK_ARG(feats && weights, GENS_EINVAL, "%s: null table", who);
    if (int e = gens_fill_maps(who, &A.fs, feats, hw, n_levels)) return e;
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "%s: 1..5 feature levels (d_feature <= 20), got %d", who, n_levels);
    GENS_CHECK_ARG(nv >= 3 && nv <= 5, GENS_ELIMIT, "%s: built for two to four source views (nv = 3..5), got nv=%d (use gens_blend_train_bwd_acc)", who, nv);
    GENS_CHECK_ARG(imgs && w2c && intr && c2w, GENS_EINVAL, "%s: null camera / image pointer", who);
    for (int k = 0; k < 23; ++k) {
        GENS_CHECK_ARG(weights[k], GENS_EINVAL, "%s: weight %d is null", who, k);
        A.w[k] = weights[k];
    }
    A.imgs = (const float4*)imgs;
    A.w2c = w2c; A.intr = intr; A.c2w = c2w; A.nv = nv;
    A.csz = gens_blend_train_acc_floats(n_levels);
    const unsigned grid = (unsigned)gens_blend_train_t_parts(A.n, nv);
    hipStream_t st = (hipStream_t)stream;
    static GensLdsOptIn once[6][2][2];
#define KT_GO(NL, G_, D_)                                                                                                              \
    {                                                                                                                                  \
        constexpr int lds_ = (int)kt_lds_bytes<3 + 4 * NL>();                                                                          \
        if (int e_ = gens_lds_opt_in(once[NL][G_ == 4][D_], (const void*)blend_train_t_k<NL, G_, D_>, lds_, who)) return e_;           \
        blend_train_t_k<NL, G_, D_><<<grid, KT_THREADS, lds_, st>>>(A);                                                                \
    }
#define KT_LAUNCH(NL)                                                                                                                  \
    {                                                                                                                                  \
        if (dump) { if (nv == 3) KT_GO(NL, 2, true) else KT_GO(NL, 4, true) }                                                          \
        else { if (nv == 3) KT_GO(NL, 2, false) else KT_GO(NL, 4, false) }                                                             \
    }
    switch (n_levels) {
        case 1: KT_LAUNCH(1) break;
        case 2: KT_LAUNCH(2) break;
        case 3: KT_LAUNCH(3) break;
        case 4: KT_LAUNCH(4) break;
        default: KT_LAUNCH(5) break;
    }
#undef KT_LAUNCH
#undef KT_GO
    return gens_launch_status(who);
}

// The backward launch of the colour branch, transposed: `parts` = gens_blend_train_t_parts(n, nv) blocks of gens_blend_train_acc_floats(n_levels)
// floats (one per workgroup), `cc` = their fixed-order sum = what gens_gemm_tn_batch returns for the eleven products (the input of
// gens_blend_train_wgrad); s_part: one partial of d loss / d |s| per workgroup (gens_blend_train_t_parts of them).
extern "C" int gens_blend_train_bwd_t(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                      const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                                      const int32_t* n_device, const float* g_rgb, float* g_feat, float* s_part, float* parts, float* cc, void* stream) {
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_rgb && s_part && parts && cc)), GENS_EINVAL, "gens_blend_train_bwd_t: null pointer");
    if (n == 0) return 0;
    KtArgs A = {};
    A.pts = pts; A.index = index; A.n_dev = n_device; A.n = n; A.g_rgb = g_rgb; A.g_feat = g_feat; A.s_part = s_part; A.parts = parts;
    if (int e = kt_launch("gens_blend_train_bwd_t", false, feats, hw, n_levels, imgs, w2c, intr, c2w, nv, weights, A, stream)) return e;
    blend_train_t_reduce_k<<<gens_blocks(A.csz, 32), 256, 0, (hipStream_t)stream>>>(parts, gens_blend_train_t_parts(n, nv), A.csz, cc);
    return gens_launch_status("gens_blend_train_bwd_t");
}

// The same launch leaving, besides, the operand rows of every layer in the layout of gens_blend_train_bwd (r_ops[l]: (rows, even(in_l + 1)) =
// [input | 1 | 0]; l_ops[l]: (rows, even(out_l))), rows = 16 per tile of 16 / G points in THIS kernel's order (row 16 tile + G point + view; three
// views: every fourth row is empty).  A test and debugging aid: the two kernels can be compared layer by layer.
extern "C" int gens_blend_train_bwd_t_dump(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                           const float* c2w, int nv, const float* const* weights, const float* pts, const int64_t* index, int64_t n,
                                           const int32_t* n_device, const float* g_rgb, float* g_feat, float* s_part, float* parts, float* cc,
                                           float* const* r_ops, float* const* l_ops, void* stream) {
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_rgb && s_part && parts && cc && r_ops && l_ops)), GENS_EINVAL, "gens_blend_train_bwd_t_dump: null pointer");
    if (n == 0) return 0;
    KtArgs A = {};
    A.pts = pts; A.index = index; A.n_dev = n_device; A.n = n; A.g_rgb = g_rgb; A.g_feat = g_feat; A.s_part = s_part; A.parts = parts;
    for (int l = 0; l < KT_NLAYER; ++l) {
        GENS_CHECK_ARG(r_ops[l] && l_ops[l], GENS_EINVAL, "gens_blend_train_bwd_t_dump: operand buffer %d is null", l);
        A.dbg_r[l] = r_ops[l];
        A.dbg_l[l] = l_ops[l];
    }
    if (int e = kt_launch("gens_blend_train_bwd_t_dump", true, feats, hw, n_levels, imgs, w2c, intr, c2w, nv, weights, A, stream)) return e;
    blend_train_t_reduce_k<<<gens_blocks(A.csz, 32), 256, 0, (hipStream_t)stream>>>(parts, gens_blend_train_t_parts(n, nv), A.csz, cc);
    return gens_launch_status("gens_blend_train_bwd_t_dump");
}

#ifdef GENS_K18T_STAMPS
extern "C" int gens_debug_k18t_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k18t_stamps), sizeof(unsigned long long) * 4 * 64) == hipSuccess ? 0 : -1;
}
#endif
