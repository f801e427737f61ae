// K7t: source-view feature look-up + the whole IBRNet-style BlendingNetwork for TWO, THREE or FOUR source views (the view counts the
// reference ships: num_src_view = 2 in the DTU / BlendedMVS test protocol and the fine-tune configs, confs/gens.conf:24,
// confs/gens_finetune.conf:15; 4 in training, confs/gens.conf:9), TRANSPOSED (the dataflow of
// k6t_sdf_value.hip applied to k7_blend.hip's eleven small layers; replaces for validation rendering lookup_feature + compute_angle,
// /root/reference/models/modules/projector.py:278-349, and BlendingNetwork.forward, models/modules/blending_network.py:69-118, as
// called from implicit_surface.py:196-199).
//
//   * weights are the A operand of v_mfma_f32_16x16x4_f32 (16 output features x 4 K, exact float32), the (point, view) rows the B
//     operand: one wavefront owns 64 rows = 16 points x 4 views as four N tiles of 16 columns; column n' = 4 * point + view of a tile
//     lives in the four lanes (n', q), q = 0..3.  Every activation vector is kept in "quad layout": feature f = 4 kq + q is register
//     kq of lane group q -- which is both what an accumulator tile T delivers (register i of lane group q = output row 4 q + i, and
//     the host orders the rows of every matrix so that row 4 q + i of tile T is feature 4 (4 T + i) + q) and what the B operand of
//     the K quad kq wants.  The activations of all eleven layers stay in registers: k7_blend.hip moved them through an LDS tile
//     (362 LDS instructions and their address arithmetic per 32 rows) and loaded every weight once per 32 rows; here a weight float4
//     is loaded once per 64 rows and feeds 16 MFMAs.
//   * the S views of a point are G ADJACENT lanes (G = 4 for S = 3, 4; G = 2 for S = 2): min / sum / max over views are one or two
//     quad_perm DPP butterflies.  S = 2: a wave owns 32 points (8 per N tile) and the per-point product below has two N tiles of
//     points.  S = 3: the fourth lane of a quad is a DEAD row -- no camera is read for it, its mask is 0, it is +inf in the minimum of
//     the anti-alias weights and -inf in the soft-max, so the three live views see exactly the reference's sums.
//   * base_fc.0 reads cat([mean, var, x]): mean and var are the same for the four views of a point, so their 2 F columns are
//     multiplied ONCE per point -- an N tile of the wave's 16 points (operands gathered with ds_bpermute) -- and the result, broadcast
//     back to the rows, is the initial value of the x-part's accumulators: 48 instead of 192 MFMAs per 64 rows.
//   * biases ride in spare K slots where the K padding is free (base_fc.0, rgb_fc.0), else in the accumulators' initial value; the
//     single-output layers (vis_fc's 33rd row, vis_fc2.2, rgb_fc.4) are per-lane dot products reduced over the four lane groups.
#include "k4_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define KT_NT 4                 // N tiles per wave (16 columns each): 16 points x 4 views
#define KT_XS 29                // row stride of the gathered-feature tile (floats): odd, so the per-column reads spread over the banks

struct BlendTWeights {
    const float4* stream;       // A fragments in consumption order: per (M tile, group of 4 K quads) 64 lanes x float4
    const float* tab;           // per lane group q: [entry][q][8] floats (accumulator-layout biases, dot-product rows)
    float v2_last_b, u2_b, r3_b, s_abs;
    const float* sdev;          // the same four scalars in DEVICE memory (gens_blend_pack_t: weights that change every step), or NULL
};
enum { KT_RD1_B = 0, KT_RD2_B, KT_B2_B, KT_V1_B, KT_V2_B, KT_U1_B, KT_R2_B, KT_V2_LAST, KT_U2, KT_R3, KT_TAB_ENTRIES };

__device__ __forceinline__ float elu1t(float x) { return __builtin_amdgcn_fmed3f(x, hw_exp(x) - 1.0f, 0.0f); }   // see k7_blend.hip::elu1
template <int G>
__device__ __forceinline__ float group_sum(float v) {     // sum over the G adjacent lanes of a point (= its views), in every lane
    v += dpp_move<0xB1, 0xF>(v, v);                       // quad_perm:[1,0,3,2]
    if (G == 4) v += dpp_move<0x4E, 0xF>(v, v);           // quad_perm:[2,3,0,1]
    return v;
}
template <int G>
__device__ __forceinline__ float group_min(float v) {
    v = fminf(v, dpp_move<0xB1, 0xF>(v, v));
    if (G == 4) v = fminf(v, dpp_move<0x4E, 0xF>(v, v));
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
    if (G == 4) v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
    return v;
}
__device__ __forceinline__ float lanes_q_sum(float v) {   // sum over the four lane groups q of a column
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int NLEV, int S>
__global__ __launch_bounds__(64, 2) void blend_t_k(BlendTWeights W, MapSet fs, const float4* __restrict__ imgs, const float* __restrict__ w2c,
                                                   const float* __restrict__ intr, const float* __restrict__ c2w, const float* __restrict__ pts,
                                                   const int64_t* __restrict__ index, int64_t n_max, const int32_t* __restrict__ n_dev,
                                                   float* __restrict__ rgb_out, uint8_t* __restrict__ vis_out) {
    constexpr int F = 3 + 4 * NLEV;
    constexpr int XQ = NLEV + 1;             // K quads of a feature vector: F + 1 = 4 (NLEV + 1) slots, the last one carries the constant one
    constexpr int XT = (XQ + 3) / 4;         // accumulator tiles of a feature vector
    constexpr int G = S == 2 ? 2 : 4;        // lanes per point (S = 3: one dead lane)
    constexpr int PPT = 16 / G;              // points per N tile
    constexpr int PPW = 64 / G;              // points per wave
    constexpr int NPT = PPW / 16;            // N tiles of POINTS of the per-point product (mean / variance columns)
    static_assert(S >= 2 && S <= 4, "two to four source views");
    static_assert(4 * XQ <= KT_XS, "feature tile too narrow");
    __shared__ float X[64 * KT_XS];          // gathered rows: rgb (3), features (4 NLEV), one
    __shared__ float RD[64 * 5];             // ray difference (4)
    __shared__ float R[64];                  // mask
    const int lane = threadIdx.x;
    const int np = lane & 15, q = lane >> 4;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    const int64_t first = (int64_t)blockIdx.x * PPW;
    if (first >= n) return;
    if (W.sdev) { W.v2_last_b = W.sdev[0]; W.u2_b = W.sdev[1]; W.r3_b = W.sdev[2]; W.s_abs = W.sdev[3]; }

    // weights: scalar base, three float4 registers rotate (this group's, the next two in flight)
    const float4* wp = W.stream;
    float4 wb[3];
    int par = 0;
    wb[0] = wp[lane];
    wb[1] = wp[lane + 64];
    wp += 128;

    // ---------------------------------------------------------------- phase 0: one lane per (point, view) row gathers it
    {
        const int pl = lane / G, sv = (lane % G) + 1;
        const bool live = first + pl < n && sv <= S;          // (S = 3: the fourth lane of a quad carries no view)
        const int64_t src = live ? (index ? index[first + pl] : first + pl) : 0;
        float x = 0.f, y = 0.f, z = 0.f;
        if (live) { x = pts[3 * src]; y = pts[3 * src + 1]; z = pts[3 * src + 2]; }
        bool inside = true;
        float* xr = X + lane * KT_XS;
        const int svc = sv <= S ? sv : S;                       // camera read by the dead lane (never used)
        const SrcBase pb = project_src_base(w2c + 16 * svc, intr + 16 * svc, x, y, z);
#pragma unroll
        for (int l = 0; l < NLEV; ++l) {
            const int h = fs.h[l], w = fs.w[l];
            const SrcProj p = project_src_level(pb, exp2f(-(float)l), h, w, fs.cw[l], fs.ch[l], fs.rcw[l], fs.rch[l]);
            inside = inside && p.inside;
            float4 f = f4_zero(), c = f4_zero();
            if (live) {
                const Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
                f = sample_texel(fs.data[l] + (int64_t)svc * h * w, h, w, 1, 0, t);
                if (l == 0) c = sample_texel(imgs + (int64_t)svc * h * w, h, w, 1, 0, t);
            }
            xr[3 + 4 * l] = f.x; xr[4 + 4 * l] = f.y; xr[5 + 4 * l] = f.z; xr[6 + 4 * l] = f.w;
            if (l == 0) { xr[0] = c.x; xr[1] = c.y; xr[2] = c.z; }
        }
        xr[F] = 1.0f;
        // (not-a-number inputs must come out as not-a-number: the median form of the ELU would drop them, so the row's mask carries a
        // poison term 0 * (sum of its inputs) -- the mask multiplies the view weights, the visibilities and gates the score)
        float acc_in = x + y + z;
#pragma unroll
        for (int k = 0; k < F; ++k) acc_in += xr[k];
        R[lane] = ((live && inside) ? 1.0f : 0.0f) + 0.0f * acc_in;
        if (live && vis_out) vis_out[src * S + (sv - 1)] = inside ? 1 : 0;
        // compute_angle (projector.py:278-291), hardware sqrt / rcp as in k7_blend.hip
        float rx = c2w[3] - x, ry = c2w[7] - y, rz = c2w[11] - z;
        const float rn = hw_rcp(__builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz) + 1e-6f);
        rx *= rn; ry *= rn; rz *= rn;
        const float* cs = c2w + 16 * svc;
        float sx = cs[3] - x, sy = cs[7] - y, sz = cs[11] - z;
        const float sn = hw_rcp(__builtin_amdgcn_sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f);
        sx *= sn; sy *= sn; sz *= sn;
        const float dx = rx - sx, dy = ry - sy, dz = rz - sz;
        const float dn = hw_rcp(fmaxf(__builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f));
        float* rd = RD + lane * 5;
        rd[0] = live ? dx * dn : 0.0f;
        rd[1] = live ? dy * dn : 0.0f;
        rd[2] = live ? dz * dn : 0.0f;
        rd[3] = live ? rx * sx + ry * sy + rz * sz : 0.0f;
    }
    __syncthreads();

    // ---------------------------------------------------------------- this lane's operand slots of the four N tiles
    float xq[KT_NT][XQ];        // x in quad layout (slot F = the one)
    float rgbc[KT_NT];          // colour channel q of the column (q < 3)
    float rdq[KT_NT], dotv[KT_NT], mask[KT_NT], rd3one[KT_NT], rdsh[KT_NT];
#pragma unroll
    for (int j = 0; j < KT_NT; ++j) {
        const int row = 16 * j + np;
#pragma unroll
        for (int kq = 0; kq < XQ; ++kq) xq[j][kq] = X[row * KT_XS + 4 * kq + q];
        rgbc[j] = xq[j][0];
        rdq[j] = RD[row * 5 + q];
        dotv[j] = RD[row * 5 + 3];
        mask[j] = R[row];
        rdsh[j] = q ? RD[row * 5 + q - 1] : 0.0f;                      // rgb_fc.0's quad [vis, rd0, rd1, rd2] (vis filled in later)
        rd3one[j] = q == 0 ? dotv[j] : (q == 1 ? 1.0f : 0.0f);        // ... and [rd3, one, 0, 0]
    }
    const float* tab = W.tab + q * 8;
#define KT_TAB(E, K) tab[(E) * 32 + (K)]

    // one group of the weight stream: NQ_ (<= 4) K quads of one M tile for the NT_ N tiles whose accumulators are ACC_(j) and whose
    // operands are B_(j, quad); requested two groups ahead
#define KT_GROUP(NT_, NQ_, ACC_, C0_, B_, Q0_)                                                           \
    {                                                                                                    \
        const float4 a_ = wb[par];                                                                       \
        wb[(par + 2) % 3] = wp[lane];                                                                    \
        wp += 64;                                                                                        \
        par = (par + 1) % 3;                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < (NT_); ++j_) ACC_(j_) = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.x, B_(j_, (Q0_)), C0_(j_), 0, 0, 0); \
        if ((NQ_) > 1) { _Pragma("unroll") for (int j_ = 0; j_ < (NT_); ++j_) ACC_(j_) = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.y, B_(j_, (Q0_) + 1), ACC_(j_), 0, 0, 0); } \
        if ((NQ_) > 2) { _Pragma("unroll") for (int j_ = 0; j_ < (NT_); ++j_) ACC_(j_) = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.z, B_(j_, (Q0_) + 2), ACC_(j_), 0, 0, 0); } \
        if ((NQ_) > 3) { _Pragma("unroll") for (int j_ = 0; j_ < (NT_); ++j_) ACC_(j_) = __builtin_amdgcn_mfma_f32_16x16x4f32(a_.w, B_(j_, (Q0_) + 3), ACC_(j_), 0, 0, 0); } \
    }
    // a whole product: M tiles MT_, K quads NQ_ (groups of 4), accumulators ACC_T_(j); the FIRST MFMA of a chain reads its C operand
    // from INIT_T_(j) -- the bias vector of the tile, shared by the four N tiles -- instead of from a copy of it in the accumulator
#define KT_PRODUCT(NT_, MT_, NQ_, ACC2_, B_)                                                             \
    _Pragma("unroll") for (int T_ = 0; T_ < (MT_); ++T_)                                                 \
        _Pragma("unroll") for (int g_ = 0; g_ < ((NQ_) + 3) / 4; ++g_) {                                 \
            if (g_ == 0) KT_GROUP(NT_, ((NQ_) < 4 ? (NQ_) : 4), ACC_T_, INIT_T_, B_, 0)                  \
            else KT_GROUP(NT_, ((NQ_) - 4 * g_ < 4 ? (NQ_) - 4 * g_ : 4), ACC_T_, ACC_T_, B_, 4 * g_)   \
        }

    // ---------------------------------------------------------------- ray_dir_fc (blending_network.py:36-39, 87)
    f32x4 D[KT_NT];
    {
        const f32x4 b = {KT_TAB(KT_RD1_B, 0), KT_TAB(KT_RD1_B, 1), KT_TAB(KT_RD1_B, 2), KT_TAB(KT_RD1_B, 3)};
#define ACC_T_(j) D[j]
#define INIT_T_(j) b
#define B_RD(j, k) rdq[j]
        KT_GROUP(KT_NT, 1, ACC_T_, INIT_T_, B_RD, 0)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
        for (int j = 0; j < KT_NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) D[j][i] = elu1t(D[j][i]);
    }
    {
        f32x4 E[XT][KT_NT], bias[XT];
#pragma unroll
        for (int T = 0; T < XT; ++T)
            bias[T] = (f32x4){KT_TAB(KT_RD2_B, 4 * T), KT_TAB(KT_RD2_B, 4 * T + 1), KT_TAB(KT_RD2_B, 4 * T + 2), KT_TAB(KT_RD2_B, 4 * T + 3)};
#define ACC_T_(j) E[T_][j]
#define INIT_T_(j) bias[T_]
#define B_D(j, k) D[j][k]
        KT_PRODUCT(KT_NT, XT, 4, E, B_D)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
        for (int j = 0; j < KT_NT; ++j)
#pragma unroll
            for (int kq = 0; kq < XQ; ++kq) xq[j][kq] += elu1t(E[kq >> 2][j][kq & 3]);      // x = rgb_feat + direction_feat (:89); the one's row is zero
    }

    // ---------------------------------------------------------------- view weights, weighted mean / variance (:93-101)
    float wn[KT_NT];
    float pm[NPT][XQ], pv[NPT][XQ];          // mean / variance of the wave's points as the operands of NPT N tiles of 16 points
    const bool dead = S == 3 && (np & 3) == 3;
    {
        // point p = 16 t + np of point tile t is row tile p / PPT, column G (p % PPT) (its first view), same lane group
        const int src = (G * (np % PPT) + 16 * q) * 4;          // byte address for ds_bpermute
#pragma unroll
        for (int t = 0; t < NPT; ++t)
#pragma unroll
            for (int kq = 0; kq < XQ; ++kq) { pm[t][kq] = 0.0f; pv[t][kq] = 0.0f; }
#pragma unroll
        for (int j = 0; j < KT_NT; ++j) {
            const float e = hw_exp(W.s_abs * (dotv[j] - 1.0f));
            const float mn = group_min<G>(dead ? __builtin_inff() : e);
            const float wr = dead ? 0.0f : (e - mn) * mask[j];
            wn[j] = wr / (group_sum<G>(wr) + 1e-8f);
#pragma unroll
            for (int kq = 0; kq < XQ; ++kq) {
                const float mean = group_sum<G>(wn[j] * xq[j][kq]);
                const float d = xq[j][kq] - mean;
                const float var = group_sum<G>(wn[j] * (d * d));
                const float tm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, mean)));
                const float tv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, var)));
#pragma unroll
                for (int t = 0; t < NPT; ++t) {
                    const bool sel = (16 * t + np) / PPT == j;
                    pm[t][kq] = sel ? tm : pm[t][kq];
                    pv[t][kq] = sel ? tv : pv[t][kq];
                }
            }
        }
    }

    // ---------------------------------------------------------------- base_fc (:103-104)
    f32x4 H1[4][KT_NT];           // base_fc.0's 64 outputs per N tile
    {
        f32x4 P[NPT][4];           // the mean / variance columns, once per POINT (N tiles of 16 points)
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define INIT_T_(j) zero4
#define ACC_T_(j) P[j][T_]
#define B_PM(j, k) ((k) < XQ ? pm[j][(k) < XQ ? (k) : 0] : pv[j][(k) < XQ ? 0 : (k) - XQ])
        KT_PRODUCT(NPT, 4, 2 * XQ, P, B_PM)
#undef ACC_T_
#undef INIT_T_
        // back to the rows: column n' of row tile j is point PPT j + n' / G = column (PPT j) % 16 + n' / G of point tile (PPT j) / 16
#pragma unroll
        for (int j = 0; j < KT_NT; ++j) {
            const int src = (((PPT * j) & 15) + np / G + 16 * q) * 4;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = P[(PPT * j) >> 4][T][i];
                    asm volatile("" : "+v"(v));      // (hipcc 7.2 otherwise replaces the four moves of a tile by ONE and splats its result)
                    H1[T][j][i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v)));
                }
        }
#define ACC_T_(j) H1[T_][j]
#define INIT_T_(j) H1[T_][j]
#define B_X(j, k) xq[j][k]
        KT_PRODUCT(KT_NT, 4, XQ, H1, B_X)          // + x's columns and the bias (slot F)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int j = 0; j < KT_NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) H1[T][j][i] = elu1t(H1[T][j][i]);
    }
    f32x4 XH[2][KT_NT];           // the 32-wide hidden state x
    {
        f32x4 bias[2];
#pragma unroll
        for (int T = 0; T < 2; ++T)
            bias[T] = (f32x4){KT_TAB(KT_B2_B, 4 * T), KT_TAB(KT_B2_B, 4 * T + 1), KT_TAB(KT_B2_B, 4 * T + 2), KT_TAB(KT_B2_B, 4 * T + 3)};
#define ACC_T_(j) XH[T_][j]
#define INIT_T_(j) bias[T_]
#define B_H1(j, k) H1[(k) >> 2][j][(k) & 3]
        KT_PRODUCT(KT_NT, 2, 16, XH, B_H1)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int j = 0; j < KT_NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) XH[T][j][i] = elu1t(XH[T][j][i]);
    }

    // a 32 -> 32 layer on SCALE(j) * x: G1 = elu(W (s x) + b)
    f32x4 G1[2][KT_NT];
#define B_XS(j, k) xs[j][k]
#define KT_LAYER32(ENTRY, SCALE)                                                                                      \
    {                                                                                                                 \
        float xs[KT_NT][8];                                                                                           \
        _Pragma("unroll") for (int j = 0; j < KT_NT; ++j)                                                             \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) xs[j][k] = XH[k >> 2][j][k & 3] * SCALE[j];                 \
        f32x4 bias[2];                                                                                                \
        _Pragma("unroll") for (int T = 0; T < 2; ++T)                                                                 \
            bias[T] = (f32x4){KT_TAB(ENTRY, 4 * T), KT_TAB(ENTRY, 4 * T + 1), KT_TAB(ENTRY, 4 * T + 2), KT_TAB(ENTRY, 4 * T + 3)}; \
        KT_PRODUCT(KT_NT, 2, 8, G1, B_XS)                                                                             \
        _Pragma("unroll") for (int T = 0; T < 2; ++T)                                                                 \
            _Pragma("unroll") for (int j = 0; j < KT_NT; ++j)                                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) G1[T][j][i] = elu1t(G1[T][j][i]);                       \
    }
    // dot product of G1 with a 32-float row given in quad layout, summed over the lane groups: every lane of the column gets it
#define KT_DOT32(ENTRY, OUT)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < KT_NT; ++j) {                                                               \
        float s_ = 0.0f;                                                                                              \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) s_ = __builtin_fmaf(G1[k >> 2][j][k & 3], KT_TAB(ENTRY, k), s_); \
        OUT[j] = lanes_q_sum(s_);                                                                                     \
    }

    // ---------------------------------------------------------------- vis_fc on x * weight (:106-109)
    float vis[KT_NT];
#define ACC_T_(j) G1[T_][j]
#define INIT_T_(j) bias[T_]
    KT_LAYER32(KT_V1_B, wn)
    KT_DOT32(KT_V2_LAST, vis)                                      // the 33rd output of vis_fc.2 reads the same hidden layer
#undef ACC_T_
#undef INIT_T_
    {
        f32x4 V2[2][KT_NT], bias[2];
#pragma unroll
        for (int T = 0; T < 2; ++T)
            bias[T] = (f32x4){KT_TAB(KT_V2_B, 4 * T), KT_TAB(KT_V2_B, 4 * T + 1), KT_TAB(KT_V2_B, 4 * T + 2), KT_TAB(KT_V2_B, 4 * T + 3)};
#define ACC_T_(j) V2[T_][j]
#define INIT_T_(j) bias[T_]
#define B_G1(j, k) G1[(k) >> 2][j][(k) & 3]
        KT_PRODUCT(KT_NT, 2, 8, V2, B_G1)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int j = 0; j < KT_NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) XH[T][j][i] += elu1t(V2[T][j][i]);              // x = x + x_res
    }
#pragma unroll
    for (int j = 0; j < KT_NT; ++j) vis[j] = hw_sigmoid(elu1t(vis[j] + W.v2_last_b)) * mask[j];

    // ---------------------------------------------------------------- vis_fc2 on x * vis (:110)
    float vis2[KT_NT];
#define ACC_T_(j) G1[T_][j]
#define INIT_T_(j) bias[T_]
    KT_LAYER32(KT_U1_B, vis)
    KT_DOT32(KT_U2, vis2)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
    for (int j = 0; j < KT_NT; ++j) vis2[j] = hw_sigmoid(vis2[j] + W.u2_b) * mask[j];

    // ---------------------------------------------------------------- rgb_fc on cat([x, vis, ray_diff]) (:113-115)
    float score[KT_NT];
    {
        f32x4 C1[KT_NT];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};                                            // (bias: the one of quad 9)
#define INIT_T_(j) zero4
#define ACC_T_(j) C1[j]
#define B_R1(j, k) ((k) < 8 ? XH[((k) < 8 ? (k) : 0) >> 2][j][(k) & 3] : (k) == 8 ? (q == 0 ? vis2[j] : rdsh[j]) : rd3one[j])
        KT_PRODUCT(KT_NT, 1, 10, C1, B_R1)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
        for (int j = 0; j < KT_NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) C1[j][i] = elu1t(C1[j][i]);
        f32x4 C2[KT_NT];
        const f32x4 b2 = {KT_TAB(KT_R2_B, 0), KT_TAB(KT_R2_B, 1), KT_TAB(KT_R2_B, 2), KT_TAB(KT_R2_B, 3)};
#define ACC_T_(j) C2[j]
#define INIT_T_(j) b2
#define B_C1(j, k) C1[j][k]
        KT_PRODUCT(KT_NT, 1, 4, C2, B_C1)
#undef ACC_T_
#undef INIT_T_
#pragma unroll
        for (int j = 0; j < KT_NT; ++j) {
            const float s = elu1t(C2[j][0]) * KT_TAB(KT_R3, 0) + elu1t(C2[j][1]) * KT_TAB(KT_R3, 1);      // features 0..7 = registers 0, 1 of the four groups
            score[j] = (mask[j] == 0.0f ? -1e9f : lanes_q_sum(s) + W.r3_b) + 0.0f * mask[j];              // masked_fill(mask == 0, -1e9)  (:115); NaN mask = poisoned row
            if (dead) score[j] = -__builtin_inff();                                                       // no such view: weight exactly 0 in the soft-max
        }
    }

    // ---------------------------------------------------------------- softmax over views, colour (:116-117)
#pragma unroll
    for (int j = 0; j < KT_NT; ++j) {
        const float mx = group_max<G>(score[j]);
        const float e = hw_exp(score[j] - mx);
        const float den = group_sum<G>(e);
        const float col = group_sum<G>(rgbc[j] * e) / den;
        const int64_t pt = first + PPT * j + np / G;
        if ((np % G) == 0 && q < 3 && pt < n) {
            const int64_t dst = index ? index[pt] : pt;
            rgb_out[3 * dst + q] = col;
        }
    }
#undef KT_TAB
#undef KT_GROUP
#undef KT_PRODUCT
#undef KT_LAYER32
#undef KT_DOT32
}

int gens_fill_maps(const char* who, MapSet* ms, const float* const* feats, const int* hw, int n_levels);

// number of float4-per-lane groups of the weight stream (without the two zero groups the kernel reads ahead); the same for every view count
extern "C" int gens_blend_views_t_groups(int n_levels) {
    if (n_levels < 1 || n_levels > 5) return 0;
    const int xq = n_levels + 1, xt = (xq + 3) / 4;
    return 1 + xt + 4 * ((2 * xq + 3) / 4) + 4 * ((xq + 3) / 4) + 2 * 4 + 2 * 2 + 2 * 2 + 2 * 2 + 3 + 1;
}
extern "C" int gens_blend_views4_groups(int n_levels) { return gens_blend_views_t_groups(n_levels); }

struct BlendTWeights;
static int blend_t_run(int n_levels, int nv, const BlendTWeights& W, const MapSet& fs, const float* imgs, const float* w2c, const float* intr,
                       const float* c2w, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out,
                       uint8_t* vis_out, void* stream);

template <int S>
static void blend_t_launch(int n_levels, unsigned grid, hipStream_t st, const BlendTWeights& W, const MapSet& fs, const float* imgs, const float* w2c,
                           const float* intr, const float* c2w, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device,
                           float* rgb_out, uint8_t* vis_out) {
#define BLEND_LAUNCH(NL) blend_t_k<NL, S><<<grid, 64, 0, st>>>(W, fs, (const float4*)imgs, w2c, intr, c2w, pts, index, n, n_device, rgb_out, vis_out)
    switch (n_levels) {
        case 1: BLEND_LAUNCH(1); break;
        case 2: BLEND_LAUNCH(2); break;
        case 3: BLEND_LAUNCH(3); break;
        case 4: BLEND_LAUNCH(4); break;
        default: BLEND_LAUNCH(5); break;
    }
#undef BLEND_LAUNCH
}

extern "C" int gens_blend_views_t(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                  const float* c2w, int nv, const float* wstream, const float* tab, const float* scalars, const float* pts,
                                  const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream) {
    MapSet fs;
    GENS_CHECK_ARG(feats && wstream && tab && scalars, GENS_EINVAL, "gens_blend_views_t: null table");
    if (int e = gens_fill_maps("gens_blend_views_t", &fs, feats, hw, n_levels)) return e;
    GENS_CHECK_ARG(n_levels <= 5, GENS_ELIMIT, "gens_blend_views_t: at most 5 feature levels (d_feature <= 20), got %d", n_levels);
    GENS_CHECK_ARG(nv >= 3 && nv <= 5, GENS_ELIMIT, "gens_blend_views_t: built for two to four source views (nv = 3..5), got nv=%d (use gens_blend_views)", nv);
    GENS_CHECK_ARG(imgs && w2c && intr && c2w, GENS_EINVAL, "gens_blend_views_t: null camera / image pointer");
    GENS_CHECK_ARG(((uintptr_t)wstream & 15) == 0, GENS_EINVAL, "gens_blend_views_t: the weight stream must be 16-byte aligned");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && rgb_out)), GENS_EINVAL, "gens_blend_views_t: null pts / output");
    if (n == 0) return 0;
    BlendTWeights W;
    W.stream = (const float4*)wstream;
    W.tab = tab;
    W.v2_last_b = scalars[0]; W.u2_b = scalars[1]; W.r3_b = scalars[2]; W.s_abs = scalars[3];
    W.sdev = nullptr;
    return blend_t_run(n_levels, nv, W, fs, imgs, w2c, intr, c2w, pts, index, n, n_device, rgb_out, vis_out, stream);
}

static int blend_t_run(int n_levels, int nv, const BlendTWeights& W, const MapSet& fs, const float* imgs, const float* w2c, const float* intr,
                       const float* c2w, const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out,
                       uint8_t* vis_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nv == 3) blend_t_launch<2>(n_levels, gens_blocks(n, 32), st, W, fs, imgs, w2c, intr, c2w, pts, index, n, n_device, rgb_out, vis_out);
    else if (nv == 4) blend_t_launch<3>(n_levels, gens_blocks(n, 16), st, W, fs, imgs, w2c, intr, c2w, pts, index, n, n_device, rgb_out, vis_out);
    else blend_t_launch<4>(n_levels, gens_blocks(n, 16), st, W, fs, imgs, w2c, intr, c2w, pts, index, n, n_device, rgb_out, vis_out);
    return gens_launch_status("gens_blend_views_t");
}

// (ABI version 4 name: four source views only)
extern "C" int gens_blend_views4(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                 const float* c2w, int nv, const float* wstream, const float* tab, const float* scalars, const float* pts,
                                 const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream) {
    GENS_CHECK_ARG(nv == 5, GENS_ELIMIT, "gens_blend_views4: built for four source views (nv = 5), got nv=%d (use gens_blend_views_t / gens_blend_views)", nv);
    return gens_blend_views_t(feats, hw, n_levels, imgs, w2c, intr, c2w, nv, wstream, tab, scalars, pts, index, n, n_device, rgb_out, vis_out, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The weight stream, tables and scalars of gens_blend_views_t straight from the 23 RAW nn.Linear parameters of a BlendingNetwork
// (order of gens_blend_train_fwd), in one launch: a training step's colour network changes every step, and gens_amd.ops._pack_blend_t
// builds the same arrays with ~100 PyTorch launches.  Layout as documented at _pack_blend_t / in this file's header.
// ---------------------------------------------------------------------------------------------------------------------------
struct BlendPackT {
    const float* p[23];
    int f;                       // 3 + 4 NLEV
    float4* stream;
    float* tab;                  // (10, 4, 8)
    float* scalars;              // (4) DEVICE
    int n_groups;                // without the two trailing zero groups
};

__device__ __forceinline__ float bpt_entry(const BlendPackT& A, int prod, int row, int slot, int xq) {
    const int f = A.f;
    int wi, O, I, col;            // parameter index of the weight, its shape, the source column (-1 bias, -2 nothing)
    bool has_bias = false;
    switch (prod) {
        case 0: wi = 0; O = 16; I = 4; col = slot < 4 ? slot : -2; break;                                          // ray_dir_fc.0
        case 1: wi = 2; O = f; I = 16; col = slot < 16 ? slot : -2; break;                                         // ray_dir_fc.2
        case 2: wi = 4; O = 64; I = 3 * f;                                                                          // base_fc.0: mean | var
            col = slot < 4 * xq ? (slot < f ? slot : -2) : (slot - 4 * xq < f ? f + slot - 4 * xq : -2); break;
        case 3: wi = 4; O = 64; I = 3 * f; has_bias = true; col = slot < f ? 2 * f + slot : (slot == f ? -1 : -2); break;   // base_fc.0: x | bias
        case 4: wi = 6; O = 32; I = 64; col = slot < 64 ? slot : -2; break;                                        // base_fc.2
        case 5: wi = 8; O = 32; I = 32; col = slot < 32 ? slot : -2; break;                                        // vis_fc.0
        case 6: wi = 10; O = 32; I = 32; col = slot < 32 ? slot : -2; break;                                       // vis_fc.2 rows 0..31
        case 7: wi = 12; O = 32; I = 32; col = slot < 32 ? slot : -2; break;                                       // vis_fc2.0
        case 8: wi = 16; O = 16; I = 37; has_bias = true; col = slot < 37 ? slot : (slot == 37 ? -1 : -2); break;  // rgb_fc.0
        default: wi = 18; O = 8; I = 16; col = slot < 16 ? slot : -2; break;                                       // rgb_fc.2
    }
    if (row >= O || col == -2) return 0.0f;
    if (col == -1) return has_bias ? A.p[wi + 1][row] : 0.0f;
    return A.p[wi][(int64_t)row * I + col];
}

__global__ __launch_bounds__(256) void blend_pack_t_k(BlendPackT A) {
    const int f = A.f, xq = (f + 1) / 4, xt = (xq + 3) / 4;
    const int nq[10] = {1, 4, 2 * xq, xq, 16, 8, 8, 8, 10, 4};
    const int mt[10] = {1, xt, 4, 4, 2, 2, 2, 2, 1, 1};
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int n_stream = (A.n_groups + 2) * 64;
    if (gid < n_stream) {
        const int grp = gid >> 6, lane = gid & 63;
        float4 v = f4_zero();
        if (grp < A.n_groups) {
            int prod = 0, g0 = 0;
            for (; prod < 10; ++prod) {
                const int gp = mt[prod] * ((nq[prod] + 3) / 4);
                if (grp < g0 + gp) break;
                g0 += gp;
            }
            const int gq = (nq[prod] + 3) / 4, t = (grp - g0) / gq, g = (grp - g0) % gq;
            const int m = lane & 15, qk = lane >> 4;
            const int row = 16 * t + 4 * (m & 3) + (m >> 2);
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kq = 4 * g + j;
                e[j] = kq < nq[prod] ? bpt_entry(A, prod, row, 4 * kq + qk, xq) : 0.0f;
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
        }
        A.stream[gid] = v;
        return;
    }
    const int k = gid - n_stream;
    if (k < 10 * 32) {                                     // tab[entry][q][c]
        const int entry = k >> 5, q = (k >> 3) & 3, c = k & 7;
        float v = 0.0f;
        if (entry < 7) {                                   // accumulator-layout bias: [q][4 T + i] = b[16 T + 4 i + q]
            const int bi[7] = {1, 3, 7, 9, 11, 13, 19}, len[7] = {16, f, 32, 32, 32, 32, 8}, tiles[7] = {1, xt, 2, 2, 2, 2, 1};
            const int T = c >> 2, i = c & 3, src = 16 * T + 4 * i + q;
            if (T < tiles[entry] && src < len[entry]) v = A.p[bi[entry]][src];
        } else {                                           // dot-product row: [q][kq] = w[4 kq + q]
            const int src = 4 * c + q;
            if (entry == 7) v = A.p[10][32 * 32 + src];                        // row 32 of vis_fc.2 (33 x 32)
            else if (entry == 8) v = A.p[14][src];                             // vis_fc2.2 (1 x 32)
            else v = src < 8 ? A.p[20][src] : 0.0f;                            // rgb_fc.4 (1 x 8)
        }
        A.tab[k] = v;
    } else if (k == 10 * 32) {
        A.scalars[0] = A.p[11][32];                        // vis_fc.2 bias[32]
        A.scalars[1] = A.p[15][0];                         // vis_fc2.2 bias
        A.scalars[2] = A.p[21][0];                         // rgb_fc.4 bias
        A.scalars[3] = fabsf(A.p[22][0]);                  // |s|
    }
}

extern "C" int gens_blend_pack_t(const float* const* weights, int n_levels, float* wstream, float* tab, float* scalars_dev, void* stream) {
    GENS_CHECK_ARG(weights && wstream && tab && scalars_dev, GENS_EINVAL, "gens_blend_pack_t: null pointer");
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= 5, GENS_ELIMIT, "gens_blend_pack_t: 1..5 feature levels, got %d", n_levels);
    GENS_CHECK_ARG(((uintptr_t)wstream & 15) == 0, GENS_EINVAL, "gens_blend_pack_t: the weight stream must be 16-byte aligned");
    BlendPackT A;
    for (int k = 0; k < 23; ++k) {
        GENS_CHECK_ARG(weights[k], GENS_EINVAL, "gens_blend_pack_t: weight %d is null", k);
        A.p[k] = weights[k];
    }
    A.f = 3 + 4 * n_levels;
    A.stream = (float4*)wstream;
    A.tab = tab;
    A.scalars = scalars_dev;
    A.n_groups = gens_blend_views_t_groups(n_levels);
    const int work = (A.n_groups + 2) * 64 + 10 * 32 + 1;
    blend_pack_t_k<<<gens_blocks(work, 256), 256, 0, (hipStream_t)stream>>>(A);
    return gens_launch_status("gens_blend_pack_t");
}

// gens_blend_views_t with the four scalars read from DEVICE memory (scalars_dev of gens_blend_pack_t)
extern "C" int gens_blend_views_t_dev(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c, const float* intr,
                                      const float* c2w, int nv, const float* wstream, const float* tab, const float* scalars_dev, const float* pts,
                                      const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out, void* stream) {
    MapSet fs;
    GENS_CHECK_ARG(feats && wstream && tab && scalars_dev, GENS_EINVAL, "gens_blend_views_t_dev: null table");
    if (int e = gens_fill_maps("gens_blend_views_t_dev", &fs, feats, hw, n_levels)) return e;
    GENS_CHECK_ARG(n_levels <= 5, GENS_ELIMIT, "gens_blend_views_t_dev: at most 5 feature levels (d_feature <= 20), got %d", n_levels);
    GENS_CHECK_ARG(nv >= 3 && nv <= 5, GENS_ELIMIT, "gens_blend_views_t_dev: built for two to four source views (nv = 3..5), got nv=%d", nv);
    GENS_CHECK_ARG(imgs && w2c && intr && c2w, GENS_EINVAL, "gens_blend_views_t_dev: null camera / image pointer");
    GENS_CHECK_ARG(((uintptr_t)wstream & 15) == 0, GENS_EINVAL, "gens_blend_views_t_dev: the weight stream must be 16-byte aligned");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && rgb_out)), GENS_EINVAL, "gens_blend_views_t_dev: null pts / output");
    if (n == 0) return 0;
    BlendTWeights W;
    W.stream = (const float4*)wstream;
    W.tab = tab;
    W.v2_last_b = W.u2_b = W.r3_b = W.s_abs = 0.0f;
    W.sdev = scalars_dev;
    return blend_t_run(n_levels, nv, W, fs, imgs, w2c, intr, c2w, pts, index, n, n_device, rgb_out, vis_out, stream);
}
