// K7: source-view feature look-up (K4) + the whole IBRNet-style BlendingNetwork in ONE kernel, inference only.
//
// Replaces, for validation rendering, lookup_feature + compute_angle (/root/reference/models/modules/projector.py:278-349)
// followed by BlendingNetwork.forward (models/modules/blending_network.py:69-118) as called from
// implicit_surface.py:196-199.  The (N, S, 23) feature tensor, the (N, S, 4) ray-difference tensor and every
// activation of the eleven small linear layers stay in LDS / registers; HBM sees 12 B per point in and 12 + S bytes out.
//
// Mapping: one wavefront owns 32 (point, source-view) rows = floor(32/S) points and runs every layer on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact float32) with the rows as the M dimension; waves are independent.
// Cross-view operations (softmax over views, weighted mean / variance, min over views) read the sibling rows from
// the wave's LDS tile.  Weights (43 KB) are pre-packed in B-fragment order and stay in L1/L2.
#include "k4_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BL_WAVES 1          // wavefronts per workgroup: waves are independent, a 1-wave workgroup makes every barrier free
#define BL_TS 73            // row stride of the wide tile (K <= 72: 3F = 69 padded to whole groups of 8)
#define BL_VS 65            // row stride of the 64-wide tile
#define BL_HS 33            // row stride of the hidden tile
#define BL_MAXF 23          // 3 + 4*5 feature columns
#define BL_RS 9             // row stride of the ray-difference tile (4 values + 4 zero columns: one group of 8)

struct BlendWeights {
    const float *rd1, *rd1_b, *rd2, *rd2_b;   // ray_dir_fc: 4 -> 16 -> F
    const float *b1, *b1_b, *b2, *b2_b;       // base_fc:    3F -> 64 -> 32
    const float *v1, *v1_b, *v2, *v2_b;       // vis_fc:     32 -> 32 -> 33 (rows 0..31 packed; row 32 = v2_last)
    const float *v2_last;                     // (32) weights of vis_fc output 32, bias v2_last_b
    const float *u1, *u1_b, *u2;              // vis_fc2:    32 -> 32 -> 1
    const float *r1, *r1_b, *r2, *r2_b, *r3;  // rgb_fc:     37 -> 16 -> 8 -> 1
    float v2_last_b, u2_b, r3_b, s_abs;
};

// ELU as ONE median: exp(x) - 1 >= x for every x, with exp(x) - 1 >= 0 exactly when x >= 0, so elu(x) = med3(x, exp(x) - 1, 0) -- x on
// the positive side (x lies between 0 and exp(x) - 1, also when that is inf), exp(x) - 1 on the negative one (it lies between x and 0):
// v_med3_f32 instead of a compare + select (136 of them per 32-row tile; the kernel's time is its MFMA cycles PLUS its VALU issue cycles).
__device__ __forceinline__ float elu1(float x) { return __builtin_amdgcn_fmed3f(x, hw_exp(x) - 1.0f, 0.0f); }
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// One 32 x 32 output tile: acc = bias + A(32 x 8G, LDS rows of stride rs) * B.  The reduction index runs in groups of 8:
// within group j the four MFMAs i = 0..3 take k = 8j + 4h + i from lane half h, so a lane's B values for a group are ONE
// global_load_dwordx4 of the host-packed stream (gens_amd.ops._pack_b_groups); all G loads of a tile are issued before its
// first MFMA (one exposed L2 latency per tile instead of one per four MFMAs).
template <int G>
struct BGroups {
    float4 v[G];
};
template <int G>
__device__ __forceinline__ void load_b(BGroups<G>& b, const float* __restrict__ wp, int lane) {
    const float4* p = (const float4*)wp + lane;
#pragma unroll
    for (int j = 0; j < G; ++j) b.v[j] = p[64 * j];
}
template <int G>
__device__ __forceinline__ f32x16 tile_mfma(const float* __restrict__ a_lds, int rs, const BGroups<G>& b, float bias, int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias;
    const float* a = a_lds + (lane & 31) * rs + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < G; ++j) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * j + 0], b.v[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * j + 1], b.v[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * j + 2], b.v[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * j + 3], b.v[j].w, acc, 0, 0, 0);
    }
    return acc;
}

// Narrow layers (<= 16 output columns: ray_dir_fc.0, rgb_fc.0, rgb_fc.2): the 32 rows x 16 columns are TWO 16 x 16 tiles of
// v_mfma_f32_16x16x4_f32 (same exact float32, 32 cycles each) instead of one half-empty 32 x 32 tile -- half the matrix cycles and half
// the activation evaluations (4 + 4 accumulator registers per lane instead of 16).  The reduction runs in groups of 4 S columns: at
// step s lane (q = lane / 16, j = lane % 16) contributes k = k0 + S q + s, so its S values of A are consecutive floats of its LDS row
// and its S values of B consecutive floats of the host-packed stream (gens_amd.ops._pack_b16).  Tile t, register r of a lane is
// output (row 16 t + 4 q + r, column j).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int S>
__device__ __forceinline__ void narrow_group(const float* __restrict__ a_lds, int rs, int k0, const float* __restrict__ wp, int lane, f32x4& c0,
                                             f32x4& c1) {
    const int q = lane >> 4, j = lane & 15;
    const float* a0 = a_lds + j * rs + k0 + S * q;
    const float* a1 = a0 + 16 * rs;
    float b[S];
#pragma unroll
    for (int s = 0; s < S; ++s) b[s] = wp[lane * S + s];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[s], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[s], c1, 0, 0, 0);
    }
}
__device__ __forceinline__ int nrow(int t, int r, int lane) { return 16 * t + 4 * (lane >> 4) + r; }

template <int NLEV>
__global__ __launch_bounds__(64 * BL_WAVES) void blend_k(BlendWeights W, MapSet fs, const float4* __restrict__ imgs,
                                                         const float* __restrict__ w2c, const float* __restrict__ intr,
                                                         const float* __restrict__ c2w, int nv, const float* __restrict__ pts,
                                                         const int64_t* __restrict__ index, int64_t n_max, const int32_t* __restrict__ n_dev,
                                                         float* __restrict__ rgb_out, uint8_t* __restrict__ vis_out) {
    __shared__ float T_[BL_WAVES][32 * BL_TS];
    __shared__ float RD_[BL_WAVES][32 * BL_RS];
    __shared__ float R_[BL_WAVES][32 * 8];     // per-row scalars: 0 mask, 1 e, 2 w, 3 w normalised, 4 vis, 5 vis2, 6 score
    __shared__ float C_[BL_WAVES][32 * 3];     // rgb_in
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* T = T_[wave];
    // V (64-wide tile, stride BL_VS) ALIASES T: a wave finishes every MFMA read of a tile before its epilogue writes, and
    // the two layouts are never live together (T: until base_fc.0 has been read; again from rgb_fc.0's output on).
    float* V = T_[wave];
    // D (ray_dir_fc hidden layer, 16 columns) lives in T's columns D_OFF.., which are free until the mean / variance are written
    float* RD = RD_[wave];
    float* R = R_[wave];
    float* C = C_[wave];
    constexpr int F = 3 + 4 * NLEV;
    constexpr int G_B1 = (3 * F + 7) / 8;      // reduction groups of base_fc.0
    static_assert(8 * G_B1 <= BL_TS, "wide tile too narrow");
    // columns of the 16-wide ray_dir_fc hidden layer D inside T: 0..15 while x (columns 2F..3F) starts behind them; with one feature level
    // (F = 7: x at 14..20) behind base_fc.0's padded reduction instead
    constexpr int D_OFF = (2 * F >= 16) ? 0 : 8 * G_B1;
    static_assert(D_OFF + 16 <= BL_TS, "no room for the ray_dir_fc hidden layer");
    const int S = nv - 1, PPW = 32 / S;
    const int64_t first = ((int64_t)blockIdx.x * BL_WAVES + wave) * PPW;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    if ((int64_t)blockIdx.x * BL_WAVES * PPW >= n) return;
    const int row = lane & 31, half = lane >> 5;
    const int pl = row / S, sv = row % S + 1;
    const bool live = pl < PPW && first + pl < n;
    const int64_t src = live ? (index ? index[first + pl] : first + pl) : 0;
    const int col = lane & 31;

    // ---------------------------------------------------------------- phase 0: K4 look-up into T[:, 2F..3F)
    {
        float x = 0.f, y = 0.f, z = 0.f;
        if (live) { x = pts[3 * src]; y = pts[3 * src + 1]; z = pts[3 * src + 2]; }
        bool inside = true;
        const int l_begin = half ? 2 : 0, l_end = half ? NLEV : min(2, NLEV);
        float* xr = T + row * BL_TS + 2 * F;
        const SrcBase pb = project_src_base(w2c + 16 * sv, intr + 16 * sv, x, y, z);     // once per (point, view): see k4_common.h
        for (int l = l_begin; l < l_end; ++l) {
            const int h = fs.h[l], w = fs.w[l];
            SrcProj p = project_src_level(pb, exp2f(-(float)l), h, w, fs.cw[l], fs.ch[l], fs.rcw[l], fs.rch[l]);
            inside = inside && p.inside;
            float4 f = f4_zero(), c = f4_zero();
            if (live) {
                Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
                f = sample_texel(fs.data[l] + (int64_t)sv * h * w, h, w, 1, 0, t);
                if (l == 0) c = sample_texel(imgs + (int64_t)sv * h * w, h, w, 1, 0, t);
            }
            xr[3 + 4 * l] = f.x; xr[4 + 4 * l] = f.y; xr[5 + 4 * l] = f.z; xr[6 + 4 * l] = f.w;
            if (l == 0) {
                xr[0] = c.x; xr[1] = c.y; xr[2] = c.z;
                C[row * 3] = c.x; C[row * 3 + 1] = c.y; C[row * 3 + 2] = c.z;
            }
        }
        const bool other = __shfl_xor((int)inside, 32, 64) != 0;
        inside = inside && other;
        // (not-a-number inputs must come out as not-a-number: the median form of the ELU would drop them, so the row's mask carries a poison
        // term 0 * (sum of its inputs) -- the mask multiplies the view weights, the visibilities and is added to the score)
        float acc_in = x + y + z;
        for (int l = l_begin; l < l_end; ++l) acc_in += xr[3 + 4 * l] + xr[4 + 4 * l] + xr[5 + 4 * l] + xr[6 + 4 * l] + (l == 0 ? xr[0] + xr[1] + xr[2] : 0.0f);
        acc_in += __shfl_xor(acc_in, 32, 64);
        if (half == 0) {
            R[row * 8] = ((live && inside) ? 1.0f : 0.0f) + 0.0f * acc_in;
            if (live && vis_out) vis_out[src * S + (sv - 1)] = inside ? 1 : 0;
            // compute_angle (projector.py:278-291)
            float rx = c2w[3] - x, ry = c2w[7] - y, rz = c2w[11] - z;
            // normalisations with v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sqrt / division sequences (3 + 9 of them)
            float rn = hw_rcp(__builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz) + 1e-6f);
            rx *= rn; ry *= rn; rz *= rn;
            const float* cs = c2w + 16 * sv;
            float sx = cs[3] - x, sy = cs[7] - y, sz = cs[11] - z;
            float sn = hw_rcp(__builtin_amdgcn_sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f);
            sx *= sn; sy *= sn; sz *= sn;
            float dx = rx - sx, dy = ry - sy, dz = rz - sz;
            float dn = hw_rcp(fmaxf(__builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f));
            float* rd = RD + row * BL_RS;
            rd[0] = live ? dx * dn : 0.0f;
            rd[1] = live ? dy * dn : 0.0f;
            rd[2] = live ? dz * dn : 0.0f;
            rd[3] = live ? rx * sx + ry * sy + rz * sz : 0.0f;
            rd[4] = rd[5] = rd[6] = rd[7] = 0.0f;                                       // K padding of ray_dir_fc.0 (4 -> 8)
#pragma unroll
            for (int k = 3 * F; k < 8 * G_B1; ++k) T[row * BL_TS + k] = 0.0f;          // K padding of base_fc.0 (3F -> whole groups)
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- ray_dir_fc (blending_network.py:36-39, 87)
    {
        const int j = lane & 15;
        f32x4 c0, c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[r] = c1[r] = W.rd1_b[j];
        narrow_group<2>(RD, BL_RS, 0, W.rd1, lane, c0, c1);                              // K = 4 (+4 zero columns)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T[nrow(0, r, lane) * BL_TS + D_OFF + j] = elu1(c0[r]);
            T[nrow(1, r, lane) * BL_TS + D_OFF + j] = elu1(c1[r]);
        }
    }
    __syncthreads();
    {
        BGroups<2> w;
        load_b(w, W.rd2, lane);
        f32x16 a = tile_mfma(T + D_OFF, BL_TS, w, W.rd2_b[col], lane);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (col < F) T[crow(r, lane) * BL_TS + 2 * F + col] += elu1(a[r]);        // x = rgb_feat + direction_feat (:89)
        if (half == 0) R[row * 8 + 1] = hw_exp(W.s_abs * (RD[row * BL_RS + 3] - 1.0f));     // exp(|s| (dot - 1))  (:93)
    }
    __syncthreads();

    // ---------------------------------------------------------------- view weights, weighted mean / variance (:94-101)
    if (S == 4) {
        // four source views = the four lanes of a DPP quad: min / sum over the views are two quad_perm butterflies in
        // registers (all four lanes end with the same value), no LDS round trips and no barriers
        const float e = R[row * 8 + 1], m = R[row * 8];
        float mn = fminf(e, dpp_move<0xB1, 0xF>(e, e));                 // quad_perm:[1,0,3,2]
        mn = fminf(mn, dpp_move<0x4E, 0xF>(mn, mn));                    // quad_perm:[2,3,0,1]
        const float wr = (e - mn) * m;
        float sum = wr + dpp_move<0xB1, 0xF>(wr, wr);
        sum = sum + dpp_move<0x4E, 0xF>(sum, sum);
        const float wn = wr / (sum + 1e-8f);
        if (half == 0) R[row * 8 + 3] = wn;
        // each lane owns its (point, view) row and every second feature column: mean / variance through the same butterflies
        for (int c = half; c < F; c += 2) {
            const float x = T[row * BL_TS + 2 * F + c];
            float mean = wn * x;
            mean += dpp_move<0xB1, 0xF>(mean, mean);
            mean += dpp_move<0x4E, 0xF>(mean, mean);
            const float d = x - mean;
            float var = wn * (d * d);
            var += dpp_move<0xB1, 0xF>(var, var);
            var += dpp_move<0x4E, 0xF>(var, var);
            T[row * BL_TS + c] = mean;
            T[row * BL_TS + F + c] = var;
        }
        __syncthreads();
    } else {
        if (half == 0) {
            const int base = pl * S;
            float mn = 3.4e38f;
            if (pl < PPW) for (int v = 0; v < S; ++v) mn = fminf(mn, R[(base + v) * 8 + 1]);
            R[row * 8 + 2] = (pl < PPW) ? (R[row * 8 + 1] - mn) * R[row * 8] : 0.0f;
        }
        __syncthreads();
        if (half == 0) {
            const int base = pl * S;
            float sum = 0.0f;
            if (pl < PPW) for (int v = 0; v < S; ++v) sum += R[(base + v) * 8 + 2];
            R[row * 8 + 3] = R[row * 8 + 2] / (sum + 1e-8f);
        }
        __syncthreads();
        for (int it = lane; it < PPW * F; it += 64) {
            const int p = it / F, c = it % F, base = p * S;
            float mean = 0.0f, var = 0.0f;
            for (int v = 0; v < S; ++v) mean += T[(base + v) * BL_TS + 2 * F + c] * R[(base + v) * 8 + 3];
            for (int v = 0; v < S; ++v) {
                float d = T[(base + v) * BL_TS + 2 * F + c] - mean;
                var += R[(base + v) * 8 + 3] * (d * d);
            }
            for (int v = 0; v < S; ++v) {
                T[(base + v) * BL_TS + c] = mean;
                T[(base + v) * BL_TS + F + c] = var;
            }
        }
        for (int it = lane; it < (32 - PPW * S) * 2 * F; it += 64) {      // unused rows (32 % S != 0): keep them finite
            const int rr = PPW * S + it / (2 * F);
            T[rr * BL_TS + it % (2 * F)] = 0.0f;
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- base_fc (:103-104)
    {
        BGroups<G_B1> w0, w1;
        load_b(w0, W.b1, lane);
        load_b(w1, W.b1 + (size_t)G_B1 * 256, lane);
        f32x16 a0 = tile_mfma(T, BL_TS, w0, W.b1_b[col], lane);
        f32x16 a1 = tile_mfma(T, BL_TS, w1, W.b1_b[32 + col], lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            V[crow(r, lane) * BL_VS + col] = elu1(a0[r]);
            V[crow(r, lane) * BL_VS + 32 + col] = elu1(a1[r]);
        }
    }
    __syncthreads();
    f32x16 h;
    {
        BGroups<8> w;
        load_b(w, W.b2, lane);
        h = tile_mfma(V, BL_VS, w, W.b2_b[col], lane);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = elu1(h[r]);
    __syncthreads();                                           // every lane has finished reading V
    // ---------------------------------------------------------------- vis_fc on x * weight (:106-109)
#pragma unroll
    for (int r = 0; r < 16; ++r) V[crow(r, lane) * BL_VS + col] = h[r] * R[crow(r, lane) * 8 + 3];
    __syncthreads();
    {
        BGroups<4> w;
        load_b(w, W.v1, lane);
        f32x16 a = tile_mfma(V, BL_VS, w, W.v1_b[col], lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) V[crow(r, lane) * BL_VS + 32 + col] = elu1(a[r]);
    }
    __syncthreads();
    {
        BGroups<4> w;
        load_b(w, W.v2, lane);
        f32x16 a = tile_mfma(V + 32, BL_VS, w, W.v2_b[col], lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] += elu1(a[r]);                          // x = x + x_res
        {                                                                        // 33rd output -> vis: the two lane halves of a row take
            float s = 0.0f;                                                      // 16 terms each (a half-masked 32-term loop issues the same
            for (int k = 0; k < 16; ++k) s += V[row * BL_VS + 32 + 16 * half + k] * W.v2_last[16 * half + k];   // instructions for half the work)
            s += __shfl_xor(s, 32, 64);
            if (half == 0) R[row * 8 + 4] = hw_sigmoid(elu1(s + W.v2_last_b)) * R[row * 8];
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- vis_fc2 on x * vis (:110)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = crow(r, lane);
        V[rr * BL_VS + col] = h[r] * R[rr * 8 + 4];
    }
    __syncthreads();
    {
        BGroups<4> w;
        load_b(w, W.u1, lane);
        f32x16 a = tile_mfma(V, BL_VS, w, W.u1_b[col], lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) V[crow(r, lane) * BL_VS + 32 + col] = elu1(a[r]);
    }
    __syncthreads();
    {
        float s = 0.0f;
        for (int k = 0; k < 16; ++k) s += V[row * BL_VS + 32 + 16 * half + k] * W.u2[16 * half + k];
        s += __shfl_xor(s, 32, 64);
        if (half == 0) R[row * 8 + 5] = hw_sigmoid(s + W.u2_b) * R[row * 8];
    }
    __syncthreads();
    // ---------------------------------------------------------------- rgb_fc on cat([x, vis, ray_diff]) (:113-114)
#pragma unroll
    for (int r = 0; r < 16; ++r) V[crow(r, lane) * BL_VS + col] = h[r];
    if (half == 0) {
        float* vr = V + row * BL_VS;
        vr[32] = R[row * 8 + 5];
        vr[33] = RD[row * BL_RS]; vr[34] = RD[row * BL_RS + 1]; vr[35] = RD[row * BL_RS + 2]; vr[36] = RD[row * BL_RS + 3];
        vr[37] = vr[38] = vr[39] = 0.0f;                                             // K padding of rgb_fc.0 (37 -> 40)
    }
    __syncthreads();
    {
        const int j = lane & 15;
        f32x4 c0, c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[r] = c1[r] = W.r1_b[j];
        narrow_group<4>(V, BL_VS, 0, W.r1, lane, c0, c1);                                // K = 37 (+3 zero columns) = 16 + 16 + 8
        narrow_group<4>(V, BL_VS, 16, W.r1 + 256, lane, c0, c1);
        narrow_group<2>(V, BL_VS, 32, W.r1 + 512, lane, c0, c1);
        // (T aliases V: a wave's LDS instructions execute in order, so the A reads above precede the writes below)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T[nrow(0, r, lane) * BL_TS + j] = elu1(c0[r]);
            T[nrow(1, r, lane) * BL_TS + j] = elu1(c1[r]);
        }
    }
    __syncthreads();
    {
        const int j = lane & 15;
        f32x4 c0, c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[r] = c1[r] = W.r2_b[j];
        narrow_group<4>(T, BL_TS, 0, W.r2, lane, c0, c1);                                // K = 16, 8 output columns
        if (j < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                T[nrow(0, r, lane) * BL_TS + 32 + j] = elu1(c0[r]);
                T[nrow(1, r, lane) * BL_TS + 32 + j] = elu1(c1[r]);
            }
        }
    }
    __syncthreads();
    if (half == 0) {
        float s = W.r3_b;
        for (int k = 0; k < 8; ++k) s += T[row * BL_TS + 32 + k] * W.r3[k];
        R[row * 8 + 6] = ((R[row * 8] == 0.0f) ? -1e9f : s) + 0.0f * R[row * 8];    // masked_fill(mask == 0, -1e9)  (:115); NaN mask = poisoned row
    }
    __syncthreads();
    // ---------------------------------------------------------------- softmax over views, colour (:116-117)
    if (lane < PPW && first + lane < n) {
        const int base = lane * S;
        float mx = -3.4e38f;
        for (int v = 0; v < S; ++v) mx = fmaxf(mx, R[(base + v) * 8 + 6]);
        float den = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        for (int v = 0; v < S; ++v) {
            float e = hw_exp(R[(base + v) * 8 + 6] - mx);
            den += e;
            cr += C[(base + v) * 3] * e;
            cg += C[(base + v) * 3 + 1] * e;
            cb += C[(base + v) * 3 + 2] * e;
        }
        const int64_t dst = index ? index[first + lane] : first + lane;
        rgb_out[3 * dst] = cr / den;
        rgb_out[3 * dst + 1] = cg / den;
        rgb_out[3 * dst + 2] = cb / den;
    }
}

int gens_fill_maps(const char* who, MapSet* ms, const float* const* feats, const int* hw, int n_levels);

extern "C" int gens_blend_views(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c,
                                const float* intr, const float* c2w, int nv, const float* const* weights, const float* scalars,
                                const float* pts, const int64_t* index, int64_t n, const int32_t* n_device, float* rgb_out, uint8_t* vis_out,
                                void* stream) {
    MapSet fs;
    GENS_CHECK_ARG(feats && weights && scalars, GENS_EINVAL, "gens_blend_views: null table");
    if (int e = gens_fill_maps("gens_blend_views", &fs, feats, hw, n_levels)) return e;
    GENS_CHECK_ARG(n_levels <= 5, GENS_ELIMIT, "gens_blend_views: at most 5 feature levels (d_feature <= 20), got %d", n_levels);
    GENS_CHECK_ARG(nv >= 2 && nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "gens_blend_views: nv=%d not in 2..%d", nv, GENS_MAX_VIEWS);
    GENS_CHECK_ARG(imgs && w2c && intr && c2w, GENS_EINVAL, "gens_blend_views: null camera / image pointer");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && rgb_out)), GENS_EINVAL, "gens_blend_views: null pts / output");
    if (n == 0) return 0;
    for (int k = 0; k < 21; ++k) GENS_CHECK_ARG(weights[k], GENS_EINVAL, "gens_blend_views: weight %d is null", k);
    BlendWeights W;
    const float* const* w = weights;
    W.rd1 = w[0]; W.rd1_b = w[1]; W.rd2 = w[2]; W.rd2_b = w[3];
    W.b1 = w[4]; W.b1_b = w[5]; W.b2 = w[6]; W.b2_b = w[7];
    W.v1 = w[8]; W.v1_b = w[9]; W.v2 = w[10]; W.v2_b = w[11]; W.v2_last = w[12];
    W.u1 = w[13]; W.u1_b = w[14]; W.u2 = w[15];
    W.r1 = w[16]; W.r1_b = w[17]; W.r2 = w[18]; W.r2_b = w[19]; W.r3 = w[20];
    W.v2_last_b = scalars[0]; W.u2_b = scalars[1]; W.r3_b = scalars[2]; W.s_abs = scalars[3];
    const int ppw = 32 / (nv - 1);
    const int64_t waves = (n + ppw - 1) / ppw;
    const unsigned grid = gens_blocks(waves, BL_WAVES);
    hipStream_t st = (hipStream_t)stream;
#define BLEND_LAUNCH(NL) blend_k<NL><<<grid, 64 * BL_WAVES, 0, st>>>(W, fs, (const float4*)imgs, w2c, intr, c2w, nv, pts, index, n, n_device, rgb_out, vis_out)
    switch (n_levels) {
        case 1: BLEND_LAUNCH(1); break;
        case 2: BLEND_LAUNCH(2); break;
        case 3: BLEND_LAUNCH(3); break;
        case 4: BLEND_LAUNCH(4); break;
        default: BLEND_LAUNCH(5); break;
    }
#undef BLEND_LAUNCH
    return gens_launch_status("gens_blend_views");
}
