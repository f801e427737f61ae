// K6h: the fused SDF network of k6_sdfmlp.hip on the f16 matrix cores with SPLIT operands.
//
// Every float32 operand x is carried as an (hi, lo) pair of halfs, hi = f16(x), lo = f16(x - hi) -- 22 mantissa bits --
// and every product a*b is evaluated as hi*hi + hi*lo + lo*hi with three v_mfma_f32_32x32x16_f16 (float32 accumulate):
// 3 x 32 cycles per 16-deep K block instead of 8 x 64 cycles of v_mfma_f32_32x32x2_f32, i.e. 5.3x less matrix-pipe time
// at ~1e-6 relative error (the dropped lo*lo term and the pair representation are both 2^-22).  Same structure as the
// float32 kernel: 32 points per workgroup, wave w owns output columns [32w, 32w+32) of every layer, softplus' stays in
// registers for the reverse pass.  Activations live in LDS as two half tiles (same bytes as one float tile); weights are
// pre-split and pre-packed in B-fragment order (one 16-byte load per lane per operand).
//
// Range: halfs overflow at 65504.  Weights are checked by the host when the plan is built; activations / volume
// features beyond 3e4 raise *overflow_flag, and the caller re-runs that batch on the float32 kernel (gens_sdf_mlp).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define HM_M 32
#define HM_H 128
#define HM_PE 27
#define HM_PE_STRIDE 29
#define HM_PEH_STRIDE 40     // halfs; 80-byte rows keep the 16-byte A reads conflict-free
#define HM_SKIP_H 101
#define HM_NLAYER 6

struct SdfMlpWeightsH {
    const f16x8* wf_h[HM_NLAYER];   // forward B fragments, hi part  [n_tile(4)][kb][64]
    const f16x8* wf_l[HM_NLAYER];
    const float* bias[HM_NLAYER];
    const f16x8* wb_h[HM_NLAYER];   // backward B fragments [n_tile][8][64]; wb[0]: 1 tile
    const f16x8* wb_l[HM_NLAYER];
    const float* w_last;
    float b_last, inv_scale, scale;
    int* overflow;
};

__device__ __forceinline__ float softplus100h(float x, float& dsig) {
    float t = 100.0f * x;
    float e = hw_exp(fminf(t, 20.0f));
    float u = 1.0f + e;
    bool lin = t > 20.0f;
    dsig = lin ? 1.0f : e * hw_rcp(u);
    return lin ? x : hw_log(u) * 0.01f;
}

__device__ __forceinline__ void put_split(_Float16* hi, _Float16* lo, int off, float x, bool& big) {
    _Float16 h = (_Float16)x;
    hi[off] = h;
    lo[off] = (_Float16)(x - (float)h);
    big = big || !(fabsf(x) < 3.0e4f);
}

__device__ __forceinline__ int hrow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Weight fragments are streamed from L2 in chunks of HM_CH K-blocks held in registers.  A chunk is fetched long before
// it is used: chunk 0 of a GEMM is loaded BEFORE the barrier / activation epilogue that precedes the GEMM (weights do
// not depend on activations), the following chunks while the previous chunk's MFMAs run.
#define HM_CH 6
#define HM_MINW_F 2   // launch-bounds occupancy targets (waves per SIMD) of the forward / gradient variants; tighter
#define HM_MINW_G 1   // bounds spill (measured): the register-resident weight chunks + softplus' need the space
struct BFrag {
    f16x8 h[HM_CH], l[HM_CH];
};
template <int CNT>
__device__ __forceinline__ void bload(BFrag& f, const f16x8* __restrict__ bh, const f16x8* __restrict__ bl, int kb0, int lane) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        f.h[i] = bh[(kb0 + i) * 64 + lane];
        f.l[i] = bl[(kb0 + i) * 64 + lane];
    }
}
// acc_hi += Ah*Bh ; acc_lo += Ah*Bl + Al*Bh for CNT 16-deep K blocks starting at kb0.  A tiles: [32][rs] halfs in LDS.
template <int CNT>
__device__ __forceinline__ void bmfma(const BFrag& f, const _Float16* __restrict__ ah_lds, const _Float16* __restrict__ al_lds, int a_off,
                                      int kb0, f32x16& hi, f32x16& lo) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const f16x8 ah = *(const f16x8*)(ah_lds + a_off + 16 * (kb0 + i));
        const f16x8 al = *(const f16x8*)(al_lds + a_off + 16 * (kb0 + i));
        hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, f.h[i], hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, f.l[i], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, f.h[i], lo, 0, 0, 0);
    }
}
template <int KBL>
__device__ __forceinline__ void bpreload(BFrag& f, const f16x8* __restrict__ bh, const f16x8* __restrict__ bl, int lane) {
    bload<(KBL < HM_CH ? KBL : HM_CH)>(f, bh, bl, 0, lane);
}
// GEMM over KBL K-blocks whose first chunk is already in `pre`
template <int KBL>
__device__ __forceinline__ void gemm_split(const BFrag& pre, const _Float16* __restrict__ ah_lds, const _Float16* __restrict__ al_lds, int rs,
                                           const f16x8* __restrict__ bh, const f16x8* __restrict__ bl, f32x16& hi, f32x16& lo, int lane) {
    static_assert(KBL <= 3 * HM_CH, "at most three chunks");
    const int a_off = (lane & 31) * rs + 8 * (lane >> 5);
    if constexpr (KBL <= HM_CH) {
        bmfma<KBL>(pre, ah_lds, al_lds, a_off, 0, hi, lo);
    } else if constexpr (KBL <= 2 * HM_CH) {
        BFrag n1;
        bload<KBL - HM_CH>(n1, bh, bl, HM_CH, lane);
        bmfma<HM_CH>(pre, ah_lds, al_lds, a_off, 0, hi, lo);
        bmfma<KBL - HM_CH>(n1, ah_lds, al_lds, a_off, HM_CH, hi, lo);
    } else {
        BFrag n1, n2;
        bload<HM_CH>(n1, bh, bl, HM_CH, lane);
        bmfma<HM_CH>(pre, ah_lds, al_lds, a_off, 0, hi, lo);
        bload<KBL - 2 * HM_CH>(n2, bh, bl, 2 * HM_CH, lane);
        bmfma<HM_CH>(n1, ah_lds, al_lds, a_off, HM_CH, hi, lo);
        bmfma<KBL - 2 * HM_CH>(n2, ah_lds, al_lds, a_off, 2 * HM_CH, hi, lo);
    }
}

template <int FE, bool GRAD>
__global__ __launch_bounds__(256, GRAD ? HM_MINW_G : HM_MINW_F) void sdf_mlp_h_k(SdfMlpWeightsH W, LevelSet vols, const float* __restrict__ pts,
                                                   const int64_t* __restrict__ index, int64_t n_max, const int32_t* __restrict__ n_dev,
                                                   float* __restrict__ sdf_out, float* __restrict__ grad_out) {
    constexpr int CF = FE / 5;
    constexpr int KIN = HM_H + FE;
    constexpr int KP = (KIN + 15) / 16 * 16;     // 192 / 240
    constexpr int RSH = KP + 8;                  // 200 / 248 halfs: (RSH/2) mod 64 is an odd multiple of 4 -> conflict-free b128 reads
    constexpr int KB = KP / 16;
    constexpr int NT_B = (KIN + 31) / 32;
    constexpr int GFS = FE + 1;                  // row stride of the float conditioning-gradient tile (aliases XH)
    __shared__ __attribute__((aligned(16))) _Float16 XH[HM_M * RSH];
    __shared__ __attribute__((aligned(16))) _Float16 XL[HM_M * RSH];
    __shared__ __attribute__((aligned(16))) _Float16 PEH[HM_M * HM_PEH_STRIDE];
    __shared__ __attribute__((aligned(16))) _Float16 PEL[HM_M * HM_PEH_STRIDE];
    __shared__ float PE[HM_M * HM_PE_STRIDE];
    __shared__ float FEF[GRAD ? HM_M * FE : 1];
    __shared__ float GPE[GRAD ? HM_M * HM_PE_STRIDE : 1];
    __shared__ float JAC[GRAD ? HM_M * CF * 3 : 1];
    __shared__ float RED[HM_M * 8];
    static_assert(sizeof(_Float16) * HM_M * RSH >= sizeof(float) * HM_M * GFS, "gradient tile must fit in XH");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * HM_M;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    if (m0 >= n) return;
    bool big = false;

    // ------------------------------------------------------------------ prologue
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
            float* pe = PE + p * HM_PE_STRIDE;
            _Float16* ph = PEH + p * HM_PEH_STRIDE;
            _Float16* pl = PEL + p * HM_PEH_STRIDE;
            pe[a] = v;
            put_split(ph, pl, a, v, big);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s, c;
                hw_sincos(v * (float)(1 << k), s, c);
                pe[3 + 6 * k + a] = s;
                pe[6 + 6 * k + a] = c;
                put_split(ph, pl, 3 + 6 * k + a, s, big);
                put_split(ph, pl, 6 + 6 * k + a, c, big);
            }
            if (a == 0) {
                pe[27] = 0.0f; pe[28] = 0.0f;
                for (int k = 27; k < 32; ++k) { ph[k] = (_Float16)0.0f; pl[k] = (_Float16)0.0f; }
            }
        }
        if (sub == 7) {   // K padding columns of the wide tile
            for (int k = KIN; k < KP; ++k) { XH[p * RSH + k] = (_Float16)0.0f; XL[p * RSH + k] = (_Float16)0.0f; }
        }
        if (sub < vols.n) {
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
            _Float16* xh = XH + p * RSH + HM_H;
            _Float16* xl = XL + p * RSH + HM_H;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ch = 4 * l + c;
                float e[5];
                e[0] = fv[c];
                hw_sincos(fv[c], e[1], e[2]);
                hw_sincos(2.0f * fv[c], e[3], e[4]);
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    put_split(xh, xl, q * CF + ch, e[q], big);
                    if constexpr (GRAD) FEF[p * FE + q * CF + ch] = e[q];
                }
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

    // ------------------------------------------------------------------ forward: layers 0..5
    f32x16 dsig[GRAD ? HM_NLAYER : 1];
    const int col = 32 * wave + (lane & 31);
    BFrag pre;
    bpreload<2>(pre, W.wf_h[0] + (size_t)wave * 2 * 64, W.wf_l[0] + (size_t)wave * 2 * 64, lane);   // (in flight during the prologue sync)
    __syncthreads();
#pragma unroll
    for (int l = 0; l < HM_NLAYER; ++l) {
        f32x16 hi, lo;
        const float bias = W.bias[l][col];
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[r] = bias; lo[r] = 0.0f; }
        if (l == 0)
            gemm_split<2>(pre, PEH, PEL, HM_PEH_STRIDE, W.wf_h[0] + (size_t)wave * 2 * 64, W.wf_l[0] + (size_t)wave * 2 * 64, hi, lo, lane);
        else
            gemm_split<KB>(pre, XH, XL, RSH, W.wf_h[l] + (size_t)wave * KB * 64, W.wf_l[l] + (size_t)wave * KB * 64, hi, lo, lane);
        if (l + 1 < HM_NLAYER)      // next layer's first weight chunk: its latency hides behind the barrier + epilogue
            bpreload<KB>(pre, W.wf_h[l + 1] + (size_t)wave * KB * 64, W.wf_l[l + 1] + (size_t)wave * KB * 64, lane);
        else if constexpr (GRAD)
            bpreload<8>(pre, W.wb_h[5] + (size_t)wave * 8 * 64, W.wb_l[5] + (size_t)wave * 8 * 64, lane);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = hrow(r, lane);
            float ds;
            float h = softplus100h(hi[r] + lo[r], ds);
            if (l == 2) {
                if (col < HM_SKIP_H) {
                    h *= 0.70710678118654752440f;
                } else {
                    h = PE[row * HM_PE_STRIDE + (col - HM_SKIP_H)] * 0.70710678118654752440f;
                    ds = 0.0f;
                }
            }
            put_split(XH, XL, row * RSH + col, h, big);
            if constexpr (GRAD) dsig[l][r] = ds;
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ layer 6: dot product per point
    {
        const int p = tid >> 3, sub = tid & 7;
        float s = 0.0f;
        for (int k = sub; k < KIN; k += 8) s += ((float)XH[p * RSH + k] + (float)XL[p * RSH + k]) * W.w_last[k];
        RED[p * 8 + sub] = s;
    }
    __syncthreads();
    if (tid < HM_M) {
        const int64_t row = m0 + tid;
        if (row < n) {
            float s = W.b_last;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += RED[tid * 8 + k];
            sdf_out[index ? index[row] : row] = s * W.inv_scale;
        }
    }
    if constexpr (!GRAD) {
        if (__any(big) && lane == 0) atomicOr(W.overflow, 1);
        return;
    }

    // ------------------------------------------------------------------ reverse pass
    f32x16 gfe_h, gfe_l;
#pragma unroll
    for (int r = 0; r < 16; ++r) { gfe_h[r] = 0.0f; gfe_l[r] = 0.0f; }
    constexpr bool SPLIT_K = (NT_B - 4) == 2;
    const int fe_tile = SPLIT_K ? 4 + (wave & 1) : 4 + wave;
    const int fe_kb0 = SPLIT_K ? 4 * (wave >> 1) : 0;
    constexpr int FE_KB = SPLIT_K ? 4 : 8;
    __syncthreads();
    {
        const float wl = W.w_last[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) put_split(XH, XL, hrow(r, lane) * RSH + col, wl * dsig[5][r], big);
    }
    __syncthreads();
#pragma unroll
    for (int l = 5; l >= 1; --l) {
        f32x16 gh, gl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { gh[r] = 0.0f; gl[r] = 0.0f; }
        BFrag fe0;
        const f16x8* fbh = W.wb_h[l] + ((size_t)fe_tile * 8 + fe_kb0) * 64;
        const f16x8* fbl = W.wb_l[l] + ((size_t)fe_tile * 8 + fe_kb0) * 64;
        bpreload<FE_KB>(fe0, fbh, fbl, lane);                     // lands while the h-tile MFMAs run
        gemm_split<8>(pre, XH, XL, RSH, W.wb_h[l] + (size_t)wave * 8 * 64, W.wb_l[l] + (size_t)wave * 8 * 64, gh, gl, lane);
        gemm_split<FE_KB>(fe0, XH + 16 * fe_kb0, XL + 16 * fe_kb0, RSH, fbh, fbl, gfe_h, gfe_l, lane);
        if (l > 1) bpreload<8>(pre, W.wb_h[l - 1] + (size_t)wave * 8 * 64, W.wb_l[l - 1] + (size_t)wave * 8 * 64, lane);
        else if (wave == 0) bpreload<8>(pre, W.wb_h[0], W.wb_l[0], lane);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = hrow(r, lane);
            float g = gh[r] + gl[r];
            if (l == 3) {
                g *= 0.70710678118654752440f;
                if (col >= HM_SKIP_H) GPE[row * HM_PE_STRIDE + (col - HM_SKIP_H)] = g;
            }
            put_split(XH, XL, row * RSH + col, g * dsig[l - 1][r], big);
        }
        __syncthreads();
    }
    if (wave == 0) {   // d/d(point encoding) through layer 0: one n-tile
        f32x16 gh, gl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { gh[r] = 0.0f; gl[r] = 0.0f; }
        gemm_split<8>(pre, XH, XL, RSH, W.wb_h[0], W.wb_l[0], gh, gl, lane);
        const int c = lane & 31;
        if (c < HM_PE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) GPE[hrow(r, lane) * HM_PE_STRIDE + c] += gh[r] + gl[r];
        }
    }
    __syncthreads();
    float* GF = (float*)XH;      // the G tile is dead: park d sdf / d fe (float) in its place
    {
        const int c = 32 * (fe_tile - 4) + (lane & 31);
        if (c < FE && (!SPLIT_K || wave < 2)) {
            const float wl = W.w_last[HM_H + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) GF[hrow(r, lane) * GFS + c] = gfe_h[r] + gfe_l[r] + wl;
        }
        if (SPLIT_K) {
            __syncthreads();
            if (c < FE && wave >= 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) GF[hrow(r, lane) * GFS + c] += gfe_h[r] + gfe_l[r];
            }
        }
    }
    __syncthreads();
    if (tid < HM_M * 3) {
        const int p = tid / 3, a = tid % 3;
        const int64_t row = m0 + p;
        if (row < n) {
            const float* gpe = GPE + p * HM_PE_STRIDE;
            const float* pe = PE + p * HM_PE_STRIDE;
            float g = gpe[a];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = (float)(1 << k);
                g += f * (gpe[3 + 6 * k + a] * pe[6 + 6 * k + a] - gpe[6 + 6 * k + a] * pe[3 + 6 * k + a]);
            }
            g *= W.scale;
            const float* gf = GF + p * GFS;
            const float* fe = FEF + p * FE;
            for (int c = 0; c < CF; ++c) {
                float df = gf[c] + gf[CF + c] * fe[2 * CF + c] - gf[2 * CF + c] * fe[CF + c] +
                           2.0f * (gf[3 * CF + c] * fe[4 * CF + c] - gf[4 * CF + c] * fe[3 * CF + c]);
                g += df * JAC[(p * CF + c) * 3 + a];
            }
            grad_out[3 * (index ? index[row] : row) + a] = g * W.inv_scale;
        }
    }
    if (__any(big) && lane == 0) atomicOr(W.overflow, 1);
}

int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

extern "C" int gens_sdf_mlp_f16(const float* const* vols_packed, const int* dims, int n_levels, const void* const* wf_hi,
                                const void* const* wf_lo, const float* const* bias, const void* const* wb_hi, const void* const* wb_lo,
                                const float* w_last, float b_last, float scale, const float* pts, const int64_t* index, int64_t n,
                                const int32_t* n_device, float* sdf_out, float* grad_out, int* overflow_flag, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_sdf_mlp_f16", &vs, vols_packed, dims, n_levels)) return e;
    GENS_CHECK_ARG(n_levels == 3 || n_levels == 5, GENS_ELIMIT, "gens_sdf_mlp_f16: built for 3 or 5 volume levels, got %d", n_levels);
    GENS_CHECK_ARG(wf_hi && wf_lo && bias && w_last && overflow_flag && ((wb_hi && wb_lo) || !grad_out), GENS_EINVAL,
                   "gens_sdf_mlp_f16: null weight table / flag");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && sdf_out)), GENS_EINVAL, "gens_sdf_mlp_f16: null pts / output");
    GENS_CHECK_ARG(scale != 0.0f, GENS_EINVAL, "gens_sdf_mlp_f16: scale must be non-zero");
    if (n == 0) return 0;
    SdfMlpWeightsH W;
    for (int l = 0; l < HM_NLAYER; ++l) {
        GENS_CHECK_ARG(wf_hi[l] && wf_lo[l] && bias[l] && (!grad_out || (wb_hi[l] && wb_lo[l])), GENS_EINVAL,
                       "gens_sdf_mlp_f16: layer %d weights are null", l);
        W.wf_h[l] = (const f16x8*)wf_hi[l];
        W.wf_l[l] = (const f16x8*)wf_lo[l];
        W.bias[l] = bias[l];
        W.wb_h[l] = grad_out ? (const f16x8*)wb_hi[l] : nullptr;
        W.wb_l[l] = grad_out ? (const f16x8*)wb_lo[l] : nullptr;
    }
    W.w_last = w_last;
    W.b_last = b_last;
    W.scale = scale;
    W.inv_scale = 1.0f / scale;
    W.overflow = overflow_flag;
    unsigned grid = gens_blocks(n, HM_M);
    hipStream_t s = (hipStream_t)stream;
    if (n_levels == 3) {
        if (grad_out) sdf_mlp_h_k<60, true><<<grid, 256, 0, s>>>(W, vs, pts, index, n, n_device, sdf_out, grad_out);
        else sdf_mlp_h_k<60, false><<<grid, 256, 0, s>>>(W, vs, pts, index, n, n_device, sdf_out, grad_out);
    } else {
        if (grad_out) sdf_mlp_h_k<100, true><<<grid, 256, 0, s>>>(W, vs, pts, index, n, n_device, sdf_out, grad_out);
        else sdf_mlp_h_k<100, false><<<grid, 256, 0, s>>>(W, vs, pts, index, n, n_device, sdf_out, grad_out);
    }
    return gens_launch_status("gens_sdf_mlp_f16");
}
