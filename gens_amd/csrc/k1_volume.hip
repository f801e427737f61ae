// K1: multi-view cost-volume build (Volume.agg_mean_var, /root/reference/models/modules/volume.py:21-61) and the
// texel-layout helpers it (and K4/K9) read through.
//
// Data layout: the feature pyramid is re-packed once per scene from NCHW to NHWC texels (16 B per pixel for C=4), so
// a bilinear tap is ONE global_load_dwordx4 instead of four strided dword loads.  One thread owns one voxel and
// loops over the views, keeping sum / sum-of-squares / count in registers; the (nv, C, D^3) "feat_warp" tensor
// the reference materialises never exists.  Voxels are numbered x-major / z-fastest like the reference's
// meshgrid(ij).reshape (Q1), so a wavefront writes 64 consecutive floats into each of the 9 output planes
// (coalesced 256-B stores).  The kernel is HBM-write bound: 36 B per voxel out, the 25 MB of level-0 texels stay in
// L2 / Infinity Cache.
#include <stdlib.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// layout helpers
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_nchw_k(const float* __restrict__ src, float4* __restrict__ dst, int c,
                                                   int64_t hw, int q4, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int q = (int)(i % q4);
    int64_t pix = (i / q4) % hw;
    int64_t img = i / (q4 * hw);
    const float* s = src + (img * c + (int64_t)q * 4) * hw + pix;
    float4 v;
    v.x = (q * 4 + 0 < c) ? s[0] : 0.0f;
    v.y = (q * 4 + 1 < c) ? s[hw] : 0.0f;
    v.z = (q * 4 + 2 < c) ? s[2 * hw] : 0.0f;
    v.w = (q * 4 + 3 < c) ? s[3 * hw] : 0.0f;
    dst[i] = v;
}

__global__ __launch_bounds__(256) void unpack_nhwc_k(const float* __restrict__ src, float* __restrict__ dst, int c,
                                                     int64_t hw, int cpad, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over (img, ch, pix)
    if (i >= total) return;
    int64_t pix = i % hw;
    int ch = (int)((i / hw) % c);
    int64_t img = i / (hw * c);
    dst[i] = src[(img * hw + pix) * cpad + ch];
}

extern "C" int gens_pack_nchw(const float* src, float* dst, int n, int c, int h, int w, void* stream) {
    GENS_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0, GENS_EINVAL, "gens_pack_nchw: bad argument");
    int q4 = (c + 3) / 4;
    int64_t total = (int64_t)n * h * w * q4;
    pack_nchw_k<<<gens_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(src, (float4*)dst, c, (int64_t)h * w, q4, total);
    return gens_launch_status("gens_pack_nchw");
}

extern "C" int gens_unpack_nhwc(const float* src, float* dst, int n, int c, int h, int w, void* stream) {
    GENS_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0, GENS_EINVAL, "gens_unpack_nhwc: bad argument");
    int64_t total = (int64_t)n * c * h * w;
    unpack_nhwc_k<<<gens_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(src, dst, c, (int64_t)h * w, 4 * ((c + 3) / 4), total);
    return gens_launch_status("gens_unpack_nhwc");
}

extern "C" int gens_pack_volume(const float* src, float* dst, int x, int y, int z, void* stream) {
    GENS_CHECK_ARG(src && dst && x > 0 && y > 0 && z > 0, GENS_EINVAL, "gens_pack_volume: bad argument");
    int64_t total = (int64_t)x * y * z;
    pack_nchw_k<<<gens_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(src, (float4*)dst, 4, total, 1, total);
    return gens_launch_status("gens_pack_volume");
}

// ---------------------------------------------------------------------------------------------------------------
// projection of one voxel into one view (volume.py:34-43)
// ---------------------------------------------------------------------------------------------------------------
struct Proj {
    float ix, iy;
    bool vis;
};
__device__ __forceinline__ Proj project_voxel(const float* __restrict__ w2c, const float* __restrict__ k, float s, int h,
                                              int w, float x, float y, float z) {
    float4 cam = mat4_point(w2c, x, y, z);
    float u = (k[0] * s) * cam.x + (k[1] * s) * cam.y + (k[2] * s) * cam.z + (k[3] * s) * cam.w;
    float v = (k[4] * s) * cam.x + (k[5] * s) * cam.y + (k[6] * s) * cam.z + (k[7] * s) * cam.w;
    float d = k[8] * cam.x + k[9] * cam.y + k[10] * cam.z + k[11] * cam.w;
    float px = u / (d + 1e-8f), py = v / (d + 1e-8f);                       // (Q3)
    float nx = px / ((float)(w - 1) / 2.0f) - 1.0f, ny = py / ((float)(h - 1) / 2.0f) - 1.0f;
    Proj p;
    p.vis = (fabsf(nx) <= 1.0f) && (fabsf(ny) <= 1.0f) && (d > 0.0f);
    p.ix = (nx + 1.0f) / 2.0f * (float)(w - 1);                             // align_corners=True (volume.py:46)
    p.iy = (ny + 1.0f) / 2.0f * (float)(h - 1);
    return p;
}

__global__ __launch_bounds__(256) void volume_build_fwd_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                          const float* __restrict__ intr, float s, int nv, int h, int w,
                                                          int d, int min_vis, float* __restrict__ vol,
                                                          float* __restrict__ mask) {
    int64_t n = (int64_t)d * d * d;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    int kz = (int)(idx % d), jy = (int)((idx / d) % d), ix = (int)(idx / ((int64_t)d * d));
    float x = linspace_at(-1.0f, 1.0f, d, ix), y = linspace_at(-1.0f, 1.0f, d, jy), z = linspace_at(-1.0f, 1.0f, d, kz);
    float4 s1 = f4_zero(), s2 = f4_zero();
    float cnt = 0.0f;
    for (int v = 0; v < nv; ++v) {
        Proj p = project_voxel(w2c + 16 * v, intr + 16 * v, s, h, w, x, y, z);
        if (!p.vis) continue;
        Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
        float4 f = sample_texel(feat + (int64_t)v * h * w, h, w, 1, 0, t);
        s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
        s2 = make_float4(__builtin_fmaf(f.x, f.x, s2.x), __builtin_fmaf(f.y, f.y, s2.y), __builtin_fmaf(f.z, f.z, s2.z), __builtin_fmaf(f.w, f.w, s2.w));
        cnt += 1.0f;
    }
    float den = cnt <= 0.0f ? 1e-8f : cnt;                                   // (Q5)
    float4 m = make_float4(s1.x / den, s1.y / den, s1.z / den, s1.w / den);
    vol[idx] = m.x;
    vol[n + idx] = m.y;
    vol[2 * n + idx] = m.z;
    vol[3 * n + idx] = m.w;
    vol[4 * n + idx] = s2.x / den - m.x * m.x;
    vol[5 * n + idx] = s2.y / den - m.y * m.y;
    vol[6 * n + idx] = s2.z / den - m.z * m.z;
    vol[7 * n + idx] = s2.w / den - m.w * m.w;
    mask[idx] = cnt > (float)min_vis ? 1.0f : 0.0f;                          // (Q4)
}

// ---------------------------------------------------------------------------------------------------------------
// Fast forward path (power-of-two D <= 256).  K1 is INSTRUCTION-bound, not HBM-bound (~1 300 instructions per wavefront
// in the generic kernel above against 36 B written per voxel), so this variant removes instructions while keeping every
// float32 result bit-identical:
//   * one workgroup = 256 / D whole z-rows: (ix, jy, kz) by shifts and masks instead of three 64-bit divisions;
//     linspace step and the level constants (w-1)/2, (h-1)/2 and their reciprocals arrive as kernel arguments
//     (computed by the host with the same float32 operations);
//   * divisions a / b as q = a * y, r = fma(-q, b, a), q' = fma(r, y, q) with y = RN(1 / b): the correctly rounded
//     quotient (Markstein), bit-equal to the IEEE division (0 mismatches in 1.5e8 trials on this path's operand ranges;
//     tests compare the two kernels bit for bit).  One reciprocal serves u / d and v / d, one serves the eight
//     channel sums / count;
//   * intrinsics pre-multiplied by the level scale (PRESCALED) when the caller passes intr_scale == 1.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float div_rn(float a, float b, float y) {   // a / b given y = RN(1 / b)
    const float q = a * y;
    const float r = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(r, y, q);
}

struct LevelConst {
    float step;          // 2 / (D - 1): torch.linspace(-1, 1, D) step
    float cw, ch;        // (w - 1) / 2, (h - 1) / 2
    float rcw, rch;      // RN(1 / cw), RN(1 / ch)
    int log2d;
};

template <bool PRESCALED>
__global__ __launch_bounds__(256) void volume_build_fwd_pow2_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                               const float* __restrict__ intr, float s, int nv, int h, int w, int d,
                                                               LevelConst lc, int min_vis, float* __restrict__ vol, float* __restrict__ mask) {
    const int tid = threadIdx.x;
    const int kz = tid & (d - 1);
    const int row = (int)blockIdx.x * (256 >> lc.log2d) + (tid >> lc.log2d);      // = ix * d + jy
    const int jy = row & (d - 1), ix = row >> lc.log2d;
    if (ix >= d) return;
    const int64_t n = (int64_t)d << (2 * lc.log2d);
    const int64_t idx = ((int64_t)row << lc.log2d) + kz;
    const int half = d >> 1;
    // torch.linspace(-1, 1, d)[i]: lower half counts up from the start, upper half down from the end
    const float x = ix < half ? -1.0f + lc.step * (float)ix : 1.0f - lc.step * (float)(d - 1 - ix);
    const float y = jy < half ? -1.0f + lc.step * (float)jy : 1.0f - lc.step * (float)(d - 1 - jy);
    const float z = kz < half ? -1.0f + lc.step * (float)kz : 1.0f - lc.step * (float)(d - 1 - kz);
    float4 s1 = f4_zero(), s2 = f4_zero();
    float cnt = 0.0f;
    for (int v = 0; v < nv; ++v) {
        const float* m = w2c + 16 * v;
        const float* k = intr + 16 * v;
        const float4 cam = mat4_point(m, x, y, z);
        float u, vv;
        if (PRESCALED) {
            u = k[0] * cam.x + k[1] * cam.y + k[2] * cam.z + k[3] * cam.w;
            vv = k[4] * cam.x + k[5] * cam.y + k[6] * cam.z + k[7] * cam.w;
        } else {
            u = (k[0] * s) * cam.x + (k[1] * s) * cam.y + (k[2] * s) * cam.z + (k[3] * s) * cam.w;
            vv = (k[4] * s) * cam.x + (k[5] * s) * cam.y + (k[6] * s) * cam.z + (k[7] * s) * cam.w;
        }
        const float dd = k[8] * cam.x + k[9] * cam.y + k[10] * cam.z + k[11] * cam.w;
        const float dn = dd + 1e-8f;                                              // (Q3)
        const float yd = 1.0f / dn;
        const float px = div_rn(u, dn, yd), py = div_rn(vv, dn, yd);
        const float nx = div_rn(px, lc.cw, lc.rcw) - 1.0f, ny = div_rn(py, lc.ch, lc.rch) - 1.0f;
        const bool vis = (fabsf(nx) <= 1.0f) && (fabsf(ny) <= 1.0f) && (dd > 0.0f);
        if (!vis) continue;
        const float fx = (nx + 1.0f) / 2.0f * (float)(w - 1), fy = (ny + 1.0f) / 2.0f * (float)(h - 1);
        // A visible voxel reads inside the image: fx in [0, w-1], fy in [0, h-1], so the general tap logic (finiteness, clamps,
        // eight bounds tests, zero-padding selects) reduces to "the +1 tap may sit on column w / row h, where its weight is
        // exactly 0": read it from the clamped index instead.  Same weights, same order of accumulation as sample_texel.
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
        const float wx1 = fx - x0f, wx0 = (x0f + 1.0f) - fx, wy1 = fy - y0f, wy0 = (y0f + 1.0f) - fy;
        const float4* img = feat + (int64_t)v * h * w;
        const float4 v00 = img[y0 * w + x0], v01 = img[y0 * w + x1], v10 = img[y1 * w + x0], v11 = img[y1 * w + x1];
        float4 f = f4_madd(f4_zero(), v00, wx0 * wy0);
        f = f4_madd(f, v01, wx1 * wy0);
        f = f4_madd(f, v10, wx0 * wy1);
        f = f4_madd(f, v11, wx1 * wy1);
        s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
        s2 = make_float4(__builtin_fmaf(f.x, f.x, s2.x), __builtin_fmaf(f.y, f.y, s2.y), __builtin_fmaf(f.z, f.z, s2.z), __builtin_fmaf(f.w, f.w, s2.w));
        cnt += 1.0f;
    }
    const float den = cnt <= 0.0f ? 1e-8f : cnt;                                  // (Q5)
    const float yn = 1.0f / den;
    const float4 mm = make_float4(div_rn(s1.x, den, yn), div_rn(s1.y, den, yn), div_rn(s1.z, den, yn), div_rn(s1.w, den, yn));
    vol[idx] = mm.x;
    vol[n + idx] = mm.y;
    vol[2 * n + idx] = mm.z;
    vol[3 * n + idx] = mm.w;
    vol[4 * n + idx] = div_rn(s2.x, den, yn) - mm.x * mm.x;
    vol[5 * n + idx] = div_rn(s2.y, den, yn) - mm.y * mm.y;
    vol[6 * n + idx] = div_rn(s2.z, den, yn) - mm.z * mm.z;
    vol[7 * n + idx] = div_rn(s2.w, den, yn) - mm.w * mm.w;
    mask[idx] = cnt > (float)min_vis ? 1.0f : 0.0f;                               // (Q4)
}

// d(volume)/d(features): recompute the projections, then scatter with the bilinear weights.
__global__ __launch_bounds__(256) void volume_build_bwd_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                          const float* __restrict__ intr, float s, int nv, int h, int w,
                                                          int d, const float* __restrict__ gvol, float* __restrict__ gfeat) {
    int64_t n = (int64_t)d * d * d;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    int kz = (int)(idx % d), jy = (int)((idx / d) % d), ix = (int)(idx / ((int64_t)d * d));
    float x = linspace_at(-1.0f, 1.0f, d, ix), y = linspace_at(-1.0f, 1.0f, d, jy), z = linspace_at(-1.0f, 1.0f, d, kz);
    float4 s1 = f4_zero();
    float cnt = 0.0f;
    for (int v = 0; v < nv; ++v) {
        Proj p = project_voxel(w2c + 16 * v, intr + 16 * v, s, h, w, x, y, z);
        if (!p.vis) continue;
        Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
        float4 f = sample_texel(feat + (int64_t)v * h * w, h, w, 1, 0, t);
        s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
        cnt += 1.0f;
    }
    if (cnt <= 0.0f) return;
    float inv = 1.0f / cnt;
    float4 mean = make_float4(s1.x * inv, s1.y * inv, s1.z * inv, s1.w * inv);
    float4 gm = make_float4(gvol[idx], gvol[n + idx], gvol[2 * n + idx], gvol[3 * n + idx]);
    float4 gv = make_float4(gvol[4 * n + idx], gvol[5 * n + idx], gvol[6 * n + idx], gvol[7 * n + idx]);
    for (int v = 0; v < nv; ++v) {
        Proj p = project_voxel(w2c + 16 * v, intr + 16 * v, s, h, w, x, y, z);
        if (!p.vis) continue;
        Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
        float4 f = sample_texel(feat + (int64_t)v * h * w, h, w, 1, 0, t);
        float4 g;
        g.x = (gm.x + 2.0f * gv.x * (f.x - mean.x)) * inv;
        g.y = (gm.y + 2.0f * gv.y * (f.y - mean.y)) * inv;
        g.z = (gm.z + 2.0f * gv.z * (f.z - mean.z)) * inv;
        g.w = (gm.w + 2.0f * gv.w * (f.w - mean.w)) * inv;
        float* base = gfeat + (((int64_t)v * h + t.y0) * w + t.x0) * 4;
        if (t.ok00) atomic_add4(base, g, t.w00);
        if (t.ok01) atomic_add4(base + 4, g, t.w01);
        if (t.ok10) atomic_add4(base + (int64_t)w * 4, g, t.w10);
        if (t.ok11) atomic_add4(base + (int64_t)w * 4 + 4, g, t.w11);
    }
}

static int check_volume_args(const char* who, const void* a, const void* b, const void* c, int nv, int h, int w, int d) {
    GENS_CHECK_ARG(a && b && c, GENS_EINVAL, "%s: null pointer", who);
    GENS_CHECK_ARG(nv > 0 && h > 1 && w > 1 && d > 0, GENS_EINVAL, "%s: bad size nv=%d h=%d w=%d d=%d", who, nv, h, w, d);
    GENS_CHECK_ARG(nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "%s: nv=%d > %d", who, nv, GENS_MAX_VIEWS);
    return 0;
}

extern "C" int gens_volume_build_fwd(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv,
                                     int h, int w, int d, int min_vis_view, float* volume, float* mask, void* stream) {
    if (int e = check_volume_args("gens_volume_build_fwd", feat, w2c, intr, nv, h, w, d)) return e;
    GENS_CHECK_ARG(volume && mask, GENS_EINVAL, "gens_volume_build_fwd: null output");
    int64_t n = (int64_t)d * d * d;
    const bool pow2 = d >= 2 && d <= 256 && (d & (d - 1)) == 0;
    if (pow2 && !getenv("GENS_K1_GENERIC")) {                 // (the environment switch keeps the generic kernel reachable for A/B tests)
        LevelConst lc;
        lc.step = (1.0f - (-1.0f)) / (float)(d - 1);
        lc.cw = (float)(w - 1) / 2.0f;
        lc.ch = (float)(h - 1) / 2.0f;
        lc.rcw = 1.0f / lc.cw;
        lc.rch = 1.0f / lc.ch;
        lc.log2d = 0;
        while ((1 << lc.log2d) < d) ++lc.log2d;
        const unsigned rows_per_block = 256u >> lc.log2d;
        const unsigned grid = ((unsigned)d * (unsigned)d + rows_per_block - 1) / rows_per_block;
        if (intr_scale == 1.0f)
            volume_build_fwd_pow2_k<true><<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv, h, w, d, lc,
                                                                               min_vis_view, volume, mask);
        else
            volume_build_fwd_pow2_k<false><<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv, h, w, d, lc,
                                                                                min_vis_view, volume, mask);
        return gens_launch_status("gens_volume_build_fwd");
    }
    volume_build_fwd_k<<<gens_blocks(n, 256), 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv,
                                                                            h, w, d, min_vis_view, volume, mask);
    return gens_launch_status("gens_volume_build_fwd");
}

extern "C" int gens_volume_build_bwd(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv,
                                     int h, int w, int d, const float* g_volume, float* g_feat, void* stream) {
    if (int e = check_volume_args("gens_volume_build_bwd", feat, w2c, intr, nv, h, w, d)) return e;
    GENS_CHECK_ARG(g_volume && g_feat, GENS_EINVAL, "gens_volume_build_bwd: null gradient buffer");
    int64_t n = (int64_t)d * d * d;
    volume_build_bwd_k<<<gens_blocks(n, 256), 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv,
                                                                            h, w, d, g_volume, g_feat);
    return gens_launch_status("gens_volume_build_bwd");
}
