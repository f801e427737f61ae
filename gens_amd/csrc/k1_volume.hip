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
        s2.x += f.x * f.x; s2.y += f.y * f.y; s2.z += f.z * f.z; s2.w += f.w * f.w;
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
