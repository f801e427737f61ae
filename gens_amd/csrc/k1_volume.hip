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

#include <algorithm>
#include "common.h"

#define K1_FAST_VIEWS 8   // the frustum-culled kernels map (z-row, view) pairs onto 32 x 8 threads; more views run the generic kernel

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
    // streaming (non-temporal) stores: keep the texels in L2
    __builtin_nontemporal_store(mm.x, vol + idx);
    __builtin_nontemporal_store(mm.y, vol + n + idx);
    __builtin_nontemporal_store(mm.z, vol + 2 * n + idx);
    __builtin_nontemporal_store(mm.w, vol + 3 * n + idx);
    __builtin_nontemporal_store(div_rn(s2.x, den, yn) - mm.x * mm.x, vol + 4 * n + idx);
    __builtin_nontemporal_store(div_rn(s2.y, den, yn) - mm.y * mm.y, vol + 5 * n + idx);
    __builtin_nontemporal_store(div_rn(s2.z, den, yn) - mm.z * mm.z, vol + 6 * n + idx);
    __builtin_nontemporal_store(div_rn(s2.w, den, yn) - mm.w * mm.w, vol + 7 * n + idx);
    __builtin_nontemporal_store(cnt > (float)min_vis ? 1.0f : 0.0f, mask + idx);   // (Q4)
}

// ---------------------------------------------------------------------------------------------------------------
// Production forward kernel (power-of-two D >= 8, pre-scaled intrinsics): volume_build_fwd_pow2_k with fewer operations, every
// float32 result still bit-identical (tests compare the kernels bit for bit at full size).
// What bounds it (scripts/probe/k1_probe.py, D = 256, warm clocks): arithmetic alone 152 us, + texel reads 182 us, + stores
// 248 us; 30 us of that are the stores sweeping the texels out of L2, which streaming (non-temporal) stores avoid (222 us).
// Issuing all views' gathers before the first use (one exposed memory latency per wave instead of nv) was measured SLOWER
// (249 us): it gives up the skip of views no lane of the wave sees and half the occupancy.  So the lever is the instruction count:
//   * RN(1/b) as y0 = v_rcp_f32(b) (<= 1 ulp), e = fma(-b, y0, 1), y = fma(e, y0, y0): the correctly rounded reciprocal of every
//     normal b (gens_selftest_division compares all 2^32 bit patterns with the IEEE division), 3 instructions instead of 10;
//   * pinhole matrices (w2c row 3 = 0 0 0 1, intrinsics [[fx 0 cx 0] [0 fy cy 0] [0 0 1 0]], tested per view on the scalar unit)
//     skip the products with an exact 0 / 1 (x + (+-0) = x, 1 * x = x in IEEE arithmetic);
//   * texel reads through a buffer descriptor: a 32-bit byte offset per lane, the view's offset in an SGPR (no 64-bit address
//     arithmetic, one multiply-add per tap).
// Tried and measured no faster (everything lands on the same ~200 us): two voxels per lane, packed (v_pk_*_f32) or scalar; all views'
// gathers issued before the first use; several chunks per workgroup.  The file is compiled without the SLP vectoriser, whose
// v_pk_*_f32 pairs plus register shuffles cost 4 % here.  Backward: XCD-private copies of the gradient image with workgroup-scope
// atomics were measured equal to the direct device-scope atomics (15.3 ms at 256^3: ~30 G float atomics / s either way).
// ---------------------------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp_rn(float b) {                    // RN(1 / b), see above
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

// One affine constraint g(t) >= 0 on the row parameter t in [0, 1] (g0 = g(0), g1 = g(1), `scale` bounds the magnitude of the
// terms g was summed from): narrows [lo, hi] conservatively.  Anything doubtful (flat or NaN g) leaves the interval alone.
__device__ __forceinline__ void row_constraint(float g0, float g1, float scale, float& lo, float& hi, bool& empty) {
    const float tol = 1e-5f * scale;
    if (g0 < -tol && g1 < -tol) { empty = true; return; }
    if (!(fabsf(g0 - g1) >= 1e-2f * scale)) return;
    const float t = g0 / (g0 - g1);                                               // zero crossing, |error| < 1e-3 (a quarter voxel at d = 256)
    if (g0 < g1) lo = fmaxf(lo, t); else hi = fminf(hi, t);
}

__device__ __forceinline__ void volume_build_chunk(uint32_t chunk, const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                   const float* __restrict__ intr, int nv, int h, int w, int d, LevelConst lc, int min_vis,
                                                   float* __restrict__ vol, float* __restrict__ mask) {
    // Chunk -> voxels.  d >= 64: the four waves take the SAME 64 z of four x-adjacent rows (ix = 4 g + wave), whose image footprints
    // overlap, so most of a wave's texel lines are already in the CU's L1 (the z-contiguous 256-voxel chunk sent 3-4x as many requests
    // to L2); smaller volumes: 256 consecutive voxels = 256 / d whole rows.
    const uint32_t dm = (uint32_t)d - 1u;
    const bool tiled = lc.log2d >= 6;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t zq_bits = (uint32_t)lc.log2d - 6u;                              // (tiled) a row is 2^zq_bits pieces of 64 voxels
    const uint32_t t_kz0 = (chunk & ((1u << zq_bits) - 1u)) << 6, t_jy = (chunk >> zq_bits) & dm, t_ix0 = (chunk >> (zq_bits + lc.log2d)) << 2;
    // within the tile a wave takes 16 z of all four rows (lane = 16 row + z): its footprints in a view then span ~30 pixels instead of ~90
    const uint32_t t_row = lane >> 4, t_z = (wv << 4) | (lane & 15u);
    const uint32_t idx = tiled ? (((t_ix0 + t_row) << (2 * lc.log2d)) | (t_jy << lc.log2d) | (t_kz0 + t_z)) : chunk * 256u + threadIdx.x;
    const int kz = (int)(idx & dm), jy = (int)((idx >> lc.log2d) & dm), ix = (int)(idx >> (2 * lc.log2d));
    const int half = d >> 1;
    // ---- frustum culling per (z-row, view).  Along a z-row the homogeneous image coordinates (u, v, depth) are affine in z, so the
    // voxels of the row a view can see form ONE interval of kz.  Thread (row r, view v) of the workgroup intersects the five half-lines
    // depth > 0, 0 <= u <= (w-1) depth, 0 <= v <= (h-1) depth from their values at the two ends of the row, widens the result by two
    // voxels, and leaves it in LDS; in the view loop a wave whose lanes all sit outside the interval skips the view before any per-voxel
    // arithmetic (64 % of all (wave, view) pairs at the benchmark geometry, where that arithmetic was 55 % of the kernel's instructions).
    // The exact per-voxel test below still decides visibility; the intervals only have to be supersets, and the bit-for-bit
    // comparisons with the unculled kernels check that they are.
    __shared__ int2 row_span[32][K1_FAST_VIEWS];
    {
        const int rows = tiled ? 4 : 256 >> lc.log2d;                             // z-rows the workgroup touches (d <= 256)
        const int r = threadIdx.x >> 3, v = threadIdx.x & 7;
        if (r < rows && v < nv) {
            const uint32_t row = tiled ? (((t_ix0 + (uint32_t)r) << lc.log2d) | t_jy) : chunk * (uint32_t)rows + (uint32_t)r;   // = ix * d + jy
            const int rj = (int)(row & dm), ri = (int)(row >> lc.log2d);
            const float rx = ri < half ? -1.0f + lc.step * (float)ri : 1.0f - lc.step * (float)(d - 1 - ri);
            const float ry = rj < half ? -1.0f + lc.step * (float)rj : 1.0f - lc.step * (float)(d - 1 - rj);
            const float* m = w2c + 16 * v;
            const float* k = intr + 16 * v;
            float c0[4], c1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float base = m[4 * q] * rx + m[4 * q + 1] * ry + m[4 * q + 3];
                c0[q] = base - m[4 * q + 2];                                      // z = -1
                c1[q] = base + m[4 * q + 2];                                      // z = +1
            }
            const float u0 = k[0] * c0[0] + k[1] * c0[1] + k[2] * c0[2] + k[3] * c0[3], u1 = k[0] * c1[0] + k[1] * c1[1] + k[2] * c1[2] + k[3] * c1[3];
            const float v0 = k[4] * c0[0] + k[5] * c0[1] + k[6] * c0[2] + k[7] * c0[3], v1 = k[4] * c1[0] + k[5] * c1[1] + k[6] * c1[2] + k[7] * c1[3];
            const float d0 = k[8] * c0[0] + k[9] * c0[1] + k[10] * c0[2] + k[11] * c0[3], d1 = k[8] * c1[0] + k[9] * c1[1] + k[10] * c1[2] + k[11] * c1[3];
            const float wm = (float)(w - 1), hm = (float)(h - 1);
            const float sd = fabsf(d0) + fabsf(d1) + 1e-6f;
            const float su = fabsf(u0) + fabsf(u1) + wm * sd, sv = fabsf(v0) + fabsf(v1) + hm * sd;
            float lo = 0.0f, hi = 1.0f;
            bool empty = false;
            row_constraint(d0, d1, sd, lo, hi, empty);
            row_constraint(u0, u1, su, lo, hi, empty);
            row_constraint(wm * d0 - u0, wm * d1 - u1, su, lo, hi, empty);
            row_constraint(v0, v1, sv, lo, hi, empty);
            row_constraint(hm * d0 - v0, hm * d1 - v1, sv, lo, hi, empty);
            int2 span;
            span.x = max((int)floorf(lo * (float)(d - 1)) - 2, 0);
            span.y = min((int)ceilf(hi * (float)(d - 1)) + 2, d - 1);
            if (empty) span = make_int2(1, 0);
            row_span[r][v] = span;
        }
    }
    __syncthreads();
    const int my_row = tiled ? (int)t_row : (int)(threadIdx.x >> lc.log2d);
    // torch.linspace(-1, 1, d)[i]: lower half counts up from the start, upper half down from the end
    const float x = ix < half ? -1.0f + lc.step * (float)ix : 1.0f - lc.step * (float)(d - 1 - ix);
    const float y = jy < half ? -1.0f + lc.step * (float)jy : 1.0f - lc.step * (float)(d - 1 - jy);
    const float z = kz < half ? -1.0f + lc.step * (float)kz : 1.0f - lc.step * (float)(d - 1 - kz);
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const uint32_t row_bytes = (uint32_t)w * 16u, view_bytes = (uint32_t)h * row_bytes;
    const __amdgpu_buffer_rsrc_t texels = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)((uint32_t)nv * view_bytes), 0x00020000);
    float4 s1 = f4_zero(), s2 = f4_zero();
    float cnt = 0.0f;
    for (int v = 0; v < nv; ++v) {
        const int2 span = row_span[my_row][v];
        if (kz < span.x || kz > span.y) continue;                                 // outside the view's frustum for sure
        const float* m = w2c + 16 * v;
        const float* k = intr + 16 * v;
        const float cx = m[0] * x + m[1] * y + m[2] * z + m[3];
        const float cy = m[4] * x + m[5] * y + m[6] * z + m[7];
        const float cz = m[8] * x + m[9] * y + m[10] * z + m[11];
        // integer tests on the scalar unit (a float compare would be a VALU instruction + vcc branch per matrix entry)
        const uint32_t* mb = (const uint32_t*)m;
        const uint32_t* kb = (const uint32_t*)k;
        const uint32_t must_be_zero = (mb[12] | mb[13] | mb[14] | kb[1] | kb[3] | kb[4] | kb[7] | kb[8] | kb[9] | kb[11]) << 1;   // +-0
        const uint32_t must_be_one = (mb[15] ^ 0x3f800000u) | (kb[10] ^ 0x3f800000u);
        float u, vv, dd;
        if ((must_be_zero | must_be_one) == 0u) {
            u = k[0] * cx + k[2] * cz;
            vv = k[5] * cy + k[6] * cz;
            dd = cz;
        } else {
            const float cw = m[12] * x + m[13] * y + m[14] * z + m[15];
            u = k[0] * cx + k[1] * cy + k[2] * cz + k[3] * cw;
            vv = k[4] * cx + k[5] * cy + k[6] * cz + k[7] * cw;
            dd = k[8] * cx + k[9] * cy + k[10] * cz + k[11] * cw;
        }
        const float dn = dd + 1e-8f;                                              // (Q3)
        const float yd = rcp_rn(dn);
        const float px = div_rn(u, dn, yd), py = div_rn(vv, dn, yd);
        const float nx = div_rn(px, lc.cw, lc.rcw) - 1.0f, ny = div_rn(py, lc.ch, lc.rch) - 1.0f;
        const bool vis = (fmaxf(fabsf(nx), fabsf(ny)) <= 1.0f) && (dd > 0.0f);    // a NaN coordinate implies dn == 0, i.e. dd < 0
        if (!vis) continue;
        // A visible voxel reads inside the image (fx in [0, w-1], fy in [0, h-1]): the +1 tap may sit on column w / row h, where its
        // weight is exactly 0 -- read it from the clamped index.  Same weights and order of accumulation as sample_texel.
        const float fx = (nx + 1.0f) / 2.0f * wm1, fy = (ny + 1.0f) / 2.0f * hm1; // align_corners=True (volume.py:46)
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const uint32_t xo0 = (uint32_t)x0 << 4, xo1 = (uint32_t)min(x0 + 1, w - 1) << 4;
        const uint32_t y1 = (uint32_t)min(y0 + 1, h - 1);
        const uint32_t view_off = (uint32_t)v * view_bytes;
        const f4 v00 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(texels, __umul24((uint32_t)y0, row_bytes) + xo0, view_off, 0));
        const f4 v01 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(texels, __umul24((uint32_t)y0, row_bytes) + xo1, view_off, 0));
        const f4 v10 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(texels, __umul24(y1, row_bytes) + xo0, view_off, 0));
        const f4 v11 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(texels, __umul24(y1, row_bytes) + xo1, view_off, 0));
        const float wx1 = fx - x0f, wx0 = (x0f + 1.0f) - fx, wy1 = fy - y0f, wy0 = (y0f + 1.0f) - fy;
        const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
        float4 f;
        f.x = __builtin_fmaf(v11.x, w11, __builtin_fmaf(v10.x, w10, __builtin_fmaf(v01.x, w01, __builtin_fmaf(v00.x, w00, 0.0f))));
        f.y = __builtin_fmaf(v11.y, w11, __builtin_fmaf(v10.y, w10, __builtin_fmaf(v01.y, w01, __builtin_fmaf(v00.y, w00, 0.0f))));
        f.z = __builtin_fmaf(v11.z, w11, __builtin_fmaf(v10.z, w10, __builtin_fmaf(v01.z, w01, __builtin_fmaf(v00.z, w00, 0.0f))));
        f.w = __builtin_fmaf(v11.w, w11, __builtin_fmaf(v10.w, w10, __builtin_fmaf(v01.w, w01, __builtin_fmaf(v00.w, w00, 0.0f))));
        s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
        s2 = make_float4(__builtin_fmaf(f.x, f.x, s2.x), __builtin_fmaf(f.y, f.y, s2.y), __builtin_fmaf(f.z, f.z, s2.z), __builtin_fmaf(f.w, f.w, s2.w));
        cnt += 1.0f;
    }
    const float den = cnt <= 0.0f ? 1e-8f : cnt;                                  // (Q5)
    const float yn = rcp_rn(den);
    const float4 mm = make_float4(div_rn(s1.x, den, yn), div_rn(s1.y, den, yn), div_rn(s1.z, den, yn), div_rn(s1.w, den, yn));
    // Output: the workgroup's 256 voxels are 256 consecutive floats in each of the 9 planes.  Staged through LDS so that a store
    // instruction carries 16 B per lane (one plane's 1 KiB per wave-store: 9 store instructions per workgroup instead of 36 -- the
    // texture path of a CU was busy 72 % of the kernel with 4-byte stores); streaming (non-temporal), so that the 36 B / voxel do not
    // sweep the texels out of L2; through buffer descriptors (lane offset in 32 bits; d^3 <= 2^24 voxels: 8 planes are 512 MiB).
    __shared__ float stage[9][256];
    const int tid = threadIdx.x;
    const int slot = tiled ? (int)(t_row * 64u + t_z) : tid;                        // position in the workgroup's tile: row-piece, then z
    stage[0][slot] = mm.x;
    stage[1][slot] = mm.y;
    stage[2][slot] = mm.z;
    stage[3][slot] = mm.w;
    stage[4][slot] = div_rn(s2.x, den, yn) - mm.x * mm.x;
    stage[5][slot] = div_rn(s2.y, den, yn) - mm.y * mm.y;
    stage[6][slot] = div_rn(s2.z, den, yn) - mm.z * mm.z;
    stage[7][slot] = div_rn(s2.w, den, yn) - mm.w * mm.w;
    stage[8][slot] = cnt > (float)min_vis ? 1.0f : 0.0f;                            // (Q4)
    __syncthreads();
    const uint32_t plane = (uint32_t)d << (2 * lc.log2d + 2);                      // bytes per plane
    const __amdgpu_buffer_rsrc_t planes = __builtin_amdgcn_make_buffer_rsrc((void*)vol, 0, (int)(8u * plane), 0x00020000);
    const __amdgpu_buffer_rsrc_t mplane = __builtin_amdgcn_make_buffer_rsrc((void*)mask, 0, (int)plane, 0x00020000);
    constexpr int NT = 2;                                                          // cache policy bits of the buffer builtins: 2 = nt
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t q = (uint32_t)tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: an SGPR plane offset
    // lane q stores four consecutive voxels of plane `wave`, `wave + 4` and (wave 0) the mask: staged floats [4q, 4q+4), which are
    // voxels 4 (q & 15) .. of row-piece q >> 4 when tiled (four 256-B pieces per plane), else voxels 4q .. of the 1-KiB chunk
    const uint32_t off = tiled ? ((((t_ix0 + (q >> 4)) << (2 * lc.log2d)) | (t_jy << lc.log2d) | (t_kz0 + ((q & 15u) << 2))) << 2) : chunk * 1024u + q * 16u;
    __builtin_amdgcn_raw_buffer_store_b128(*(const u4*)&stage[wave][4 * q], planes, off, wave * plane, NT);
    __builtin_amdgcn_raw_buffer_store_b128(*(const u4*)&stage[wave + 4][4 * q], planes, off, (wave + 4u) * plane, NT);
    if (wave == 0) __builtin_amdgcn_raw_buffer_store_b128(*(const u4*)&stage[8][4 * q], mplane, off, 0u, NT);
}

// All levels of a scene in ONE launch: the small levels (a few hundred workgroups, latency-bound on their own: 28 + 10 us for
// 128^3 + 64^3) run in the shadow of the large one.  Blocks [first[l], first[l + 1]) belong to level l.
struct VolumeLevels {
    const float4* feat[GENS_MAX_LEVELS];
    const float* intr[GENS_MAX_LEVELS];          // (nv, 4, 4) per level, rows 0-1 pre-scaled by 0.5^level
    float* vol[GENS_MAX_LEVELS];
    float* mask[GENS_MAX_LEVELS];
    int h[GENS_MAX_LEVELS], w[GENS_MAX_LEVELS], d[GENS_MAX_LEVELS];
    LevelConst lc[GENS_MAX_LEVELS];
    uint32_t first[GENS_MAX_LEVELS + 1];
    int n;
};

__global__ __launch_bounds__(256) void volume_build_fwd_levels_k(VolumeLevels lv, const float* __restrict__ w2c, int nv, int min_vis) {
    int l = 0;
    while (l + 1 < lv.n && blockIdx.x >= lv.first[l + 1]) ++l;                     // scalar: blockIdx and the table are uniform
    volume_build_chunk(blockIdx.x - lv.first[l], lv.feat[l], w2c, lv.intr[l], nv, lv.h[l], lv.w[l], lv.d[l], lv.lc[l], min_vis, lv.vol[l], lv.mask[l]);
}

__global__ __launch_bounds__(256) void volume_build_fwd_lean_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                               const float* __restrict__ intr, int nv, int h, int w, int d, LevelConst lc,
                                                               int min_vis, float* __restrict__ vol, float* __restrict__ mask) {
    volume_build_chunk(blockIdx.x, feat, w2c, intr, nv, h, w, d, lc, min_vis, vol, mask);
}

// Self-test of the exact-division shortcuts used above: every float32 bit pattern b with a normal, finite reciprocal is
// compared with the IEEE division (counts[0]: RN(1/b) mismatches), and for each b a pseudo-random numerator a is divided both
// ways (counts[1]: a/b mismatches among quotients in the normal range).  Both must be 0.
__global__ __launch_bounds__(256) void selftest_division_k(unsigned long long* __restrict__ counts) {
    unsigned long long bad_rcp = 0, bad_div = 0;
    for (uint32_t part = 0; part < 16u; ++part) {                                  // 2^28 threads x 16 bit patterns each
        const uint32_t bits = (part << 28) | (blockIdx.x * 256u + threadIdx.x);
        const float b = __uint_as_float(bits);
        const float ab = fabsf(b);
        if (!(ab >= 1.1754944e-38f * 4.0f && ab <= 8.5e37f)) continue;             // 1/b normal (NaN fails the test too)
        const float ref = 1.0f / b;
        bad_rcp += __float_as_uint(rcp_rn(b)) != __float_as_uint(ref);
        // numerator: hash of the bit pattern, exponent within 2^+-16 of b's so the quotient stays normal
        uint32_t hsh = bits * 2654435761u + 0x9e3779b9u;
        hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        const int eb = (int)((bits >> 23) & 0xffu);
        const int ea = min(max(eb + (int)(hsh >> 27) - 16, 30), 220);
        const float a = __uint_as_float((hsh & 0x807fffffu) | ((uint32_t)ea << 23));
        const float q_ref = a / b;
        if (!(fabsf(q_ref) >= 1.1754944e-38f * 4.0f && fabsf(q_ref) <= 8.5e37f)) continue;
        bad_div += __float_as_uint(div_rn(a, b, ref)) != __float_as_uint(q_ref);
    }
    if (bad_rcp) atomicAdd(counts + 0, bad_rcp);
    if (bad_div) atomicAdd(counts + 1, bad_div);
}

extern "C" int gens_selftest_division(unsigned long long* counts, void* stream) {
    GENS_CHECK_ARG(counts, GENS_EINVAL, "gens_selftest_division: null pointer");
    selftest_division_k<<<1u << 20, 256, 0, (hipStream_t)stream>>>(counts);
    return gens_launch_status("gens_selftest_division");
}

// d(volume)/d(features): recompute the projections, then scatter with the bilinear weights.
//
// The scatter is bound by float atomics (~30 G/s on this chip: 0.48 G of them, 15.6 ms, at 256^3).  Neighbouring voxels have overlapping
// 2 x 2 footprints: along z a projection moves 0 .. 1.4 pixels per voxel at the benchmark geometry (all on the same texels in the
// reference view), between x-adjacent rows ~4 pixels.  A wave therefore owns a compact tile of the volume -- 16 consecutive z of four
// x-adjacent rows -- whose footprints in one view cover a few dozen pixels of two or three image rows: it first adds its 64 x 4 taps into
// a window of LDS over the bounding box of those footprints (LDS atomics; a wave's LDS operations execute in order, so no barrier is
// needed), then sends ONE global atomic per touched texel and channel.  Bounding boxes larger than BWD_CAP texels (very wide or very
// oblique views) fall back to direct atomics, lane by lane.
#define BWD_CAP 512          // texels per wave window (8 KB; 32 KB per workgroup)

__device__ __forceinline__ void lds_add4(float* p, float4 v, float s) {
    atomicAdd(p + 0, v.x * s);
    atomicAdd(p + 1, v.y * s);
    atomicAdd(p + 2, v.z * s);
    atomicAdd(p + 3, v.w * s);
}

__global__ __launch_bounds__(256) void volume_build_bwd_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                          const float* __restrict__ intr, float s, int nv, int h, int w,
                                                          int d, int cap, const float* __restrict__ gvol, float* __restrict__ gfeat) {
    __shared__ float4 window[4][BWD_CAP];
    const int tid = threadIdx.x, lane = tid & 63;
    float4* win = window[tid >> 6];
    const int64_t n = (int64_t)d * d * d;
    // wave -> voxels: d a multiple of 16 (and of 4): 16 consecutive z of 4 x-adjacent rows (lane = 16 row + z); otherwise 64 consecutive voxels
    int64_t idx = (int64_t)blockIdx.x * 256 + tid;
    if ((d & 15) == 0) {
        const int zp = d >> 4;                                                      // 16-voxel pieces per row
        const int64_t q = (int64_t)blockIdx.x * 4 + (tid >> 6);                     // wave number: (ix / 4, jy, z piece), z piece fastest
        const int kz0 = (int)(q % zp) * 16, jy_ = (int)((q / zp) % d), ix0 = (int)(q / ((int64_t)zp * d)) * 4;
        idx = ((int64_t)(ix0 + (lane >> 4)) * d + jy_) * d + kz0 + (lane & 15);
    }
    const bool in = idx < n;                                                        // (no early return: the wave scatters together)
    const int64_t vox = in ? idx : 0;
    const int kz = (int)(vox % d), jy = (int)((vox / d) % d), ix = (int)(vox / ((int64_t)d * d));
    const float x = linspace_at(-1.0f, 1.0f, d, ix), y = linspace_at(-1.0f, 1.0f, d, jy), z = linspace_at(-1.0f, 1.0f, d, kz);
    float4 s1 = f4_zero();
    float cnt = 0.0f;
    if (in) {
        for (int v = 0; v < nv; ++v) {
            Proj p = project_voxel(w2c + 16 * v, intr + 16 * v, s, h, w, x, y, z);
            if (!p.vis) continue;
            Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
            float4 f = sample_texel(feat + (int64_t)v * h * w, h, w, 1, 0, t);
            s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
            cnt += 1.0f;
        }
    }
    const bool live = in && cnt > 0.0f;
    const float inv = live ? 1.0f / cnt : 0.0f;
    const float4 mean = make_float4(s1.x * inv, s1.y * inv, s1.z * inv, s1.w * inv);
    float4 gm = f4_zero(), gv = f4_zero();
    if (live) {
        gm = make_float4(gvol[vox], gvol[n + vox], gvol[2 * n + vox], gvol[3 * n + vox]);
        gv = make_float4(gvol[4 * n + vox], gvol[5 * n + vox], gvol[6 * n + vox], gvol[7 * n + vox]);
    }
    for (int v = 0; v < nv; ++v) {
        bool vis = false;
        Taps2 t;
        float4 g = f4_zero();
        t.x0 = t.y0 = 0;
        t.ok00 = t.ok01 = t.ok10 = t.ok11 = false;
        t.w00 = t.w01 = t.w10 = t.w11 = 0.0f;
        if (live) {
            Proj p = project_voxel(w2c + 16 * v, intr + 16 * v, s, h, w, x, y, z);
            vis = p.vis;
            if (vis) {
                t = bilinear_taps(p.ix, p.iy, h, w);
                float4 f = sample_texel(feat + (int64_t)v * h * w, h, w, 1, 0, t);
                g.x = (gm.x + 2.0f * gv.x * (f.x - mean.x)) * inv;
                g.y = (gm.y + 2.0f * gv.y * (f.y - mean.y)) * inv;
                g.z = (gm.z + 2.0f * gv.z * (f.z - mean.z)) * inv;
                g.w = (gm.w + 2.0f * gv.w * (f.w - mean.w)) * inv;
            }
        }
        if (!__any(vis)) continue;
        float* img = gfeat + (int64_t)v * h * w * 4;
        // bounding box of the wave's footprints, clipped to the image (taps outside it are dropped by their ok flags)
        const float big = 1.0e9f;
        const int x_lo = max((int)-wave_max(vis ? -(float)t.x0 : -big), 0), x_hi = min((int)wave_max(vis ? (float)(t.x0 + 1) : -big), w - 1);
        const int y_lo = max((int)-wave_max(vis ? -(float)t.y0 : -big), 0), y_hi = min((int)wave_max(vis ? (float)(t.y0 + 1) : -big), h - 1);
        const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
        if (bw > 0 && bh > 0 && bw * bh <= cap) {                                    // (wave-uniform)
            for (int i = lane; i < bw * bh; i += 64) win[i] = f4_zero();
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                  // (program order is execution order for a wave's LDS operations)
            if (vis) {
                float* b = (float*)&win[(t.y0 - y_lo) * bw + (t.x0 - x_lo)];
                if (t.ok00) lds_add4(b, g, t.w00);
                if (t.ok01) lds_add4(b + 4, g, t.w01);
                if (t.ok10) lds_add4(b + 4 * bw, g, t.w10);
                if (t.ok11) lds_add4(b + 4 * bw + 4, g, t.w11);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (int i = lane; i < 4 * bw * bh; i += 64) {                          // a lane per FLOAT: consecutive lanes, consecutive addresses (L2 serves
                const float a = ((const float*)win)[i];                             // atomic requests, not lanes: 190 - 280 against 66 G/s)
                if (a != 0.0f) {
                    const int texel = i >> 2, r = texel / bw, c = texel - r * bw;
                    atomicAdd(img + ((int64_t)(y_lo + r) * w + x_lo + c) * 4 + (i & 3), a);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        } else if (vis) {
            float* base = img + ((int64_t)t.y0 * w + t.x0) * 4;
            if (t.ok00) atomic_add4(base, g, t.w00);
            if (t.ok01) atomic_add4(base + 4, g, t.w01);
            if (t.ok10) atomic_add4(base + (int64_t)w * 4, g, t.w10);
            if (t.ok11) atomic_add4(base + (int64_t)w * 4 + 4, g, t.w11);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K1 backward, second generation: the IMAGE TILE owns the sum.
//
// The wave windows above reduce what one 4 x 16 voxel tile sends to a view by ~4 x (a z run slides 0 .. 1.4 pixels per voxel), and what is
// left -- ~0.3 G global float atomics at 256^3 -- is the kernel's whole time (~66 G atomics/s on this chip, whatever the tile shape:
// scripts/probe/README.md).  But a view's gradient image has only nv x H x W x 4 = 6 M floats for 1.3 G taps: nearly all of the reduction
// is between voxels far apart in the volume that lie along the same viewing rays.  So the sum is turned around:
//   prep   one pass over the voxels (the same 4 x 16 wave tiles): count / mean of the visible views, the per-voxel factors of
//          g_view = A + B * (f_view - M)   (A = g_mean / count, B = 2 g_var / count, M = mean)   to scratch, and for every view the image
//          tiles (BT_W x BT_H texels, by the north-west tap) the wave's voxels fall into: at most 2 x 2, else that wave / view pair
//          scatters directly as before;
//   bins   counting sort of the (wave tile, view) pairs by image tile (count, one-workgroup scan, fill: the lanes of a wave that address one
//          bin send one atomic between them), cut into work items of at most `seg` wave tiles;
//   tiles  a workgroup per work item keeps the gradient of its image tile (+ one texel of halo for the south / east taps) in LDS, walks
//          its wave tiles -- reads A, B, M, re-projects into ITS view with the same arithmetic as prep, and adds the taps of the voxels
//          that belong to the tile with LDS atomics (64-bit fixed point: tile_add4) -- and sends each touched texel to memory once.
// Every (voxel, view) pair is owned by exactly one image tile, so the sums are those of the direct scatter in another order.
#define BT_W 64
#define BT_H 30          // (65 x 31 texels x 4 channels x 8 bytes = 63 KB of LDS per workgroup; 480 / 240 / 120 rows are whole tiles)
#define BT_SEG 1024
#define BT_THREADS 512
#define BT_WIN ((BT_W + 1) * (BT_H + 1))

struct BwdScratch {
    float4 *a, *b, *m;              // per voxel
    uint32_t* range;                // per (wave tile, view): tile range code, 0 = none
    uint32_t *count, *cursor, *offset;   // per bin = (view, tile row, tile column); offset has one more
    uint32_t *n_items, *gmax;       // gmax: bits of an upper bound of |g_view| over all voxels, views and channels (fixed-point scale)
    uint4* items;                   // (bin, begin, end, -)
    uint32_t* list;                 // wave tiles, bin after bin
    int tiles_x, tiles_y, n_bins, seg;   // seg: wave tiles per work item
    uint32_t max_items;
    int64_t n_waves;
};

static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

static int64_t bwd_scratch_layout(int nv, int h, int w, int d, char* base, BwdScratch* out) {
    const int64_t n = (int64_t)d * d * d, nw = n / 64;
    BwdScratch sc;
    sc.tiles_x = (w + BT_W - 1) / BT_W;
    sc.tiles_y = (h + BT_H - 1) / BT_H;
    sc.n_bins = nv * sc.tiles_x * sc.tiles_y;
    sc.n_waves = nw;
    sc.seg = (int)std::min<int64_t>(BT_SEG, std::max<int64_t>(64, nw * nv / 512 / 64 * 64));     // (small levels: shorter items, more of them)
    sc.max_items = (uint32_t)((nw * nv * 4 + sc.seg - 1) / sc.seg + sc.n_bins);
    int64_t at = 0;
    auto take = [&](int64_t bytes) { char* p = base ? base + at : nullptr; at += align256(bytes); return p; };
    sc.a = (float4*)take(n * 16);
    sc.b = (float4*)take(n * 16);
    sc.m = (float4*)take(n * 16);
    sc.range = (uint32_t*)take(nw * nv * 4);
    sc.count = (uint32_t*)take((int64_t)sc.n_bins * 4);        // count, cursor, n_items, gmax: one block, zeroed per call
    sc.cursor = (uint32_t*)take((int64_t)sc.n_bins * 4);
    sc.n_items = (uint32_t*)take(4);
    sc.gmax = (uint32_t*)take(4);
    sc.offset = (uint32_t*)take((int64_t)(sc.n_bins + 1) * 4);
    sc.items = (uint4*)take((int64_t)sc.max_items * 16);
    sc.list = (uint32_t*)take(nw * nv * 4 * 4);
    if (out) *out = sc;
    return at;
}

// wave tile q = (ix / 4, jy, z piece), z piece fastest, as (z piece | jy << 8 | ix / 4 << 20): 16 consecutive z of 4 x-adjacent rows, lane = 16 row + z
__device__ __forceinline__ uint32_t bwd_wave_code(uint32_t q, int d) {
    const uint32_t zp = (uint32_t)d >> 4, r = q / zp;
    return (q - r * zp) | ((r % (uint32_t)d) << 8) | ((r / (uint32_t)d) << 20);
}
struct WaveVoxel {
    int ix, jy, kz;
    int64_t vox;
};
__device__ __forceinline__ WaveVoxel bwd_wave_voxel(uint32_t code, int lane, int d) {
    WaveVoxel o;
    o.kz = (int)(code & 0xFFu) * 16 + (lane & 15);
    o.jy = (int)((code >> 8) & 0xFFFu);
    o.ix = (int)(code >> 20) * 4 + (lane >> 4);
    o.vox = ((int64_t)o.ix * d + o.jy) * d + o.kz;
    return o;
}

__global__ __launch_bounds__(256) void volume_build_bwd_prep_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                               const float* __restrict__ intr, float s, int nv, int h, int w, int d,
                                                               const float* __restrict__ gvol, float* __restrict__ gfeat, BwdScratch sc) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t n = (int64_t)d * d * d;
    const uint32_t q = blockIdx.x * 4u + (uint32_t)(tid >> 6);
    const WaveVoxel wv = bwd_wave_voxel(bwd_wave_code(q, d), lane, d);              // (d a multiple of 16: every lane has a voxel)
    const int64_t vox = wv.vox;
    const float x = linspace_at(-1.0f, 1.0f, d, wv.ix), y = linspace_at(-1.0f, 1.0f, d, wv.jy), z = linspace_at(-1.0f, 1.0f, d, wv.kz);
    float4 s1 = f4_zero();
    float cnt = 0.0f;
    uint32_t direct = 0;                                                            // (wave-uniform) views whose footprint spans more than 2 x 2 tiles
    const float big = 1.0e9f;
    float4 f_lo = make_float4(big, big, big, big), f_hi = make_float4(-big, -big, -big, -big);
    for (int v = 0; v < nv; ++v) {
        Proj p = project_voxel(w2c + 16 * v, intr + 16 * v, s, h, w, x, y, z);
        Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
        if (p.vis) {
            float4 f = sample_texel(feat + (int64_t)v * h * w, h, w, 1, 0, t);
            s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
            cnt += 1.0f;
            f_lo = make_float4(fminf(f_lo.x, f.x), fminf(f_lo.y, f.y), fminf(f_lo.z, f.z), fminf(f_lo.w, f.w));
            f_hi = make_float4(fmaxf(f_hi.x, f.x), fmaxf(f_hi.y, f.y), fmaxf(f_hi.z, f.z), fmaxf(f_hi.w, f.w));
        }
        uint32_t code = 0;
        if (__any(p.vis)) {                                                         // visible: 0 <= x0 <= w - 1, 0 <= y0 <= h - 1
            const int tx_lo = (int)-wave_max(p.vis ? -(float)t.x0 : -big) / BT_W, tx_hi = (int)wave_max(p.vis ? (float)t.x0 : -big) / BT_W;
            const int ty_lo = (int)-wave_max(p.vis ? -(float)t.y0 : -big) / BT_H, ty_hi = (int)wave_max(p.vis ? (float)t.y0 : -big) / BT_H;
            if (tx_hi - tx_lo > 1 || ty_hi - ty_lo > 1) {
                direct |= 1u << v;
            } else {
                code = 0x80000000u | (uint32_t)tx_lo | ((uint32_t)ty_lo << 12) | ((uint32_t)(tx_hi - tx_lo) << 24) | ((uint32_t)(ty_hi - ty_lo) << 25);
            }
        }
        if (lane == 0) sc.range[(int64_t)v * sc.n_waves + q] = code;
    }
    const bool live = cnt > 0.0f;
    const float inv = live ? 1.0f / cnt : 0.0f;
    const float4 mean = make_float4(s1.x * inv, s1.y * inv, s1.z * inv, s1.w * inv);
    float4 ga = f4_zero(), gb = f4_zero();
    if (live) {
        ga = make_float4(gvol[vox] * inv, gvol[n + vox] * inv, gvol[2 * n + vox] * inv, gvol[3 * n + vox] * inv);
        gb = make_float4(2.0f * gvol[4 * n + vox] * inv, 2.0f * gvol[5 * n + vox] * inv, 2.0f * gvol[6 * n + vox] * inv, 2.0f * gvol[7 * n + vox] * inv);
    }
    sc.a[vox] = ga;
    sc.b[vox] = gb;
    sc.m[vox] = mean;
    // |g_view| <= |A| + |B| (max f - min f) in every channel: the scale of the tile kernel's fixed-point sums.  Compared as BITS of
    // non-negative floats, so a NaN or an infinity (which sends that kernel to its float path) wins over every finite bound.
    uint32_t bound = 0;
    if (live) {
        bound = max(max(__float_as_uint(fabsf(ga.x) + fabsf(gb.x) * (f_hi.x - f_lo.x)), __float_as_uint(fabsf(ga.y) + fabsf(gb.y) * (f_hi.y - f_lo.y))),
                    max(__float_as_uint(fabsf(ga.z) + fabsf(gb.z) * (f_hi.z - f_lo.z)), __float_as_uint(fabsf(ga.w) + fabsf(gb.w) * (f_hi.w - f_lo.w))));
        bound = max(bound, max(max(__float_as_uint(fabsf(mean.x)), __float_as_uint(fabsf(mean.y))), max(__float_as_uint(fabsf(mean.z)), __float_as_uint(fabsf(mean.w)))) >= 0x7f800000u ? 0x7fc00000u : 0u);
    }
    for (int o = 32; o; o >>= 1) bound = max(bound, (uint32_t)__shfl_xor((int)bound, o));
    if (lane == 0 && bound > *(volatile uint32_t*)sc.gmax) atomicMax(sc.gmax, bound);     // (a stale read only costs an atomic)
    for (int v = 0; direct >> v; ++v) {                                             // (rare: very oblique or very wide views)
        if (!((direct >> v) & 1u) || !live) continue;
        Proj p = project_voxel(w2c + 16 * v, intr + 16 * v, s, h, w, x, y, z);
        if (!p.vis) continue;
        Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
        float4 f = sample_texel(feat + (int64_t)v * h * w, h, w, 1, 0, t);
        const float4 g = make_float4(ga.x + gb.x * (f.x - mean.x), ga.y + gb.y * (f.y - mean.y), ga.z + gb.z * (f.z - mean.z), ga.w + gb.w * (f.w - mean.w));
        float* base = gfeat + (((int64_t)v * h + t.y0) * w + t.x0) * 4;
        if (t.ok00) atomic_add4(base, g, t.w00);
        if (t.ok01) atomic_add4(base + 4, g, t.w01);
        if (t.ok10) atomic_add4(base + (int64_t)w * 4, g, t.w10);
        if (t.ok11) atomic_add4(base + (int64_t)w * 4 + 4, g, t.w11);
    }
}

// offsets of the bins in the list and the work items (one workgroup; the bins number a few hundred)
__global__ __launch_bounds__(256) void volume_build_bwd_scan_k(BwdScratch sc) {
    __shared__ uint32_t part_c[256], part_i[256];
    const int tid = threadIdx.x, per = (sc.n_bins + 255) / 256;
    const uint32_t seg = (uint32_t)sc.seg;
    const int lo = min(tid * per, sc.n_bins), hi = min(lo + per, sc.n_bins);
    uint32_t c = 0, it = 0;
    for (int b = lo; b < hi; ++b) {
        c += sc.count[b];
        it += (sc.count[b] + seg - 1) / seg;
    }
    part_c[tid] = c;
    part_i[tid] = it;
    __syncthreads();
    if (tid == 0) {
        uint32_t rc = 0, ri = 0;
        for (int i = 0; i < 256; ++i) {
            const uint32_t tc = part_c[i], ti = part_i[i];
            part_c[i] = rc;
            part_i[i] = ri;
            rc += tc;
            ri += ti;
        }
        sc.offset[sc.n_bins] = rc;
        *sc.n_items = ri;
    }
    __syncthreads();
    c = part_c[tid];
    it = part_i[tid];
    for (int b = lo; b < hi; ++b) {
        const uint32_t k = sc.count[b];
        sc.offset[b] = c;
        for (uint32_t at = 0; at < k; at += seg) sc.items[it++] = make_uint4((uint32_t)b, c + at, c + min(at + seg, k), 0u);
        c += k;
    }
}

// The (wave tile, view) pairs of one view, counted per bin (FILL = false) or written to their bins' lists (FILL = true).  Neighbouring wave tiles
// mostly share their bins: the lanes of a wave that address one bin send ONE atomic for all of them (the bins are a few hundred addresses).
template <bool FILL>
__global__ __launch_bounds__(256) void volume_build_bwd_bins_k(BwdScratch sc, int d) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t code = q < sc.n_waves ? sc.range[(int64_t)v * sc.n_waves + q] : 0u;
    const uint32_t wave_code = FILL ? bwd_wave_code((uint32_t)min(q, sc.n_waves - 1), d) : 0u;
    const int tx_lo = code & 0xFFF, ty_lo = (code >> 12) & 0xFFF, nx = (code >> 24) & 1, ny = (code >> 25) & 1;
    for (int k = 0; k < 4; ++k) {                                                   // the up to 2 x 2 tiles of the pair
        const int dx = k & 1, dy = k >> 1;
        int bin = (code && dx <= nx && dy <= ny) ? (v * sc.tiles_y + ty_lo + dy) * sc.tiles_x + tx_lo + dx : -1;
        unsigned long long todo = __ballot(bin >= 0);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int b = __builtin_amdgcn_readlane(bin, lead);
            const unsigned long long same = __ballot(bin == b);
            uint32_t base = 0;
            if (lane == lead) base = atomicAdd((FILL ? sc.cursor : sc.count) + b, (uint32_t)__popcll(same));
            if (FILL && bin == b) {
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, lead);
                sc.list[sc.offset[b] + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = wave_code;
            }
            todo &= ~same;
            if (bin == b) bin = -1;
        }
    }
}

// One tap into the tile window.  FIXED: 64-bit fixed-point sums (ds_add_u64: ~18 cycles per wave instruction, against ~200-250 for ds_add_f32
// on this chip -- scripts/probe/lds_atomic_probe.py -- and the window's adds are this kernel's whole time).  v * w * scale is rounded to an integer
// by the 1.5 x 2^52 trick (|v w scale| < 2^42, one tap is exact to 2^-41 of the largest |g| of the call, a work item sums < 2^18 taps per texel:
// no overflow, and the error of a texel is far below one float32 rounding of its sum).  Otherwise (a NaN / infinity among the gradients) float adds.
template <bool FIXED>
__device__ __forceinline__ void tile_add4(void* win, int at, float4 v, float w, double scale) {
    if (FIXED) {
        unsigned long long* q = (unsigned long long*)win + at;
        const double ws = (double)w * scale, magic = 6755399441055744.0;
        atomicAdd(q, (unsigned long long)__double_as_longlong(__builtin_fma((double)v.x, ws, magic)) - 0x4338000000000000ull);
        atomicAdd(q + BT_WIN, (unsigned long long)__double_as_longlong(__builtin_fma((double)v.y, ws, magic)) - 0x4338000000000000ull);
        atomicAdd(q + 2 * BT_WIN, (unsigned long long)__double_as_longlong(__builtin_fma((double)v.z, ws, magic)) - 0x4338000000000000ull);
        atomicAdd(q + 3 * BT_WIN, (unsigned long long)__double_as_longlong(__builtin_fma((double)v.w, ws, magic)) - 0x4338000000000000ull);
    } else {
        float* q = (float*)win + at;
        atomicAdd(q, v.x * w);
        atomicAdd(q + BT_WIN, v.y * w);
        atomicAdd(q + 2 * BT_WIN, v.z * w);
        atomicAdd(q + 3 * BT_WIN, v.w * w);
    }
}

template <bool FIXED>
__device__ __forceinline__ void bwd_tile_item(unsigned long long* win, const float4* __restrict__ feat, const float* __restrict__ w2c,
                                              const float* __restrict__ intr, float s, int h, int w, int d, float* __restrict__ gfeat,
                                              const BwdScratch& sc, double scale, double inv_scale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint4 item = sc.items[blockIdx.x];
    const int bin = (int)item.x, tx = bin % sc.tiles_x, ty = (bin / sc.tiles_x) % sc.tiles_y, v = bin / (sc.tiles_x * sc.tiles_y);
    const int x_org = tx * BT_W, y_org = ty * BT_H;
    for (int i = tid; i < 4 * BT_WIN; i += BT_THREADS) win[i] = 0ull;              // (the float path uses the first half)
    __syncthreads();
    const float4* img = feat + (int64_t)v * h * w;
    const float *m = w2c + 16 * v, *k = intr + 16 * v;
    for (uint32_t e0 = item.y + 64u * wave; e0 < item.z; e0 += 64u * (BT_THREADS / 64)) {
        const uint32_t mine = e0 + lane < item.z ? sc.list[e0 + lane] : 0u;       // the wave's next 64 wave tiles, one per lane
        const int cnt = (int)min(64u, item.z - e0);
        for (int j = 0; j < cnt; ++j) {
            const WaveVoxel wv = bwd_wave_voxel((uint32_t)__builtin_amdgcn_readlane((int)mine, j), lane, d);
            const float4 ga = sc.a[wv.vox], gb = sc.b[wv.vox], mean = sc.m[wv.vox];
            const float x = linspace_at(-1.0f, 1.0f, d, wv.ix), y = linspace_at(-1.0f, 1.0f, d, wv.jy), z = linspace_at(-1.0f, 1.0f, d, wv.kz);
            Proj p = project_voxel(m, k, s, h, w, x, y, z);
            Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
            const int cx = t.x0 - x_org, cy = t.y0 - y_org;
            const bool live = ga.x != 0.0f || ga.y != 0.0f || ga.z != 0.0f || ga.w != 0.0f || gb.x != 0.0f || gb.y != 0.0f || gb.z != 0.0f || gb.w != 0.0f;
            if (p.vis && live && cx >= 0 && cx < BT_W && cy >= 0 && cy < BT_H) {
                float4 f = sample_texel(img, h, w, 1, 0, t);
                const float4 g = make_float4(ga.x + gb.x * (f.x - mean.x), ga.y + gb.y * (f.y - mean.y), ga.z + gb.z * (f.z - mean.z), ga.w + gb.w * (f.w - mean.w));
                const int at = cy * (BT_W + 1) + cx;
                if (t.ok00) tile_add4<FIXED>(win, at, g, t.w00, scale);
                if (t.ok01) tile_add4<FIXED>(win, at + 1, g, t.w01, scale);
                if (t.ok10) tile_add4<FIXED>(win, at + (BT_W + 1), g, t.w10, scale);
                if (t.ok11) tile_add4<FIXED>(win, at + (BT_W + 1) + 1, g, t.w11, scale);
            }
        }
    }
    __syncthreads();
    float* out = gfeat + (int64_t)v * h * w * 4;
    for (int i = tid; i < 4 * BT_WIN; i += BT_THREADS) {      // a lane per FLOAT: the lanes of an atomic instruction on consecutive addresses (L2 serves requests, not lanes)
        const int texel = i >> 2, c = i & 3;
        const float a = FIXED ? (float)((double)(long long)win[c * BT_WIN + texel] * inv_scale) : ((const float*)win)[c * BT_WIN + texel];
        if (a != 0.0f) {
            const int r = texel / (BT_W + 1), cc = texel - r * (BT_W + 1);
            atomicAdd(out + ((int64_t)(y_org + r) * w + x_org + cc) * 4 + c, a);      // (only in-image taps were added: the texel exists)
        }
    }
}

__global__ __launch_bounds__(BT_THREADS) void volume_build_bwd_tiles_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                                      const float* __restrict__ intr, float s, int h, int w, int d,
                                                                      float* __restrict__ gfeat, BwdScratch sc) {
    __shared__ unsigned long long win[4 * BT_WIN];      // channel planes: the lanes of one add (one channel) spread over all banks
    if (blockIdx.x >= *sc.n_items) return;
    const uint32_t e = (*sc.gmax >> 23) & 0xFFu;                                    // biased exponent of the bound: 2^(e - 127) <= bound < 2^(e - 126)
    if (e != 0xFFu) {
        const double scale = __longlong_as_double((long long)(1023 + 40 + 127 - (int)e) << 52);      // bound * scale in [2^40, 2^41)
        const double inv_scale = __longlong_as_double((long long)(1023 - 40 - 127 + (int)e) << 52);
        bwd_tile_item<true>(win, feat, w2c, intr, s, h, w, d, gfeat, sc, scale, inv_scale);
    } else {
        bwd_tile_item<false>(win, feat, w2c, intr, s, h, w, d, gfeat, sc, 1.0, 1.0);
    }
}

static int check_volume_args(const char* who, const void* a, const void* b, const void* c, int nv, int h, int w, int d) {
    GENS_CHECK_ARG(a && b && c, GENS_EINVAL, "%s: null pointer", who);
    GENS_CHECK_ARG(nv > 0 && h > 1 && w > 1 && d > 0, GENS_EINVAL, "%s: bad size nv=%d h=%d w=%d d=%d", who, nv, h, w, d);
    GENS_CHECK_ARG(nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "%s: nv=%d > %d", who, nv, GENS_MAX_VIEWS);
    return 0;
}

static LevelConst level_const(int h, int w, int d) {      // the float32 operations the reference performs per voxel, done once
    LevelConst lc;
    lc.step = (1.0f - (-1.0f)) / (float)(d - 1);
    lc.cw = (float)(w - 1) / 2.0f;
    lc.ch = (float)(h - 1) / 2.0f;
    lc.rcw = 1.0f / lc.cw;
    lc.rch = 1.0f / lc.ch;
    lc.log2d = 0;
    while ((1 << lc.log2d) < d) ++lc.log2d;
    return lc;
}

extern "C" int gens_volume_build_fwd(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv,
                                     int h, int w, int d, int min_vis_view, float* volume, float* mask, void* stream) {
    if (int e = check_volume_args("gens_volume_build_fwd", feat, w2c, intr, nv, h, w, d)) return e;
    GENS_CHECK_ARG(volume && mask, GENS_EINVAL, "gens_volume_build_fwd: null output");
    int64_t n = (int64_t)d * d * d;
    const bool pow2 = d >= 2 && d <= 256 && (d & (d - 1)) == 0;
    if (pow2 && !getenv("GENS_K1_GENERIC")) {                 // (the environment switch keeps the generic kernel reachable for A/B tests)
        const LevelConst lc = level_const(h, w, d);
        if (d >= 8 && nv <= K1_FAST_VIEWS && intr_scale == 1.0f && !getenv("GENS_K1_SINGLE")) {   // production path (the switch keeps the previous kernel reachable for A/B runs)
            volume_build_fwd_lean_k<<<(unsigned)(n / 256), 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, nv, h, w, d, lc,
                                                                                     min_vis_view, volume, mask);
            return gens_launch_status("gens_volume_build_fwd");
        }
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

extern "C" int gens_volume_build_levels(const float* const* feat, const int* hw, const int* dims, int n_levels, const float* w2c,
                                        const float* const* intr, int nv, int min_vis_view, float* const* volumes, float* const* masks,
                                        void* stream) {
    GENS_CHECK_ARG(feat && hw && dims && w2c && intr && volumes && masks, GENS_EINVAL, "gens_volume_build_levels: null table");
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= GENS_MAX_LEVELS, GENS_ELIMIT, "gens_volume_build_levels: %d levels (1..%d)", n_levels, GENS_MAX_LEVELS);
    bool one_launch = nv <= K1_FAST_VIEWS && !getenv("GENS_K1_GENERIC") && !getenv("GENS_K1_SINGLE") && !getenv("GENS_K1_PER_LEVEL");
    for (int l = 0; l < n_levels; ++l) {
        if (int e = check_volume_args("gens_volume_build_levels", feat[l], w2c, intr[l], nv, hw[2 * l], hw[2 * l + 1], dims[l])) return e;
        GENS_CHECK_ARG(volumes[l] && masks[l], GENS_EINVAL, "gens_volume_build_levels: null output (level %d)", l);
        const int d = dims[l];
        one_launch = one_launch && d >= 8 && d <= 256 && (d & (d - 1)) == 0;
    }
    if (!one_launch) {                                                             // sizes the fused kernel does not cover: level by level
        for (int l = 0; l < n_levels; ++l)
            if (int e = gens_volume_build_fwd(feat[l], w2c, intr[l], 1.0f, nv, hw[2 * l], hw[2 * l + 1], dims[l], min_vis_view, volumes[l], masks[l], stream))
                return e;
        return 0;
    }
    VolumeLevels lv;
    lv.n = n_levels;
    lv.first[0] = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int d = dims[l];
        lv.feat[l] = (const float4*)feat[l];
        lv.intr[l] = intr[l];
        lv.vol[l] = volumes[l];
        lv.mask[l] = masks[l];
        lv.h[l] = hw[2 * l];
        lv.w[l] = hw[2 * l + 1];
        lv.d[l] = d;
        lv.lc[l] = level_const(hw[2 * l], hw[2 * l + 1], d);
        lv.first[l + 1] = lv.first[l] + (uint32_t)(((int64_t)d * d * d) / 256);
    }
    volume_build_fwd_levels_k<<<lv.first[n_levels], 256, 0, (hipStream_t)stream>>>(lv, w2c, nv, min_vis_view);
    return gens_launch_status("gens_volume_build_levels");
}

extern "C" int64_t gens_volume_build_bwd_scratch_bytes(int nv, int h, int w, int d) {
    if (nv <= 0 || nv > GENS_MAX_VIEWS || h <= 1 || w <= 1 || d <= 0 || d > 4096 || (d & 15) || (w + BT_W - 1) / BT_W > 4095 || (h + BT_H - 1) / BT_H > 4095) return 0;
    if ((int64_t)d * d * d / 64 * nv * 4 > 0xFFFFFFFFll) return 0;                  // (list offsets are 32-bit)
    return bwd_scratch_layout(nv, h, w, d, nullptr, nullptr);
}

extern "C" int gens_volume_build_bwd_tiled(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv,
                                           int h, int w, int d, const float* g_volume, float* g_feat, void* scratch,
                                           int64_t scratch_bytes, void* stream) {
    if (int e = check_volume_args("gens_volume_build_bwd_tiled", feat, w2c, intr, nv, h, w, d)) return e;
    GENS_CHECK_ARG(g_volume && g_feat && scratch, GENS_EINVAL, "gens_volume_build_bwd_tiled: null buffer");
    GENS_CHECK_ARG(((uintptr_t)feat & 15) == 0, GENS_EINVAL, "gens_volume_build_bwd_tiled: the texels are read as float4 and must be 16-byte aligned");
    const int64_t need = gens_volume_build_bwd_scratch_bytes(nv, h, w, d);
    GENS_CHECK_ARG(need > 0, GENS_ELIMIT, "gens_volume_build_bwd_tiled: d=%d must be a multiple of 16 (image %d x %d, %d views): use gens_volume_build_bwd", d, h, w, nv);
    GENS_CHECK_ARG(scratch_bytes >= need && ((uintptr_t)scratch & 15) == 0, GENS_EINVAL,
                   "gens_volume_build_bwd_tiled: scratch of %lld bytes, 16-byte aligned, needed (got %lld)", (long long)need, (long long)scratch_bytes);
    BwdScratch sc;
    bwd_scratch_layout(nv, h, w, d, (char*)scratch, &sc);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sc.count, 0, (char*)sc.offset - (char*)sc.count, st) != hipSuccess) return gens_launch_status("gens_volume_build_bwd_tiled");
    volume_build_bwd_prep_k<<<(unsigned)(sc.n_waves / 4), 256, 0, st>>>((const float4*)feat, w2c, intr, intr_scale, nv, h, w, d, g_volume, g_feat, sc);
    const dim3 bins_grid(gens_blocks(sc.n_waves, 256), (unsigned)nv);
    volume_build_bwd_bins_k<false><<<bins_grid, 256, 0, st>>>(sc, d);
    volume_build_bwd_scan_k<<<1, 256, 0, st>>>(sc);
    volume_build_bwd_bins_k<true><<<bins_grid, 256, 0, st>>>(sc, d);
    volume_build_bwd_tiles_k<<<sc.max_items, BT_THREADS, 0, st>>>((const float4*)feat, w2c, intr, intr_scale, h, w, d, g_feat, sc);
    return gens_launch_status("gens_volume_build_bwd_tiled");
}

extern "C" int gens_volume_build_bwd(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv,
                                     int h, int w, int d, const float* g_volume, float* g_feat, void* stream) {
    if (int e = check_volume_args("gens_volume_build_bwd", feat, w2c, intr, nv, h, w, d)) return e;
    GENS_CHECK_ARG(g_volume && g_feat, GENS_EINVAL, "gens_volume_build_bwd: null gradient buffer");
    int64_t n = (int64_t)d * d * d;
    const int cap = getenv("GENS_K1_BWD_DIRECT") ? 0 : BWD_CAP;                    // (switch: every tap a global atomic, for A/B runs)
    volume_build_bwd_k<<<gens_blocks(n, 256), 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv,
                                                                            h, w, d, cap, g_volume, g_feat);
    return gens_launch_status("gens_volume_build_bwd");
}
