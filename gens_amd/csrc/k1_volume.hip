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
    GENS_CHECK_ARG(n < (1 << 15) && c < (1 << 15) && h < (1 << 15) && w < (1 << 15), GENS_EINVAL, "gens_pack_nchw: a dimension of 32768 or more");
    int q4 = (c + 3) / 4;
    int64_t total = (int64_t)n * h * w * q4;
    pack_nchw_k<<<gens_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(src, (float4*)dst, c, (int64_t)h * w, q4, total);
    return gens_launch_status("gens_pack_nchw");
}

extern "C" int gens_unpack_nhwc(const float* src, float* dst, int n, int c, int h, int w, void* stream) {
    GENS_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0, GENS_EINVAL, "gens_unpack_nhwc: bad argument");
    GENS_CHECK_ARG(n < (1 << 15) && c < (1 << 15) && h < (1 << 15) && w < (1 << 15), GENS_EINVAL, "gens_unpack_nhwc: a dimension of 32768 or more");
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
                                                          float* __restrict__ mask, uint8_t* __restrict__ count) {
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
    if (count) count[idx] = (uint8_t)cnt;                                    // the visible views, kept for the backward pass
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
    int no_shortcut;     // (A/B switch GENS_K1_NO_EMPTY_SHORTCUT: tiles no view reaches take the general path too)
    int pieces;          // 64-voxel pieces of its rows a workgroup of the frustum-culled kernel takes (1, 2 or 4; GENS_K1_PIECES: A/B)
};

template <bool PRESCALED>
__global__ __launch_bounds__(256) void volume_build_fwd_pow2_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                               const float* __restrict__ intr, float s, int nv, int h, int w, int d,
                                                               LevelConst lc, int min_vis, float* __restrict__ vol, float* __restrict__ mask,
                                                               uint8_t* __restrict__ count) {
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
    if (count) count[idx] = (uint8_t)cnt;
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
    const float t = g0 * __builtin_amdgcn_rcpf(g0 - g1);                          // zero crossing, |error| < 1e-3 (a quarter voxel at d = 256); 1 ulp of the reciprocal is nothing beside the two voxels the span is widened by
    if (g0 < g1) lo = fmaxf(lo, t); else hi = fminf(hi, t);
}

typedef const __attribute__((address_space(4))) float* k1_cfloat_p;
typedef const __attribute__((address_space(4))) uint32_t* k1_cuint_p;
// the backward kernels' matrix pointers: plain global pointers (K1_BWD_CONST_AS: the constant address space there too -- measured: no gain, and a
// captured training step then faulted on its second replay, profiles/r05_k1_bwd_const_as_fault.txt)
#ifdef K1_BWD_CONST_AS
typedef k1_cfloat_p k1b_float_p;
typedef k1_cuint_p k1b_uint_p;
__device__ __forceinline__ k1b_float_p k1_const(const float* p) { return (k1b_float_p)(uintptr_t)p; }
#else
typedef const float* k1b_float_p;
typedef const uint32_t* k1b_uint_p;
__device__ __forceinline__ k1b_float_p k1_const(const float* p) { return p; }
#endif

__device__ __forceinline__ void volume_build_chunk(uint32_t group, const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                   const float* __restrict__ intr, int nv, int h, int w, int d, LevelConst lc, int min_vis,
                                                   float* __restrict__ vol, float* __restrict__ mask, uint8_t* __restrict__ count,
                                                   uint32_t* __restrict__ bits = nullptr) {
    // Chunk -> voxels.  d >= 64: the four waves take the SAME 64 z of four x-adjacent rows (ix = 4 g + wave), whose image footprints
    // overlap, so most of a wave's texel lines are already in the CU's L1 (the z-contiguous 256-voxel chunk sent 3-4x as many requests
    // to L2); smaller volumes: 256 consecutive voxels = 256 / d whole rows.
    const uint32_t dm = (uint32_t)d - 1u;
    const bool tiled = lc.log2d >= 6;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t zq_bits = (uint32_t)lc.log2d - 6u;                              // (tiled) a row is 2^zq_bits pieces of 64 voxels
    // A workgroup takes lc.pieces CONSECUTIVE pieces of its four rows (all of them up to d = 256), one after the other: the frustum spans and the
    // z-row-invariant halves of the projection below belong to the ROWS -- a workgroup per piece computed them once per piece (~170 dependent vector
    // instructions and 28 loads on one wave in front of everything else: most of what a tile no view reaches costs, an eighth of the launch's
    // instructions), and there are a quarter as many workgroups to start.
    const uint32_t np = tiled ? (uint32_t)lc.pieces : 1u, chunk = group * np;       // (np is a power of two <= 2^zq_bits: the pieces share ix, jy)
    const uint32_t t_jy = (chunk >> zq_bits) & dm, t_ix0 = (chunk >> (zq_bits + lc.log2d)) << 2;
    // within the tile a wave takes 16 z of all four rows (lane = 16 row + z): its footprints in a view then span ~30 pixels instead of ~90
    const uint32_t t_row = lane >> 4, t_z = (wv << 4) | (lane & 15u);
    const int half = d >> 1;
    // ---- frustum culling per (z-row, view).  Along a z-row the homogeneous image coordinates (u, v, depth) are affine in z, so the
    // voxels of the row a view can see form ONE interval of kz.  Thread (row r, view v) of the workgroup intersects the five half-lines
    // depth > 0, 0 <= u <= (w-1) depth, 0 <= v <= (h-1) depth from their values at the two ends of the row, widens the result by two
    // voxels, and leaves it in LDS; in the view loop a wave whose lanes all sit outside the interval skips the view before any per-voxel
    // arithmetic (64 % of all (wave, view) pairs at the benchmark geometry, where that arithmetic was 55 % of the kernel's instructions).
    // The exact per-voxel test below still decides visibility; the intervals only have to be supersets, and the bit-for-bit
    // comparisons with the unculled kernels check that they are.
    __shared__ int2 row_span[32][K1_FAST_VIEWS];
    // ... and the part of the camera-space coordinates that is the same along a z-row: m[4 q] x + m[4 q + 1] y, the first two terms of
    // m[4 q] x + m[4 q + 1] y + m[4 q + 2] z + m[4 q + 3] as the reference's left-to-right float32 sums form them (two products, one addition:
    // the same three roundings here as there, so the voxel's  (p + m[4 q + 2] z) + m[4 q + 3]  is bit for bit the four-term expression).  Nine vector
    // instructions per (voxel, view) become one broadcast 16-byte LDS read: the kernel is bound by its instruction stream (78 % VALU-busy).
    __shared__ float4 row_pre[32][K1_FAST_VIEWS];
    __shared__ uint32_t piece_seen;                                               // bit p: some (row, view) span of the tile reaches piece p
    if (threadIdx.x == 0) piece_seen = lc.no_shortcut ? 0xFFFFFFFFu : 0u;
    __syncthreads();
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
            row_pre[r][v] = make_float4(m[0] * rx + m[1] * ry, m[4] * rx + m[5] * ry, m[8] * rx + m[9] * ry, m[12] * rx + m[13] * ry);
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
            if (span.x <= span.y) {
                const int kz_first = tiled ? (int)((chunk & ((1u << zq_bits) - 1u)) << 6) : 0;
                uint32_t bits = 0u;
                for (uint32_t p = 0; p < np; ++p) {
                    const int lo_p = tiled ? kz_first + (int)(p << 6) : 0, hi_p = tiled ? lo_p + 63 : d - 1;
                    bits |= (span.x <= hi_p && span.y >= lo_p) ? 1u << p : 0u;
                }
                if (bits) atomicOr(&piece_seen, bits);
            }
        }
    }
    __syncthreads();
    const uint32_t seen_bits = piece_seen;
    __shared__ float stage[9][256];
    __shared__ __attribute__((aligned(16))) uint8_t stage_count[256];
#pragma unroll 1
    for (uint32_t piece = 0; piece < np; ++piece) {
    const uint32_t t_kz0 = ((chunk + piece) & ((1u << zq_bits) - 1u)) << 6;
    const uint32_t idx = tiled ? (((t_ix0 + t_row) << (2 * lc.log2d)) | (t_jy << lc.log2d) | (t_kz0 + t_z)) : chunk * 256u + threadIdx.x;
    const int kz = (int)(idx & dm);                                             // (ix, jy: through row_pre)
    if (!((seen_bits >> piece) & 1u)) {
        // No view reaches any voxel of the tile -- more than half of the tiles at the benchmark geometry (five 31 x 23 degree frusta
        // cover a third of the cube): mean = var = mask = count = +0, exactly what the general path computes for cnt = 0
        // (div_rn(0, 1e-8, y) = +0, 0 - 0 * 0 = +0), stored straight from registers: no view loop, no divisions, no staging,
        // no second barrier.  185.8 -> 178.2 us at 256^3 (42.3 -> 44.1 % of the HBM peak by algorithmic bytes, back-to-back launches;
        // scripts/probe/k1_fwd_ab.py).  The same test per WAVE inside the other tiles (skip the divisions of 64 unseen voxels) measured
        // nothing (179.4): their waves wait at the staging barrier for the slowest anyway.
        const uint32_t plane = (uint32_t)d << (2 * lc.log2d + 2);
        const __amdgpu_buffer_rsrc_t planes = __builtin_amdgcn_make_buffer_rsrc((void*)vol, 0, (int)(8u * plane), 0x00020000);
        const __amdgpu_buffer_rsrc_t mplane = __builtin_amdgcn_make_buffer_rsrc((void*)mask, 0, (int)plane, 0x00020000);
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 zero = {0u, 0u, 0u, 0u};
        const uint32_t q = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const uint32_t off = tiled ? ((((t_ix0 + (q >> 4)) << (2 * lc.log2d)) | (t_jy << lc.log2d) | (t_kz0 + ((q & 15u) << 2))) << 2) : chunk * 1024u + q * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(zero, planes, off, wave * plane, 2);
        __builtin_amdgcn_raw_buffer_store_b128(zero, planes, off, (wave + 4u) * plane, 2);
        if (wave == 0) __builtin_amdgcn_raw_buffer_store_b128(zero, mplane, off, 0u, 2);
        if (count && wave == 1 && q < 16u) {
            const uint32_t at = tiled ? (((t_ix0 + (q >> 2)) << (2 * lc.log2d)) | (t_jy << lc.log2d) | (t_kz0 + ((q & 3u) << 4))) : chunk * 256u + q * 16u;
            *(u4*)(count + at) = zero;
        }
        if (bits && wave == 2 && q < 8u)                                           // (the mask as bits, see below: this tile's eight words)
            bits[tiled ? (((((t_ix0 + (q >> 1)) << (2 * lc.log2d)) | (t_jy << lc.log2d) | t_kz0) >> 5) + (q & 1u)) : chunk * 8u + q] = 0u;
        continue;
    }
    const int my_row = tiled ? (int)t_row : (int)(threadIdx.x >> lc.log2d);
    // torch.linspace(-1, 1, d)[i]: lower half counts up from the start, upper half down from the end
    // (x and y of the voxel enter through row_pre: they are the same along the z-row)
    const float z = kz < half ? -1.0f + lc.step * (float)kz : 1.0f - lc.step * (float)(d - 1 - kz);
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const uint32_t row_bytes = (uint32_t)w * 16u, view_bytes = (uint32_t)h * row_bytes;
    const __amdgpu_buffer_rsrc_t texels = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)((uint32_t)nv * view_bytes), 0x00020000);
    float4 s1 = f4_zero(), s2 = f4_zero();
    float cnt = 0.0f;
    for (int v = 0; v < nv; ++v) {
        const int2 span = row_span[my_row][v];
        if (kz < span.x || kz > span.y) continue;                                 // outside the view's frustum for sure
        // The camera matrices are read through the CONSTANT address space: wave-uniform addresses of memory nothing in this launch writes, i.e.
        // scalar loads whatever the compiler can prove about aliasing.  As plain global pointers they were scalar loads in the one-level kernel
        // (__restrict__ kernel arguments) but per-lane global_load_dword in the all-levels kernel (pointers out of the level table: every store
        // might alias them) -- a vector-memory round trip per view and wave, 60 us of that launch's 280 (profiles/r05_k1_levels_ab.txt).
        const k1_cfloat_p m = (k1_cfloat_p)(uintptr_t)(w2c + 16 * v);
        const k1_cfloat_p k = (k1_cfloat_p)(uintptr_t)(intr + 16 * v);
        const float4 pre = row_pre[my_row][v];                                    // (m[4 q] x + m[4 q + 1] y)_q of this lane's z-row
        const float cx = pre.x + m[2] * z + m[3];
        const float cy = pre.y + m[6] * z + m[7];
        const float cz = pre.z + m[10] * z + m[11];
        // integer tests on the scalar unit (a float compare would be a VALU instruction + vcc branch per matrix entry)
        const k1_cuint_p mb = (k1_cuint_p)m;
        const k1_cuint_p kb = (k1_cuint_p)k;
        const uint32_t must_be_zero = (mb[12] | mb[13] | mb[14] | kb[1] | kb[3] | kb[4] | kb[7] | kb[8] | kb[9] | kb[11]) << 1;   // +-0
        const uint32_t must_be_one = (mb[15] ^ 0x3f800000u) | (kb[10] ^ 0x3f800000u);
        float u, vv, dd;
        if ((must_be_zero | must_be_one) == 0u) {
            u = k[0] * cx + k[2] * cz;
            vv = k[5] * cy + k[6] * cz;
            dd = cz;
        } else {
            const float cw = pre.w + m[14] * z + m[15];
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
    if (bits) {
        // The mask once more as BITS (bit i & 31 of word i >> 5 = mask[i] > 0), the form the ray-point and nearest look-up kernels read: a training
        // step packed them from the float planes in a launch per level (gens_pack_mask_bits: 75 MB read again, 50 us per step).  A wave's ballot
        // holds 16 z of four rows (tiled: half a word per row) or 64 consecutive voxels (two words).
        const unsigned long long b = __ballot(cnt > (float)min_vis);
        if (tiled) {
            if (lane < 4u)
                ((uint16_t*)bits)[((((t_ix0 + lane) << (2 * lc.log2d)) | (t_jy << lc.log2d) | (t_kz0 + (wv << 4))) >> 4)] = (uint16_t)(b >> (16u * lane));
        } else if ((lane & 31u) == 0u) {
            bits[((chunk * 256u + (wv << 6)) >> 5) + (lane >> 5)] = (uint32_t)(b >> lane);
        }
    }
    const float den = cnt <= 0.0f ? 1e-8f : cnt;                                  // (Q5)
    const float yn = rcp_rn(den);
    const float4 mm = make_float4(div_rn(s1.x, den, yn), div_rn(s1.y, den, yn), div_rn(s1.z, den, yn), div_rn(s1.w, den, yn));
    // Output: the workgroup's 256 voxels are 256 consecutive floats in each of the 9 planes.  Staged through LDS so that a store
    // instruction carries 16 B per lane (one plane's 1 KiB per wave-store: 9 store instructions per workgroup instead of 36 -- the
    // texture path of a CU was busy 72 % of the kernel with 4-byte stores); streaming (non-temporal), so that the 36 B / voxel do not
    // sweep the texels out of L2; through buffer descriptors (lane offset in 32 bits; d^3 <= 2^24 voxels: 8 planes are 512 MiB).
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
    stage_count[slot] = (uint8_t)cnt;
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
    if (count && wave == 1 && q < 16u) {      // the visible-view counts (a byte per voxel, for the backward pass): sixteen lanes, sixteen voxels each
        const uint32_t at = tiled ? (((t_ix0 + (q >> 2)) << (2 * lc.log2d)) | (t_jy << lc.log2d) | (t_kz0 + ((q & 3u) << 4))) : chunk * 256u + q * 16u;
        *(u4*)(count + at) = *(const u4*)&stage_count[16u * q];
    }
    if (piece + 1u < np) __syncthreads();                                         // (the staging rows are written again by the next piece with work)
    }
}


// All levels of a scene in ONE launch: the small levels (a few hundred workgroups, latency-bound on their own: 28 + 10 us for
// 128^3 + 64^3) run in the shadow of the large one.  Blocks [first[l], first[l + 1]) belong to level l.
struct VolumeLevels {
    const float4* feat[GENS_MAX_LEVELS];
    const float* intr[GENS_MAX_LEVELS];          // (nv, 4, 4) per level, rows 0-1 pre-scaled by 0.5^level
    float* vol[GENS_MAX_LEVELS];
    float* mask[GENS_MAX_LEVELS];
    uint8_t* count[GENS_MAX_LEVELS];             // visible views per voxel (may be null)
    uint32_t* bits[GENS_MAX_LEVELS];             // the mask as bits (may be null)
    int h[GENS_MAX_LEVELS], w[GENS_MAX_LEVELS], d[GENS_MAX_LEVELS];
    LevelConst lc[GENS_MAX_LEVELS];
    uint32_t first[GENS_MAX_LEVELS + 1];
    int n;
};

// Texel warm-up.  Inside a step this launch follows a render (or a backward pass) that left L2 and the Infinity Cache full of volume data: every
// first touch of a texel line then goes to HBM, ~2 us each, paid wave by wave all through the launch -- 255 - 275 us where the same launch takes 217 - 222
// with the 33 MB of texels resident (scripts/probe/k1_instep_probe.py: a 1 GiB fill before the launch costs + 55 us, one pass over the texels
// between the fill and the launch takes it back).  So the FIRST workgroups of the launch ask for every 128-byte line of the tables -- one dword per
// line, two lines per thread, ~500 workgroups of the ~2 000 that start together -- before they turn to their own voxels; the loads are waited for
// only at the very end of those workgroups (their values feed a store that never happens), HBM streams the tables in ~6 us under everybody's
// first projections.  Nothing is computed differently.
#define K1_WARM_LINES_PER_BLOCK 512
__device__ __forceinline__ float k1_warm_texels(const VolumeLevels& lv, int nv) {
    float acc = 0.0f;
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < K1_WARM_LINES_PER_BLOCK / 256; ++k) {
        uint32_t line = blockIdx.x * K1_WARM_LINES_PER_BLOCK + (uint32_t)k * 256u + threadIdx.x;      // over the tables of all levels, one after the other
        base = 0;
        for (int l = 0; l < lv.n; ++l) {
            const uint32_t lines = ((uint32_t)nv * (uint32_t)lv.h[l] * (uint32_t)lv.w[l] * 16u + 127u) >> 7;
            if (line >= base && line < base + lines) {
                const uint32_t bytes = (uint32_t)nv * (uint32_t)lv.h[l] * (uint32_t)lv.w[l] * 16u, at = min((line - base) << 7, bytes - 4u);
                acc += *(const float*)((const char*)lv.feat[l] + at);
            }
            base += lines;
        }
    }
    return acc;
}

// ... or as a launch of its own in front (GENS_K1_WARM=2): one thread per line
__global__ __launch_bounds__(256) void volume_warm_texels_k(VolumeLevels lv, int nv) {
    uint32_t line = blockIdx.x * 256u + threadIdx.x, base = 0;
    float acc = 0.0f;
    for (int l = 0; l < lv.n; ++l) {
        const uint32_t bytes = (uint32_t)nv * (uint32_t)lv.h[l] * (uint32_t)lv.w[l] * 16u, lines = (bytes + 127u) >> 7;
        if (line >= base && line < base + lines) acc += *(const float*)((const char*)lv.feat[l] + min((line - base) << 7, bytes - 4u));
        base += lines;
    }
    if (acc == 1.2345678e-31f) lv.mask[0][0] = acc;                                 // (never)
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void volume_build_fwd_levels_k(VolumeLevels lv, const float* __restrict__ w2c, int nv, int min_vis, uint32_t warm_blocks,
                                                                 uint32_t groups) {
    float warm = 0.0f;
    if (blockIdx.x < warm_blocks) warm = k1_warm_texels(lv, nv);
    int l = 0;
    while (l + 1 < lv.n && blockIdx.x >= lv.first[l + 1]) ++l;                     // scalar: blockIdx and the table are uniform
    // Chunk order (A/B switch GENS_K1_INTERLEAVE=<parts>, off by default).  In index order the launch walks the volume slab by slab, and the tiles no view
    // reaches (stores only: two thirds of the volume at the benchmark geometry) and the tiles with work (instruction-bound) come in long runs of one
    // kind; with the switch consecutive workgroups take their chunks from <parts> different parts of the volume (chunk i of part g runs as workgroup
    // i * parts + g).  Isolated and cold, 128 parts at 256^3 (two per x-slab) measured 230 -> 221 us and -2.5 .. -6 % on four of five camera set-ups
    // (16, 32, 256 parts: 260 - 277 us: not monotonic) -- but inside the bench step nothing: 223.4 against 221.6 us (profiles/r06_k1_interleave_ab.txt).
    uint32_t chunk = blockIdx.x - lv.first[l];
    const uint32_t n_chunks = lv.first[l + 1] - lv.first[l];
    if (groups > 1u && n_chunks % groups == 0u) chunk = (chunk % groups) * (n_chunks / groups) + chunk / groups;
    volume_build_chunk(chunk, lv.feat[l], w2c, lv.intr[l], nv, lv.h[l], lv.w[l], lv.d[l], lv.lc[l], min_vis, lv.vol[l], lv.mask[l], lv.count[l], lv.bits[l]);
    if (warm == 1.2345678e-31f) lv.mask[0][0] = warm;                              // (never: the warm-up loads have to be loads of something)
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void volume_build_fwd_lean_k(const float4* __restrict__ feat, const float* __restrict__ w2c,
                                                               const float* __restrict__ intr, int nv, int h, int w, int d, LevelConst lc,
                                                               int min_vis, float* __restrict__ vol, float* __restrict__ mask,
                                                               uint8_t* __restrict__ count) {
    volume_build_chunk(blockIdx.x, feat, w2c, intr, nv, h, w, d, lc, min_vis, vol, mask, count);
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
    lc.no_shortcut = getenv("GENS_K1_NO_EMPTY_SHORTCUT") != nullptr;
    const int want = getenv("GENS_K1_PIECES") ? atoi(getenv("GENS_K1_PIECES")) : 4;
    lc.pieces = d >= 256 && want >= 4 ? 4 : d >= 128 && want >= 2 ? 2 : 1;
    return lc;
}

static int volume_build_fwd_level(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv, int h, int w, int d,
                                  int min_vis_view, float* volume, float* mask, uint8_t* count, void* stream) {
    if (int e = check_volume_args("gens_volume_build_fwd", feat, w2c, intr, nv, h, w, d)) return e;
    GENS_CHECK_ARG(volume && mask, GENS_EINVAL, "gens_volume_build_fwd: null output");
    int64_t n = (int64_t)d * d * d;
    const bool pow2 = d >= 2 && d <= 256 && (d & (d - 1)) == 0;
    if (pow2 && !getenv("GENS_K1_GENERIC")) {                 // (the environment switch keeps the generic kernel reachable for A/B tests)
        const LevelConst lc = level_const(h, w, d);
        if (d >= 8 && nv <= K1_FAST_VIEWS && intr_scale == 1.0f && !getenv("GENS_K1_SINGLE")) {   // production path (the switch keeps the previous kernel reachable for A/B runs)
            volume_build_fwd_lean_k<<<(unsigned)(n / 256 / lc.pieces), 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, nv, h, w, d, lc,
                                                                                     min_vis_view, volume, mask, count);
            return gens_launch_status("gens_volume_build_fwd");
        }
        const unsigned rows_per_block = 256u >> lc.log2d;
        const unsigned grid = ((unsigned)d * (unsigned)d + rows_per_block - 1) / rows_per_block;
        if (intr_scale == 1.0f)
            volume_build_fwd_pow2_k<true><<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv, h, w, d, lc,
                                                                               min_vis_view, volume, mask, count);
        else
            volume_build_fwd_pow2_k<false><<<grid, 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv, h, w, d, lc,
                                                                                min_vis_view, volume, mask, count);
        return gens_launch_status("gens_volume_build_fwd");
    }
    volume_build_fwd_k<<<gens_blocks(n, 256), 256, 0, (hipStream_t)stream>>>((const float4*)feat, w2c, intr, intr_scale, nv,
                                                                            h, w, d, min_vis_view, volume, mask, count);
    return gens_launch_status("gens_volume_build_fwd");
}

extern "C" int gens_volume_build_fwd(const float* feat, const float* w2c, const float* intr, float intr_scale, int nv,
                                     int h, int w, int d, int min_vis_view, float* volume, float* mask, void* stream) {
    return volume_build_fwd_level(feat, w2c, intr, intr_scale, nv, h, w, d, min_vis_view, volume, mask, nullptr, stream);
}

extern "C" int gens_pack_mask_bits(const float* mask, int64_t n, uint32_t* bits, void* stream);

extern "C" int gens_volume_build_levels_bits(const float* const* feat, const int* hw, const int* dims, int n_levels, const float* w2c,
                                             const float* const* intr, int nv, int min_vis_view, float* const* volumes, float* const* masks,
                                             uint8_t* const* counts, uint32_t* const* mask_bits, void* stream) {
    GENS_CHECK_ARG(feat && hw && dims && w2c && intr && volumes && masks, GENS_EINVAL, "gens_volume_build_levels: null table");
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= GENS_MAX_LEVELS, GENS_ELIMIT, "gens_volume_build_levels: %d levels (1..%d)", n_levels, GENS_MAX_LEVELS);
    bool one_launch = nv <= K1_FAST_VIEWS && !getenv("GENS_K1_GENERIC") && !getenv("GENS_K1_SINGLE") && !getenv("GENS_K1_PER_LEVEL");
    for (int l = 0; l < n_levels; ++l) {
        if (int e = check_volume_args("gens_volume_build_levels", feat[l], w2c, intr[l], nv, hw[2 * l], hw[2 * l + 1], dims[l])) return e;
        GENS_CHECK_ARG(volumes[l] && masks[l], GENS_EINVAL, "gens_volume_build_levels: null output (level %d)", l);
        GENS_CHECK_ARG(!counts || !counts[l] || ((uintptr_t)counts[l] & 15) == 0, GENS_EINVAL, "gens_volume_build_levels: the count plane of level %d must be 16-byte aligned", l);
        const int d = dims[l];
        one_launch = one_launch && d >= 8 && d <= 256 && (d & (d - 1)) == 0;
    }
    if (!one_launch) {                                                             // sizes the fused kernel does not cover: level by level
        for (int l = 0; l < n_levels; ++l)
            if (int e = volume_build_fwd_level(feat[l], w2c, intr[l], 1.0f, nv, hw[2 * l], hw[2 * l + 1], dims[l], min_vis_view, volumes[l], masks[l],
                                               counts ? counts[l] : nullptr, stream))
                return e;
        for (int l = 0; mask_bits && l < n_levels; ++l)
            if (mask_bits[l])
                if (int e = gens_pack_mask_bits(masks[l], (int64_t)dims[l] * dims[l] * dims[l], mask_bits[l], stream)) return e;
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
        lv.count[l] = counts ? counts[l] : nullptr;
        lv.bits[l] = mask_bits ? mask_bits[l] : nullptr;
        lv.h[l] = hw[2 * l];
        lv.w[l] = hw[2 * l + 1];
        lv.d[l] = d;
        lv.lc[l] = level_const(hw[2 * l], hw[2 * l + 1], d);
        lv.first[l + 1] = lv.first[l] + (uint32_t)(((int64_t)d * d * d) / 256 / lv.lc[l].pieces);
    }
    // the texel warm-up (k1_warm_texels): enough leading workgroups to ask for every 128-byte line of the tables once; GENS_K1_WARM=0 switches it off
    // (A/B runs: scripts/probe/k1_instep_probe.py)
    int64_t lines = 0;
    for (int l = 0; l < n_levels; ++l) lines += ((int64_t)nv * hw[2 * l] * hw[2 * l + 1] * 16 + 127) >> 7;
    uint32_t warm_blocks = (uint32_t)std::min<int64_t>((lines + K1_WARM_LINES_PER_BLOCK - 1) / K1_WARM_LINES_PER_BLOCK, lv.first[n_levels]);
    // Measured (scripts/probe/k1_instep_probe.py, profiles/r06_k1_warm_ab.txt): cold 266 us; warm-up inside the launch 244; as a launch of its own 229 --
    // it costs 6 us when the texels are resident already (214 -> 221)
    const int warm_mode = getenv("GENS_K1_WARM") ? atoi(getenv("GENS_K1_WARM")) : 2;
    if (lines >= (1ll << 31) || warm_mode != 1) warm_blocks = 0;
    if (warm_mode == 2 && lines < (1ll << 31)) volume_warm_texels_k<<<(unsigned)((lines + 255) / 256), 256, 0, (hipStream_t)stream>>>(lv, nv);
    const uint32_t groups = getenv("GENS_K1_INTERLEAVE") ? (uint32_t)atoi(getenv("GENS_K1_INTERLEAVE")) : 0u;
    volume_build_fwd_levels_k<<<lv.first[n_levels], 256, 0, (hipStream_t)stream>>>(lv, w2c, nv, min_vis_view, warm_blocks, groups);
    return gens_launch_status("gens_volume_build_levels");
}

extern "C" int gens_volume_build_levels(const float* const* feat, const int* hw, const int* dims, int n_levels, const float* w2c,
                                        const float* const* intr, int nv, int min_vis_view, float* const* volumes, float* const* masks,
                                        uint8_t* const* counts, void* stream) {
    return gens_volume_build_levels_bits(feat, hw, dims, n_levels, w2c, intr, nv, min_vis_view, volumes, masks, counts, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// K1 backward on the image-tile plan: the IMAGE owns the sum, ALL LEVELS in one launch set, nothing recomputed that the forward pass knows.
//
// The wave windows above reduce what one 4 x 16 voxel tile sends to a view by ~4 x (a z run slides 0 .. 1.4 pixels per voxel), and what is
// left -- ~0.3 G global float atomics at 256^3 -- is that kernel's whole time (~66 G atomics/s on this chip, whatever the tile shape:
// scripts/probe/README.md).  But a view's gradient image has only nv x H x W x 4 = 6 M floats for 1.3 G taps: nearly all of the reduction
// is between voxels far apart in the volume that lie along the same viewing rays.  So the sum is turned around: the (wave tile, view) pairs
// are sorted by the image tile (BL_W x BL_H texels, by the north-west tap) their voxels fall into, and a workgroup per image tile keeps that
// tile's gradient (+ one texel of halo for the south / east taps) in LDS, walks its pairs, and sends each touched texel to memory once.
// Every (voxel, view) pair is owned by exactly one image tile, so the sums are those of the direct scatter in another order.
//   plan    a thread per (wave tile, view) pair: its range of image tiles from the FOUR CORNERS of the wave tile (a 4 x 16 rectangle of the
//           lattice in the x-z plane: its image under a pinhole is a convex quadrilateral when all four depths are positive; the few pairs
//           with a corner at or behind the camera walk their 64 voxels): <= 2 x 2 tiles, else the level's `direct` bin; counted per bin
//   scan    offsets of all levels' bins in one list, work items of <= seg pairs
//   fill    the pairs written to their bins
//   tiles   a workgroup (16 waves) per work item.  Per voxel: 8 cotangent floats, 4 means (the forward's own output: the volume's first four
//           planes), the count byte gens_volume_build_levels leaves -- planar, coalesced along z, through buffer descriptors --, the projection
//           into the item's view with the forward kernel's arithmetic (same visibility and taps, bit for bit), the four texels, and sixteen
//           adds into the window; direct items scatter with global atomics.
// Five launches for any number of levels (memset, plan, scan, fill, tiles); scratch: 20 B per pair.
//
// History: the second generation re-derived count and mean per voxel in a `prep` pass (0.53 of its 1.27 ms at 256^3: every voxel
// projected into every view with the generic arithmetic, 48 B of records per voxel written and read back once per visible view, four wave
// reductions per pair for exact tile ranges) and summed in 64-bit FIXED POINT, which needed a bound of |g_view| over the whole call.
// The window sums are DOUBLES now: ds_add_f64 costs what ds_add_u64 does on this chip (18 cycles per wave instruction on distinct words,
// scripts/probe/lds_atomic_probe.py; ds_add_f32: 190 - 250), a float32 tap converts exactly, a sum of < 2^29 of them is exact to 2^-24 of one
// float32 rounding -- no bound, no pass over the cotangent planes to find one, and NaN / infinity propagate like in the direct scatter.
#define BL_W 64
#define BL_H 60          // (65 x 61 texels x 4 channels x 8 bytes = 124 KB of LDS: one workgroup of 16 waves per CU; 480 / 240 / 120 rows are whole tiles)
#define BL_SEG 512          // voxel tiles per work item (32 trips per wave; the window's zeroing and flush are a few per cent of that)
#define BL_THREADS 1024
#define BL_SEG_DIRECT 16     // pairs per work item of a direct bin
#define BL_WIN ((BL_W + 1) * (BL_H + 1))
#define BL_LDS_BYTES (4 * BL_WIN * 8)

static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

// Voxel tile q = (ix / 4, jy, z piece), z piece fastest, as (z piece | jy << 8 | ix / 4 << 20): 16 consecutive z of 4 x-adjacent rows, lane = 16 row + z.
// (Tiles of 128 voxels -- two z-adjacent voxels per lane, whole 128-byte lines per row and plane -- were measured slower: 0.99 against 0.81 ms at
// 256^3.  Their longer z runs span three image tiles more often, and two voxels per lane leave no registers to keep the next tile's texels in flight.)
__device__ __forceinline__ uint32_t bwd_tile_code(uint32_t q, int d) {
    const uint32_t zp = (uint32_t)d >> 4, r = q / zp;
    return (q - r * zp) | ((r % (uint32_t)d) << 8) | ((r / (uint32_t)d) << 20);
}
struct TileVoxel {
    int ix, jy, kz;
    uint32_t vox;                   // (d^3 < 2^30 is checked by the host: byte offsets into a plane fit 32 bits)
};
__device__ __forceinline__ TileVoxel bwd_tile_voxel(uint32_t code, int lane, int d) {
    TileVoxel o;
    o.kz = (int)(code & 0xFFu) * 16 + (lane & 15);
    o.jy = (int)((code >> 8) & 0xFFFu);
    o.ix = (int)(code >> 20) * 4 + (lane >> 4);
    o.vox = ((uint32_t)o.ix * (uint32_t)d + (uint32_t)o.jy) * (uint32_t)d + (uint32_t)o.kz;
    return o;
}

struct BwdLevel {
    const float4* feat;
    const float* intr;
    const float* vol;               // forward output: planes 0-3 = mean
    const uint8_t* count;           // forward output: visible views per voxel
    const float* gvol;
    float* gfeat;
    uint32_t* codes;                // per (view, wave tile)
    int h, w, d, tiles_x, tiles_y, seg;
    uint32_t bin0, n_tile_bins;     // the level's bins are [bin0, bin0 + n_tile_bins] -- the last one is its direct bin
    uint32_t n_waves;               // voxel tiles (64 voxels: one per lane)
    uint32_t plan_b;                // first block of the level's pairs in plan_k and fill_k
    LevelConst lc;
};
struct BwdLevels {
    BwdLevel lv[GENS_MAX_LEVELS];
    int n, nv;
    uint32_t *count, *cursor, *offset, *n_items, *next_item;
    uint4* items;                   // (bin, begin, end, level)
    uint32_t* list;
    uint32_t max_items, n_bins, plan_blocks;
};

struct LeanProj {
    float fx, fy;
    bool vis;
};
// volume_build_chunk's projection (same operations in the same order: the visibility and the taps of the backward pass are the forward's)
__device__ __forceinline__ LeanProj project_lean(k1b_float_p m, k1b_float_p k, bool pinhole, const LevelConst& lc,
                                                 float wm1, float hm1, float x, float y, float z) {
    const float cx = m[0] * x + m[1] * y + m[2] * z + m[3];
    const float cy = m[4] * x + m[5] * y + m[6] * z + m[7];
    const float cz = m[8] * x + m[9] * y + m[10] * z + m[11];
    float u, vv, dd;
    if (pinhole) {
        u = k[0] * cx + k[2] * cz;
        vv = k[5] * cy + k[6] * cz;
        dd = cz;
    } else {
        const float cw = m[12] * x + m[13] * y + m[14] * z + m[15];
        u = k[0] * cx + k[1] * cy + k[2] * cz + k[3] * cw;
        vv = k[4] * cx + k[5] * cy + k[6] * cz + k[7] * cw;
        dd = k[8] * cx + k[9] * cy + k[10] * cz + k[11] * cw;
    }
    const float dn = dd + 1e-8f;                                                  // (Q3)
    const float yd = rcp_rn(dn);
    const float px = div_rn(u, dn, yd), py = div_rn(vv, dn, yd);
    const float nx = div_rn(px, lc.cw, lc.rcw) - 1.0f, ny = div_rn(py, lc.ch, lc.rch) - 1.0f;
    LeanProj o;
    o.vis = (fmaxf(fabsf(nx), fabsf(ny)) <= 1.0f) && (dd > 0.0f);
    o.fx = (nx + 1.0f) / 2.0f * wm1;
    o.fy = (ny + 1.0f) / 2.0f * hm1;
    return o;
}
__device__ __forceinline__ bool is_pinhole(k1b_float_p m, k1b_float_p k) {
    const k1b_uint_p mb = (k1b_uint_p)m;
    const k1b_uint_p kb = (k1b_uint_p)k;
    const uint32_t must_be_zero = (mb[12] | mb[13] | mb[14] | kb[1] | kb[3] | kb[4] | kb[7] | kb[8] | kb[9] | kb[11]) << 1;   // +-0
    const uint32_t must_be_one = (mb[15] ^ 0x3f800000u) | (kb[10] ^ 0x3f800000u);
    return (must_be_zero | must_be_one) == 0u;
}
__device__ __forceinline__ float lattice_at(const LevelConst& lc, int d, int i) {      // torch.linspace(-1, 1, d)[i]
    return i < (d >> 1) ? -1.0f + lc.step * (float)i : 1.0f - lc.step * (float)(d - 1 - i);
}

// Range of image tiles of one (wave tile, view) pair: 0 = no voxel of the wave tile is visible in the view for sure; bit 31 = some may be,
// bit 30 = more than 2 x 2 tiles (the pair goes to the direct bin), else tx_lo | ty_lo << 12 | (nx - 1) << 24 | (ny - 1) << 25.
__device__ uint32_t bwd_pair_code(const BwdLevel& L, k1b_float_p m, k1b_float_p k, uint32_t wave_code) {
    const int d = L.d, w = L.w, h = L.h;
    constexpr int zlen = 16, rows = 4;
    const int kz0 = (int)(wave_code & 0xFFu) * zlen, jy = (int)((wave_code >> 8) & 0xFFFu), ix0 = (int)(wave_code >> 20) * rows;
    const float y = lattice_at(L.lc, d, jy);
    float x_lo = 1e30f, x_hi = -1e30f, y_lo = 1e30f, y_hi = -1e30f;
    bool sure = true;
    for (int c = 0; c < 4; ++c) {
        const float x = lattice_at(L.lc, d, ix0 + ((c & 1) ? rows - 1 : 0)), z = lattice_at(L.lc, d, kz0 + ((c & 2) ? zlen - 1 : 0));
        float4 cam;                                                                // (= mat4_point, on the constant-address-space matrix)
        cam.x = m[0] * x + m[1] * y + m[2] * z + m[3];
        cam.y = m[4] * x + m[5] * y + m[6] * z + m[7];
        cam.z = m[8] * x + m[9] * y + m[10] * z + m[11];
        cam.w = m[12] * x + m[13] * y + m[14] * z + m[15];
        const float u = k[0] * cam.x + k[1] * cam.y + k[2] * cam.z + k[3] * cam.w;
        const float vv = k[4] * cam.x + k[5] * cam.y + k[6] * cam.z + k[7] * cam.w;
        const float dd = k[8] * cam.x + k[9] * cam.y + k[10] * cam.z + k[11] * cam.w;
        const float mag = fabsf(k[8] * cam.x) + fabsf(k[9] * cam.y) + fabsf(k[10] * cam.z) + fabsf(k[11] * cam.w);
        sure = sure && dd > 1e-3f * mag;
        const float px = u / dd, py = vv / dd;
        x_lo = fminf(x_lo, px);
        x_hi = fmaxf(x_hi, px);
        y_lo = fminf(y_lo, py);
        y_hi = fmaxf(y_hi, py);
    }
    int x0_lo, x0_hi, y0_lo, y0_hi;
    const float big = 1.0e8f;
    if (sure && fabsf(x_lo) < big && fabsf(x_hi) < big && fabsf(y_lo) < big && fabsf(y_hi) < big) {
        // every voxel of the wave tile projects into the corners' bounding box (+- the float32 rounding of the per-voxel arithmetic, < 0.01 texel)
        const float margin = 0.05f;
        x0_lo = max((int)floorf(x_lo - margin), 0);
        x0_hi = min((int)floorf(x_hi + margin), w - 1);
        y0_lo = max((int)floorf(y_lo - margin), 0);
        y0_hi = min((int)floorf(y_hi + margin), h - 1);
    } else {                                                                       // a corner at or behind the camera: the 64 voxels one by one
        const bool pinhole = is_pinhole(m, k);
        x0_lo = y0_lo = 0x7fffffff;
        x0_hi = y0_hi = -1;
        for (int i = 0; i < 64; ++i) {
            const LeanProj p = project_lean(m, k, pinhole, L.lc, (float)(w - 1), (float)(h - 1), lattice_at(L.lc, d, ix0 + i / zlen), y, lattice_at(L.lc, d, kz0 + i % zlen));
            if (!p.vis) continue;
            const int x0 = (int)floorf(p.fx), y0 = (int)floorf(p.fy);
            x0_lo = min(x0_lo, x0);
            x0_hi = max(x0_hi, x0);
            y0_lo = min(y0_lo, y0);
            y0_hi = max(y0_hi, y0);
        }
    }
    if (x0_lo > x0_hi || y0_lo > y0_hi) return 0u;
    const int tx_lo = x0_lo / BL_W, tx_hi = x0_hi / BL_W, ty_lo = y0_lo / BL_H, ty_hi = y0_hi / BL_H;
    if (tx_hi - tx_lo > 1 || ty_hi - ty_lo > 1) return 0xC0000000u;
    return 0x80000000u | (uint32_t)tx_lo | ((uint32_t)ty_lo << 12) | ((uint32_t)(tx_hi - tx_lo) << 24) | ((uint32_t)(ty_hi - ty_lo) << 25);
}

// The pairs of one block of threads -- 256 consecutive wave tiles in ONE view -- counted per bin (FILL = false) or written to their bins' lists
// (FILL = true).  Neighbouring wave tiles mostly share their bins, so the block first counts in LDS (a histogram over the view's tiles + the
// direct bin; ds_add_rtn gives every pair its rank) and then sends ONE global atomic per bin it touched, all of them independent (the first
// version elected a leader per wave and bin: four to eight dependent round trips to L2 per wave, 0.13 ms per pass at 256^3).
#define BWD_HIST 2048           // tiles of a view + 1 the histogram holds (larger images: one global atomic per pair and bin)
template <bool FILL>
__device__ __forceinline__ void bwd_bin_pairs(const BwdLevels& a, const BwdLevel& L, uint32_t code, int v, uint32_t entry, uint32_t pair) {
    __shared__ uint32_t hist[BWD_HIST], base[FILL ? BWD_HIST : 1];
    const int tid = threadIdx.x;
    const int per_view = L.tiles_x * L.tiles_y, nb = per_view + 1;
    const int tx_lo = code & 0xFFF, ty_lo = (code >> 12) & 0xFFF, nx = (code >> 24) & 1, ny = (code >> 25) & 1;
    const bool direct = (code >> 30) & 1u;
    int bin[4];                                                                    // slot in the histogram: tile of the view, or per_view = the direct bin; -1 = none
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                                   // the up to 2 x 2 tiles of the pair
        const int dx = k & 1, dy = k >> 1;
        bin[k] = -1;
        if (code && direct && k == 0) bin[k] = per_view;
        if (code && !direct && dx <= nx && dy <= ny) bin[k] = (ty_lo + dy) * L.tiles_x + tx_lo + dx;
    }
    uint32_t* global = FILL ? a.cursor : a.count;
    auto slot_bin = [&](int slot) { return slot == per_view ? L.bin0 + L.n_tile_bins : L.bin0 + (uint32_t)(v * per_view + slot); };
    if (nb > BWD_HIST) {                                                            // (very large images)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (bin[k] < 0) continue;
            const uint32_t b = slot_bin(bin[k]), at = atomicAdd(global + b, 1u);
            if (FILL) a.list[a.offset[b] + at] = direct ? pair : entry;
        }
        return;
    }
    for (int i = tid; i < nb; i += 256) hist[i] = 0u;
    __syncthreads();
    uint32_t rank[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (bin[k] >= 0) rank[k] = atomicAdd(&hist[bin[k]], 1u);
    __syncthreads();
    for (int i = tid; i < nb; i += 256) {
        const uint32_t c = hist[i];
        if (!c) continue;
        if (FILL) base[i] = a.offset[slot_bin(i)] + atomicAdd(global + slot_bin(i), c);
        else atomicAdd(global + slot_bin(i), c);
    }
    if (!FILL) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (bin[k] >= 0) a.list[base[bin[k]] + rank[k]] = direct ? pair : entry;
}

__global__ __launch_bounds__(256) void volume_bwd_zero_k(uint32_t* __restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = 0u;
}

__global__ __launch_bounds__(256) void volume_bwd_plan_k(BwdLevels a, const float* __restrict__ w2c) {
    int l = 0;
    while (l + 1 < a.n && blockIdx.x >= a.lv[l + 1].plan_b) ++l;                   // (scalar)
    const BwdLevel& L = a.lv[l];
    // a block per 256 wave tiles of one view
    const uint32_t per_view = (L.n_waves + 255u) / 256u, b = blockIdx.x - L.plan_b;
    const int v = (int)(b / per_view);
    const uint32_t q = (b - (uint32_t)v * per_view) * 256u + (uint32_t)threadIdx.x;
    uint32_t code = 0;
    if (q < L.n_waves) {
        code = bwd_pair_code(L, k1_const(w2c + 16 * v), k1_const(L.intr + 16 * v), bwd_tile_code(q, L.d));
        L.codes[(uint32_t)v * L.n_waves + q] = code;
    }
    bwd_bin_pairs<false>(a, L, code, v, 0u, 0u);
}

// offsets of all levels' bins in the list and the work items (one workgroup; the bins number a few hundred to a few thousand)
__global__ __launch_bounds__(256) void volume_bwd_scan_k(BwdLevels a) {
    __shared__ uint32_t part_c[256], part_i[256];
    const int tid = threadIdx.x, per = ((int)a.n_bins + 255) / 256;
    const int lo = min(tid * per, (int)a.n_bins), hi = min(lo + per, (int)a.n_bins);
    auto level_of = [&](int b) {
        int l = 0;
        while (l + 1 < a.n && (uint32_t)b >= a.lv[l + 1].bin0) ++l;
        return l;
    };
    auto seg_of = [&](int b, int l) {       // pairs per work item: the direct bin (global atomics, no window) in pieces of a pair per wave
        return (uint32_t)b == a.lv[l].bin0 + a.lv[l].n_tile_bins ? (uint32_t)BL_SEG_DIRECT : (uint32_t)a.lv[l].seg;
    };
    uint32_t c = 0, it = 0;
    for (int b = lo; b < hi; ++b) {
        const uint32_t seg = seg_of(b, level_of(b));
        c += a.count[b];
        it += (a.count[b] + seg - 1) / seg;
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
        a.offset[a.n_bins] = rc;
        *a.n_items = ri;
    }
    __syncthreads();
    c = part_c[tid];
    it = part_i[tid];
    for (int b = lo; b < hi; ++b) {
        const int l = level_of(b);
        const uint32_t seg = seg_of(b, l), k = a.count[b];
        a.offset[b] = c;
        for (uint32_t at = 0; at < k; at += seg) a.items[it++] = make_uint4((uint32_t)b, c + at, c + min(at + seg, k), (uint32_t)l);
        c += k;
    }
}

__global__ __launch_bounds__(256) void volume_bwd_fill_k(BwdLevels a) {
    int l = 0;
    while (l + 1 < a.n && blockIdx.x >= a.lv[l + 1].plan_b) ++l;
    const BwdLevel& L = a.lv[l];
    const uint32_t per_view = (L.n_waves + 255u) / 256u, b = blockIdx.x - L.plan_b;
    const int v = (int)(b / per_view);
    const uint32_t q = (b - (uint32_t)v * per_view) * 256u + threadIdx.x;
    const uint32_t code = q < L.n_waves ? L.codes[(uint32_t)v * L.n_waves + q] : 0u;
    bwd_bin_pairs<true>(a, L, code, v, bwd_tile_code(min(q, L.n_waves - 1u), L.d), (uint32_t)v * L.n_waves + q);
}

// One voxel of a tile in the item's view, in two steps so that ALL loads of the next tile are in flight while the current one is added to the window
// (a wave waits 2 - 4 us for its 17 loads and the window leaves room for 4 waves per SIMD):
//   bwd_issue    the projection (it depends on the lattice position only) and the loads: 8 cotangent floats, 4 means, the count byte -- planar, through
//                buffer descriptors (a 32-bit offset per lane, the plane in an SGPR) --, 4 texels
//   bwd_finish   g_view = (g_mean + 2 g_var (f_view - mean)) / count (the reference's formula term by term) and the taps
struct BwdLoads {
    float gm0, gm1, gm2, gm3, gv0, gv1, gv2, gv3, m0, m1, m2, m3;
    uint32_t cnt;
    f4 v00, v01, v10, v11;
    float fx, fy;
    bool vis;
};
struct BwdVoxel {
    float4 g;
    int x0, y0;
    float w00, w01, w10, w11;
    bool ok_x1, ok_y1, on;
};
struct BwdBuffers {                 // buffer descriptors of one level
    __amdgpu_buffer_rsrc_t gvol, vol, count, texels;
    uint32_t plane_bytes, row_bytes, view_bytes;
};
__device__ __forceinline__ BwdBuffers bwd_buffers(const BwdLevel& L, int nv) {
    BwdBuffers b;
    b.plane_bytes = L.n_waves * 256u;
    b.row_bytes = (uint32_t)L.w * 16u;
    b.view_bytes = (uint32_t)L.h * b.row_bytes;
    b.gvol = __builtin_amdgcn_make_buffer_rsrc((void*)L.gvol, 0, (int)(8u * b.plane_bytes), 0x00020000);
    b.vol = __builtin_amdgcn_make_buffer_rsrc((void*)L.vol, 0, (int)(4u * b.plane_bytes), 0x00020000);
    b.count = __builtin_amdgcn_make_buffer_rsrc((void*)L.count, 0, (int)(L.n_waves * 64u), 0x00020000);
    b.texels = __builtin_amdgcn_make_buffer_rsrc((void*)L.feat, 0, (int)((uint32_t)nv * b.view_bytes), 0x00020000);
    return b;
}
__device__ __forceinline__ float buffer_f32(const __amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t uniform_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, uniform_bytes, 0));
}
__device__ __forceinline__ BwdLoads bwd_issue(const BwdLevel& L, const BwdBuffers& b, uint32_t view_off, k1b_float_p m, k1b_float_p k, bool pinhole,
                                              const int4 box, const TileVoxel& tv) {
    const int d = L.d, w = L.w, h = L.h;
    const uint32_t at = tv.vox * 4u, pb = b.plane_bytes;
    BwdLoads o = {};
#if defined(GENS_K1_BWD_PROBE) && GENS_K1_BWD_PROBE == 4
    // (timing probe, WRONG sums: no loads at all -- lane-dependent constants in their place: what do the projection, the taps and the window adds cost alone?)
    {
        const float c0 = (float)(tv.vox & 1023u) * 1e-3f;
        o.cnt = 1u + (tv.vox & 3u);
        o.gm0 = c0, o.gm1 = c0 + 1.0f, o.gm2 = c0 + 2.0f, o.gm3 = c0 + 3.0f, o.gv0 = c0 * 2.0f, o.gv1 = c0 * 3.0f, o.gv2 = c0 * 4.0f, o.gv3 = c0 * 5.0f;
        o.m0 = c0 - 1.0f, o.m1 = c0 - 2.0f, o.m2 = c0 - 3.0f, o.m3 = c0 - 4.0f;
        const LeanProj p = project_lean(m, k, pinhole, L.lc, (float)(w - 1), (float)(h - 1), lattice_at(L.lc, d, tv.ix), lattice_at(L.lc, d, tv.jy), lattice_at(L.lc, d, tv.kz));
        o.vis = p.vis;
        o.fx = p.vis ? p.fx : 0.0f;
        o.fy = p.vis ? p.fy : 0.0f;
        o.v00 = (f4){c0, c0, c0, c0}, o.v01 = o.v00 + 1.0f, o.v10 = o.v00 + 2.0f, o.v11 = o.v00 + 3.0f;
        return o;
    }
#endif
    const LeanProj p = project_lean(m, k, pinhole, L.lc, (float)(w - 1), (float)(h - 1), lattice_at(L.lc, d, tv.ix), lattice_at(L.lc, d, tv.jy), lattice_at(L.lc, d, tv.kz));
    // a visible voxel reads inside the image: fx in [0, w - 1], fy in [0, h - 1]; the +1 taps may sit on column w / row h with weight exactly 0
    o.vis = p.vis;
    o.fx = p.vis ? p.fx : 0.0f;
    o.fy = p.vis ? p.fy : 0.0f;
    const int x0 = (int)floorf(o.fx), y0 = (int)floorf(o.fy);
    // Only the lanes whose voxel is visible in this view AND owned by this item's image tile load anything (box = x_org, y_org, width, height of the
    // tile): the kernel is bound by the bytes it fetches, not by its arithmetic or its LDS atomics (profiles/r06_k1_bwd_floor.txt: no loads 0.43 ms,
    // no window adds 0.66, all of it 0.69), and a third of the lanes of a visit belong to a neighbouring tile's visit of the same pair or to no view.
    if (!(p.vis && (uint32_t)(x0 - box.x) < (uint32_t)box.z && (uint32_t)(y0 - box.y) < (uint32_t)box.w)) return o;
    o.cnt = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(b.count, tv.vox, 0, 0);
    o.gm0 = buffer_f32(b.gvol, at, 0), o.gm1 = buffer_f32(b.gvol, at, pb), o.gm2 = buffer_f32(b.gvol, at, 2u * pb), o.gm3 = buffer_f32(b.gvol, at, 3u * pb);
    o.gv0 = buffer_f32(b.gvol, at, 4u * pb), o.gv1 = buffer_f32(b.gvol, at, 5u * pb), o.gv2 = buffer_f32(b.gvol, at, 6u * pb), o.gv3 = buffer_f32(b.gvol, at, 7u * pb);
    o.m0 = buffer_f32(b.vol, at, 0), o.m1 = buffer_f32(b.vol, at, pb), o.m2 = buffer_f32(b.vol, at, 2u * pb), o.m3 = buffer_f32(b.vol, at, 3u * pb);
    const uint32_t xo0 = (uint32_t)x0 << 4, xo1 = (uint32_t)min(x0 + 1, w - 1) << 4, y1 = (uint32_t)min(y0 + 1, h - 1);
    o.v00 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(b.texels, __umul24((uint32_t)y0, b.row_bytes) + xo0, view_off, 0));
    o.v01 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(b.texels, __umul24((uint32_t)y0, b.row_bytes) + xo1, view_off, 0));
    o.v10 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(b.texels, __umul24(y1, b.row_bytes) + xo0, view_off, 0));
    o.v11 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(b.texels, __umul24(y1, b.row_bytes) + xo1, view_off, 0));
    return o;
}
__device__ __forceinline__ BwdVoxel bwd_finish(const BwdLevel& L, const BwdLoads& r) {
    BwdVoxel o;
    // 1 / count: the count is a small integer, RN(1 / count) from the reciprocal instruction + one correction (rcp_rn) is the IEEE quotient
    const float inv = r.cnt ? rcp_rn((float)r.cnt) : 0.0f;
    const float4 ga = make_float4(r.gm0 * inv, r.gm1 * inv, r.gm2 * inv, r.gm3 * inv);
    const float4 gb = make_float4(2.0f * r.gv0 * inv, 2.0f * r.gv1 * inv, 2.0f * r.gv2 * inv, 2.0f * r.gv3 * inv);
    o.on = r.vis;                                                                   // (a visible voxel has count >= 1; zero cotangents add zeros)
    const float x0f = floorf(r.fx), y0f = floorf(r.fy);
    o.x0 = (int)x0f;
    o.y0 = (int)y0f;
    o.ok_x1 = o.x0 + 1 <= L.w - 1;
    o.ok_y1 = o.y0 + 1 <= L.h - 1;
    const float wx1 = r.fx - x0f, wx0 = (x0f + 1.0f) - r.fx, wy1 = r.fy - y0f, wy0 = (y0f + 1.0f) - r.fy;
    o.w00 = wx0 * wy0;
    o.w01 = wx1 * wy0;
    o.w10 = wx0 * wy1;
    o.w11 = wx1 * wy1;
    float4 f;
    f.x = __builtin_fmaf(r.v11.x, o.w11, __builtin_fmaf(r.v10.x, o.w10, __builtin_fmaf(r.v01.x, o.w01, __builtin_fmaf(r.v00.x, o.w00, 0.0f))));
    f.y = __builtin_fmaf(r.v11.y, o.w11, __builtin_fmaf(r.v10.y, o.w10, __builtin_fmaf(r.v01.y, o.w01, __builtin_fmaf(r.v00.y, o.w00, 0.0f))));
    f.z = __builtin_fmaf(r.v11.z, o.w11, __builtin_fmaf(r.v10.z, o.w10, __builtin_fmaf(r.v01.z, o.w01, __builtin_fmaf(r.v00.z, o.w00, 0.0f))));
    f.w = __builtin_fmaf(r.v11.w, o.w11, __builtin_fmaf(r.v10.w, o.w10, __builtin_fmaf(r.v01.w, o.w01, __builtin_fmaf(r.v00.w, o.w00, 0.0f))));
    o.g = make_float4(ga.x + gb.x * (f.x - r.m0), ga.y + gb.y * (f.y - r.m1), ga.z + gb.z * (f.z - r.m2), ga.w + gb.w * (f.w - r.m3));
    return o;
}

// One tap into the window: the four channels of g * w (the float32 products the direct scatter adds, converted exactly) into the four channel planes.
// Lane l adds channel (c + l) & 3 in the c-th instruction: the z-neighbours of a lattice row mostly share their texel, and four lanes on one
// WORD cost an instruction 45 cycles instead of 18 (sixteen: 190) -- rotated, runs of four lanes address four different planes.
struct RotatedLane {
    int plane[4];                   // ((c + lane) & 3) * BL_WIN
    bool b0, b1;
};
__device__ __forceinline__ float4 rotate4(const RotatedLane& rl, float4 g) {        // -> (g[(0 + l) & 3], g[(1 + l) & 3], g[(2 + l) & 3], g[(3 + l) & 3])
    const float t0 = rl.b0 ? g.y : g.x, t1 = rl.b0 ? g.z : g.y, t2 = rl.b0 ? g.w : g.z, t3 = rl.b0 ? g.x : g.w;
    return make_float4(rl.b1 ? t2 : t0, rl.b1 ? t3 : t1, rl.b1 ? t0 : t2, rl.b1 ? t1 : t3);
}
__device__ __forceinline__ void window_add4(double* win, const RotatedLane& rl, int at, float4 g_rot, float w) {
#if defined(GENS_K1_BWD_PROBE) && GENS_K1_BWD_PROBE == 1
    // (timing probe, WRONG sums: plain stores instead of the LDS atomics -- what do the 16 ds_add_f64 per voxel cost?  scripts/probe/k1_bwd_floor_probe.sh)
    win[rl.plane[0] + at] = (double)(g_rot.x * w);
    win[rl.plane[1] + at] = (double)(g_rot.y * w);
    win[rl.plane[2] + at] = (double)(g_rot.z * w);
    win[rl.plane[3] + at] = (double)(g_rot.w * w);
#elif defined(GENS_K1_BWD_PROBE) && GENS_K1_BWD_PROBE == 2
    // (timing probe, WRONG sums: one atomic per tap instead of four)
    atomicAdd(win + rl.plane[0] + at, (double)((g_rot.x + g_rot.y + g_rot.z + g_rot.w) * w));
#else
    atomicAdd(win + rl.plane[0] + at, (double)(g_rot.x * w));
    atomicAdd(win + rl.plane[1] + at, (double)(g_rot.y * w));
    atomicAdd(win + rl.plane[2] + at, (double)(g_rot.z * w));
    atomicAdd(win + rl.plane[3] + at, (double)(g_rot.w * w));
#endif
}
__device__ __forceinline__ void window_voxel(double* win, const RotatedLane& rl, const BwdVoxel& o, int x_org, int y_org) {
    const int cx = o.x0 - x_org, cy = o.y0 - y_org;
#if defined(GENS_K1_BWD_PROBE) && GENS_K1_BWD_PROBE == 3
    // (timing probe, WRONG sums: the loads and the arithmetic stay alive, no window adds -- what do the loads cost alone?)
    if (o.on && o.g.x + o.g.y + o.g.z + o.g.w + o.w00 == 1.2345e33f) win[cx] = 1.0;
    return;
#endif
    if (o.on && cx >= 0 && cx < BL_W && cy >= 0 && cy < BL_H) {                     // the voxel belongs to this image tile
        const int at = cy * (BL_W + 1) + cx;
        const float4 gr = rotate4(rl, o.g);
        window_add4(win, rl, at, gr, o.w00);
        if (o.ok_x1) window_add4(win, rl, at + 1, gr, o.w01);
        if (o.ok_y1) window_add4(win, rl, at + (BL_W + 1), gr, o.w10);
        if (o.ok_x1 && o.ok_y1) window_add4(win, rl, at + (BL_W + 1) + 1, gr, o.w11);
    }
}
__device__ __forceinline__ void direct_voxel(const BwdLevel& L, int v, const BwdVoxel& o) {
    if (!o.on) return;
    float* base = L.gfeat + (((int64_t)v * L.h + o.y0) * L.w + o.x0) * 4;
    atomic_add4(base, o.g, o.w00);
    if (o.ok_x1) atomic_add4(base + 4, o.g, o.w01);
    if (o.ok_y1) atomic_add4(base + (int64_t)L.w * 4, o.g, o.w10);
    if (o.ok_x1 && o.ok_y1) atomic_add4(base + (int64_t)L.w * 4 + 4, o.g, o.w11);
}

#ifdef GENS_K1_BWD_STAMPS
// (dev build: cycles of wave 0 per phase, summed over the launch -- 0 take an item, 1 zero the window, 2 the pairs, 3 the flush, 4 direct items, 5 items)
__device__ unsigned long long k1_bwd_stamps[8];
extern "C" int gens_debug_k1_bwd_stamps(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        return hipMemcpyToSymbol(HIP_SYMBOL(k1_bwd_stamps), z, sizeof(z)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k1_bwd_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#define K1B_STAMP(i)                                                                   \
    do {                                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter();                  \
        if (tid == 0) atomicAdd(&k1_bwd_stamps[i], now_ - stamp_);                     \
        stamp_ = now_;                                                                 \
    } while (0)
#else
#define K1B_STAMP(i)
#endif
__global__ __launch_bounds__(BL_THREADS) void volume_bwd_tiles_k(BwdLevels a, const float* __restrict__ w2c) {
    extern __shared__ double win[];                    // 4 channel planes of BL_WIN sums: the lanes of one add spread over all banks
    __shared__ uint32_t taken;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // One workgroup per CU takes work items off a counter until none is left: the items differ in length (a bin's last piece, the short pieces
    // of the direct bins), and a grid of one workgroup per POSSIBLE item was mostly empty workgroups (10 000 launched for 1 700 items at 256^3).
    const uint32_t n_items = *a.n_items;
#ifdef GENS_K1_BWD_STAMPS
    unsigned long long stamp_ = __builtin_readcyclecounter();
#endif
    for (;;) {
    if (tid == 0) taken = atomicAdd(a.next_item, 1u);
    __syncthreads();
    const uint32_t mine_item = taken;
    __syncthreads();
    if (mine_item >= n_items) return;
    K1B_STAMP(0);
    const uint4 item = a.items[mine_item];
    const BwdLevel& L = a.lv[__builtin_amdgcn_readfirstlane((int)item.w)];
    const int d = L.d, w = L.w, h = L.h;
    const uint32_t local = item.x - L.bin0;
    const BwdBuffers buf = bwd_buffers(L, a.nv);
    constexpr uint32_t NW = BL_THREADS / 64;
    if (local == L.n_tile_bins) {                                                  // the direct bin: pairs whose footprint spans more than 2 x 2 tiles
        for (uint32_t e = item.y + (uint32_t)wave; e < item.z; e += NW) {
            const uint32_t pair = a.list[e];
            const int v = (int)(pair / L.n_waves);
            const TileVoxel tv = bwd_tile_voxel(bwd_tile_code(pair - (uint32_t)v * L.n_waves, d), lane, d);
            const k1b_float_p m = k1_const(w2c + 16 * v), k = k1_const(L.intr + 16 * v);
            direct_voxel(L, v, bwd_finish(L, bwd_issue(L, buf, (uint32_t)v * buf.view_bytes, m, k, is_pinhole(m, k), make_int4(0, 0, 1 << 30, 1 << 30), tv)));
        }
#ifdef GENS_K1_BWD_STAMPS
        __syncthreads();
        K1B_STAMP(4);
#endif
        continue;
    }
    const int tx = (int)(local % (uint32_t)L.tiles_x), ty = (int)((local / (uint32_t)L.tiles_x) % (uint32_t)L.tiles_y), v = (int)(local / (uint32_t)(L.tiles_x * L.tiles_y));
    const int x_org = tx * BL_W, y_org = ty * BL_H;
    const int4 box = make_int4(x_org, y_org, BL_W, BL_H);
    for (int i = tid; i < 4 * BL_WIN; i += BL_THREADS) win[i] = 0.0;
    __syncthreads();
    K1B_STAMP(1);
    const k1b_float_p m = k1_const(w2c + 16 * v), k = k1_const(L.intr + 16 * v);      // (constant address space: see volume_build_chunk)
    const bool pinhole = is_pinhole(m, k);
    const uint32_t view_off = (uint32_t)v * buf.view_bytes;
    RotatedLane rl;
    rl.b0 = lane & 1;
    rl.b1 = lane & 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) rl.plane[c] = ((c + lane) & 3) * BL_WIN;
    // the waves take interleaved entries (wave w: e0 + w, e0 + w + 16, ...): neighbours in the list are neighbours in the volume and read the same texels
    // and the two halves of the same 128-byte lines of the planes at the same moment (64 consecutive entries per wave: 0.60 -> 0.70 ms; a streaming
    // hint on the planar loads: 0.76; work items ordered by slab of the volume so that the views meet in the memory-side cache: 0.62 -- r06_k1_bwd_floor.txt)
    for (uint32_t e0 = item.y; e0 + (uint32_t)wave < item.z; e0 += 64u * NW) {
        const uint32_t mine_at = e0 + (uint32_t)wave + NW * (uint32_t)lane;
        const uint32_t mine = mine_at < item.z ? a.list[mine_at] : 0u;            // the wave's next 64 tiles, one per lane
        const int cnt = (int)min(64u, (item.z - e0 - (uint32_t)wave + NW - 1u) / NW);
        // two tiles per trip, their loads in two named sets of registers (one set copied into the other per tile was 39 moves of ~230 instructions)
        BwdLoads even = bwd_issue(L, buf, view_off, m, k, pinhole, box, bwd_tile_voxel((uint32_t)__builtin_amdgcn_readlane((int)mine, 0), lane, d));
        for (int j = 0; j < cnt; j += 2) {
            // (past the end a trip loads the last tile again: no branch around the loads, and the lines are in L1)
            const BwdLoads odd = bwd_issue(L, buf, view_off, m, k, pinhole, box, bwd_tile_voxel((uint32_t)__builtin_amdgcn_readlane((int)mine, min(j + 1, cnt - 1)), lane, d));
            window_voxel(win, rl, bwd_finish(L, even), x_org, y_org);
            even = bwd_issue(L, buf, view_off, m, k, pinhole, box, bwd_tile_voxel((uint32_t)__builtin_amdgcn_readlane((int)mine, min(j + 2, cnt - 1)), lane, d));
            if (j + 1 < cnt) window_voxel(win, rl, bwd_finish(L, odd), x_org, y_org);
        }
    }
    __syncthreads();
    K1B_STAMP(2);
    float* out = L.gfeat + (int64_t)v * h * w * 4;
    for (int i = tid; i < 4 * BL_WIN; i += BL_THREADS) {      // a lane per FLOAT: the lanes of an atomic instruction on consecutive addresses (L2 serves requests, not lanes)
        const int texel = i >> 2, c = i & 3;
        const float s = (float)win[c * BL_WIN + texel];
        if (s != 0.0f) {
            const int r = texel / (BL_W + 1), cc = texel - r * (BL_W + 1);
            // Taps on row h / column w (the +1 neighbours of the last row / column) carry weight exactly 0, so their sums are 0 and stay in LDS --
            // unless the cotangent is NaN or infinite: 0 x NaN = NaN != 0 then reached this add, and for the last row of the last view that is
            // a write PAST THE END of the gradient buffer (a captured training step whose parameters went non-finite faulted there on its second
            // replay; eagerly the write landed in whatever the allocator had put behind the buffer).  Only texels of the image are written.
            if (y_org + r < h && x_org + cc < w) atomicAdd(out + ((int64_t)(y_org + r) * w + x_org + cc) * 4 + c, s);
        }
    }
    __syncthreads();                                                               // (the window is zeroed again by the next item)
    K1B_STAMP(3);
#ifdef GENS_K1_BWD_STAMPS
    if (tid == 0) atomicAdd(&k1_bwd_stamps[5], 1ull);
#endif
    }
}

static int64_t bwd_levels_layout(const int* hw, const int* dims, int n_levels, int nv, const bool* on, char* base, BwdLevels* out) {
    BwdLevels a = {};
    a.n = n_levels;
    a.nv = nv;
    int64_t pairs = 0, items = 0;
    uint32_t bins = 0, plan = 0;
    for (int l = 0; l < n_levels; ++l) {
        BwdLevel& L = a.lv[l];
        const int d = dims[l], h = hw[2 * l], w = hw[2 * l + 1];
        const bool live = !on || on[l];
        const int64_t n = (int64_t)d * d * d, nw = live ? n / 64 : 0;                  // voxel tiles
        L.h = h;
        L.w = w;
        L.d = d;
        L.tiles_x = (w + BL_W - 1) / BL_W;
        L.tiles_y = (h + BL_H - 1) / BL_H;
        L.n_tile_bins = live ? (uint32_t)(nv * L.tiles_x * L.tiles_y) : 0u;
        L.n_waves = (uint32_t)nw;
        L.seg = (int)std::min<int64_t>(BL_SEG, std::max<int64_t>(64, nw * nv / 512 / 64 * 64));       // (small levels: shorter items, more of them)
        L.bin0 = bins;
        L.lc = level_const(h, w, d);
        L.plan_b = plan;
        plan += (uint32_t)((nw + 255) / 256 * nv);                                 // (blocks do not straddle views)
        if (live) {
            bins += L.n_tile_bins + 1;
            items += (nw * nv * 4 + L.seg - 1) / L.seg + L.n_tile_bins + (nw * nv + BL_SEG_DIRECT - 1) / BL_SEG_DIRECT + 1;
            pairs += nw * nv;
        }
    }
    a.n_bins = bins;
    a.plan_blocks = plan;
    a.max_items = (uint32_t)items;
    int64_t at = 0;
    auto take = [&](int64_t bytes) { char* p = base ? base + at : nullptr; at += align256(bytes); return p; };
    a.count = (uint32_t*)take((int64_t)bins * 4);               // count, cursor, n_items: one block, zeroed per call
    a.cursor = (uint32_t*)take((int64_t)bins * 4);
    a.n_items = (uint32_t*)take(4);
    a.next_item = (uint32_t*)take(4);
    a.offset = (uint32_t*)take((int64_t)(bins + 1) * 4);
    a.items = (uint4*)take(items * 16);
    a.list = (uint32_t*)take(pairs * 4 * 4);
    for (int l = 0; l < n_levels; ++l) a.lv[l].codes = (uint32_t*)take((int64_t)a.lv[l].n_waves * nv * 4);
    if (out) *out = a;
    return at;
}

static bool bwd_levels_supported(const int* hw, const int* dims, int n_levels, int nv) {
    if (!hw || !dims || n_levels < 1 || n_levels > GENS_MAX_LEVELS || nv <= 0 || nv > GENS_MAX_VIEWS) return false;
    int64_t pairs = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int d = dims[l], h = hw[2 * l], w = hw[2 * l + 1];
        if (h <= 1 || w <= 1 || d <= 0 || d > 4096 || (d & 15) || (w + BL_W - 1) / BL_W > 4095 || (h + BL_H - 1) / BL_H > 4095) return false;
        if ((int64_t)nv * h * w * 16 > 0x7fffffffll) return false;                 // (texels are addressed with 32-bit byte offsets)
        if ((int64_t)d * d * d >= (1ll << 30)) return false;                       // (byte offsets into a plane are 32-bit)
        pairs += (int64_t)d * d * d / 64 * nv;
    }
    return pairs * 4 <= 0xFFFFFFFFll;                                               // (list offsets are 32-bit)
}

extern "C" int64_t gens_volume_build_bwd_levels_scratch_bytes(const int* hw, const int* dims, int n_levels, int nv) {
    if (!bwd_levels_supported(hw, dims, n_levels, nv)) return 0;
    return bwd_levels_layout(hw, dims, n_levels, nv, nullptr, nullptr, nullptr);
}

extern "C" int gens_volume_build_bwd_levels(const float* const* feat, const int* hw, const int* dims, int n_levels, const float* w2c,
                                            const float* const* intr, int nv, const float* const* volumes, const uint8_t* const* counts,
                                            const float* const* g_volumes, float* const* g_feat, void* scratch, int64_t scratch_bytes,
                                            void* stream) {
    GENS_CHECK_ARG(feat && hw && dims && w2c && intr && volumes && counts && g_volumes && g_feat && scratch, GENS_EINVAL, "gens_volume_build_bwd_levels: null table");
    GENS_CHECK_ARG(bwd_levels_supported(hw, dims, n_levels, nv), GENS_ELIMIT,
                   "gens_volume_build_bwd_levels: every volume side must be a multiple of 16 (%d levels, %d views): use gens_volume_build_bwd", n_levels, nv);
    bool on[GENS_MAX_LEVELS];
    bool any = false;
    for (int l = 0; l < n_levels; ++l) {
        on[l] = g_volumes[l] != nullptr;
        any = any || on[l];
        if (!on[l]) continue;
        if (int e = check_volume_args("gens_volume_build_bwd_levels", feat[l], w2c, intr[l], nv, hw[2 * l], hw[2 * l + 1], dims[l])) return e;
        GENS_CHECK_ARG(volumes[l] && counts[l] && g_feat[l], GENS_EINVAL, "gens_volume_build_bwd_levels: null buffer (level %d)", l);
        GENS_CHECK_ARG(((uintptr_t)feat[l] & 15) == 0, GENS_EINVAL, "gens_volume_build_bwd_levels: the texels are read as float4 and must be 16-byte aligned (level %d)", l);
    }
    if (!any) return 0;
    const int64_t need = bwd_levels_layout(hw, dims, n_levels, nv, nullptr, nullptr, nullptr);    // (the caller sizes for all levels)
    GENS_CHECK_ARG(scratch_bytes >= need && ((uintptr_t)scratch & 15) == 0, GENS_EINVAL,
                   "gens_volume_build_bwd_levels: scratch of %lld bytes, 16-byte aligned, needed (got %lld)", (long long)need, (long long)scratch_bytes);
    BwdLevels a;
    bwd_levels_layout(hw, dims, n_levels, nv, on, (char*)scratch, &a);
    for (int l = 0; l < n_levels; ++l) {
        if (!on[l]) continue;
        a.lv[l].feat = (const float4*)feat[l];
        a.lv[l].intr = intr[l];
        a.lv[l].vol = volumes[l];
        a.lv[l].count = counts[l];
        a.lv[l].gvol = g_volumes[l];
        a.lv[l].gfeat = g_feat[l];
    }
    static GensLdsOptIn lds_set;                                                    // (the window is more than the 64 KB a kernel gets by default)
    if (int e = gens_lds_opt_in(lds_set, (const void*)volume_bwd_tiles_k, BL_LDS_BYTES, "gens_volume_build_bwd_levels")) return e;
    hipStream_t st = (hipStream_t)stream;
    // A kernel, not hipMemsetAsync: captured into a HIP graph (ROCm 7.0 runtime of the GPU boxes) the memset NODE did not order against the kernel
    // nodes behind it -- the plan kernel counted on top of what the previous replay had left, the offsets ran past the end of the lists, and the
    // fill kernel of a captured training step's SECOND replay wrote into unmapped memory (profiles/r05_k1_bwd_graph_fault.txt).
    {
        const uint32_t words = (uint32_t)(((char*)a.offset - (char*)a.count) / 4);
        volume_bwd_zero_k<<<(words + 255u) / 256u, 256, 0, st>>>(a.count, words);
    }
    volume_bwd_plan_k<<<a.plan_blocks, 256, 0, st>>>(a, w2c);
    volume_bwd_scan_k<<<1, 256, 0, st>>>(a);
    volume_bwd_fill_k<<<a.plan_blocks, 256, 0, st>>>(a);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    volume_bwd_tiles_k<<<std::min<uint32_t>((uint32_t)n_cu, a.max_items), BL_THREADS, BL_LDS_BYTES, st>>>(a, w2c);
    return gens_launch_status("gens_volume_build_bwd_levels");
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
