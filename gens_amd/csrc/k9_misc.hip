// K9 (patch reads of surface_patch_warp), K10 (total-variation regulariser), K11 (SDF lattice), and the error slot.
//   K9  projector.py:406-416 (+ the F.interpolate of implicit_surface.py:316-325)
//   K10 implicit_surface.py:135-150        K11 implicit_surface.py:407-418
#include <stdarg.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// error slot
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
void gens_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gens_last_error(void) { return g_err; }
extern "C" int gens_abi_version(void) { return 12; }

// ---------------------------------------------------------------------------------------------------------------
// K9: bilinear read of a texel image at pixel coordinates (align_corners=True after the reference's own
// normalisation cancels), forward and d/d(xy).  One thread per (point, texel) forward, per point backward.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_fwd_k(const float4* __restrict__ img, int h, int w, int c, int q4,
                                                   const float* __restrict__ xy, int64_t p, float* __restrict__ out) {
    int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= p * q4) return;
    int q = (int)(gid % q4);
    int64_t i = gid / q4;
    Taps2 t = bilinear_taps(xy[2 * i], xy[2 * i + 1], h, w);
    float4 v = sample_texel(img, h, w, q4, q, t);
    float* o = out + i * c + 4 * q;
    o[0] = v.x;
    if (4 * q + 1 < c) o[1] = v.y;
    if (4 * q + 2 < c) o[2] = v.z;
    if (4 * q + 3 < c) o[3] = v.w;
}

__device__ __forceinline__ float dot_c(float4 v, const float* g, int c, int q) {
    float s = v.x * g[0];
    if (4 * q + 1 < c) s += v.y * g[1];
    if (4 * q + 2 < c) s += v.z * g[2];
    if (4 * q + 3 < c) s += v.w * g[3];
    return s;
}

__global__ __launch_bounds__(256) void patch_bwd_k(const float4* __restrict__ img, int h, int w, int c, int q4,
                                                   const float* __restrict__ xy, const float* __restrict__ g_out, int64_t p,
                                                   float* __restrict__ g_xy) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p) return;
    float ix = xy[2 * i], iy = xy[2 * i + 1];
    Taps2 t = bilinear_taps(ix, iy, h, w);
    bool fin = isfinite(ix) && isfinite(iy);
    float fx = (float)t.x0, fy = (float)t.y0;
    float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    float gx = 0.0f, gy = 0.0f;
    if (fin) {
        int64_t base = ((int64_t)t.y0 * w + t.x0) * q4;
        for (int q = 0; q < q4; ++q) {
            const float* g = g_out + i * c + 4 * q;
            float d00 = t.ok00 ? dot_c(img[base + q], g, c, q) : 0.0f;
            float d01 = t.ok01 ? dot_c(img[base + q4 + q], g, c, q) : 0.0f;
            float d10 = t.ok10 ? dot_c(img[base + (int64_t)w * q4 + q], g, c, q) : 0.0f;
            float d11 = t.ok11 ? dot_c(img[base + (int64_t)w * q4 + q4 + q], g, c, q) : 0.0f;
            gx += (d01 - d00) * wy0 + (d11 - d10) * wy1;
            gy += (d10 - d00) * wx0 + (d11 - d01) * wx1;
        }
    }
    g_xy[2 * i] = gx;
    g_xy[2 * i + 1] = gy;
}

// F.interpolate(mode="bilinear", align_corners=False) of an NCHW map, written into a channel slice of a texel tensor
__global__ __launch_bounds__(256) void upsample2d_into_k(const float* __restrict__ src, int c, int hs, int ws, float* __restrict__ dst,
                                                         int h, int w, int cpad, int c_off, int64_t total) {
    int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    int ch = (int)(gid % c);
    int x = (int)((gid / c) % w);
    int y = (int)((gid / ((int64_t)c * w)) % h);
    int64_t img = gid / ((int64_t)c * w * h);
    float sy = fmaxf(((float)y + 0.5f) * ((float)hs / (float)h) - 0.5f, 0.0f);
    float sx = fmaxf(((float)x + 0.5f) * ((float)ws / (float)w) - 0.5f, 0.0f);
    int y0 = min((int)sy, hs - 1), x0 = min((int)sx, ws - 1);
    int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
    float ty = sy - (float)y0, tx = sx - (float)x0;
    const float* s = src + (img * c + ch) * (int64_t)hs * ws;
    float top = s[(int64_t)y0 * ws + x0] * (1.0f - tx) + s[(int64_t)y0 * ws + x1] * tx;
    float bot = s[(int64_t)y1 * ws + x0] * (1.0f - tx) + s[(int64_t)y1 * ws + x1] * tx;
    dst[((img * h + y) * (int64_t)w + x) * cpad + c_off + ch] = top * (1.0f - ty) + bot * ty;
}

extern "C" int gens_patch_sample_fwd(const float* image, int h, int w, int c, const float* xy, int64_t p, float* out, void* stream) {
    GENS_CHECK_ARG(image && h > 1 && w > 1 && c > 0, GENS_EINVAL, "gens_patch_sample_fwd: bad image");
    GENS_CHECK_ARG(p >= 0 && (p == 0 || (xy && out)), GENS_EINVAL, "gens_patch_sample_fwd: null xy/out");
    if (p == 0) return 0;
    int q4 = (c + 3) / 4;
    patch_fwd_k<<<gens_blocks(p * q4, 256), 256, 0, (hipStream_t)stream>>>((const float4*)image, h, w, c, q4, xy, p, out);
    return gens_launch_status("gens_patch_sample_fwd");
}

extern "C" int gens_patch_sample_bwd(const float* image, int h, int w, int c, const float* xy, const float* g_out, int64_t p,
                                     float* g_xy, void* stream) {
    GENS_CHECK_ARG(image && h > 1 && w > 1 && c > 0, GENS_EINVAL, "gens_patch_sample_bwd: bad image");
    GENS_CHECK_ARG(p >= 0 && (p == 0 || (xy && g_out && g_xy)), GENS_EINVAL, "gens_patch_sample_bwd: null pointer");
    if (p == 0) return 0;
    patch_bwd_k<<<gens_blocks(p, 256), 256, 0, (hipStream_t)stream>>>((const float4*)image, h, w, c, (c + 3) / 4, xy, g_out, p, g_xy);
    return gens_launch_status("gens_patch_sample_bwd");
}

extern "C" int gens_upsample2d_into(const float* src, int n, int c, int hs, int ws, float* dst, int h, int w, int c_pad_dst, int c_off,
                                    void* stream) {
    GENS_CHECK_ARG(src && dst && n > 0 && c > 0 && hs > 0 && ws > 0 && h > 0 && w > 0, GENS_EINVAL, "gens_upsample2d_into: bad argument");
    GENS_CHECK_ARG(c_off >= 0 && c_off + c <= c_pad_dst, GENS_EINVAL, "gens_upsample2d_into: channel slice [%d,%d) outside %d", c_off,
                   c_off + c, c_pad_dst);
    int64_t total = (int64_t)n * h * w * c;
    upsample2d_into_k<<<gens_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(src, c, hs, ws, dst, h, w, c_pad_dst, c_off, total);
    return gens_launch_status("gens_upsample2d_into");
}

// cat([f0, up(f1), up(f2), ...], 1) as texels in ONE pass: a thread per pixel writes all its channels (whole 16-byte pieces of a contiguous
// C_pad x 4-byte texel), where a launch of upsample2d_into_k per level wrote a quarter of every texel each time, behind a zero fill of the whole
// tensor (three launches + fill: 135 us per training step at 5 x 480 x 640 x 12; this one: see DESIGN.md).  Per value the arithmetic of
// upsample2d_into_k, operation for operation.
struct UpsampleCat {
    const float* src[GENS_MAX_LEVELS];
    int c[GENS_MAX_LEVELS], hs[GENS_MAX_LEVELS], ws[GENS_MAX_LEVELS];
    int n;
};
template <bool FOUR>
__global__ __launch_bounds__(256) void upsample2d_cat_k(UpsampleCat A, float* __restrict__ dst, int h, int w, int cpad, int64_t pixels) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= pixels) return;
    const int x = (int)(gid % w), y = (int)((gid / w) % h);
    const int64_t img = gid / ((int64_t)w * h);
    float* out = dst + gid * cpad;
    int off = 0;
    for (int l = 0; l < A.n; ++l) {
        const int c = A.c[l], hs = A.hs[l], ws = A.ws[l];
        const float sy = fmaxf(((float)y + 0.5f) * ((float)hs / (float)h) - 0.5f, 0.0f);
        const float sx = fmaxf(((float)x + 0.5f) * ((float)ws / (float)w) - 0.5f, 0.0f);
        const int y0 = min((int)sy, hs - 1), x0 = min((int)sx, ws - 1);
        const int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
        const float ty = sy - (float)y0, tx = sx - (float)x0;
        const int64_t plane = (int64_t)hs * ws;
        const float* s = A.src[l] + img * c * plane;
        const int64_t i00 = (int64_t)y0 * ws + x0, i01 = (int64_t)y0 * ws + x1, i10 = (int64_t)y1 * ws + x0, i11 = (int64_t)y1 * ws + x1;
        if (FOUR) {      // (every level has four channels: one 16-byte store per level)
            float v[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float* sc = s + ch * plane;
                const float top = sc[i00] * (1.0f - tx) + sc[i01] * tx;
                const float bot = sc[i10] * (1.0f - tx) + sc[i11] * tx;
                v[ch] = top * (1.0f - ty) + bot * ty;
            }
            *(float4*)(out + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int ch = 0; ch < c; ++ch) {
                const float* sc = s + ch * plane;
                const float top = sc[i00] * (1.0f - tx) + sc[i01] * tx;
                const float bot = sc[i10] * (1.0f - tx) + sc[i11] * tx;
                out[off + ch] = top * (1.0f - ty) + bot * ty;
            }
        }
        off += c;
    }
    for (int ch = off; ch < cpad; ++ch) out[ch] = 0.0f;                             // the pad channels (what the zero fill left there)
}

extern "C" int gens_upsample2d_cat(const float* const* srcs, const int* chw, int n_maps, int n, float* dst, int h, int w, int c_pad_dst, void* stream) {
    GENS_CHECK_ARG(srcs && chw && dst && n > 0 && h > 0 && w > 0, GENS_EINVAL, "gens_upsample2d_cat: bad argument");
    GENS_CHECK_ARG(n_maps >= 1 && n_maps <= GENS_MAX_LEVELS, GENS_ELIMIT, "gens_upsample2d_cat: %d maps (at most %d)", n_maps, GENS_MAX_LEVELS);
    UpsampleCat A = {};
    A.n = n_maps;
    int ctot = 0;
    bool four = (c_pad_dst & 3) == 0 && ((uintptr_t)dst & 15) == 0;
    for (int l = 0; l < n_maps; ++l) {
        A.src[l] = srcs[l];
        A.c[l] = chw[3 * l], A.hs[l] = chw[3 * l + 1], A.ws[l] = chw[3 * l + 2];
        GENS_CHECK_ARG(A.src[l] && A.c[l] > 0 && A.hs[l] > 0 && A.ws[l] > 0, GENS_EINVAL, "gens_upsample2d_cat: bad map %d", l);
        four = four && A.c[l] == 4;
        ctot += A.c[l];
    }
    GENS_CHECK_ARG(ctot <= c_pad_dst, GENS_EINVAL, "gens_upsample2d_cat: %d channels into texels of %d", ctot, c_pad_dst);
    const int64_t pixels = (int64_t)n * h * w;
    if (four)
        upsample2d_cat_k<true><<<gens_blocks(pixels, 256), 256, 0, (hipStream_t)stream>>>(A, dst, h, w, c_pad_dst, pixels);
    else
        upsample2d_cat_k<false><<<gens_blocks(pixels, 256), 256, 0, (hipStream_t)stream>>>(A, dst, h, w, c_pad_dst, pixels);
    return gens_launch_status("gens_upsample2d_cat");
}

// ---------------------------------------------------------------------------------------------------------------
// K10: TV.  One thread per voxel (z fastest => coalesced), forward differences along the three axes for the four
// channels, block reduction to one float4 per block (deterministic; the host sums the partials).
// ---------------------------------------------------------------------------------------------------------------
#define TV_BLOCK 256
extern "C" int gens_tv_blocks(int64_t n_voxels) { return (int)gens_blocks(n_voxels, TV_BLOCK); }

__global__ __launch_bounds__(TV_BLOCK) void tv_fwd_k(const float* __restrict__ vol, const float* __restrict__ mask, int X, int Y, int Z,
                                                     float4* __restrict__ partial) {
    __shared__ float4 red[TV_BLOCK / 64];
    int64_t n = (int64_t)X * Y * Z;
    int64_t i = (int64_t)blockIdx.x * TV_BLOCK + threadIdx.x;
    float4 acc = f4_zero();
    if (i < n) {
        int kz = (int)(i % Z), jy = (int)((i / Z) % Y), ix = (int)(i / ((int64_t)Z * Y));
        float m = mask[i];
        bool mx = (ix + 1 < X) && (m * mask[i + (int64_t)Y * Z] > 0.0f);
        bool my = (jy + 1 < Y) && (m * mask[i + Z] > 0.0f);
        bool mz = (kz + 1 < Z) && (m * mask[i + 1] > 0.0f);
        for (int c = 0; c < 4; ++c) {
            const float* v = vol + c * n + i;
            float v0 = v[0];
            if (mx) { float d = v[(int64_t)Y * Z] - v0; acc.x += d * d; }
            if (my) { float d = v[Z] - v0; acc.y += d * d; }
            if (mz) { float d = v[1] - v0; acc.z += d * d; }
        }
        acc.w = mx ? 1.0f : 0.0f;
    }
    acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y); acc.z = wave_sum(acc.z); acc.w = wave_sum(acc.w);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 t = red[0];
        for (int k = 1; k < TV_BLOCK / 64; ++k) { t.x += red[k].x; t.y += red[k].y; t.z += red[k].z; t.w += red[k].w; }
        partial[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(TV_BLOCK) void tv_bwd_k(const float* __restrict__ vol, const float* __restrict__ mask, int X, int Y, int Z,
                                                     float coef, const float* __restrict__ coef_dev, float* __restrict__ g_vol) {
    int64_t n = (int64_t)X * Y * Z;
    int64_t i = (int64_t)blockIdx.x * TV_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (coef_dev != nullptr) coef *= coef_dev[0];
    int kz = (int)(i % Z), jy = (int)((i / Z) % Y), ix = (int)(i / ((int64_t)Z * Y));
    int64_t sx = (int64_t)Y * Z, sy = Z;
    float m = mask[i];
    bool px = (ix + 1 < X) && (m * mask[i + sx] > 0.0f), nx = (ix > 0) && (m * mask[i - sx] > 0.0f);
    bool py = (jy + 1 < Y) && (m * mask[i + sy] > 0.0f), ny = (jy > 0) && (m * mask[i - sy] > 0.0f);
    bool pz = (kz + 1 < Z) && (m * mask[i + 1] > 0.0f), nz = (kz > 0) && (m * mask[i - 1] > 0.0f);
    for (int c = 0; c < 4; ++c) {
        const float* v = vol + c * n + i;
        float v0 = v[0], g = 0.0f;
        if (px) g -= 2.0f * (v[sx] - v0);
        if (nx) g += 2.0f * (v0 - v[-sx]);
        if (py) g -= 2.0f * (v[sy] - v0);
        if (ny) g += 2.0f * (v0 - v[-sy]);
        if (pz) g -= 2.0f * (v[1] - v0);
        if (nz) g += 2.0f * (v0 - v[-1]);
        g_vol[c * n + i] = coef * g;
    }
}

// Vectorised forms (Z % 4 == 0, fewer than 2^31 voxels: every shipped level): one thread owns FOUR consecutive z of one (x, y) row, so
// every plane access is one 16-byte load / store (the scalar kernels issued 16 / 28 four-byte loads per voxel and spent ~100 instructions
// on 64-bit index division: 20 % of HBM peak); the z + 1 / z - 1 neighbours across a thread boundary come from one extra scalar load.
// 32-bit indices.  Same arithmetic per voxel, so the partial sums differ from the scalar kernel's only in summation order.
__global__ __launch_bounds__(TV_BLOCK) void tv_fwd4_k(const float* __restrict__ vol, const float* __restrict__ mask, int X, int Y, int Z,
                                                      float4* __restrict__ partial) {
    __shared__ float4 red[TV_BLOCK / 64];
    const uint32_t n = (uint32_t)X * Y * Z, q = n >> 2;
    const uint32_t t = blockIdx.x * TV_BLOCK + threadIdx.x;
    float4 acc = f4_zero();
    if (t < q) {
        const uint32_t i = t << 2, zq = (uint32_t)Z >> 2;
        const uint32_t kz = (t % zq) << 2, row = t / zq, jy = row % (uint32_t)Y, ix = row / (uint32_t)Y;
        const uint32_t sx = (uint32_t)Y * Z, sy = (uint32_t)Z;
        const bool hx = ix + 1 < (uint32_t)X, hy = jy + 1 < (uint32_t)Y, hz = kz + 4 < (uint32_t)Z;
        const float4 m = *(const float4*)(mask + i);
        const float4 mxv = hx ? *(const float4*)(mask + i + sx) : f4_zero();
        const float4 myv = hy ? *(const float4*)(mask + i + sy) : f4_zero();
        const float mzn = hz ? mask[i + 4] : 0.0f;
        const float mm[4] = {m.x, m.y, m.z, m.w}, mxa[4] = {mxv.x, mxv.y, mxv.z, mxv.w}, mya[4] = {myv.x, myv.y, myv.z, myv.w};
        const float mza[4] = {m.y, m.z, m.w, mzn};
        bool bx[4], by[4], bz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bx[k] = hx && (mm[k] * mxa[k] > 0.0f);
            by[k] = hy && (mm[k] * mya[k] > 0.0f);
            bz[k] = (k < 3 || hz) && (mm[k] * mza[k] > 0.0f);
            acc.w += bx[k] ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* v = vol + (size_t)c * n + i;
            const float4 v0 = *(const float4*)v;
            const float4 vx = hx ? *(const float4*)(v + sx) : f4_zero();
            const float4 vy = hy ? *(const float4*)(v + sy) : f4_zero();
            const float vzn = hz ? v[4] : 0.0f;
            const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, ax[4] = {vx.x, vx.y, vx.z, vx.w}, ay[4] = {vy.x, vy.y, vy.z, vy.w};
            const float az[4] = {v0.y, v0.z, v0.w, vzn};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (bx[k]) { const float d = ax[k] - a0[k]; acc.x += d * d; }
                if (by[k]) { const float d = ay[k] - a0[k]; acc.y += d * d; }
                if (bz[k]) { const float d = az[k] - a0[k]; acc.z += d * d; }
            }
        }
    }
    acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y); acc.z = wave_sum(acc.z); acc.w = wave_sum(acc.w);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 r = red[0];
        for (int k = 1; k < TV_BLOCK / 64; ++k) { r.x += red[k].x; r.y += red[k].y; r.z += red[k].z; r.w += red[k].w; }
        partial[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(TV_BLOCK) void tv_bwd4_k(const float* __restrict__ vol, const float* __restrict__ mask, int X, int Y, int Z,
                                                      float coef, const float* __restrict__ coef_dev, float* __restrict__ g_vol) {
    const uint32_t n = (uint32_t)X * Y * Z, q = n >> 2;
    const uint32_t t = blockIdx.x * TV_BLOCK + threadIdx.x;
    if (t >= q) return;
    if (coef_dev != nullptr) coef *= coef_dev[0];
    const uint32_t i = t << 2, zq = (uint32_t)Z >> 2;
    const uint32_t kz = (t % zq) << 2, row = t / zq, jy = row % (uint32_t)Y, ix = row / (uint32_t)Y;
    const uint32_t sx = (uint32_t)Y * Z, sy = (uint32_t)Z;
    const bool hxp = ix + 1 < (uint32_t)X, hxn = ix > 0, hyp = jy + 1 < (uint32_t)Y, hyn = jy > 0, hzp = kz + 4 < (uint32_t)Z, hzn = kz > 0;
    const float4 m = *(const float4*)(mask + i);
    const float4 mxp = hxp ? *(const float4*)(mask + i + sx) : f4_zero(), mxn = hxn ? *(const float4*)(mask + i - sx) : f4_zero();
    const float4 myp = hyp ? *(const float4*)(mask + i + sy) : f4_zero(), myn = hyn ? *(const float4*)(mask + i - sy) : f4_zero();
    const float mzp = hzp ? mask[i + 4] : 0.0f, mzn = hzn ? mask[i - 1] : 0.0f;
    const float mm[4] = {m.x, m.y, m.z, m.w};
    const float axp[4] = {mxp.x, mxp.y, mxp.z, mxp.w}, axn[4] = {mxn.x, mxn.y, mxn.z, mxn.w};
    const float ayp[4] = {myp.x, myp.y, myp.z, myp.w}, ayn[4] = {myn.x, myn.y, myn.z, myn.w};
    const float azp[4] = {m.y, m.z, m.w, mzp}, azn[4] = {mzn, m.x, m.y, m.z};
    bool px[4], nx[4], py[4], ny[4], pz[4], nz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = hxp && (mm[k] * axp[k] > 0.0f);
        nx[k] = hxn && (mm[k] * axn[k] > 0.0f);
        py[k] = hyp && (mm[k] * ayp[k] > 0.0f);
        ny[k] = hyn && (mm[k] * ayn[k] > 0.0f);
        pz[k] = (k < 3 || hzp) && (mm[k] * azp[k] > 0.0f);
        nz[k] = (k > 0 || hzn) && (mm[k] * azn[k] > 0.0f);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* v = vol + (size_t)c * n + i;
        const float4 v0 = *(const float4*)v;
        const float4 vxp = hxp ? *(const float4*)(v + sx) : f4_zero(), vxn = hxn ? *(const float4*)(v - sx) : f4_zero();
        const float4 vyp = hyp ? *(const float4*)(v + sy) : f4_zero(), vyn = hyn ? *(const float4*)(v - sy) : f4_zero();
        const float vzp = hzp ? v[4] : 0.0f, vzn = hzn ? v[-1] : 0.0f;
        const float a0[4] = {v0.x, v0.y, v0.z, v0.w};
        const float bxp[4] = {vxp.x, vxp.y, vxp.z, vxp.w}, bxn[4] = {vxn.x, vxn.y, vxn.z, vxn.w};
        const float byp[4] = {vyp.x, vyp.y, vyp.z, vyp.w}, byn[4] = {vyn.x, vyn.y, vyn.z, vyn.w};
        const float bzp[4] = {v0.y, v0.z, v0.w, vzp}, bzn[4] = {vzn, v0.x, v0.y, v0.z};
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {      // the scalar kernel's order of additions, so both kernels give the same bits
            float s = 0.0f;
            if (px[k]) s -= 2.0f * (bxp[k] - a0[k]);
            if (nx[k]) s += 2.0f * (a0[k] - bxn[k]);
            if (py[k]) s -= 2.0f * (byp[k] - a0[k]);
            if (ny[k]) s += 2.0f * (a0[k] - byn[k]);
            if (pz[k]) s -= 2.0f * (bzp[k] - a0[k]);
            if (nz[k]) s += 2.0f * (a0[k] - bzn[k]);
            g[k] = coef * s;
        }
        *(float4*)(g_vol + (size_t)c * n + i) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

// the float4 kernels need 16-byte aligned planes (a view carved out of a flat buffer at an odd offset takes the scalar kernels)
static bool tv_vectorised(int x, int y, int z, const void* a, const void* b, const void* c = nullptr) {
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
    return aligned && (z & 3) == 0 && (int64_t)x * y * z < ((int64_t)1 << 31) && !getenv("GENS_TV_SCALAR");     // (the switch keeps the scalar kernels reachable for A/B tests)
}

extern "C" int gens_tv_fwd(const float* vol, const float* mask, int x, int y, int z, float* partial, void* stream) {
    GENS_CHECK_ARG(vol && mask && partial && x > 0 && y > 0 && z > 0, GENS_EINVAL, "gens_tv_fwd: bad argument");
    int64_t n = (int64_t)x * y * z;
    if (tv_vectorised(x, y, z, vol, mask)) {   // same number of partial blocks as the scalar kernel (gens_tv_blocks): the tail blocks write zeros
        tv_fwd4_k<<<gens_blocks(n, TV_BLOCK), TV_BLOCK, 0, (hipStream_t)stream>>>(vol, mask, x, y, z, (float4*)partial);
        return gens_launch_status("gens_tv_fwd");
    }
    tv_fwd_k<<<gens_blocks(n, TV_BLOCK), TV_BLOCK, 0, (hipStream_t)stream>>>(vol, mask, x, y, z, (float4*)partial);
    return gens_launch_status("gens_tv_fwd");
}

extern "C" int gens_tv_bwd(const float* vol, const float* mask, int x, int y, int z, float coef, float* g_vol, void* stream) {
    GENS_CHECK_ARG(vol && mask && g_vol && x > 0 && y > 0 && z > 0, GENS_EINVAL, "gens_tv_bwd: bad argument");
    int64_t n = (int64_t)x * y * z;
    if (tv_vectorised(x, y, z, vol, mask, g_vol)) {
        tv_bwd4_k<<<gens_blocks(n / 4, TV_BLOCK), TV_BLOCK, 0, (hipStream_t)stream>>>(vol, mask, x, y, z, coef, nullptr, g_vol);
        return gens_launch_status("gens_tv_bwd");
    }
    tv_bwd_k<<<gens_blocks(n, TV_BLOCK), TV_BLOCK, 0, (hipStream_t)stream>>>(vol, mask, x, y, z, coef, nullptr, g_vol);
    return gens_launch_status("gens_tv_bwd");
}

extern "C" int gens_tv_bwd_scaled(const float* vol, const float* mask, int x, int y, int z, float coef, const float* coef_dev, float* g_vol,
                                  void* stream) {
    GENS_CHECK_ARG(vol && mask && coef_dev && g_vol && x > 0 && y > 0 && z > 0, GENS_EINVAL, "gens_tv_bwd_scaled: bad argument");
    int64_t n = (int64_t)x * y * z;
    if (tv_vectorised(x, y, z, vol, mask, g_vol)) {
        tv_bwd4_k<<<gens_blocks(n / 4, TV_BLOCK), TV_BLOCK, 0, (hipStream_t)stream>>>(vol, mask, x, y, z, coef, coef_dev, g_vol);
        return gens_launch_status("gens_tv_bwd_scaled");
    }
    tv_bwd_k<<<gens_blocks(n, TV_BLOCK), TV_BLOCK, 0, (hipStream_t)stream>>>(vol, mask, x, y, z, coef, coef_dev, g_vol);
    return gens_launch_status("gens_tv_bwd_scaled");
}

// ---------------------------------------------------------------------------------------------------------------
// K11: lattice points
// ---------------------------------------------------------------------------------------------------------------
struct Box { float lo[3], hi[3]; };
__global__ __launch_bounds__(256) void lattice_k(Box b, int res, int64_t first, int64_t count, float* __restrict__ pts) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    int64_t i = first + t;
    int kz = (int)(i % res), jy = (int)((i / res) % res), ix = (int)(i / ((int64_t)res * res));
    pts[3 * t] = linspace_at(b.lo[0], b.hi[0], res, ix);
    pts[3 * t + 1] = linspace_at(b.lo[1], b.hi[1], res, jy);
    pts[3 * t + 2] = linspace_at(b.lo[2], b.hi[2], res, kz);
}

extern "C" int gens_lattice_points(const float* bmin3_host, const float* bmax3_host, int res, int64_t first, int64_t count, float* pts,
                                   void* stream) {
    GENS_CHECK_ARG(bmin3_host && bmax3_host && res > 0 && first >= 0 && count >= 0, GENS_EINVAL, "gens_lattice_points: bad argument");
    GENS_CHECK_ARG(first + count <= (int64_t)res * res * res, GENS_EINVAL, "gens_lattice_points: range beyond res^3");
    if (count == 0) return 0;
    GENS_CHECK_ARG(pts, GENS_EINVAL, "gens_lattice_points: null output");
    Box b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = bmin3_host[a]; b.hi[a] = bmax3_host[a]; }
    lattice_k<<<gens_blocks(count, 256), 256, 0, (hipStream_t)stream>>>(b, res, first, count, pts);
    return gens_launch_status("gens_lattice_points");
}
