// Measurement probe (not part of libgens_hip.so): the one-voxel-per-lane K1 forward kernel with parts of its memory traffic
// removed, to see what bounds it.  VARIANT: 0 baseline | 1 every lane reads texel 0 | 2 no texel loads | 3 no stores |
// 4 one tap instead of four | 5 no loads, no stores.  Results are meaningless except for VARIANT 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct LevelConst { float step, cw, ch, rcw, rch; int log2d; };

__device__ __forceinline__ float div_rn(float a, float b, float y) {
    const float q = a * y;
    const float r = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float4 madd(float4 acc, float4 v, float s) {
    acc.x = __builtin_fmaf(v.x, s, acc.x); acc.y = __builtin_fmaf(v.y, s, acc.y);
    acc.z = __builtin_fmaf(v.z, s, acc.z); acc.w = __builtin_fmaf(v.w, s, acc.w);
    return acc;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void probe_k(const float4* __restrict__ feat, const float* __restrict__ w2c, const float* __restrict__ intr,
                                               int nv, int h, int w, int d, LevelConst lc, int min_vis, float* __restrict__ vol,
                                               float* __restrict__ mask) {
    const int tid = threadIdx.x;
    const int kz = tid & (d - 1);
    const int row = (int)blockIdx.x * (256 >> lc.log2d) + (tid >> lc.log2d);
    const int jy = row & (d - 1), ix = row >> lc.log2d;
    if (ix >= d) return;
    const int64_t n = (int64_t)d << (2 * lc.log2d);
    const int64_t idx = ((int64_t)row << lc.log2d) + kz;
    const int half = d >> 1;
    const float x = ix < half ? -1.0f + lc.step * (float)ix : 1.0f - lc.step * (float)(d - 1 - ix);
    const float y = jy < half ? -1.0f + lc.step * (float)jy : 1.0f - lc.step * (float)(d - 1 - jy);
    const float z = kz < half ? -1.0f + lc.step * (float)kz : 1.0f - lc.step * (float)(d - 1 - kz);
    float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
    float cnt = 0.0f;
    for (int v = 0; v < nv; ++v) {
        const float* m = w2c + 16 * v;
        const float* k = intr + 16 * v;
        float4 cam;
        cam.x = m[0] * x + m[1] * y + m[2] * z + m[3];
        cam.y = m[4] * x + m[5] * y + m[6] * z + m[7];
        cam.z = m[8] * x + m[9] * y + m[10] * z + m[11];
        cam.w = m[12] * x + m[13] * y + m[14] * z + m[15];
        const float u = k[0] * cam.x + k[1] * cam.y + k[2] * cam.z + k[3] * cam.w;
        const float vv = k[4] * cam.x + k[5] * cam.y + k[6] * cam.z + k[7] * cam.w;
        const float dd = k[8] * cam.x + k[9] * cam.y + k[10] * cam.z + k[11] * cam.w;
        const float dn = dd + 1e-8f;
        const float yd = 1.0f / dn;
        const float px = div_rn(u, dn, yd), py = div_rn(vv, dn, yd);
        const float nx = div_rn(px, lc.cw, lc.rcw) - 1.0f, ny = div_rn(py, lc.ch, lc.rch) - 1.0f;
        const bool vis = (fabsf(nx) <= 1.0f) && (fabsf(ny) <= 1.0f) && (dd > 0.0f);
        if (!vis) continue;
        const float fx = (nx + 1.0f) / 2.0f * (float)(w - 1), fy = (ny + 1.0f) / 2.0f * (float)(h - 1);
        const float x0f = floorf(fx), y0f = floorf(fy);
        int x0 = (int)x0f, y0 = (int)y0f;
        int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
        const float wx1 = fx - x0f, wx0 = (x0f + 1.0f) - fx, wy1 = fy - y0f, wy0 = (y0f + 1.0f) - fy;
        const float4* img = feat + (int64_t)v * h * w;
        float4 v00, v01, v10, v11;
        if (VARIANT == 1) { x0 = y0 = x1 = y1 = 0; }
        if (VARIANT == 2 || VARIANT == 5) {
            v00 = make_float4(wx0, wx1, wy0, wy1); v01 = make_float4(wx1, wx0, wy0, wy1); v10 = make_float4(wy0, wx1, wx0, wy1); v11 = make_float4(wy1, wx1, wy0, wx0);
        } else if (VARIANT == 4) {
            v00 = img[y0 * w + x0]; v01 = v00; v10 = v00; v11 = v00;
        } else {
            v00 = img[y0 * w + x0]; v01 = img[y0 * w + x1]; v10 = img[y1 * w + x0]; v11 = img[y1 * w + x1];
        }
        float4 f = madd(make_float4(0, 0, 0, 0), v00, wx0 * wy0);
        f = madd(f, v01, wx1 * wy0);
        f = madd(f, v10, wx0 * wy1);
        f = madd(f, v11, wx1 * wy1);
        s1.x += f.x; s1.y += f.y; s1.z += f.z; s1.w += f.w;
        s2 = make_float4(__builtin_fmaf(f.x, f.x, s2.x), __builtin_fmaf(f.y, f.y, s2.y), __builtin_fmaf(f.z, f.z, s2.z), __builtin_fmaf(f.w, f.w, s2.w));
        cnt += 1.0f;
    }
    const float den = cnt <= 0.0f ? 1e-8f : cnt;
    const float yn = 1.0f / den;
    const float4 mm = make_float4(div_rn(s1.x, den, yn), div_rn(s1.y, den, yn), div_rn(s1.z, den, yn), div_rn(s1.w, den, yn));
    if ((VARIANT == 3 || VARIANT == 5) && cnt != 12345.0f) return;
    if (VARIANT == 6) {                                       // streaming (non-temporal) stores: keep the texels in L2
        __builtin_nontemporal_store(mm.x, vol + idx);
        __builtin_nontemporal_store(mm.y, vol + n + idx);
        __builtin_nontemporal_store(mm.z, vol + 2 * n + idx);
        __builtin_nontemporal_store(mm.w, vol + 3 * n + idx);
        __builtin_nontemporal_store(div_rn(s2.x, den, yn) - mm.x * mm.x, vol + 4 * n + idx);
        __builtin_nontemporal_store(div_rn(s2.y, den, yn) - mm.y * mm.y, vol + 5 * n + idx);
        __builtin_nontemporal_store(div_rn(s2.z, den, yn) - mm.z * mm.z, vol + 6 * n + idx);
        __builtin_nontemporal_store(div_rn(s2.w, den, yn) - mm.w * mm.w, vol + 7 * n + idx);
        __builtin_nontemporal_store(cnt > (float)min_vis ? 1.0f : 0.0f, mask + idx);
        return;
    }
    vol[idx] = mm.x;
    vol[n + idx] = mm.y;
    vol[2 * n + idx] = mm.z;
    vol[3 * n + idx] = mm.w;
    vol[4 * n + idx] = div_rn(s2.x, den, yn) - mm.x * mm.x;
    vol[5 * n + idx] = div_rn(s2.y, den, yn) - mm.y * mm.y;
    vol[6 * n + idx] = div_rn(s2.z, den, yn) - mm.z * mm.z;
    vol[7 * n + idx] = div_rn(s2.w, den, yn) - mm.w * mm.w;
    mask[idx] = cnt > (float)min_vis ? 1.0f : 0.0f;
}

extern "C" int k1_probe(const float* feat, const float* w2c, const float* intr, int nv, int h, int w, int d, float* volume, float* mask,
                        int variant, void* stream) {
    LevelConst lc;
    lc.step = 2.0f / (float)(d - 1);
    lc.cw = (float)(w - 1) / 2.0f; lc.ch = (float)(h - 1) / 2.0f;
    lc.rcw = 1.0f / lc.cw; lc.rch = 1.0f / lc.ch;
    lc.log2d = 0;
    while ((1 << lc.log2d) < d) ++lc.log2d;
    const unsigned rpb = 256u >> lc.log2d;
    const unsigned grid = ((unsigned)d * (unsigned)d + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
#define GO(V) probe_k<V><<<grid, 256, 0, s>>>((const float4*)feat, w2c, intr, nv, h, w, d, lc, 1, volume, mask)
    switch (variant) {
        case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break; case 5: GO(5); break; default: GO(6); break;
    }
    return (int)hipGetLastError();
}
