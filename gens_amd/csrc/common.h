// Internal helpers shared by the gfx950 kernels of libgens_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gens_hip.h"

#define GENS_WAVE 64

void gens_set_error(const char* fmt, ...);

#define GENS_CHECK_ARG(cond, code, ...)  \
    do {                                 \
        if (!(cond)) {                   \
            gens_set_error(__VA_ARGS__); \
            return (code);               \
        }                                \
    } while (0)

static inline int gens_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gens_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline unsigned gens_blocks(int64_t work, int per_block) { return (unsigned)((work + per_block - 1) / per_block); }

// Kernels that need more LDS than the 64 KB a launch gets by default opt in with hipFuncSetAttribute -- which acts on the CURRENT device's
// copy of the function.  Done once per (kernel, device): a bit per device, set after the call succeeded (setting it twice from two threads is
// harmless); a refusal is an error of the entry point, not a launch that fails later.
#include <atomic>
struct GensLdsOptIn {
    std::atomic<uint64_t> done{0};
};
static inline int gens_lds_opt_in(GensLdsOptIn& state, const void* kernel, int bytes, const char* who) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) {
        gens_set_error("%s: no usable current device (hipGetDevice -> %d; at most 64 devices per process)", who, dev);
        (void)hipGetLastError();
        return GENS_EINVAL;
    }
    const uint64_t bit = 1ull << dev;
    if (state.done.load(std::memory_order_acquire) & bit) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        gens_set_error("%s: device %d does not grant %d bytes of LDS to one workgroup", who, dev, bytes);
        return GENS_ELIMIT;
    }
    state.done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// The hardware-id fields the stash slots of K6g / K6gh are built from (HW_REG_HW_ID: SIMD [5:4], CU [11:8], SE [15:13]; HW_REG_XCC_ID [2:0])
// are gfx950's; another target needs its own decode (the slot index is masked into range and every slot is guarded by a lock word, so a
// wrong decode costs speed, never correctness -- but it must not go unnoticed).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libgens_hip is written for gfx950 (MI355X): the hardware-id decode of the K6g / K6gh stash slots and the MFMA shapes are that target's"
#endif

// Bounded wait for a stash slot's lock word: a slot is always found free unless two waves were mapped to one slot (another partition mode: a
// few sleeps) or an earlier launch died holding it (nothing will ever free it: trap, so that the host sees a failed launch instead of a hang;
// gens_sdf_grad*_stash_reset re-zeroes the stash afterwards).  ~2^22 x s_sleep(32) is seconds.
__device__ __forceinline__ void gens_lock_slot(uint32_t* lock, uint32_t seen) {
    uint32_t tries = 0;
    while (seen != 0u) {
        __builtin_amdgcn_s_sleep(32);
        seen = atomicCAS(lock, 0u, 1u);
        if (++tries > (1u << 22)) __builtin_trap();
    }
}

// Up to GENS_MAX_LEVELS volumes / maps passed by value in the kernel argument block (scalar loads, no indirection
// through HBM).
struct LevelSet {
    const float* data[GENS_MAX_LEVELS];
    float* grad[GENS_MAX_LEVELS];
    const float* aux[GENS_MAX_LEVELS];
    int dx[GENS_MAX_LEVELS], dy[GENS_MAX_LEVELS], dz[GENS_MAX_LEVELS];
    int n;
    int bits;   // mask pyramids only: data[] are bit-packed words (gens_pack_mask_bits), bit i = voxel i (C order) is set
};

// torch.linspace(start, end, steps)[i] in float32 (ATen's symmetric formula: the upper half counts down from end).
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// 4x4 row-major matrix times (x,y,z,1): full homogeneous result.
__device__ __forceinline__ float4 mat4_point(const float* __restrict__ m, float x, float y, float z) {
    float4 r;
    r.x = m[0] * x + m[1] * y + m[2] * z + m[3];
    r.y = m[4] * x + m[5] * y + m[6] * z + m[7];
    r.z = m[8] * x + m[9] * y + m[10] * z + m[11];
    r.w = m[12] * x + m[13] * y + m[14] * z + m[15];
    return r;
}

// wave-level helpers (wave = 64 lanes on gfx950).  Cross-lane traffic goes through DPP (data-parallel primitives: the
// operand of a VALU instruction is taken from a neighbouring lane, no LDS round trip as with ds_bpermute / __shfl):
// row_shr:n inside the 16-lane rows, then row_bcast:15 / row_bcast:31 to carry the row totals -- 6 VALU steps per scan.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {   // lanes without a source (or in masked-off rows) get `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
#define GENS_DPP_SCAN(v, ident, OP)                          \
    do {                                                     \
        v = OP(v, dpp_move<0x111, 0xF>(ident, v)); /* row_shr:1 */    \
        v = OP(v, dpp_move<0x112, 0xF>(ident, v)); /* row_shr:2 */    \
        v = OP(v, dpp_move<0x114, 0xF>(ident, v)); /* row_shr:4 */    \
        v = OP(v, dpp_move<0x118, 0xF>(ident, v)); /* row_shr:8 */    \
        v = OP(v, dpp_move<0x142, 0xA>(ident, v)); /* row_bcast:15 -> rows 1, 3 */ \
        v = OP(v, dpp_move<0x143, 0xC>(ident, v)); /* row_bcast:31 -> rows 2, 3 */ \
    } while (0)
__device__ __forceinline__ float op_add_(float a, float b) { return a + b; }
__device__ __forceinline__ float op_mul_(float a, float b) { return a * b; }
__device__ __forceinline__ float op_max_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float lane_value(float v, int lane) {   // lane must be wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// inclusive prefix product / sum across the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
    (void)lane;
    GENS_DPP_SCAN(v, 1.0f, op_mul_);
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
    (void)lane;
    GENS_DPP_SCAN(v, 0.0f, op_add_);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    GENS_DPP_SCAN(v, 0.0f, op_add_);
    return lane_value(v, 63);
}
__device__ __forceinline__ float wave_max(float v) {
    GENS_DPP_SCAN(v, -3.402823466e38f, op_max_);
    return lane_value(v, 63);
}
// value of the next / previous lane (wave_shl:1 / wave_shr:1); the lane without a neighbour gets `edge`
__device__ __forceinline__ float lane_next(float v, float edge) { return dpp_move<0x130, 0xF>(edge, v); }
__device__ __forceinline__ float lane_prev(float v, float edge) { return dpp_move<0x138, 0xF>(edge, v); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sin_f32 / v_cos_f32: ~1 ulp, one instruction each) for
// the fused MLP kernels, whose accurate-libm activations would otherwise out-weigh the MFMA work 1:1 in issue slots.
__device__ __forceinline__ float hw_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float hw_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float hw_sigmoid(float x) { return hw_rcp(1.0f + hw_exp(-x)); }
// sin / cos with a two-term Cody-Waite reduction to revolutions in [-0.5, 0.5] (argument error < 2e-7 rad for |x| < 1e4)
__device__ __forceinline__ void hw_sincos(float x, float& s, float& c) {
    const float inv2pi_hi = 0.15915494309189535f, inv2pi_lo = 6.4206322e-9f;
    float k = rintf(x * inv2pi_hi);
    float v = fmaf(x, inv2pi_lo, fmaf(x, inv2pi_hi, -k));
    s = __builtin_amdgcn_sinf(v);
    c = __builtin_amdgcn_cosf(v);
}

// nearest-neighbour mask read with grid_sample(align_corners=False) index math: ((p+1)*D-1)/2 rounded half-to-even (Q6)
__device__ __forceinline__ float mask_nearest(const float* __restrict__ m, int dx, int dy, int dz, float px, float py,
                                              float pz) {
    float fx = rintf(((px + 1.0f) * (float)dx - 1.0f) / 2.0f);
    float fy = rintf(((py + 1.0f) * (float)dy - 1.0f) / 2.0f);
    float fz = rintf(((pz + 1.0f) * (float)dz - 1.0f) / 2.0f);
    if (!(fx >= 0.0f && fx < (float)dx && fy >= 0.0f && fy < (float)dy && fz >= 0.0f && fz < (float)dz)) return 0.0f;
    return m[((int64_t)(int)fx * dy + (int)fy) * dz + (int)fz];
}

// the same read from a bit-packed mask (1 bit per voxel: the 256^3 level is 2 MB and stays in L2)
__device__ __forceinline__ bool mask_nearest_bit(const uint32_t* __restrict__ m, int dx, int dy, int dz, float px, float py, float pz) {
    float fx = rintf(((px + 1.0f) * (float)dx - 1.0f) / 2.0f);
    float fy = rintf(((py + 1.0f) * (float)dy - 1.0f) / 2.0f);
    float fz = rintf(((pz + 1.0f) * (float)dz - 1.0f) / 2.0f);
    if (!(fx >= 0.0f && fx < (float)dx && fy >= 0.0f && fy < (float)dy && fz >= 0.0f && fz < (float)dz)) return false;
    const int64_t i = ((int64_t)(int)fx * dy + (int)fy) * dz + (int)fz;
    return (m[i >> 5] >> (unsigned)(i & 31)) & 1u;
}

__device__ __forceinline__ bool any_mask(const LevelSet& ms, float px, float py, float pz) {
    bool ok = false;
    if (ms.bits) {
        for (int l = 0; l < ms.n; ++l) ok = ok || mask_nearest_bit((const uint32_t*)ms.data[l], ms.dx[l], ms.dy[l], ms.dz[l], px, py, pz);
    } else {
        for (int l = 0; l < ms.n; ++l) ok = ok || (mask_nearest(ms.data[l], ms.dx[l], ms.dy[l], ms.dz[l], px, py, pz) > 0.0f);
    }
    return ok;
}

// Bilinear taps of a zero-padded texel image: pixel coordinates (ix, iy) may be anything, incl. NaN/inf.
struct Taps2 {
    int x0, y0;
    float w00, w01, w10, w11;  // (y,x): 00 = north-west
    bool ok00, ok01, ok10, ok11;
};
__device__ __forceinline__ Taps2 bilinear_taps(float ix, float iy, int h, int w) {
    Taps2 t;
    bool fin = isfinite(ix) && isfinite(iy);
    float sx = fin ? ix : 0.0f, sy = fin ? iy : 0.0f;
    float fx = floorf(sx), fy = floorf(sy);
    fx = fminf(fmaxf(fx, -4.0f), (float)w + 4.0f);
    fy = fminf(fmaxf(fy, -4.0f), (float)h + 4.0f);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    float wx1 = sx - fx, wx0 = (fx + 1.0f) - sx;
    float wy1 = sy - fy, wy0 = (fy + 1.0f) - sy;
    t.w00 = wx0 * wy0;
    t.w01 = wx1 * wy0;
    t.w10 = wx0 * wy1;
    t.w11 = wx1 * wy1;
    bool xin0 = t.x0 >= 0 && t.x0 < w, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 < w;
    bool yin0 = t.y0 >= 0 && t.y0 < h, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 < h;
    t.ok00 = fin && xin0 && yin0;
    t.ok01 = fin && xin1 && yin0;
    t.ok10 = fin && xin0 && yin1;
    t.ok11 = fin && xin1 && yin1;
    return t;
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// acc + v * s with fused multiply-adds (the file is compiled with -ffp-contract=off so that index / mask DECISIONS keep
// their reference rounding; interpolated VALUES may use the FMA the reference's own CUDA kernels are compiled to: half the
// instructions of the gather kernels' inner loops, and one rounding less)
__device__ __forceinline__ float4 f4_madd(float4 acc, float4 v, float s) {
    acc.x = __builtin_fmaf(v.x, s, acc.x);
    acc.y = __builtin_fmaf(v.y, s, acc.y);
    acc.z = __builtin_fmaf(v.z, s, acc.z);
    acc.w = __builtin_fmaf(v.w, s, acc.w);
    return acc;
}

// texel (float4) bilinear read from img (h, w, cpad/4 float4s), texel slot q.  Branch-free: the four loads are issued
// back to back from coordinates clamped into the image (one s_waitcnt for all of them instead of four serialised
// load-use pairs under exec-mask branches); a tap outside the image keeps grid_sample's zero padding through a zero
// weight (image values are finite, so 0 * v adds exactly nothing).
__device__ __forceinline__ float4 sample_texel(const float4* __restrict__ img, int h, int w, int q4, int q, const Taps2& t) {
    const int x0 = min(max(t.x0, 0), w - 1), x1 = min(max(t.x0 + 1, 0), w - 1);
    const int y0 = min(max(t.y0, 0), h - 1), y1 = min(max(t.y0 + 1, 0), h - 1);
    const float4* r0 = img + (int64_t)y0 * w * q4 + q;
    const float4* r1 = img + (int64_t)y1 * w * q4 + q;
    const float4 v00 = r0[(int64_t)x0 * q4], v01 = r0[(int64_t)x1 * q4], v10 = r1[(int64_t)x0 * q4], v11 = r1[(int64_t)x1 * q4];
    float4 acc = f4_zero();
    acc = f4_madd(acc, v00, t.ok00 ? t.w00 : 0.0f);
    acc = f4_madd(acc, v01, t.ok01 ? t.w01 : 0.0f);
    acc = f4_madd(acc, v10, t.ok10 ? t.w10 : 0.0f);
    acc = f4_madd(acc, v11, t.ok11 ? t.w11 : 0.0f);
    return acc;
}

// The same bilinear read by the two lanes of a pair (2k, 2k + 1) TOGETHER.  The vector L1 looks up one 128-byte line per clock whatever the
// lanes want from it (scripts/probe/gather_rate_probe.py: 0.97 sixteen-byte loads per clock and CU with a line per lane, 1.87 with two adjacent
// lanes on one line), and the two taps of an image row are 32 contiguous bytes: so every load instruction serves ONE item of the pair -- lane 2k
// reads its x0 tap, lane 2k + 1 the x1 tap next to it -- and the halves are swapped back with quad_perm DPP moves.  Same number of loads per
// lane as sample_texel, half the line look-ups; same weights and order of accumulation: bit-identical.  ALL lanes of the wave must call it
// (a lane without an item passes any in-range taps); `row_off` = texel index of the item's image inside `tab` (view * h * w).
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }   // quad_perm:[1,0,3,2] (every lane has a source; bound_ctrl lets the compiler fold the move into its consumer)
__device__ __forceinline__ float pair_swap(float v) { return __builtin_bit_cast(float, pair_swap(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float4 pair_swap(float4 v) { return make_float4(pair_swap(v.x), pair_swap(v.y), pair_swap(v.z), pair_swap(v.w)); }

struct PairTaps {         // texel indices of the four loads of one lane: rows of the even lane's item (a0, a1), of the odd lane's (b0, b1)
    int a0, a1, b0, b1;
};
__device__ __forceinline__ PairTaps pair_taps(int row_off, int h, int w, const Taps2& t, int odd) {
    const int x0 = min(max(t.x0, 0), w - 1), x1 = min(max(t.x0 + 1, 0), w - 1);
    const int r0 = row_off + min(max(t.y0, 0), h - 1) * w, r1 = row_off + min(max(t.y0 + 1, 0), h - 1) * w;
    const int p_r0 = pair_swap(r0), p_r1 = pair_swap(r1), p_x0 = pair_swap(x0), p_x1 = pair_swap(x1);
    const int ca = odd ? p_x1 : x0, cb = odd ? x1 : p_x0;
    PairTaps q;
    q.a0 = (odd ? p_r0 : r0) + ca;
    q.a1 = (odd ? p_r1 : r1) + ca;
    q.b0 = (odd ? r0 : p_r0) + cb;
    q.b1 = (odd ? r1 : p_r1) + cb;
    return q;
}
__device__ __forceinline__ float4 sample_texel_pair(const float4* __restrict__ tab, const PairTaps& q, const Taps2& t, int odd) {
    const float4 a0 = tab[q.a0], a1 = tab[q.a1], b0 = tab[q.b0], b1 = tab[q.b1];
    const float4 sa0 = pair_swap(a0), sa1 = pair_swap(a1), sb0 = pair_swap(b0), sb1 = pair_swap(b1);
    // even lane: its x0 taps are a0 / a1, its x1 taps the partner's a0 / a1; odd lane: x1 = b0 / b1, x0 = the partner's b0 / b1
    const float4 v00 = odd ? sb0 : a0, v01 = odd ? b0 : sa0, v10 = odd ? sb1 : a1, v11 = odd ? b1 : sa1;
    float4 acc = f4_zero();
    acc = f4_madd(acc, v00, t.ok00 ? t.w00 : 0.0f);
    acc = f4_madd(acc, v01, t.ok01 ? t.w01 : 0.0f);
    acc = f4_madd(acc, v10, t.ok10 ? t.w10 : 0.0f);
    acc = f4_madd(acc, v11, t.ok11 ? t.w11 : 0.0f);
    return acc;
}

__device__ __forceinline__ void atomic_add4(float* p, float4 v, float s) {
    atomicAdd(p + 0, v.x * s);
    atomicAdd(p + 1, v.y * s);
    atomicAdd(p + 2, v.z * s);
    atomicAdd(p + 3, v.w * s);
}
