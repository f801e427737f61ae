// Measurement probe: float atomic adds into a gradient-map-sized region (32 MB) at DEVICE scope (one shared copy) against WORKGROUP scope into a
// PRIVATE copy per XCD (blockIdx % 8, checked against the XCC_ID hardware register): the flush side of K1 backward / K2 backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode 0: device scope, one copy | 1: workgroup scope, copy per XCD (blockIdx % 8) | 2: device scope, copy per XCD | 3: workgroup scope, one copy (WRONG across XCDs: timing only)
// | 6: device scope, float, the 64 lanes of an instruction on 64 CONSECUTIVE floats (16 texels x 4 channels: 4 cache lines per instruction)
// | 7: as 6 with 16 consecutive floats per 16 lanes (one texel quad per 16 lanes, quads anywhere)
// | 4: device scope, one copy, uint32 adds | 5: device scope, one copy, uint64 adds (two channels per word: half the texel span)
template <int MODE>
__global__ __launch_bounds__(256) void atomic_k(float* __restrict__ maps, uint32_t n_texels, int per_thread, int* __restrict__ xcc_mismatch) {
    const uint32_t xcd = blockIdx.x & 7u;
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;      // HW_REG_XCC_ID, bits [3:0]
        if (hw != xcd) atomicAdd(xcc_mismatch, 1);
    }
    float* dst = maps + ((MODE == 1 || MODE == 2) ? (size_t)xcd * n_texels * 4 : 0);
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    if (MODE == 6 || MODE == 7) {
        const uint32_t lane = threadIdx.x & 63u, wave = tid >> 6;
        for (int k = 0; k < per_thread * 4; ++k) {           // the same number of atomics per thread as the other modes
            const uint32_t t = MODE == 6 ? (hash32(wave * 977u + k) % (n_texels - 16u)) : (hash32((wave * 4u + (lane >> 4)) * 977u + k) % (n_texels - 4u));
            const uint32_t off = MODE == 6 ? lane : (lane & 15u);
            __hip_atomic_fetch_add(dst + (size_t)t * 4 + off, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    for (int k = 0; k < per_thread; ++k) {
        // neighbouring lanes hit neighbouring texels (as consecutive voxels of a row do), rows land anywhere
        const uint32_t t = (hash32((tid >> 4) * 977u + k) + (tid & 15u)) % n_texels;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (MODE == 4) __hip_atomic_fetch_add((uint32_t*)dst + (size_t)t * 4 + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 5) __hip_atomic_fetch_add((unsigned long long*)dst + (size_t)(t >> 1) * 4 + c, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1 || MODE == 3) __hip_atomic_fetch_add(dst + (size_t)t * 4 + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(dst + (size_t)t * 4 + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" void atomic_probe(float* maps, uint32_t n_texels, int per_thread, int blocks, int mode, int* mismatch, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case 0: atomic_k<0><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
        case 1: atomic_k<1><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
        case 2: atomic_k<2><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
        case 4: atomic_k<4><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
        case 6: atomic_k<6><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
        case 7: atomic_k<7><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
        case 5: atomic_k<5><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
        default: atomic_k<3><<<blocks, 256, 0, s>>>(maps, n_texels, per_thread, mismatch); break;
    }
}
