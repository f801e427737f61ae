// Measurement probe: the store side of K1 alone -- 9 planes of D^3 floats written with the kernel's pattern.
// variant 0: one dword per lane per plane (K1 today) | 1: one dwordx4 per lane per plane (4 consecutive z per lane) |
// 2: dword stores + ~400 dependent FMAs per lane (arithmetic of the K1 size, no loads) | 3: dwordx2 per lane per plane
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int VARIANT>
__global__ __launch_bounds__(256) void store_k(float* __restrict__ vol, float* __restrict__ mask, size_t n, float seed) {
    if (VARIANT == 1) {
        const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
        const float4 v = make_float4(seed, seed + 1, seed + 2, seed + 3);
#pragma unroll
        for (int p = 0; p < 8; ++p) *(float4*)(vol + p * n + i) = v;
        *(float4*)(mask + i) = v;
    } else if (VARIANT == 3) {
        const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
        const float2 v = make_float2(seed, seed + 1);
#pragma unroll
        for (int p = 0; p < 8; ++p) *(float2*)(vol + p * n + i) = v;
        *(float2*)(mask + i) = v;
    } else {
        const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        float a = seed + (float)threadIdx.x;
        if (VARIANT == 2) {
            float b = a * 0.5f, c = a + 1.0f, d = a - 2.0f;
#pragma unroll 1
            for (int k = 0; k < 100; ++k) {
                a = __builtin_fmaf(a, 0.999f, b); b = __builtin_fmaf(b, 0.998f, c); c = __builtin_fmaf(c, 0.997f, d); d = __builtin_fmaf(d, 0.996f, a);
            }
            a += b + c + d;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) vol[p * n + i] = a + (float)p;
        mask[i] = a;
    }
}

// variant 4: variant 2 with CHUNKS consecutive 256-voxel chunks per workgroup: the stores of chunk i are in flight while chunk i+1 computes
template <int CHUNKS>
__global__ __launch_bounds__(256) void store_loop_k(float* __restrict__ vol, float* __restrict__ mask, size_t n, float seed) {
#pragma unroll 1
    for (int c = 0; c < CHUNKS; ++c) {
        const size_t i = ((size_t)blockIdx.x * CHUNKS + c) * 256 + threadIdx.x;
        float a = seed + (float)threadIdx.x + (float)c;
        float b = a * 0.5f, cc = a + 1.0f, d = a - 2.0f;
#pragma unroll 1
        for (int k = 0; k < 100; ++k) {
            a = __builtin_fmaf(a, 0.999f, b); b = __builtin_fmaf(b, 0.998f, cc); cc = __builtin_fmaf(cc, 0.997f, d); d = __builtin_fmaf(d, 0.996f, a);
        }
        a += b + cc + d;
#pragma unroll
        for (int p = 0; p < 8; ++p) vol[p * n + i] = a + (float)p;
        mask[i] = a;
    }
}

extern "C" int store_probe(float* vol, float* mask, int d, int variant, void* stream) {
    const size_t n = (size_t)d * d * d;
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 0: store_k<0><<<(unsigned)(n / 256), 256, 0, s>>>(vol, mask, n, 1.0f); break;
        case 1: store_k<1><<<(unsigned)(n / 1024), 256, 0, s>>>(vol, mask, n, 1.0f); break;
        case 2: store_k<2><<<(unsigned)(n / 256), 256, 0, s>>>(vol, mask, n, 1.0f); break;
        case 4: store_loop_k<4><<<(unsigned)(n / 1024), 256, 0, s>>>(vol, mask, n, 1.0f); break;
        case 5: store_loop_k<8><<<(unsigned)(n / 2048), 256, 0, s>>>(vol, mask, n, 1.0f); break;
        case 6: store_loop_k<16><<<(unsigned)(n / 4096), 256, 0, s>>>(vol, mask, n, 1.0f); break;
        case 7: store_loop_k<32><<<(unsigned)(n / 8192), 256, 0, s>>>(vol, mask, n, 1.0f); break;
        case 3: default: store_k<3><<<(unsigned)(n / 512), 256, 0, s>>>(vol, mask, n, 1.0f); break;
    }
    return (int)hipGetLastError();
}
