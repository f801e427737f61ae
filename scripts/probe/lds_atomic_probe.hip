// Measurement probe: LDS atomic-add cost by operand type (f32 / u32 / u64 / plain read-modify-write) and address pattern of a wave's 64 lanes:
// the accumulate side of K1 backward's image-tile kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

// pattern 0: 64 distinct consecutive words | 1: all lanes one word | 2: runs of 4 lanes share a word | 3: runs of 16 lanes share a word
// | 4: distinct, stride 4 words (float4 texels, one channel)
__device__ __forceinline__ int slot(int lane, int pattern, int k) {
    const int base = (k * 67) & 1023;
    switch (pattern) {
        case 0: return base + lane;
        case 1: return base;
        case 2: return base + (lane >> 2);
        case 3: return base + (lane >> 4);
        default: return base + 4 * lane;
    }
}

// type 0: float atomic | 1: uint32 atomic | 2: uint64 atomic | 3: float load + add + store (NOT a sum: timing only) | 4: double atomic
// | 5: float atomic, relaxed at workgroup scope (__hip_atomic_fetch_add: the native ds_add_f32)
template <int TYPE>
__global__ __launch_bounds__(256) void lds_k(int pattern, int iters, float* out) {
    __shared__ unsigned long long buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float* f = (float*)buf;
    uint32_t* u = (uint32_t*)buf;
    for (int k = 0; k < iters; ++k) {
        const int s = slot(lane, pattern, k);
        if (TYPE == 0) atomicAdd(f + s, 1.0f);
        else if (TYPE == 1) atomicAdd(u + s, 1u);
        else if (TYPE == 2) atomicAdd(buf + (s & 2047), 1ull);
        else if (TYPE == 4) atomicAdd((double*)buf + (s & 2047), 1.0);
        else if (TYPE == 5) __hip_atomic_fetch_add(f + s, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else f[s] = f[s] + 1.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = f[5];
}

extern "C" void lds_probe(int type, int pattern, int iters, int blocks, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (type) {
        case 0: lds_k<0><<<blocks, 256, 0, s>>>(pattern, iters, out); break;
        case 1: lds_k<1><<<blocks, 256, 0, s>>>(pattern, iters, out); break;
        case 2: lds_k<2><<<blocks, 256, 0, s>>>(pattern, iters, out); break;
        case 4: lds_k<4><<<blocks, 256, 0, s>>>(pattern, iters, out); break;
        case 5: lds_k<5><<<blocks, 256, 0, s>>>(pattern, iters, out); break;
        default: lds_k<3><<<blocks, 256, 0, s>>>(pattern, iters, out); break;
    }
}
