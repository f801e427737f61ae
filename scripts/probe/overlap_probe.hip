// Measurement probe: do VALU work and stores of one kernel overlap on MI355X?  Each lane runs NF dependent-chain FMAs (4 chains),
// then writes NS planes.  wrap != 0 folds every store address into a 1 MiB window (L2-resident: no HBM traffic).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void overlap_k(float* __restrict__ vol, size_t n, int nf, int ns, uint32_t wrap_mask, float seed) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a = seed + (float)threadIdx.x;
    float b = a * 0.5f, c = a + 1.0f, d = a - 2.0f;
#pragma unroll 1
    for (int k = 0; k < nf; k += 4) {
        a = __builtin_fmaf(a, 0.999f, b); b = __builtin_fmaf(b, 0.998f, c); c = __builtin_fmaf(c, 0.997f, d); d = __builtin_fmaf(d, 0.996f, a);
    }
    a += b + c + d;
    if (wrap_mask) i &= wrap_mask;
    const size_t stride = wrap_mask ? 0 : n;
#pragma unroll 1
    for (int p = 0; p < ns; ++p) vol[p * stride + i] = a + (float)p;
}

extern "C" int overlap_probe(float* vol, int d, int nf, int ns, int wrap, void* stream) {
    const size_t n = (size_t)d * d * d;
    overlap_k<<<(unsigned)(n / 256), 256, 0, (hipStream_t)stream>>>(vol, n, nf, ns, wrap ? (1u << 18) - 1u : 0u, 1.0f);
    return (int)hipGetLastError();
}
