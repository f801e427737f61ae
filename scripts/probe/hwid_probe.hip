// Measurement probe: which (XCC, SE, CU, SIMD) a wave runs on -- HW_REG_HW_ID (4) and HW_REG_XCC_ID (20) of every wave of a launch, so that a kernel
// that runs ONE wave per SIMD (K6g: 512 registers) can index a SIMD-private stash in global memory.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void hwid_k(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}
extern "C" void hwid_probe(uint32_t* out, int blocks, int spin, void* stream) { hwid_k<<<blocks, 64, 0, (hipStream_t)stream>>>(out, spin); }
