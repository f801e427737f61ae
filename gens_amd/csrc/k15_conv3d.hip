// K15: 3 x 3 x 3 convolutions of the cost-volume U-Net (reference: nn.Conv3d / nn.ConvTranspose3d inside
// models/modules/reg_network.py:7-50,105-169; padding 1, stride 1 or 2, transposed: stride 2 with output_padding 1), forward, data
// gradient and weight gradient.  Few channels (8 ... 32) on up to 256^3 voxels: MIOpen's immediate-mode choice for the BACKWARD of
// these shapes takes 2.4 s per 256^3 layer on this machine (forward 9 ms), 6.2 s per training step for the whole U-Net against 41 ms
// for everything else -- scripts/probe/cnn_probe.py.
//
// One relation, three kernels.  A coarse tensor P (cp channels, X x Y x Z) and a fine tensor Q (cq channels, sX x sY x sZ, s = stride)
// are tied by W[cp][cq][3][3][3]:
//     gather    P[a][o]    = bias[a] + sum_{b, t} W[a][b][t] Q[b][s o + t - 1]        Conv3d forward;      ConvTranspose3d data gradient
//     scatter   Q[b][i]    = sum_{a, t : s | i + 1 - t} W[a][b][t] P[a][(i + 1 - t) / s]   Conv3d data gradient; ConvTranspose3d forward
//     wgrad     dW[a][b][t] = sum_o P[a][o] Q[b][s o + t - 1]                            both weight gradients
// (nn.Conv3d(in, out).weight is W with a = out, b = in; nn.ConvTranspose3d(in, out).weight is W with a = in, b = out.)  A stride-1
// scatter is a gather with the taps reversed and the channel roles swapped, which the host does on the 7-110 KB weight tensor.
//
// Mapping: a thread owns one voxel of P (lanes along z, the contiguous axis) and a block of OB output channels in registers; the tap
// offsets are computed once per voxel and reused for every input channel; taps that fall into the zero padding get an out-of-range
// buffer offset, for which the hardware returns 0 -- no masks in the inner loop, which is `loads + OB fused multiply-adds per load`
// with the weights as scalar operands (wave-uniform indices: s_load through the constant cache).  Arithmetic intensity is
// 27 cp cq MACs per voxel against 4 (cp + cq) bytes: 54 FLOP / B at 8 -> 8 channels, i.e. bound by the vector ALUs (78 TFLOP/s
// float32 FMA), not HBM: 58 GFLOP per 256^3 layer = 0.74 ms at that peak.
#include <stdlib.h>

#include "common.h"

#define CONV_OOB 0x7fffffffu     // beyond num_records of every buffer this file makes (< 2 GiB): the load returns 0

struct ConvGeom {
    int x, y, z;                 // extent of P
    int cp, cq;                  // channels of P, Q
    int cpp, cqp;                // the same, rounded up to the channel blocking of the weight tensor handed in
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const float* p, int64_t floats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(floats * 4), 0x00020000);
}
__device__ __forceinline__ float conv_load(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}

// gather: w laid out (cq, 27, cpp)
template <int OB, int S>
__global__ __launch_bounds__(256) void conv3d_gather_k(const float* __restrict__ q, const float* __restrict__ w, const float* __restrict__ bias,
                                                       ConvGeom g, float* __restrict__ p) {
    const int pn = g.x * g.y * g.z;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= pn) return;
    const int oz = v % g.z, t1 = v / g.z, oy = t1 % g.y, ox = t1 / g.y;
    const int qx = g.x * S, qy = g.y * S, qz = g.z * S;
    const uint32_t qn_bytes = (uint32_t)qx * qy * qz * 4u;
    uint32_t off[27];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const int ix = ox * S + dx - 1, iy = oy * S + dy - 1, iz = oz * S + dz - 1;
                const bool ok = (unsigned)ix < (unsigned)qx && (unsigned)iy < (unsigned)qy && (unsigned)iz < (unsigned)qz;
                off[(dx * 3 + dy) * 3 + dz] = ok ? (uint32_t)((ix * qy + iy) * qz + iz) * 4u : CONV_OOB;
            }
    const __amdgpu_buffer_rsrc_t qr = conv_rsrc(q, (int64_t)g.cq * qx * qy * qz);
    const int cb = blockIdx.y * OB;
    float acc[OB];
#pragma unroll
    for (int o = 0; o < OB; ++o) acc[o] = (bias != nullptr && cb + o < g.cp) ? bias[cb + o] : 0.0f;
    // Two register buffers: the 27 loads of input channel b + 1 are in flight while the 27 OB multiply-adds of channel b run (a single
    // buffer left every channel's load latency exposed: the loop body was "27 loads, wait, 108 packed FMAs").
    const float* wc = w + cb;
    uint32_t soff = 0;
    float xa[27], xb[27];
#define GATHER_LOAD(buf)                                                  \
    _Pragma("unroll") for (int t = 0; t < 27; ++t) buf[t] = conv_load(qr, off[t], soff); \
    soff += qn_bytes
#define GATHER_FMA(buf)                                                   \
    {                                                                     \
        float part[OB]; /* the 27 taps of one input channel in their own accumulator, then one add: round-off grows with 27 + cq, not 27 cq */ \
        _Pragma("unroll") for (int o = 0; o < OB; ++o) part[o] = 0.0f;    \
        _Pragma("unroll") for (int t = 0; t < 27; ++t)                    \
            _Pragma("unroll") for (int o = 0; o < OB; ++o) part[o] = __builtin_fmaf(wc[t * g.cpp + o], buf[t], part[o]); \
        _Pragma("unroll") for (int o = 0; o < OB; ++o) acc[o] += part[o]; \
    }                                                                     \
    wc += 27 * g.cpp
    GATHER_LOAD(xa);
    for (int b = 0; b < g.cq; b += 2) {
        if (b + 1 < g.cq) { GATHER_LOAD(xb); }
        GATHER_FMA(xa);
        if (b + 1 < g.cq) {
            if (b + 2 < g.cq) { GATHER_LOAD(xa); }
            GATHER_FMA(xb);
        }
    }
#undef GATHER_LOAD
#undef GATHER_FMA
#pragma unroll
    for (int o = 0; o < OB; ++o)
        if (cb + o < g.cp) p[(int64_t)(cb + o) * pn + v] = acc[o];
}

__device__ __forceinline__ void wave_lds_sync() {                 // orders the LDS accesses of ONE wave (wave-synchronous exchange: no s_barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// gather, stride 1, z a multiple of 64 (a wave's 64 voxels are consecutive in one z-row): per input channel a lane loads the CENTRE tap of
// its nine (dx, dy) rows and takes the z - 1 / z + 1 taps from its neighbours through a wave-private LDS strip; one more load brings the
// 18 halo values (lanes 0..8 left, 16..24 right).  10 loads per channel instead of 27: the direct kernel keeps the texture addresser
// busy 95 % of the time (PMC), this one is bound by the packed FMAs.
template <int OB>
__global__ __launch_bounds__(256) void conv3d_gather_rows_k(const float* __restrict__ q, const float* __restrict__ w, const float* __restrict__ bias,
                                                            ConvGeom g, float* __restrict__ p) {
    __shared__ float strip[4][9][66];                                             // [wave][row][halo + 64 + halo]
    const int pn = g.x * g.y * g.z;
    const int v = blockIdx.x * 256 + threadIdx.x;                                 // pn is a multiple of 64: whole waves are in or out
    if (v >= pn) return;
    const int lane = threadIdx.x & 63;
    float (*my)[66] = strip[threadIdx.x >> 6];
    const int oz = v % g.z, t1 = v / g.z, oy = t1 % g.y, ox = t1 / g.y;
    const uint32_t qn_bytes = (uint32_t)pn * 4u;
    uint32_t off[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int ix = ox + r / 3 - 1, iy = oy + r % 3 - 1;
        off[r] = ((unsigned)ix < (unsigned)g.x && (unsigned)iy < (unsigned)g.y) ? (uint32_t)((ix * g.y + iy) * g.z + oz) * 4u : CONV_OOB;
    }
    uint32_t hoff = CONV_OOB;                                                     // lane r: left halo of row r; lane 16 + r: right halo
    {
        const int r = lane & 15, hz = lane < 16 ? oz - lane - 1 : oz - lane + 64;
        const int ix = ox + r / 3 - 1, iy = oy + r % 3 - 1;
        if (lane < 32 && r < 9 && (unsigned)ix < (unsigned)g.x && (unsigned)iy < (unsigned)g.y && (unsigned)hz < (unsigned)g.z)
            hoff = (uint32_t)((ix * g.y + iy) * g.z + hz) * 4u;
    }
    const __amdgpu_buffer_rsrc_t qr = conv_rsrc(q, (int64_t)g.cq * pn);
    const int cb = blockIdx.y * OB;
    float acc[OB];
#pragma unroll
    for (int o = 0; o < OB; ++o) acc[o] = (bias != nullptr && cb + o < g.cp) ? bias[cb + o] : 0.0f;
    const float* wc = w + cb;
    uint32_t soff = 0;
    float ca[9], cn[9], ha, hn;
#pragma unroll
    for (int r = 0; r < 9; ++r) ca[r] = conv_load(qr, off[r], soff);
    ha = conv_load(qr, hoff, soff);
    for (int b = 0; b < g.cq; ++b) {
        soff += qn_bytes;
        if (b + 1 < g.cq) {                                                       // next channel's loads fly during this channel's arithmetic
#pragma unroll
            for (int r = 0; r < 9; ++r) cn[r] = conv_load(qr, off[r], soff);
            hn = conv_load(qr, hoff, soff);
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) my[r][1 + lane] = ca[r];
        if (lane < 32 && (lane & 15) < 9) my[lane & 15][lane < 16 ? 0 : 65] = ha;
        wave_lds_sync();
        float part[OB];                                                           // one input channel's 27 taps, then one add (see conv3d_gather_k)
#pragma unroll
        for (int o = 0; o < OB; ++o) part[o] = 0.0f;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const float xv[3] = {my[r][lane], ca[r], my[r][lane + 2]};
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                for (int o = 0; o < OB; ++o) part[o] = __builtin_fmaf(wc[(r * 3 + dz) * g.cpp + o], xv[dz], part[o]);
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) acc[o] += part[o];
        wave_lds_sync();
        wc += 27 * g.cpp;
#pragma unroll
        for (int r = 0; r < 9; ++r) ca[r] = cn[r];
        ha = hn;
    }
#pragma unroll
    for (int o = 0; o < OB; ++o)
        if (cb + o < g.cp) p[(int64_t)(cb + o) * pn + v] = acc[o];
}

// gather, stride 2, z a multiple of 64: the taps of coarse voxel oz are Q[2 oz - 1], Q[2 oz], Q[2 oz + 1]: one aligned 8-byte load per
// (dx, dy) row brings the last two, the first is the neighbouring lane's second value (LDS strip; one load for the nine left halos) --
// 10 loads per input channel instead of 27 loads at an 8-byte lane stride.
template <int OB>
__global__ __launch_bounds__(256) void conv3d_gather_rows2_k(const float* __restrict__ q, const float* __restrict__ w, const float* __restrict__ bias,
                                                             ConvGeom g, float* __restrict__ p) {
    __shared__ float strip[4][9][65];                                             // [wave][row][left halo + 64]
    const int pn = g.x * g.y * g.z;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= pn) return;
    const int lane = threadIdx.x & 63;
    float (*my)[65] = strip[threadIdx.x >> 6];
    const int oz = v % g.z, t1 = v / g.z, oy = t1 % g.y, ox = t1 / g.y;
    const int qx = 2 * g.x, qy = 2 * g.y, qz = 2 * g.z;
    const uint32_t qn_bytes = (uint32_t)pn * 32u;
    // (plain 8-byte global loads for the pairs: this compiler lowers __builtin_amdgcn_raw_buffer_load_b64 to a ONE-dword load; a row is
    // valid or not for the whole wave, and inside a valid row the pair is always in range)
    int off[9];                                                                   // element offset of Q[row r][2 oz] in a channel plane, -1: padding row
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int ix = 2 * ox + r / 3 - 1, iy = 2 * oy + r % 3 - 1;
        off[r] = ((unsigned)ix < (unsigned)qx && (unsigned)iy < (unsigned)qy) ? (ix * qy + iy) * qz + 2 * oz : -1;
    }
    uint32_t hoff = CONV_OOB;                                                     // lane r < 9: Q[row r][2 oz0 - 1], oz0 = the wave's first voxel
    if (lane < 9) {
        const int ix = 2 * ox + lane / 3 - 1, iy = 2 * oy + lane % 3 - 1, hz = 2 * (oz - lane) - 1;
        if ((unsigned)ix < (unsigned)qx && (unsigned)iy < (unsigned)qy && hz >= 0) hoff = (uint32_t)((ix * qy + iy) * qz + hz) * 4u;
    }
    const __amdgpu_buffer_rsrc_t qr = conv_rsrc(q, (int64_t)g.cq * pn * 8);
    const int cb = blockIdx.y * OB;
    float acc[OB];
#pragma unroll
    for (int o = 0; o < OB; ++o) acc[o] = (bias != nullptr && cb + o < g.cp) ? bias[cb + o] : 0.0f;
    const float* wc = w + cb;
    uint32_t soff = 0;
    const float* qc = q;
    for (int b = 0; b < g.cq; ++b) {
        float2 c[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) c[r] = off[r] >= 0 ? *(const float2*)(qc + off[r]) : make_float2(0.0f, 0.0f);
        const float h = conv_load(qr, hoff, soff);
#pragma unroll
        for (int r = 0; r < 9; ++r) my[r][1 + lane] = c[r].y;
        if (lane < 9) my[lane][0] = h;
        wave_lds_sync();
        float part[OB];
#pragma unroll
        for (int o = 0; o < OB; ++o) part[o] = 0.0f;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const float xv[3] = {my[r][lane], c[r].x, c[r].y};
#pragma unroll
            for (int dz = 0; dz < 3; ++dz)
#pragma unroll
                for (int o = 0; o < OB; ++o) part[o] = __builtin_fmaf(wc[(r * 3 + dz) * g.cpp + o], xv[dz], part[o]);
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) acc[o] += part[o];
        wave_lds_sync();
        wc += 27 * g.cpp;
        soff += qn_bytes;
        qc += (int64_t)pn * 8;
    }
#pragma unroll
    for (int o = 0; o < OB; ++o)
        if (cb + o < g.cp) p[(int64_t)(cb + o) * pn + v] = acc[o];
}

// scatter, stride 2: w laid out (cp, 27, cqp).  The thread of coarse voxel o produces the 2 x 2 x 2 fine voxels 2 o + e; along each axis
// an even fine index takes tap 1 from P[o], an odd one tap 2 from P[o] and tap 0 from P[o + 1].
template <int OB>
__global__ __launch_bounds__(256) void conv3d_scatter2_k(const float* __restrict__ p, const float* __restrict__ w, ConvGeom g, float* __restrict__ q) {
    const int pn = g.x * g.y * g.z;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= pn) return;
    const int oz = v % g.z, t1 = v / g.z, oy = t1 % g.y, ox = t1 / g.y;
    uint32_t off[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int ix = ox + (n >> 2), iy = oy + ((n >> 1) & 1), iz = oz + (n & 1);
        off[n] = (ix < g.x && iy < g.y && iz < g.z) ? (uint32_t)((ix * g.y + iy) * g.z + iz) * 4u : CONV_OOB;
    }
    const __amdgpu_buffer_rsrc_t pr = conv_rsrc(p, (int64_t)g.cp * pn);
    const int cb = blockIdx.y * OB;
    float acc[8][OB];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int o = 0; o < OB; ++o) acc[e][o] = 0.0f;
    const float* wc = w + cb;
    uint32_t soff = 0;
    for (int a = 0; a < g.cp; ++a) {
        float pv[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) pv[n] = conv_load(pr, off[n], soff);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tz = 0; tz < 3; ++tz) {
                    // tap 1 -> (even output, source o); tap 2 -> (odd, o); tap 0 -> (odd, o + 1)
                    const int ex = tx != 1, ey = ty != 1, ez = tz != 1;
                    const int sx = tx == 0, sy = ty == 0, sz = tz == 0;
                    const float src = pv[(sx << 2) | (sy << 1) | sz];
                    const float* wt = wc + ((tx * 3 + ty) * 3 + tz) * g.cqp;
#pragma unroll
                    for (int o = 0; o < OB; ++o) acc[(ex << 2) | (ey << 1) | ez][o] = __builtin_fmaf(wt[o], src, acc[(ex << 2) | (ey << 1) | ez][o]);
                }
        wc += 27 * g.cqp;
        soff += (uint32_t)pn * 4u;
    }
    const int qy = 2 * g.y, qz = 2 * g.z;
    const int64_t qn = (int64_t)pn * 8;
#pragma unroll
    for (int o = 0; o < OB; ++o) {
        if (cb + o >= g.cq) break;
        float* qc = q + (int64_t)(cb + o) * qn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ix = 2 * ox + (e >> 1), iy = 2 * oy + (e & 1);
            *(float2*)(qc + ((int64_t)ix * qy + iy) * qz + 2 * oz) = make_float2(acc[2 * e][o], acc[2 * e + 1][o]);
        }
    }
}

// wgrad: one (dx, dy) tap row (three taps along z), PB = 8 (or 4) channels of P and eight of Q per workgroup column; every wave leaves its
// 3 PB QB partial sums in ws (parts, cpp, cqp, 27); the host adds the parts (a fixed order: no atomics).
// ROWLDS (stride 1, z a multiple of 64: a wave's 64 voxels are consecutive in ONE z-row): a lane loads only the centre tap of its row and
// gets the z - 1 / z + 1 taps from its neighbours through a wave-private LDS strip (plus one load for the two halo values of all eight
// channels): 13 loads per trip instead of 28 -- the kernel was bound by the texture addresser (TA busy 92 %, PMC).

template <int S, int PB, bool ROWLDS>
__global__ __launch_bounds__(256) void conv3d_wgrad_k(const float* __restrict__ p, const float* __restrict__ q, ConvGeom g, int n_ranges, int chunks_per_range,
                                                      float* __restrict__ ws) {
    constexpr int QB = 8;
    const int pn = g.x * g.y * g.z;
    const int qx = g.x * S, qy = g.y * S, qz = g.z * S;
    const uint32_t qn_bytes = (uint32_t)qx * qy * qz * 4u;
    // Work item = (range of consecutive voxel chunks, column) with column = (tap row, Q block, P block).  All columns of a range
    // re-read the same planes of P and Q: they are given consecutive slots on ONE XCD (workgroup id % 8), so that the range comes
    // from HBM once and from that XCD's L2 for the other columns (with the columns spread over the launch, each streamed the
    // planes from HBM again: 14.5 GB per 8 -> 8 layer at 256^3, 5 TB/s -- HBM-bound).
    const int nqb = g.cqp / QB, ny = 9 * nqb * (g.cpp / PB);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int col = slot % ny, range = xcd + 8 * (slot / ny);
    if (range >= n_ranges) return;
    const int row = col % 9, qb = (col / 9) % nqb * QB, pb = col / (9 * nqb) * PB;
    const int dx = row / 3, dy = row % 3;
    const __amdgpu_buffer_rsrc_t qr = conv_rsrc(q, (int64_t)g.cq * qx * qy * qz);
    const __amdgpu_buffer_rsrc_t pr = conv_rsrc(p, (int64_t)g.cp * pn);
    float acc[PB][QB][3];
#pragma unroll
    for (int a = 0; a < PB; ++a)
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[a][b][t] = 0.0f;
    const int v_end = min(pn, (range + 1) * chunks_per_range * 256);
    const int v0 = range * chunks_per_range * 256 + threadIdx.x;
    int oz = v0 % g.z, oy = (v0 / g.z) % g.y, ox = v0 / g.z / g.y;                // decoded once, then advanced by 256 voxels per trip
    for (int v = v0; v < v_end; v += 256) {                                       // (four divisions per trip were 40 % of the instructions)
        const int ix = ox * S + dx - 1, iy = oy * S + dy - 1, iz = oz * S - 1;
        const bool okr = (unsigned)ix < (unsigned)qx && (unsigned)iy < (unsigned)qy;
        const uint32_t base = (uint32_t)((ix * qy + iy) * qz + iz) * 4u;
        uint32_t off[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) off[t] = (okr && (unsigned)(iz + t) < (unsigned)qz) ? base + 4u * t : CONV_OOB;
        float pv[PB];
#pragma unroll
        for (int a = 0; a < PB; ++a) pv[a] = pb + a < g.cp ? conv_load(pr, (uint32_t)v * 4u, (uint32_t)(pb + a) * (uint32_t)pn * 4u) : 0.0f;
        if constexpr (ROWLDS) {
            __shared__ float strip[4][QB][66];                                   // [wave][channel][halo + 64 + halo]
            const int lane = threadIdx.x & 63;
            float (*my)[66] = strip[threadIdx.x >> 6];
            float c[QB];
#pragma unroll
            for (int b = 0; b < QB; ++b) c[b] = qb + b < g.cq ? conv_load(qr, off[1], (uint32_t)(qb + b) * qn_bytes) : 0.0f;
            // lanes 0..7: the value left of the wave's segment for channel `lane`; lanes 8..15: the value right of it
            uint32_t hoff = CONV_OOB;
            if (lane < 2 * QB) {
                const int j = lane & (QB - 1), hz = lane < QB ? oz - lane - 1 : oz - lane + 64;
                if (okr && qb + j < g.cq && (unsigned)hz < (unsigned)qz) hoff = (uint32_t)((ix * qy + iy) * qz + hz) * 4u + (uint32_t)(qb + j) * qn_bytes;
            }
            const float h = conv_load(qr, hoff, 0u);
#pragma unroll
            for (int b = 0; b < QB; ++b) my[b][1 + lane] = c[b];
            if (lane < 2 * QB) my[lane & (QB - 1)][lane < QB ? 0 : 65] = h;
            wave_lds_sync();
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                const float qv[3] = {my[b][lane], c[b], my[b][lane + 2]};
#pragma unroll
                for (int a = 0; a < PB; ++a)
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[a][b][t] = __builtin_fmaf(pv[a], qv[t], acc[a][b][t]);
            }
            wave_lds_sync();
        } else {
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                if (qb + b >= g.cq) break;                                       // (uniform) padding channels of the last block
                const uint32_t soff = (uint32_t)(qb + b) * qn_bytes;
                float qv[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) qv[t] = conv_load(qr, off[t], soff);
#pragma unroll
                for (int a = 0; a < PB; ++a)
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[a][b][t] = __builtin_fmaf(pv[a], qv[t], acc[a][b][t]);
            }
        }
        oz += 256;
        while (oz >= g.z) {
            oz -= g.z;
            if (++oy == g.y) { oy = 0; ++ox; }
        }
    }
    const int lane = threadIdx.x & 63, part = range * 4 + (threadIdx.x >> 6);
#pragma unroll
    for (int a = 0; a < PB; ++a)
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float s = acc[a][b][t];
                GENS_DPP_SCAN(s, 0.0f, op_add_);                                 // lane 63 holds the wave's total
                if (lane == 63) ws[(((int64_t)part * g.cpp + pb + a) * g.cqp + qb + b) * 27 + row * 3 + t] = s;
            }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// wgrad on the matrix cores (stride 1, z a multiple of 64): dW[a][b][tx][ty][tz] = sum_o P[a][o] Q[b][o + t - 1] as three 32 x 32 products
// per (x, y) row -- one per tx -- whose reduction index runs along z:
//     D_tx[(a, tz)][(b, ty)] += sum_z  P[a][x][y][z - tz + 1]  *  Q[b][x + tx - 1][y + ty - 1][z]
// (24 of 32 rows and columns carry eight channels x three taps: 56 % of the MFMA's products are useful, against the 23 - 30 TFLOP/s the
// vector-ALU kernel above reaches on a quarter of the float32 peak).  A workgroup = four waves = four consecutive y of one 64-voxel z segment,
// marching along x: the three x-planes of Q it needs (six y rows x eight channels x 64 z) live in LDS as a ring -- one new plane per step, 12 KB --
// beside the step's four P rows with their z halo; both are loaded as 16-byte buffer reads one step AHEAD of their use (out-of-volume rows are
// out-of-range offsets: the hardware returns the zero padding).  MFMA operands come from LDS one float per lane and step (row pitches 65 / 67
// floats: conflict-free).  Every (P voxel, Q voxel) pair is counted once: the z sum is partitioned by the Q position (the P halo supplies the
// neighbouring segment's voxels), x and y by the P position.  A workgroup leaves its 8 x 8 x 27 sums in ws[part]; the host adds the parts.
// ---------------------------------------------------------------------------------------------------------------------------------------
#define WM_QP 65
#define WM_PP 67
typedef float wm_f32x16 __attribute__((ext_vector_type(16)));
typedef float wm_f32x4 __attribute__((ext_vector_type(4)));

// PAIR (four channels of P or fewer -- the U-Net's output heads, whose (a, tz) rows would fill 12 of the MFMA's 32): a wave takes TWO consecutive y
// rows of P, rows i = (a, y row, tz) = 24, columns j = (b, fine row fy = y row + ty) = 32; both y rows' products land in the same weight (summed on the
// way out) and a quarter of the columns (fy - y row outside 0..2) is no tap: 56 % of the products useful instead of 28 %, half the MFMAs per voxel.
template <bool PAIR>
__global__ __launch_bounds__(256) void conv3d_wgrad_mfma_k(const float* __restrict__ p, const float* __restrict__ q, ConvGeom g, int y_blocks, int z_segs,
                                                           int xc, float* __restrict__ ws) {
    constexpr int NYQ = PAIR ? 10 : 6, RY = PAIR ? 8 : 4, PCH = PAIR ? 4 : 8, NQ = NYQ * 8 * 16 / 256;
    extern __shared__ __attribute__((aligned(16))) float wm_lds[];
    float (*Ql)[NYQ][8][WM_QP] = (float (*)[NYQ][8][WM_QP])wm_lds;                        // [3][NYQ][8][WM_QP]
    float (*Pl)[WM_PP] = (float (*)[WM_PP])(wm_lds + 3 * NYQ * 8 * WM_QP);                // [32 = RY x PCH][WM_PP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqb = g.cqp / 8, n_col = ((g.cpp + PCH - 1) / PCH) * nqb;
    const int part = (int)blockIdx.x / n_col, col = (int)blockIdx.x % n_col;
    const int pb = (col / nqb) * PCH, qb = (col % nqb) * 8;
    const int zs = part % z_segs, yb = (part / z_segs) % y_blocks, xk = part / (z_segs * y_blocks);
    const int z0 = zs * 64, y0 = yb * RY, x0 = xk * xc, x1 = min(x0 + xc, g.x);
    const uint32_t pn_bytes = (uint32_t)g.x * g.y * g.z * 4u;
    const __amdgpu_buffer_rsrc_t qr = conv_rsrc(q, (int64_t)g.cq * g.x * g.y * g.z);
    const __amdgpu_buffer_rsrc_t pr = conv_rsrc(p, (int64_t)g.cp * g.x * g.y * g.z);

    // ---- this thread's share of a step's loads: NQ float4 of the Q plane, 2 float4 of the P rows, (64 threads) one halo float
    int q_row[NQ], q_f4[NQ], p_row[2], p_f4[2];
    uint32_t q_soff[NQ], p_soff[2];
    bool q_ok[NQ], p_ok[2];
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
        const int idx = tid + 256 * it;
        q_row[it] = idx >> 4;                                                       // yrow * 8 + b
        q_f4[it] = idx & 15;
        const int yy = y0 - 1 + (q_row[it] >> 3), b = q_row[it] & 7;
        q_ok[it] = (unsigned)yy < (unsigned)g.y && qb + b < g.cq;
        q_soff[it] = (uint32_t)(qb + b) * pn_bytes;
    }
    auto p_y = [&](int row) { return PAIR ? row >> 2 : row >> 3; };                 // flat P row (0..31) -> y row of the workgroup, channel of the block
    auto p_a = [&](int row) { return PAIR ? row & 3 : row & 7; };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + 256 * it;
        p_row[it] = idx >> 4;
        p_f4[it] = idx & 15;
        const int yy = y0 + p_y(p_row[it]), a = p_a(p_row[it]);
        p_ok[it] = yy < g.y && pb + a < g.cp;
        p_soff[it] = (uint32_t)(pb + a) * pn_bytes;
    }
    const int h_row = (tid >> 1) & 31, h_side = tid & 1;                           // (threads 0..63) halo of P row h_row: z0 - 1 or z0 + 64
    const int h_z = h_side ? z0 + 64 : z0 - 1;
    const bool h_ok = tid < 64 && y0 + p_y(h_row) < g.y && pb + p_a(h_row) < g.cp && (unsigned)h_z < (unsigned)g.z;
    const uint32_t h_soff = (uint32_t)(pb + p_a(h_row)) * pn_bytes;

    wm_f32x4 qv[NQ], pv[2];
    float hv = 0.0f;
    auto load_q = [&](int xq) {
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int yy = y0 - 1 + (q_row[it] >> 3);
            // (the channel's offset in the LANE offset: as the load's scalar operand it differs across the lanes -> a readfirstlane loop per load)
            const uint32_t off = (q_ok[it] && (unsigned)xq < (unsigned)g.x) ? (uint32_t)(((xq * g.y + yy) * g.z + z0 + 4 * q_f4[it]) * 4) + q_soff[it] : CONV_OOB;
            qv[it] = __builtin_bit_cast(wm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(qr, off, 0, 0));
        }
    };
    auto store_q = [&](int xq) {
        const int slot = (xq + 3) % 3;
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            float* dst = &Ql[slot][q_row[it] >> 3][q_row[it] & 7][4 * q_f4[it]];
            dst[0] = qv[it].x; dst[1] = qv[it].y; dst[2] = qv[it].z; dst[3] = qv[it].w;
        }
    };
    auto load_p = [&](int xp) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int yy = y0 + p_y(p_row[it]);
            const uint32_t off = p_ok[it] ? (uint32_t)(((xp * g.y + yy) * g.z + z0 + 4 * p_f4[it]) * 4) + p_soff[it] : CONV_OOB;
            pv[it] = __builtin_bit_cast(wm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, off, 0, 0));
        }
        const uint32_t hoff = h_ok ? (uint32_t)(((xp * g.y + y0 + p_y(h_row)) * g.z + h_z) * 4) + h_soff : CONV_OOB;
        hv = conv_load(pr, hoff, 0);
    };
    auto store_p = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float* dst = &Pl[p_row[it]][1 + 4 * p_f4[it]];
            dst[0] = pv[it].x; dst[1] = pv[it].y; dst[2] = pv[it].z; dst[3] = pv[it].w;
        }
        if (tid < 64) Pl[h_row][h_side ? 65 : 0] = hv;
    };

    // ---- MFMA operand rows of this lane.  Plain: A row i = (a, tz), B column j = (b, ty); lanes 24..31 of a half repeat rows 0..7 (their products are
    // dropped).  PAIR: A row i = (a, y row, tz) (24 used), B column j = (b, fy) (all 32)
    const int i32 = lane & 31, kk = lane >> 5;
    const int ii = i32 < 24 ? i32 : i32 - 24;
    const int oa = ii / 3, ot = ii - 3 * oa;                                        // plain: a (or b), tz (or ty)
    const int pa = ii / 6, pyr = (ii - 6 * pa) / 3;                                 // PAIR: a, y row; tz = ot
    const float* a_ptr = PAIR ? &Pl[(2 * wave + pyr) * 4 + pa][2 - ot + 32 * kk] : &Pl[wave * 8 + oa][2 - ot + 32 * kk];
    wm_f32x16 acc[3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tx][r] = 0.0f;

    load_q(x0 - 1);
    store_q(x0 - 1);
    load_q(x0);
    store_q(x0);
    load_q(x0 + 1);
    load_p(x0);
    for (int x = x0; x < x1; ++x) {
        store_q(x + 1);
        store_p();
        __syncthreads();
        if (x + 1 < x1) {                                                           // the next step's tensors are on their way while this one multiplies
            load_q(x + 2);
            load_p(x + 1);
        }
        const float* b_ptr[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
            b_ptr[tx] = PAIR ? &Ql[(x - 1 + tx + 3) % 3][2 * wave + (i32 & 3)][i32 >> 2][32 * kk] : &Ql[(x - 1 + tx + 3) % 3][wave + ot][oa][32 * kk];
#pragma unroll 8
        for (int t = 0; t < 32; ++t) {
            const float av = a_ptr[t];
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) acc[tx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_ptr[tx][t], acc[tx], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- the four waves' sums, then the workgroup's block of ws[part]
    float* R = &Ql[0][0][0][0];                                                     // 3 x 16 x 64 floats
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* dst = R + (tx * 16 + r) * 64 + lane;
                    *dst = w == 0 ? acc[tx][r] : *dst + acc[tx][r];
                }
        }
        __syncthreads();
    }
    if (PAIR) {
        // accumulator element of row i, column j: register r = (i & 3) + 4 (i >> 3), lane 32 ((i >> 2) & 1) + j
        auto at = [&](int tx, int i, int j) { return R[(tx * 16 + (i & 3) + 4 * (i >> 3)) * 64 + 32 * ((i >> 2) & 1) + j]; };
        for (int e = tid; e < 3 * 4 * 8 * 9; e += 256) {                            // (tx, a, b, ty, tz)
            const int tz = e % 3, ty = (e / 3) % 3, b = (e / 9) % 8, a = (e / 72) % 4, tx = e / 288;
            const float v = at(tx, a * 6 + tz, b * 4 + ty) + at(tx, a * 6 + 3 + tz, b * 4 + ty + 1);      // y row 0 (fy = ty) + y row 1 (fy = ty + 1)
            if (pb + a < g.cpp && qb + b < g.cqp) ws[(((int64_t)part * g.cpp + pb + a) * g.cqp + qb + b) * 27 + (tx * 3 + ty) * 3 + tz] = v;
        }
        return;
    }
    for (int e = tid; e < 3 * 16 * 64; e += 256) {
        const int tx = e >> 10, r = (e >> 6) & 15, l = e & 63;
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;            // row / column of accumulator register r in lane l
        if (i >= 24 || j >= 24) continue;
        const int a = i / 3, tz = i - 3 * a, b = j / 3, ty = j - 3 * b;
        if (pb + a < g.cpp && qb + b < g.cqp) ws[(((int64_t)part * g.cpp + pb + a) * g.cqp + qb + b) * 27 + (tx * 3 + ty) * 3 + tz] = R[e];
    }
}
#define WM_LDS_BYTES(PAIR) ((3 * ((PAIR) ? 10 : 6) * 8 * WM_QP + 32 * WM_PP) * 4)

// ---------------------------------------------------------------------------------------------------------------------------------------
// wgrad on the matrix cores, STRIDE 2 (coarse z a multiple of 64): dW[a][b][tx][ty][tz] = sum_o P[a][o] Q[b][2 o + t - 1].  Along z the three taps
// read the EVEN fine voxels (tz = 1: Q[2 o]) and the ODD ones (tz = 2: Q[2 o + 1]; tz = 0: Q[2 o - 1] = the odd voxel of o - 1), so with the fine
// rows stored de-interleaved in LDS -- [odd | even] per row and z segment -- the reduction index m runs over the segment's coarse z with unit stride:
//     D_e[(a, s)][(b, par)] += sum_m  P[a][m + s]  *  Q_par[b][2 x + tx - 1][2 y + ty - 1][m]
//         (s, par) = (0, even) -> tz = 1,  (0, odd) -> tz = 2,  (1, odd) -> tz = 0,  (1, even) -> not a tap (a quarter of the products)
// Rows i = (a, s): SIXTEEN channels of P x {P, P shifted by one}; columns j = (b, par, u): eight channels of Q x {even, odd} x two of the nine
// (tx, ty) combinations: five products per reduction step cover the nine (the tenth slot repeats the ninth and is dropped) -- 67 % of the
// MFMA's products are taps, against the 29 TFLOP/s of the vector-ALU kernel.  The z sum is partitioned by the Q position (a segment owns fine z
// [2 z0, 2 z0 + 2 ZC); the P halo P[z0 + ZC] supplies the neighbouring segment's voxel of the tz = 0 tap), x and y by the P position.
// A workgroup = four waves = four consecutive coarse y of one z segment, marching along coarse x: the three fine x planes of a step (nine fine y
// rows x eight channels x 2 ZC z) live in LDS as a ring by (fine x) mod 3 -- a step brings TWO new planes, loaded as 16-byte buffer reads into
// registers while the previous step multiplies (out-of-volume rows: out-of-range offsets, the hardware returns the zero padding).
// Measured (scripts/probe/wgrad2_ab.py, 16 x 8 channels at coarse 128^3, the U-Net's first down / last up convolution): vector-ALU kernel 0.99 ms;
// ZC = 64, one workgroup per CU 0.44 ms; the loads' channel offset moved from the scalar operand (a readfirstlane loop per load!) into the lane
// offset 0.375; ZC = 32, two workgroups per CU 0.29 ms = 50 TFLOP/s of taps.
// ---------------------------------------------------------------------------------------------------------------------------------------
// ZC = coarse z per segment: 64, or 32 -- 75 KB of LDS instead of 130: TWO workgroups per CU, so that one's loads, LDS stores and barriers run under
// the other's products (with one workgroup a step took 26 k cycles for 10 k cycles of MFMA issue).
// LDS row of Q: [odd ZC][1][even ZC] -- the even half ZC + 1 floats on, so that a column's two parities sit in neighbouring banks (64 on = the SAME
// bank: a two-way conflict on every B read); pitch 2 ZC + 3 = 3 mod 64 across the rows a wave reads.
#define W2_PP 67            // P row: ZC + the halo (67 = 3 mod 64 for either ZC)
template <int ZC>
__global__ __launch_bounds__(256, ZC == 32 ? 2 : 1) void conv3d_wgrad2_mfma_k(const float* __restrict__ p, const float* __restrict__ q, ConvGeom g, int y_blocks, int z_segs,
                                                            int xc, float* __restrict__ ws) {
    constexpr int W2_QP = 2 * ZC + 3, NF = ZC / 2, NQ = (72 * NF + 255) / 256, NPF = ZC / 4, NP = 64 * NPF / 256;
    extern __shared__ __attribute__((aligned(16))) float w2_lds[];
    float* Ql = w2_lds;                                   // [3][9][8][W2_QP]
    float* Pl = w2_lds + 3 * 9 * 8 * W2_QP;               // [4][16][W2_PP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqb = g.cqp / 8, n_col = ((g.cpp + 15) / 16) * nqb;
    const int part = (int)blockIdx.x / n_col, col = (int)blockIdx.x % n_col;
    const int pb = (col / nqb) * 16, qb = (col % nqb) * 8;
    const int zs = part % z_segs, yb = (part / z_segs) % y_blocks, xk = part / (z_segs * y_blocks);
    const int z0 = zs * ZC, y0 = yb * 4, x0 = xk * xc, x1 = min(x0 + xc, g.x);
    const int qx = 2 * g.x, qy = 2 * g.y, qz = 2 * g.z;
    const uint32_t pn_bytes = (uint32_t)g.x * g.y * g.z * 4u, qn_bytes = 8u * pn_bytes;
    const __amdgpu_buffer_rsrc_t qr = conv_rsrc(q, (int64_t)g.cq * qx * qy * qz);
    const __amdgpu_buffer_rsrc_t pr = conv_rsrc(p, (int64_t)g.cp * g.x * g.y * g.z);

    // ---- this thread's share of a plane of Q (72 rows x NF float4) and of a step's P rows (64 rows x NPF float4, + a halo float)
    uint32_t q_off[NQ];                                    // byte offset inside a fine x plane, or out of range
    uint32_t q_soff[NQ];
    int q_dst[NQ];                                         // LDS float index inside a ring slot: row * W2_QP + 2 * f4   (odd part; even part 64 further)
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
        const int idx = tid + 256 * it, row = idx / NF, f4 = idx % NF;          // row = fy row * 8 + b
        const int fy = 2 * y0 - 1 + (row >> 3), b = row & 7;
        const bool ok = idx < 72 * NF && (unsigned)fy < (unsigned)qy && qb + b < g.cq;
        q_off[it] = ok ? (uint32_t)((fy * qz + 2 * z0 + 4 * f4) * 4) : CONV_OOB;
        q_soff[it] = (uint32_t)(qb + b) * qn_bytes;
        q_dst[it] = idx < 72 * NF ? row * W2_QP + 2 * f4 : -1;
    }
    uint32_t p_off[NP], p_soff[NP];
    int p_dst[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int idx = tid + 256 * it, row = idx / NPF, f4 = idx % NPF;        // row = wave' * 16 + a
        const int yy = y0 + (row >> 4), a = row & 15;
        const bool ok = yy < g.y && pb + a < g.cp;
        p_off[it] = ok ? (uint32_t)((yy * g.z + z0 + 4 * f4) * 4) : CONV_OOB;
        p_soff[it] = (uint32_t)(pb + a) * pn_bytes;
        p_dst[it] = row * W2_PP + 4 * f4;
    }
    const int h_row = tid & 63;                                                  // (threads 0..63) the halo P[z0 + ZC] of row h_row
    const bool h_ok = tid < 64 && y0 + (h_row >> 4) < g.y && pb + (h_row & 15) < g.cp && z0 + ZC < g.z;
    const uint32_t h_off = h_ok ? (uint32_t)(((y0 + (h_row >> 4)) * g.z + z0 + ZC) * 4) : CONV_OOB;
    const uint32_t h_soff = (uint32_t)(pb + (h_row & 15)) * pn_bytes;
    const uint32_t q_plane = (uint32_t)qy * qz * 4u, p_plane = (uint32_t)g.y * g.z * 4u;

    wm_f32x4 qv[2][NQ], pv[NP];
    float hv = 0.0f;
    auto load_q = [&](int fx, wm_f32x4 (&v)[NQ]) {
        const bool in = (unsigned)fx < (unsigned)qx;
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            // (the channel's offset rides in the LANE offset: as the scalar operand of the load it differs across the lanes and the compiler wraps
            // every load in a readfirstlane loop over the wave's eight channels)
            const uint32_t off = (in && q_off[it] != CONV_OOB) ? (uint32_t)fx * q_plane + q_off[it] + q_soff[it] : CONV_OOB;
            v[it] = __builtin_bit_cast(wm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(qr, off, 0, 0));
        }
    };
    auto store_q = [&](int fx, const wm_f32x4 (&v)[NQ]) {
        float* slot = Ql + ((fx + 3) % 3) * (9 * 8 * W2_QP);
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            if (NQ * 256 != 72 * NF && q_dst[it] < 0) continue;
            float* dst = slot + q_dst[it];                                       // fine z 4 f4 .. 4 f4 + 3 = coarse m = 2 f4, 2 f4 + 1: (even, odd) each
            dst[ZC + 1] = v[it].x; dst[0] = v[it].y; dst[ZC + 2] = v[it].z; dst[1] = v[it].w;
        }
    };
    auto load_p = [&](int xp) {
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const uint32_t off = p_off[it] != CONV_OOB ? (uint32_t)xp * p_plane + p_off[it] + p_soff[it] : CONV_OOB;
            pv[it] = __builtin_bit_cast(wm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, off, 0, 0));
        }
        hv = conv_load(pr, h_ok ? (uint32_t)xp * p_plane + h_off + h_soff : CONV_OOB, 0);
    };
    auto store_p = [&]() {
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            float* dst = Pl + p_dst[it];
            dst[0] = pv[it].x; dst[1] = pv[it].y; dst[2] = pv[it].z; dst[3] = pv[it].w;
        }
        if (tid < 64) Pl[h_row * W2_PP + ZC] = hv;
    };

    // ---- MFMA operands of this lane: A row i = (a, s), B column j = (b, par, u)
    const int i32 = lane & 31, kk = lane >> 5;
    const float* a_ptr = Pl + (wave * 16 + (i32 >> 1)) * W2_PP + (i32 & 1) + (ZC / 2) * kk;
    const int ob = i32 & 7, opar = (i32 >> 3) & 1, ou = i32 >> 4;
    int b_tx[5], b_row[5];                                                       // per product: the ring plane's tx and the LDS row offset inside a slot
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int e = min(2 * c + ou, 8), tx = e / 3, ty = e - 3 * tx;
        b_tx[c] = tx;
        b_row[c] = ((2 * wave + ty) * 8 + ob) * W2_QP + (opar ? 0 : ZC + 1) + (ZC / 2) * kk;
    }
    wm_f32x16 acc[5];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

    // planes 2 x0 - 1, 2 x0, 2 x0 + 1 and the P rows of x0
    load_q(2 * x0 - 1, qv[0]);
    store_q(2 * x0 - 1, qv[0]);
    load_q(2 * x0, qv[0]);
    load_q(2 * x0 + 1, qv[1]);
    load_p(x0);
    for (int x = x0; x < x1; ++x) {
        store_q(2 * x, qv[0]);
        store_q(2 * x + 1, qv[1]);
        store_p();
        __syncthreads();
        if (x + 1 < x1) {                                                        // the next step's two planes and rows fly while this one multiplies
            load_q(2 * x + 2, qv[0]);
            load_q(2 * x + 3, qv[1]);
            load_p(x + 1);
        }
        const float* b_ptr[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) b_ptr[c] = Ql + ((2 * x - 1 + b_tx[c] + 3) % 3) * (9 * 8 * W2_QP) + b_row[c];
#pragma unroll 4
        for (int t = 0; t < ZC / 2; ++t) {
            const float av = a_ptr[t];
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_ptr[c][t], acc[c], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- the four waves' sums, then the workgroup's block of ws[part]
    float* R = Ql;                                                               // 5 x 16 x 64 floats
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* dst = R + (c * 16 + r) * 64 + lane;
                    *dst = w == 0 ? acc[c][r] : *dst + acc[c][r];
                }
        }
        __syncthreads();
    }
    for (int e = tid; e < 5 * 16 * 64; e += 256) {
        const int c = e >> 10, r = (e >> 6) & 15, l = e & 63;
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;         // row / column of accumulator register r in lane l
        const int a = i >> 1, sft = i & 1, b = j & 7, par = (j >> 3) & 1, u = j >> 4;
        const int combo = 2 * c + u;
        if (combo > 8 || (sft == 1 && par == 0)) continue;
        const int tx = combo / 3, ty = combo - 3 * tx, tz = sft ? 0 : (par ? 2 : 1);
        if (pb + a < g.cpp && qb + b < g.cqp) ws[(((int64_t)part * g.cpp + pb + a) * g.cqp + qb + b) * 27 + (tx * 3 + ty) * 3 + tz] = R[e];
    }
}
#define W2_LDS_BYTES(ZC) ((3 * 9 * 8 * (2 * (ZC) + 3) + 4 * 16 * W2_PP) * 4)

struct WgradMfmaPlan {
    bool use, pair;
    int y_blocks, z_segs, x_chunks, xc, n_col;
};
static WgradMfmaPlan wgrad_mfma_plan(int cp, int cq, const int* dims_p, int stride) {
    WgradMfmaPlan pl = {};
    pl.use = ((stride == 1 && (dims_p[2] & 63) == 0) || (stride == 2 && (dims_p[2] & 31) == 0)) && getenv("GENS_K15_NO_MFMA_WGRAD") == nullptr;
    if (stride == 2 && getenv("GENS_K15_NO_MFMA_WGRAD2") != nullptr) pl.use = false;
    if (!pl.use) return pl;
    const int cpp = (cp + 3) / 4 * 4, cqp = (cq + 7) / 8 * 8;
    pl.pair = stride == 1 && cpp <= 4 && getenv("GENS_K15_NO_WGRAD_PAIR") == nullptr;             // the output heads: two y rows per wave
    const int pch = stride == 2 ? 16 : pl.pair ? 4 : 8;                                           // channels of P per column
    pl.n_col = ((cpp + pch - 1) / pch) * (cqp / 8);
    pl.y_blocks = (dims_p[1] + (pl.pair ? 7 : 3)) / (pl.pair ? 8 : 4);
    pl.z_segs = dims_p[2] / (stride == 2 ? 32 : 64);                               // (stride 2: segments of 32 coarse z, two workgroups per CU)
    const int tiles = pl.y_blocks * pl.z_segs * pl.n_col;
    int chunks = ((stride == 2 ? 1024 : 1536) + tiles - 1) / tiles;               // ~1 500 workgroups: two or three per CU (stride 2: two per CU)
    const int max_chunks = (dims_p[0] + 3) / 4;                                   // at least four steps per chunk (each chunk loads two extra planes)
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    pl.xc = (dims_p[0] + chunks - 1) / chunks;
    pl.x_chunks = (dims_p[0] + pl.xc - 1) / pl.xc;
    return pl;
}

static int conv_geom(const int* dims_p, int cp, int cq, int stride, ConvGeom& g, const char* what) {
    GENS_CHECK_ARG(dims_p && dims_p[0] > 0 && dims_p[1] > 0 && dims_p[2] > 0 && cp > 0 && cq > 0, GENS_EINVAL, "%s: bad shape", what);
    GENS_CHECK_ARG(stride == 1 || stride == 2, GENS_EINVAL, "%s: stride %d (1 or 2)", what, stride);
    // (every factor bounded before the products are formed: a 2 GiB tensor has no side longer than 2^29 and no 2^16 channels)
    GENS_CHECK_ARG(dims_p[0] < (1 << 11) && dims_p[1] < (1 << 11) && dims_p[2] < (1 << 11) && cp < (1 << 16) && cq < (1 << 16), GENS_EINVAL,
                   "%s: a side of 2048 voxels or more, or 65536 channels or more (32-bit buffer offsets)", what);
    const int64_t pn = (int64_t)dims_p[0] * dims_p[1] * dims_p[2], qn = pn * stride * stride * stride;
    GENS_CHECK_ARG(pn * (cp + 8) * 4 < (int64_t)CONV_OOB && qn * (cq + 8) * 4 < (int64_t)CONV_OOB, GENS_EINVAL,
                   "%s: a tensor of 2 GiB or more (32-bit buffer offsets)", what);
    g.x = dims_p[0]; g.y = dims_p[1]; g.z = dims_p[2];
    g.cp = cp; g.cq = cq;
    return 0;
}

extern "C" int gens_conv3d_gather(const float* q, const float* w, const float* bias, int cp, int cq, const int* dims_p, int stride,
                                  float* p, void* stream) {
    ConvGeom g;
    if (int rc = conv_geom(dims_p, cp, cq, stride, g, "gens_conv3d_gather")) return rc;
    GENS_CHECK_ARG(q && w && p, GENS_EINVAL, "gens_conv3d_gather: null pointer");
    const int ob = cp > 4 ? 8 : 4;
    g.cpp = (cp + ob - 1) / ob * ob;
    g.cqp = cq;
    const dim3 grid(gens_blocks((int64_t)g.x * g.y * g.z, 256), g.cpp / ob);
    hipStream_t s = (hipStream_t)stream;
    if ((g.z & 63) == 0 && getenv("GENS_K15_NO_ROWLDS") == nullptr) {
        if (stride == 1) {
            if (ob == 8) hipLaunchKernelGGL((conv3d_gather_rows_k<8>), grid, dim3(256), 0, s, q, w, bias, g, p);
            else hipLaunchKernelGGL((conv3d_gather_rows_k<4>), grid, dim3(256), 0, s, q, w, bias, g, p);
        } else {
            if (ob == 8) hipLaunchKernelGGL((conv3d_gather_rows2_k<8>), grid, dim3(256), 0, s, q, w, bias, g, p);
            else hipLaunchKernelGGL((conv3d_gather_rows2_k<4>), grid, dim3(256), 0, s, q, w, bias, g, p);
        }
        return gens_launch_status("gens_conv3d_gather");
    }
    if (ob == 8) {
        if (stride == 1) hipLaunchKernelGGL((conv3d_gather_k<8, 1>), grid, dim3(256), 0, s, q, w, bias, g, p);
        else hipLaunchKernelGGL((conv3d_gather_k<8, 2>), grid, dim3(256), 0, s, q, w, bias, g, p);
    } else {
        if (stride == 1) hipLaunchKernelGGL((conv3d_gather_k<4, 1>), grid, dim3(256), 0, s, q, w, bias, g, p);
        else hipLaunchKernelGGL((conv3d_gather_k<4, 2>), grid, dim3(256), 0, s, q, w, bias, g, p);
    }
    return gens_launch_status("gens_conv3d_gather");
}

extern "C" int gens_conv3d_scatter2(const float* p, const float* w, int cp, int cq, const int* dims_p, float* q, void* stream) {
    ConvGeom g;
    if (int rc = conv_geom(dims_p, cp, cq, 2, g, "gens_conv3d_scatter2")) return rc;
    GENS_CHECK_ARG(p && w && q, GENS_EINVAL, "gens_conv3d_scatter2: null pointer");
    const int ob = cq > 4 ? 8 : 4;
    g.cqp = (cq + ob - 1) / ob * ob;
    g.cpp = cp;
    const dim3 grid(gens_blocks((int64_t)g.x * g.y * g.z, 256), g.cqp / ob);
    hipStream_t s = (hipStream_t)stream;
    if (ob == 8) hipLaunchKernelGGL((conv3d_scatter2_k<8>), grid, dim3(256), 0, s, p, w, g, q);
    else hipLaunchKernelGGL((conv3d_scatter2_k<4>), grid, dim3(256), 0, s, p, w, g, q);
    return gens_launch_status("gens_conv3d_scatter2");
}

static int wgrad_pb(int cp) { (void)cp; return 4; }             // channels of P per column (8 x 8 x 3 accumulators: 264 VGPRs, one wave per SIMD, 1.4x slower)

static void wgrad_shape(int cp, int cq, int64_t pn, int& cpp, int& cqp, int& ny, int& n_ranges, int& chunks_per_range) {
    const int pb = wgrad_pb(cp);
    cpp = (cp + pb - 1) / pb * pb;
    cqp = (cq + 7) / 8 * 8;
    ny = 9 * (cpp / pb) * (cqp / 8);
    const int64_t chunks = (pn + 255) / 256;
    int64_t want = 456;                                          // voxel ranges per launch (x 12 ... 18 columns at 8 -> 8 channels), each with >= 4 chunks if there are that many
    if (want > (chunks + 3) / 4) want = (chunks + 3) / 4;
    if (want < 1) want = 1;
    chunks_per_range = (int)((chunks + want - 1) / want);
    n_ranges = (int)((chunks + chunks_per_range - 1) / chunks_per_range);
}

static bool conv_shape_in_range(int cp, int cq, const int* dims_p) {      // conv_geom's bounds, for the helpers that are asked before any launch
    return dims_p && cp > 0 && cq > 0 && cp < (1 << 16) && cq < (1 << 16) && dims_p[0] > 0 && dims_p[1] > 0 && dims_p[2] > 0 && dims_p[0] < (1 << 11) &&
           dims_p[1] < (1 << 11) && dims_p[2] < (1 << 11);
}

extern "C" int gens_conv3d_wgrad_parts(int cp, int cq, const int* dims_p) {
    if (!conv_shape_in_range(cp, cq, dims_p)) return 0;
    int cpp, cqp, ny, n_ranges, cpr;
    wgrad_shape(cp, cq, (int64_t)dims_p[0] * dims_p[1] * dims_p[2], cpp, cqp, ny, n_ranges, cpr);
    return n_ranges * 4;
}

// the same for a given stride: the stride-1 matrix-core kernel (z a multiple of 64) cuts the volume into its own parts
extern "C" int gens_conv3d_wgrad_parts_strided(int cp, int cq, const int* dims_p, int stride) {
    if (!conv_shape_in_range(cp, cq, dims_p)) return 0;
    const WgradMfmaPlan pl = wgrad_mfma_plan(cp, cq, dims_p, stride);
    if (pl.use) return pl.x_chunks * pl.y_blocks * pl.z_segs;
    return gens_conv3d_wgrad_parts(cp, cq, dims_p);
}

extern "C" int gens_conv3d_wgrad(const float* p, const float* q, int cp, int cq, const int* dims_p, int stride, float* workspace, void* stream) {
    ConvGeom g;
    if (int rc = conv_geom(dims_p, cp, cq, stride, g, "gens_conv3d_wgrad")) return rc;
    GENS_CHECK_ARG(p && q && workspace, GENS_EINVAL, "gens_conv3d_wgrad: null pointer");
    int ny, n_ranges, cpr;
    wgrad_shape(cp, cq, (int64_t)g.x * g.y * g.z, g.cpp, g.cqp, ny, n_ranges, cpr);
    const dim3 grid(8u * (unsigned)ny * (unsigned)((n_ranges + 7) / 8));
    hipStream_t s = (hipStream_t)stream;
    const WgradMfmaPlan pl = wgrad_mfma_plan(cp, cq, dims_p, stride);
    if (pl.use) {          // (workspace: gens_conv3d_wgrad_parts_strided(..., stride) parts)
        const unsigned blocks = (unsigned)(pl.x_chunks * pl.y_blocks * pl.z_segs * pl.n_col);
        if (stride == 2) {
            static GensLdsOptIn lds;
            if (int e = gens_lds_opt_in(lds, (const void*)conv3d_wgrad2_mfma_k<32>, W2_LDS_BYTES(32), "gens_conv3d_wgrad")) return e;
            hipLaunchKernelGGL(conv3d_wgrad2_mfma_k<32>, dim3(blocks), dim3(256), W2_LDS_BYTES(32), s, p, q, g, pl.y_blocks, pl.z_segs, pl.xc, workspace);
        } else if (pl.pair) {
            static GensLdsOptIn lds;
            if (int e = gens_lds_opt_in(lds, (const void*)conv3d_wgrad_mfma_k<true>, WM_LDS_BYTES(true), "gens_conv3d_wgrad")) return e;
            hipLaunchKernelGGL(conv3d_wgrad_mfma_k<true>, dim3(blocks), dim3(256), WM_LDS_BYTES(true), s, p, q, g, pl.y_blocks, pl.z_segs, pl.xc, workspace);
        } else {
            static GensLdsOptIn lds;
            if (int e = gens_lds_opt_in(lds, (const void*)conv3d_wgrad_mfma_k<false>, WM_LDS_BYTES(false), "gens_conv3d_wgrad")) return e;
            hipLaunchKernelGGL(conv3d_wgrad_mfma_k<false>, dim3(blocks), dim3(256), WM_LDS_BYTES(false), s, p, q, g, pl.y_blocks, pl.z_segs, pl.xc, workspace);
        }
        return gens_launch_status("gens_conv3d_wgrad");
    }
    if (stride == 1 && (g.z & 63) == 0 && getenv("GENS_K15_NO_ROWLDS") == nullptr)
        hipLaunchKernelGGL((conv3d_wgrad_k<1, 4, true>), grid, dim3(256), 0, s, p, q, g, n_ranges, cpr, workspace);
    else if (stride == 1) hipLaunchKernelGGL((conv3d_wgrad_k<1, 4, false>), grid, dim3(256), 0, s, p, q, g, n_ranges, cpr, workspace);
    else hipLaunchKernelGGL((conv3d_wgrad_k<2, 4, false>), grid, dim3(256), 0, s, p, q, g, n_ranges, cpr, workspace);
    return gens_launch_status("gens_conv3d_wgrad");
}
