// K2 / K2'' / K3: multi-level trilinear volume look-up with first- and second-order derivatives, and the
// nearest-neighbour visibility-mask look-up.
//
//   forward     lookup_volume(..., "grad")  projector.py:217-245 -> cuda_gridsample.py:71-84 (F.grid_sample 3-D)
//   backward    aten::grid_sampler_3d_backward through cuda_gridsample.py:94-108
//   backward^2  grid_sampler_3d_grad2_kernel, gridsample_cuda.cu:212-533 (the reference's only native kernel)
//   nearest     lookup_volume(..., "nearest"), projector.py:231,240
//
// MI355X design: ALL levels of the pyramid are served by one launch (the reference issues one grid_sampler launch
// plus reshape/permute/cat per level).  The forward runs one thread per (point, level) with the level fastest, so
// a wavefront's 64 results are 64 consecutive float4 of the (N, 4L) output: fully coalesced 1-KiB stores, and the
// 8 corner reads of a "packed" volume are 8 independent 16-B gathers in flight per lane.  Backward passes run one
// thread per point and loop over the levels so d/dpts accumulates in registers (no atomics on it); only the
// scatter into volume gradients uses atomics, and it is skipped entirely when the caller passes no gradient
// buffers (inference, or frozen volumes) -- the reference zero-fills and scatters into a volume-sized tensor on
// every call (gridsample_cuda.cu:620-622, cuda_gridsample.py:113-114).
//
// Axis convention (Q1): p = (px,py,pz) reads element [ix(px)][iy(py)][iz(pz)]; align_corners=True, zeros padding.
#include <stdlib.h>

#include "common.h"

template <int LAYOUT>
__device__ __forceinline__ float4 vox_load(const float* __restrict__ v, int64_t nvox, int64_t lin) {
    if (LAYOUT == GENS_LAYOUT_PACKED) return ((const float4*)v)[lin];
    return make_float4(v[lin], v[nvox + lin], v[2 * nvox + lin], v[3 * nvox + lin]);
}
template <int LAYOUT>
__device__ __forceinline__ void vox_atomic_add(float* __restrict__ v, int64_t nvox, int64_t lin, float4 g, float s) {
    if (LAYOUT == GENS_LAYOUT_PACKED) {
        atomic_add4(v + lin * 4, g, s);
    } else {
        atomicAdd(v + lin, g.x * s);
        atomicAdd(v + nvox + lin, g.y * s);
        atomicAdd(v + 2 * nvox + lin, g.z * s);
        atomicAdd(v + 3 * nvox + lin, g.w * s);
    }
}

// Per-axis cell: base index, the two weights ((i0+1)-pos, pos-i0 as ATen forms them) and bounds flags.
struct Cell {
    int i0;
    float w0, w1;
    bool in0, in1;
};
__device__ __forceinline__ Cell axis_cell(float p, int size) {
    float pos = (p + 1.0f) / 2.0f * (float)(size - 1);
    float f = floorf(pos);
    f = fminf(fmaxf(f, -2.0f), (float)size + 1.0f);  // far-away points: every tap out of bounds, no int overflow
    Cell c;
    c.i0 = (int)f;
    c.w0 = (f + 1.0f) - pos;
    c.w1 = pos - f;
    c.in0 = c.i0 >= 0 && c.i0 < size;
    c.in1 = c.i0 + 1 >= 0 && c.i0 + 1 < size;
    // NaN point, or one so far outside that no tap is inside: zero WEIGHTS as well -- (f + 1) - pos of a point at 1e12 overflows in the product of
    // three weights, and 0 * inf would put a NaN where F.grid_sample (which skips out-of-bounds taps) returns exactly 0
    if (!(pos > -2.0f && pos < (float)size + 1.0f)) { c.in0 = c.in1 = false; c.w0 = c.w1 = 0.0f; }
    return c;
}

#define FOR_CORNERS(body)                                                                     \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b) \
        _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                        \
        bool ok = (a ? cx.in1 : cx.in0) && (b ? cy.in1 : cy.in0) && (c ? cz.in1 : cz.in0);    \
        int64_t lin = ((int64_t)(cx.i0 + a) * Y + (cy.i0 + b)) * Z + (cz.i0 + c);            \
        float wx = a ? cx.w1 : cx.w0, wy = b ? cy.w1 : cy.w0, wz = c ? cz.w1 : cz.w0;         \
        float sx = a ? 1.0f : -1.0f, sy = b ? 1.0f : -1.0f, sz = c ? 1.0f : -1.0f;            \
        body                                                                                   \
    }

// ---------------------------------------------------------------------------------------------------------------
// forward: one thread per (point, level)
// ---------------------------------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void lookup_fwd_k(LevelSet vs, const float* __restrict__ pts, int64_t n,
                                                    float4* __restrict__ out, int xcd_remap) {
    // (XCD x takes the x-th contiguous eighth of the blocks: neighbouring points then share their texel lines in ONE L2; see k4_feature.hip)
    uint32_t blk = blockIdx.x;
    if (xcd_remap) {
        const uint32_t per = gridDim.x >> 3;
        if (blk < 8u * per) blk = (blk & 7u) * per + (blk >> 3);
    }
    int64_t gid = (int64_t)blk * 256 + threadIdx.x;
    int L = vs.n;
    if (gid >= n * L) return;
    int l = (int)(gid % L);
    int64_t i = gid / L;
    float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    int X = vs.dx[l], Y = vs.dy[l], Z = vs.dz[l];
    int64_t nvox = (int64_t)X * Y * Z;
    const float* v = vs.data[l];
    Cell cx = axis_cell(px, X), cy = axis_cell(py, Y), cz = axis_cell(pz, Z);
    float4 acc = f4_zero();
    // branch-free zero padding: read from clamped indices (8 loads in flight), drop out-of-volume corners by a select
    FOR_CORNERS({
        (void)sx; (void)sy; (void)sz; (void)lin;
        const int qx = min(max(cx.i0 + a, 0), X - 1);
        const int qy = min(max(cy.i0 + b, 0), Y - 1);
        const int qz = min(max(cz.i0 + c, 0), Z - 1);
        float4 val = vox_load<LAYOUT>(v, nvox, ((int64_t)qx * Y + qy) * Z + qz);
        if (!ok) val = f4_zero();
        acc = f4_madd(acc, val, wx * wy * wz);
    })
    out[gid] = acc;
}

// The same look-up with the lanes of a wave PAIRED (packed layout only).  The vector L1 looks up one 128-byte line per clock whatever the
// lanes want from it (scripts/probe/gather_rate_probe.py: 0.97 loads per clock and CU with a line per lane, 1.87 with two adjacent lanes on
// one line, 3.4 with four): the forward above issues 8 loads per lane on 8 different lines although the two z-taps of a corner are 32
// contiguous bytes, and runs at that look-up ceiling.  Here lane 2k reads the z0 taps and lane 2k+1 the z1 taps of BOTH items of the pair
// -- eight loads each as before, every instruction now two lanes to a line -- and the halves are swapped back with quad_perm DPP moves; each
// lane then accumulates its own item's eight taps in the order of the kernel above: bit-identical results (pair_swap: common.h).
__global__ __launch_bounds__(256) void lookup_fwd_paired_k(LevelSet vs, const float* __restrict__ pts, uint32_t total, uint32_t magic,
                                                           float4* __restrict__ out, int xcd_remap) {
    uint32_t blk = blockIdx.x;
    if (xcd_remap) {
        const uint32_t per = gridDim.x >> 3;
        if (blk < 8u * per) blk = (blk & 7u) * per + (blk >> 3);
    }
    // (the host takes this kernel for fewer than 2^29 items: 32-bit indices, and item / L as one multiply-high by magic = ceil(2^32 / L))
    const uint32_t gid = blk * 256u + threadIdx.x;
    const uint32_t L = (uint32_t)vs.n;
    const bool active = gid < total;
    const uint32_t g = active ? gid : total - 1u;             // (a lane past the end still serves its partner: it works on the last item)
    const uint32_t i = L == 1u ? g : __umulhi(g, magic);
    const int l = (int)(g - i * L);
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int X = vs.dx[l], Y = vs.dy[l], Z = vs.dz[l];
    const Cell cx = axis_cell(px, X), cy = axis_cell(py, Y), cz = axis_cell(pz, Z);
    const int odd = threadIdx.x & 1;
    // texel indices of this lane's item (clamped, as above), then the partner's
    int rowb[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) rowb[2 * a + b] = (min(max(cx.i0 + a, 0), X - 1) * Y + min(max(cy.i0 + b, 0), Y - 1)) * Z;
    // every instruction serves ONE item of the pair: the first four loads the even lane's item, the last four the odd lane's; this lane
    // reads z-slice `odd` of both
    const int qz0 = min(max(cz.i0, 0), Z - 1), qz1 = min(max(cz.i0 + 1, 0), Z - 1);
    const int p_qz0 = pair_swap(qz0), p_qz1 = pair_swap(qz1);
    const uint64_t base = (uint64_t)vs.data[l];
    const uint64_t p_base = ((uint64_t)(uint32_t)pair_swap((int)(base >> 32)) << 32) | (uint32_t)pair_swap((int)(uint32_t)base);
    const float4* tab_a = (const float4*)(odd ? p_base : base);          // the even lane's item, seen from both lanes
    const float4* tab_b = (const float4*)(odd ? base : p_base);          // the odd lane's item
    const int q_a = odd ? p_qz1 : qz0, q_b = odd ? qz1 : p_qz0;
    float4 first[4], second[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p_row = pair_swap(rowb[k]);
        first[k] = tab_a[(odd ? p_row : rowb[k]) + q_a];
        second[k] = tab_b[(odd ? rowb[k] : p_row) + q_b];
    }
    float4 acc = f4_zero();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // even lane: its z0 tap is first[k], its z1 tap the partner's first[k]; odd lane: z1 = second[k], z0 = the partner's second[k]
        const float4 sf = pair_swap(first[k]), ss = pair_swap(second[k]);
        const int a = k >> 1, b = k & 1;
        const bool okab = (a ? cx.in1 : cx.in0) && (b ? cy.in1 : cy.in0);
        const float wab = (a ? cx.w1 : cx.w0) * (b ? cy.w1 : cy.w0);
        float4 v0 = odd ? ss : first[k], v1 = odd ? second[k] : sf;
        if (!(okab && cz.in0)) v0 = f4_zero();
        if (!(okab && cz.in1)) v1 = f4_zero();
        acc = f4_madd(acc, v0, wab * cz.w0);
        acc = f4_madd(acc, v1, wab * cz.w1);
    }
    if (active) out[gid] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// backward: one thread per point, loops levels.  g_pts always written; volume scatter only if vs.grad[l] != NULL.
// ---------------------------------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void lookup_bwd_k(LevelSet vs, const float* __restrict__ pts,
                                                    const float4* __restrict__ g_out, int64_t n, float* __restrict__ g_pts,
                                                    const uint32_t* __restrict__ order = nullptr) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (order) i = (int64_t)order[i];      // (the brick entries: thread k takes the k-th point of the BRICK order -- neighbours in the volume, whose corners the wave then finds in L1)
    float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    for (int l = 0; l < vs.n; ++l) {
        int X = vs.dx[l], Y = vs.dy[l], Z = vs.dz[l];
        int64_t nvox = (int64_t)X * Y * Z;
        const float* v = vs.data[l];
        float* gv = vs.grad[l];
        Cell cx = axis_cell(px, X), cy = axis_cell(py, Y), cz = axis_cell(pz, Z);
        float4 go = g_out[i * vs.n + l];
        float lx = 0.0f, ly = 0.0f, lz = 0.0f;
        FOR_CORNERS({
            if (ok) {
                float4 val = vox_load<LAYOUT>(v, nvox, lin);
                float dot = val.x * go.x + val.y * go.y + val.z * go.z + val.w * go.w;
                lx += dot * (sx * wy * wz);
                ly += dot * (wx * sy * wz);
                lz += dot * (wx * wy * sz);
                if (gv) vox_atomic_add<LAYOUT>(gv, nvox, lin, go, wx * wy * wz);
            }
        })
        gx += lx * ((float)(X - 1) / 2.0f);
        gy += ly * ((float)(Y - 1) / 2.0f);
        gz += lz * ((float)(Z - 1) / 2.0f);
    }
    if (g_pts) {
        g_pts[3 * i] = gx;
        g_pts[3 * i + 1] = gy;
        g_pts[3 * i + 2] = gz;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward of the backward.  Cotangents: gg_pts on g_pts, optional gg_vols (vs.aux) on g_vols.
//   gg_out[c]  = sum_k V_c[k] * T[k]            (+ sum_k ggV_c[k] * w[k])
//   g_vols2    += g_out[c] * T[k]                         T[k] = e . grad_pos(w[k]),  e = gg_pts * (size-1)/2
//   g_pts2[x]  = sx_scale * sum_c g_out[c] * ( e_y * Mxy_c + e_z * Mxz_c   (+ sum_k ggV_c[k] d w[k]/dx) ) ...
// ---------------------------------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void lookup_bwd2_k(LevelSet vs, const float* __restrict__ pts,
                                                     const float4* __restrict__ g_out, const float* __restrict__ gg_pts,
                                                     int64_t n, float4* __restrict__ gg_out, float* __restrict__ g_pts2,
                                                     const uint32_t* __restrict__ order = nullptr) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (order) i = (int64_t)order[i];
    float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    float qx = gg_pts[3 * i], qy = gg_pts[3 * i + 1], qz = gg_pts[3 * i + 2];
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    for (int l = 0; l < vs.n; ++l) {
        int X = vs.dx[l], Y = vs.dy[l], Z = vs.dz[l];
        int64_t nvox = (int64_t)X * Y * Z;
        const float* v = vs.data[l];
        const float* ggv = vs.aux[l];
        float* gv2 = vs.grad[l];
        float mx = (float)(X - 1) / 2.0f, my = (float)(Y - 1) / 2.0f, mz = (float)(Z - 1) / 2.0f;
        float ex = qx * mx, ey = qy * my, ez = qz * mz;
        Cell cx = axis_cell(px, X), cy = axis_cell(py, Y), cz = axis_cell(pz, Z);
        float4 go = g_out[i * vs.n + l];
        float4 acc = f4_zero();
        float lx = 0.0f, ly = 0.0f, lz = 0.0f;
        FOR_CORNERS({
            if (ok) {
                float4 val = vox_load<LAYOUT>(v, nvox, lin);
                float t = ex * (sx * wy * wz) + ey * (wx * sy * wz) + ez * (wx * wy * sz);
                acc = f4_madd(acc, val, t);
                float dot = val.x * go.x + val.y * go.y + val.z * go.z + val.w * go.w;
                lx += dot * (ey * (sx * sy * wz) + ez * (sx * wy * sz));
                ly += dot * (ex * (sx * sy * wz) + ez * (wx * sy * sz));
                lz += dot * (ex * (sx * wy * sz) + ey * (wx * sy * sz));
                if (gv2) vox_atomic_add<LAYOUT>(gv2, nvox, lin, go, t);
                if (ggv) {
                    float4 g2 = vox_load<LAYOUT>(ggv, nvox, lin);
                    acc = f4_madd(acc, g2, wx * wy * wz);
                    float d2 = g2.x * go.x + g2.y * go.y + g2.z * go.z + g2.w * go.w;
                    lx += d2 * (sx * wy * wz);
                    ly += d2 * (wx * sy * wz);
                    lz += d2 * (wx * wy * sz);
                }
            }
        })
        gg_out[i * vs.n + l] = acc;
        ox += lx * mx;
        oy += ly * my;
        oz += lz * mz;
    }
    g_pts2[3 * i] = ox;
    g_pts2[3 * i + 1] = oy;
    g_pts2[3 * i + 2] = oz;
}

// ---------------------------------------------------------------------------------------------------------------
// The volume scatter of both backward passes: g_vols[c][corner] += g_out[c] * coef(corner), coef = the corner's trilinear weight (first
// order) or T = e . grad_pos(w) (second order).  THIRTY-TWO lanes share a (point, level) pair, lane = (corner, channel): L2 serves float
// atomics per request, so the lanes of an instruction should hit consecutive floats (scripts/probe/atomic_scope_probe.py: 66 G atomics/s with a
// lane per point, 190 - 280 on consecutive floats) -- in a packed volume the two z-neighbours of a corner pair are 32 contiguous bytes, in a
// planar one two floats per channel plane (scripts/probe/k2_bwd_levels_probe.py: the one-lane-per-point scatter cost 0.85 - 1.07 ms per level and
// 1 M points whatever the level's density).  The per-point kernels above run without gradient buffers; the products are theirs bit for bit.
// ---------------------------------------------------------------------------------------------------------------
template <int LAYOUT, bool SECOND>
__global__ __launch_bounds__(256) void lookup_scatter_k(LevelSet vs, const float* __restrict__ pts, const float4* __restrict__ g_out,
                                                        const float* __restrict__ gg_pts, int64_t n) {
    // PACKED: 32 lanes per (level, point), lane = (corner, channel): the two z-neighbours of a corner pair are 32 contiguous bytes.
    // PLANAR: 2 lanes per (level, point), lane = z-neighbour, looping over the channel planes and the four (x, y) corner pairs: only the
    // z-neighbours are contiguous there, and consecutive points (samples along a ray) keep the neighbouring lanes on neighbouring lines.
    constexpr int LANES = LAYOUT == GENS_LAYOUT_PACKED ? 32 : 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t item = gid / LANES;
    if (item >= n * vs.n) return;
    // PLANAR: level-major, so that the lanes of a wave are consecutive points of ONE volume; PACKED: level fastest (a wave = two points' levels)
    const int l = LAYOUT == GENS_LAYOUT_PACKED ? (int)(item % vs.n) : (int)(item / n);
    const int64_t i = LAYOUT == GENS_LAYOUT_PACKED ? item / vs.n : item - (int64_t)l * n;
    float* gv = vs.grad[l];
    if (!gv) return;
    const int sub = (int)(gid % LANES);
    const int X = vs.dx[l], Y = vs.dy[l], Z = vs.dz[l];
    const int64_t nvox = (int64_t)X * Y * Z;
    const Cell cx = axis_cell(pts[3 * i], X), cy = axis_cell(pts[3 * i + 1], Y), cz = axis_cell(pts[3 * i + 2], Z);
    float ex = 0.0f, ey = 0.0f, ez = 0.0f;
    if (SECOND) {
        ex = gg_pts[3 * i] * ((float)(X - 1) / 2.0f);
        ey = gg_pts[3 * i + 1] * ((float)(Y - 1) / 2.0f);
        ez = gg_pts[3 * i + 2] * ((float)(Z - 1) / 2.0f);
    }
    const float* go = (const float*)(g_out + i * vs.n + l);
    auto coef_of = [&](int a, int b, int c) {
        const float wx = a ? cx.w1 : cx.w0, wy = b ? cy.w1 : cy.w0, wz = c ? cz.w1 : cz.w0;
        if (!SECOND) return wx * wy * wz;
        const float sx = a ? 1.0f : -1.0f, sy = b ? 1.0f : -1.0f, sz = c ? 1.0f : -1.0f;
        return ex * (sx * wy * wz) + ey * (wx * sy * wz) + ez * (wx * wy * sz);
    };
    if (LAYOUT == GENS_LAYOUT_PACKED) {
        const int ch = sub & 3, c = (sub >> 2) & 1, b = (sub >> 3) & 1, a = sub >> 4;
        const bool ok = (a ? cx.in1 : cx.in0) && (b ? cy.in1 : cy.in0) && (c ? cz.in1 : cz.in0);
        if (!ok) return;
        const int64_t lin = ((int64_t)(cx.i0 + a) * Y + (cy.i0 + b)) * Z + (cz.i0 + c);
        atomicAdd(gv + lin * 4 + ch, go[ch] * coef_of(a, b, c));
    } else {
        const int c = sub;
        if (!(c ? cz.in1 : cz.in0)) return;
        const float4 g = make_float4(go[0], go[1], go[2], go[3]);
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const int a = ab >> 1, b = ab & 1;
            if (!((a ? cx.in1 : cx.in0) && (b ? cy.in1 : cy.in0))) continue;
            const int64_t lin = ((int64_t)(cx.i0 + a) * Y + (cy.i0 + b)) * Z + (cz.i0 + c);
            const float k = coef_of(a, b, c);
            atomicAdd(gv + lin, g.x * k);
            atomicAdd(gv + nvox + lin, g.y * k);
            atomicAdd(gv + 2 * nvox + lin, g.z * k);
            atomicAdd(gv + 3 * nvox + lin, g.w * k);
        }
    }
}

template <bool SECOND>
static void launch_lookup_scatter(int layout, const LevelSet& vs, const float* pts, const float* g_out, const float* gg_pts, int64_t n, void* stream) {
    if (layout == GENS_LAYOUT_PACKED)
        lookup_scatter_k<GENS_LAYOUT_PACKED, SECOND><<<gens_blocks(n * vs.n * 32, 256), 256, 0, (hipStream_t)stream>>>(vs, pts, (const float4*)g_out, gg_pts, n);
    else
        lookup_scatter_k<GENS_LAYOUT_PLANAR, SECOND><<<gens_blocks(n * vs.n * 2, 256), 256, 0, (hipStream_t)stream>>>(vs, pts, (const float4*)g_out, gg_pts, n);
}

// ---------------------------------------------------------------------------------------------------------------
// The scatter by BRICKS (round 6; gens_lookup_volume_bwd_bricks / _bwd2_bricks, caller's scratch).
//
// The scatter above sends one float atomic per (point, level, corner, channel) -- 96 per point at three levels -- and runs at 24 - 47 G atomics/s
// depending on how the caller ordered its points (profiles/r06_k2_bwd_order_probe.txt): 2 - 4 ms per million points.  Here the points are first
// counted into bricks of 8 x 8 x 8 cells of the FINEST level (a counting sort: key, histogram, scan, fill -- four small launches), and a workgroup per
// brick then walks the levels: the brick's points can only touch a tile of <= 12^3 voxels of a level (<= 9^3 of the finest), the tile's sums live in
// LDS as DOUBLES (ds_add_f64 costs 18 cycles per wave instruction, ds_add_f32 193: lds_atomic_probe.py), 32 lanes per point = (corner, channel), and
// every touched voxel of the tile goes to memory ONCE, z-contiguous lanes on consecutive floats.  A corner that falls outside its brick's tile (points
// far outside the cube, float rounding at a tile's edge) takes the direct atomic.  The sums are those of the direct scatter in another order (double
// partial sums: closer to the exact value); NaN / infinity propagate.
// ---------------------------------------------------------------------------------------------------------------
#define BR_NB 32                 // bricks per axis at most
#define BR_MAX (BR_NB * BR_NB * BR_NB)
#define BR_EXT 10                // tile extent per axis at most (the finest level's is 9: eight cells and the + 1 neighbour)
#define BR_TILE (BR_EXT * BR_EXT * BR_EXT * 4)
#define BR_SEG 512               // points per work item
struct BrickPlan {
    int fine;                    // the level the bricks are cut from (the one with the most voxels)
    int nb[3];                   // bricks per axis
    uint32_t n_bricks;
    uint32_t* count;             // [n_bricks + 1]   (zeroed per call)
    uint32_t* offset;            // [n_bricks + 1]
    uint32_t* order;             // [n] point indices, brick by brick
    uint32_t* key;               // [n] a point's brick and
    uint32_t* rank;              // [n] its rank among the brick's points (what the counting atomic returned: the fill needs no second one)
    uint2* items;                // work items (brick, segment of <= BR_SEG of its points): a crowded brick is many workgroups' work
    uint32_t* n_items;
    uint32_t max_items;
};
__device__ __forceinline__ uint32_t brick_of(const LevelSet& vs, const BrickPlan& P, const float* __restrict__ pts, int64_t i) {
    const int size[3] = {vs.dx[P.fine], vs.dy[P.fine], vs.dz[P.fine]};
    uint32_t b[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const Cell c = axis_cell(pts[3 * i + ax], size[ax]);
        const int cell = min(max(c.i0, 0), size[ax] - 1);                        // (NaN: axis_cell leaves i0 finite -- (int) of a clamped float)
        b[ax] = (uint32_t)min(cell >> 3, P.nb[ax] - 1);
    }
    return (b[0] * (uint32_t)P.nb[1] + b[1]) * (uint32_t)P.nb[2] + b[2];
}
__global__ __launch_bounds__(256) void brick_zero_k(uint32_t* __restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = 0u;
}
__global__ __launch_bounds__(256) void brick_count_k(LevelSet vs, BrickPlan P, const float* __restrict__ pts, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const uint32_t b = brick_of(vs, P, pts, i);
        P.key[i] = b;
        P.rank[i] = atomicAdd(P.count + b, 1u);
    }
}
__global__ __launch_bounds__(1024) void brick_scan_k(BrickPlan P) {
    __shared__ uint32_t part[1024], part_i[1024];
    const uint32_t tid = threadIdx.x, per = (P.n_bricks + 1023u) / 1024u;
    const uint32_t lo = min(tid * per, P.n_bricks), hi = min(lo + per, P.n_bricks);
    // (a thread's bins through 16-byte loads where its range allows: 32 dependent 4-byte loads per thread were 70 us of a 1 ms call)
    auto segs = [](uint32_t k) { return (k + BR_SEG - 1u) / BR_SEG; };
    uint32_t c = 0, it = 0;
    const bool quads = (per & 3u) == 0u && hi - lo == per && ((uintptr_t)(P.count + lo) & 15) == 0;
    if (quads) {
#pragma unroll 8
        for (uint32_t b = lo; b < hi; b += 4u) {
            const uint4 q = *(const uint4*)(P.count + b);
            c += (q.x + q.y) + (q.z + q.w);
            it += (segs(q.x) + segs(q.y)) + (segs(q.z) + segs(q.w));
        }
    } else {
        for (uint32_t b = lo; b < hi; ++b) {
            c += P.count[b];
            it += segs(P.count[b]);
        }
    }
    part[tid] = c;
    part_i[tid] = it;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                                      // (inclusive scans of the 1 024 partial sums)
        const uint32_t add = tid >= d ? part[tid - d] : 0u, add_i = tid >= d ? part_i[tid - d] : 0u;
        __syncthreads();
        part[tid] += add;
        part_i[tid] += add_i;
        __syncthreads();
    }
    if (tid == 1023u) {
        P.offset[P.n_bricks] = part[1023];
        *P.n_items = min(part_i[1023], P.max_items);
    }
    c = part[tid] - c;                                                              // exclusive
    it = part_i[tid] - it;
    auto emit = [&](uint32_t b, uint32_t k) {
        for (uint32_t sg = 0; sg * BR_SEG < k; ++sg, ++it)
            if (it < P.max_items) P.items[it] = make_uint2(b, sg);
    };
    if (quads) {
#pragma unroll 4
        for (uint32_t b = lo; b < hi; b += 4u) {
            const uint4 q = *(const uint4*)(P.count + b);
            const uint4 o = make_uint4(c, c + q.x, c + q.x + q.y, c + q.x + q.y + q.z);
            *(uint4*)(P.offset + b) = o;
            c = o.w + q.w;
            emit(b, q.x); emit(b + 1u, q.y); emit(b + 2u, q.z); emit(b + 3u, q.w);
        }
    } else {
        for (uint32_t b = lo; b < hi; ++b) {
            const uint32_t k = P.count[b];
            P.offset[b] = c;
            c += k;
            emit(b, k);
        }
    }
}
__global__ __launch_bounds__(256) void brick_fill_k(BrickPlan P, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) P.order[P.offset[P.key[i]] + P.rank[i]] = (uint32_t)i;
}

#define BR_STAGE 64              // points of an item staged in LDS at a time (their coordinates and every level's cotangent: one round of gathers)
template <int LAYOUT, bool SECOND>
__global__ __launch_bounds__(256) void lookup_scatter_bricks_k(LevelSet vs, BrickPlan P, const float* __restrict__ pts, const float4* __restrict__ g_out,
                                                               const float* __restrict__ gg_pts) {
    __shared__ double tile[BR_TILE];
    __shared__ float sp[BR_STAGE][3], se[BR_STAGE][3];
    __shared__ float4 sg[BR_STAGE][GENS_MAX_LEVELS];
    if (blockIdx.x >= *P.n_items) return;
    const uint2 item = P.items[blockIdx.x];
    const uint32_t brick = item.x, first = P.offset[brick] + item.y * BR_SEG, cnt = min((uint32_t)BR_SEG, P.offset[brick + 1] - first);
    const int tid = threadIdx.x, slot = tid >> 5, sub = tid & 31;
    const int ch = sub & 3, c = (sub >> 2) & 1, b = (sub >> 3) & 1, a = sub >> 4;
    const uint32_t bz = brick % (uint32_t)P.nb[2], by = (brick / (uint32_t)P.nb[2]) % (uint32_t)P.nb[1], bx = brick / (uint32_t)(P.nb[1] * P.nb[2]);
    const int fine[3] = {vs.dx[P.fine], vs.dy[P.fine], vs.dz[P.fine]};
    const uint32_t bc[3] = {bx, by, bz};
    // one round of gathers for a chunk of the item's points: a thread per (point, level) for the cotangents, the level-0 threads for the coordinates
    auto stage = [&](uint32_t base, int staged) {
        for (int k = tid; k < staged * vs.n; k += 256) {
            const int p = k / vs.n, l = k - p * vs.n;
            const int64_t i = (int64_t)P.order[first + base + (uint32_t)p];
            sg[p][l] = g_out[i * vs.n + l];
            if (l == 0) {
                sp[p][0] = pts[3 * i]; sp[p][1] = pts[3 * i + 1]; sp[p][2] = pts[3 * i + 2];
                if (SECOND) { se[p][0] = gg_pts[3 * i]; se[p][1] = gg_pts[3 * i + 1]; se[p][2] = gg_pts[3 * i + 2]; }
            }
        }
    };
    const bool single = cnt <= BR_STAGE;                                             // (most items of a sparse cloud: staged once for all levels)
    if (single) stage(0u, (int)cnt);
    for (int l = 0; l < vs.n; ++l) {
        float* __restrict__ gv = vs.grad[l];
        if (!gv) continue;                                                          // (uniform)
        const int size[3] = {vs.dx[l], vs.dy[l], vs.dz[l]};
        const int64_t nvox = (int64_t)size[0] * size[1] * size[2];
        // the voxels of this level the brick's points can touch: cells [8 b, 8 b + 8) of the finest level are positions [8 b, 8 b + 8) / (fine - 1) of
        // the axis, i.e. cells floor(8 b r) .. floor((8 b + 8) r) of this one (r = (size - 1) / (fine - 1); a hundredth of a cell either way for the
        // rounding of the two float expressions) and their + 1 neighbours; the last brick of an axis also holds what lies beyond it.  Whatever still
        // falls outside takes the direct atomic below.
        int lo[3], ext[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float r = fine[ax] > 1 ? (float)(size[ax] - 1) / (float)(fine[ax] - 1) : 0.0f;
            const int cell_lo = (int)bc[ax] * 8, cell_hi = (int)bc[ax] + 1 == P.nb[ax] ? fine[ax] : cell_lo + 8;
            const bool same = size[ax] == fine[ax];                                  // (the finest level itself: cells 8 b .. 8 b + 7 and their + 1 neighbours, exactly)
            lo[ax] = same ? cell_lo : max((int)floorf((float)cell_lo * r - 0.01f), 0);
            const int hi = min(same ? cell_hi : (int)floorf((float)cell_hi * r + 0.01f) + 1, size[ax] - 1);
            ext[ax] = min(max(hi - lo[ax] + 1, 1), BR_EXT);
        }
        const int cells = ext[0] * ext[1] * ext[2];
        for (int k = tid; k < 4 * cells; k += 256) tile[k] = 0.0;
        __syncthreads();                                                            // (also: the staged chunk is complete)
        for (uint32_t base = 0; base < cnt; base += BR_STAGE) {
            const int staged = (int)min((uint32_t)BR_STAGE, cnt - base);
            if (!single) {
                stage(base, staged);
                __syncthreads();
            }
            for (int p = slot; p < staged; p += 8) {
                const Cell cx = axis_cell(sp[p][0], size[0]), cy = axis_cell(sp[p][1], size[1]), cz = axis_cell(sp[p][2], size[2]);
                const bool ok = (a ? cx.in1 : cx.in0) && (b ? cy.in1 : cy.in0) && (c ? cz.in1 : cz.in0);
                if (!ok) continue;
                const float wx = a ? cx.w1 : cx.w0, wy = b ? cy.w1 : cy.w0, wz = c ? cz.w1 : cz.w0;
                float coef;
                if (!SECOND) {
                    coef = wx * wy * wz;
                } else {
                    const float ex = se[p][0] * ((float)(size[0] - 1) / 2.0f), ey = se[p][1] * ((float)(size[1] - 1) / 2.0f), ez = se[p][2] * ((float)(size[2] - 1) / 2.0f);
                    const float sx = a ? 1.0f : -1.0f, sy = b ? 1.0f : -1.0f, sz = c ? 1.0f : -1.0f;
                    coef = ex * (sx * wy * wz) + ey * (wx * sy * wz) + ez * (wx * wy * sz);
                }
                const float val = ((const float*)&sg[p][l])[ch] * coef;
                const int x = cx.i0 + a, y = cy.i0 + b, z = cz.i0 + c;
                const int tx = x - lo[0], ty = y - lo[1], tz = z - lo[2];
                if ((unsigned)tx < (unsigned)ext[0] && (unsigned)ty < (unsigned)ext[1] && (unsigned)tz < (unsigned)ext[2]) {
                    const int at = LAYOUT == GENS_LAYOUT_PACKED ? ((tx * ext[1] + ty) * ext[2] + tz) * 4 + ch : ((ch * ext[0] + tx) * ext[1] + ty) * ext[2] + tz;
                    atomicAdd(tile + at, (double)val);
                } else {                                                            // outside the tile: the direct atomic
                    const int64_t lin = ((int64_t)x * size[1] + y) * size[2] + z;
                    atomicAdd(LAYOUT == GENS_LAYOUT_PACKED ? gv + lin * 4 + ch : gv + ch * nvox + lin, val);
                }
            }
            if (!single) __syncthreads();                                           // (the next chunk overwrites the staged rows)
        }
        __syncthreads();
        // the flush, a z-row of the tile per 16 lanes (no division per entry; the lanes of a row on consecutive floats): PLANAR rows = (channel, x, y),
        // 16 lanes along z; PACKED rows = (x, y), lanes = (z, channel) pairs, 64 along a row of <= 10 x 4 floats
        if (LAYOUT == GENS_LAYOUT_PACKED) {
            const int rows = ext[0] * ext[1], lane = tid & 63, wave = tid >> 6;
            for (int r = wave; r < rows; r += 4) {
                const int tx = r / ext[1], ty = r - tx * ext[1];
                if (lane < 4 * ext[2]) {
                    const float v = (float)tile[r * ext[2] * 4 + lane];
                    if (v != 0.0f) {                                                // (NaN != 0: travels on)
                        const int64_t lin = ((int64_t)(lo[0] + tx) * size[1] + (lo[1] + ty)) * size[2] + lo[2];
                        atomicAdd(gv + lin * 4 + lane, v);
                    }
                }
            }
        } else {
            const int rows = 4 * ext[0] * ext[1], tz = tid & 15;
            for (int r = tid >> 4; r < rows; r += 16) {
                const int cc = r / (ext[0] * ext[1]), xy = r - cc * ext[0] * ext[1], tx = xy / ext[1], ty = xy - tx * ext[1];
                if (tz < ext[2]) {
                    const float v = (float)tile[r * ext[2] + tz];
                    if (v != 0.0f) {
                        const int64_t lin = ((int64_t)(lo[0] + tx) * size[1] + (lo[1] + ty)) * size[2] + (lo[2] + tz);
                        atomicAdd(gv + cc * nvox + lin, v);
                    }
                }
            }
        }
        __syncthreads();
    }
}

static bool brick_plan(const LevelSet& vs, int64_t n, void* scratch, int64_t scratch_bytes, BrickPlan* P, int64_t* need) {
    int fine = 0;
    int64_t most = 0;
    for (int l = 0; l < vs.n; ++l) {
        const int64_t v = (int64_t)vs.dx[l] * vs.dy[l] * vs.dz[l];
        if (v > most) { most = v; fine = l; }
    }
    P->fine = fine;
    const int size[3] = {vs.dx[fine], vs.dy[fine], vs.dz[fine]};
    for (int ax = 0; ax < 3; ++ax) P->nb[ax] = std::min(BR_NB, std::max(1, (size[ax] + 7) / 8));
    P->n_bricks = (uint32_t)(P->nb[0] * P->nb[1] * P->nb[2]);
    const int64_t arr = ((int64_t)P->n_bricks + 1 + 3) / 4 * 4;                      // (each array a multiple of four words: 16-byte loads in the scan)
    P->max_items = (uint32_t)std::min<int64_t>((int64_t)P->n_bricks + n / BR_SEG + 1, 0x7fffffff);
    const int64_t n4 = (n + 3) / 4 * 4;
    const int64_t words = 2 * arr + 3 * n4 + 2 * (int64_t)P->max_items + 4;
    *need = words * 4;
    if (!scratch || scratch_bytes < *need) return false;
    uint32_t* w = (uint32_t*)scratch;
    P->count = w;
    P->offset = w + arr;
    P->order = w + 2 * arr;
    P->key = w + 2 * arr + n4;
    P->rank = w + 2 * arr + 2 * n4;
    P->n_items = w + 2 * arr + 3 * n4;
    P->items = (uint2*)(w + 2 * arr + 3 * n4 + 4);
    return true;
}
// bricks only where every level's tile fits: a level may be at most as fine as the finest per axis (always) and the finest at most 8 * BR_NB cells wide
// per axis beyond which a brick is wider than 8 cells -- then its tile would not fit and the direct scatter serves the call
static bool bricks_fit(const LevelSet& vs, const BrickPlan& P) {
    const int size[3] = {vs.dx[P.fine], vs.dy[P.fine], vs.dz[P.fine]};
    for (int ax = 0; ax < 3; ++ax)
        if ((size[ax] + 7) / 8 > BR_NB) return false;
    return true;
}

extern "C" int64_t gens_lookup_scatter_bricks_scratch_bytes(int64_t n) {
    return n < 0 ? 0 : 4 * (2 * ((int64_t)BR_MAX + 4) + 3 * ((n + 3) / 4 * 4) + 2 * ((int64_t)BR_MAX + n / BR_SEG + 1) + 4);
}

// the counting sort of a call's points by brick; P.order == nullptr afterwards: sizes the bricks do not cover (the caller takes the direct scatter)
static int bricks_sort(const char* who, const LevelSet& vs, const float* pts, int64_t n, void* scratch, int64_t scratch_bytes, void* stream, BrickPlan* P) {
    int64_t need = 0;
    const bool have = brick_plan(vs, n, scratch, scratch_bytes, P, &need);
    GENS_CHECK_ARG(have && ((uintptr_t)scratch & 15) == 0, GENS_EINVAL, "%s: scratch of %lld bytes needed (gens_lookup_scatter_bricks_scratch_bytes), got %lld", who,
                   (long long)need, (long long)scratch_bytes);
    if (!bricks_fit(vs, *P) || n >= ((int64_t)1 << 31)) {
        P->order = nullptr;
        return 0;
    }
    hipStream_t st = (hipStream_t)stream;
    brick_zero_k<<<(P->n_bricks + 1u + 255u) / 256u, 256, 0, st>>>(P->count, P->n_bricks + 1u);   // (a kernel: a captured memset node does not order, k1_volume.hip)
    brick_count_k<<<gens_blocks(n, 256), 256, 0, st>>>(vs, *P, pts, n);
    brick_scan_k<<<1, 1024, 0, st>>>(*P);
    brick_fill_k<<<gens_blocks(n, 256), 256, 0, st>>>(*P, n);
    return 0;
}
template <bool SECOND>
static void scatter_bricks(int layout, const LevelSet& vs, const BrickPlan& P, const float* pts, const float* g_out, const float* gg_pts, int64_t n, void* stream) {
    if (!P.order) {                                                                 // (sizes the bricks do not cover: the direct scatter)
        launch_lookup_scatter<SECOND>(layout, vs, pts, g_out, gg_pts, n, stream);
        return;
    }
    hipStream_t st = (hipStream_t)stream;
    if (layout == GENS_LAYOUT_PACKED)
        lookup_scatter_bricks_k<GENS_LAYOUT_PACKED, SECOND><<<P.max_items, 256, 0, st>>>(vs, P, pts, (const float4*)g_out, gg_pts);
    else
        lookup_scatter_bricks_k<GENS_LAYOUT_PLANAR, SECOND><<<P.max_items, 256, 0, st>>>(vs, P, pts, (const float4*)g_out, gg_pts);
}

// ---------------------------------------------------------------------------------------------------------------
// K3 nearest mask + fused ray-point generation
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_nearest_k(LevelSet ms, const float* __restrict__ pts, int64_t n,
                                                      uint8_t* __restrict__ valid, float* __restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    bool any = false;
    for (int l = 0; l < ms.n; ++l) {
        float m = mask_nearest(ms.data[l], ms.dx[l], ms.dy[l], ms.dz[l], px, py, pz);
        if (vals) vals[i * ms.n + l] = m;
        any = any || (m > 0.0f);
    }
    if (valid) valid[i] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void ray_points_k(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const float* __restrict__ z, int64_t total, int n, int mid,
                                                    float sample_dist, LevelSet ms, float* __restrict__ pts,
                                                    uint8_t* __restrict__ valid) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int64_t r;
    int j;
    if ((n & (n - 1)) == 0) {                           // 64 coarse / 128 final samples: a shift and a mask instead of two 64-bit divisions
        const int sh = __builtin_ctz((unsigned)n);      // (the divisions were most of this kernel's instructions)
        r = i >> sh;
        j = (int)(i & (int64_t)(n - 1));
    } else if (total < ((int64_t)1 << 31)) {
        const uint32_t iu = (uint32_t)i, q = iu / (uint32_t)n;
        r = q;
        j = (int)(iu - q * (uint32_t)n);
    } else {
        r = i / n;
        j = (int)(i % n);
    }
    float t = z[i];
    if (mid) {
        float dist = (j + 1 < n) ? z[i + 1] - t : sample_dist;               // (Q10)
        t = t + dist * 0.5f;
    }
    float px = rays_o[3 * r] + rays_d[3 * r] * t;
    float py = rays_o[3 * r + 1] + rays_d[3 * r + 1] * t;
    float pz = rays_o[3 * r + 2] + rays_d[3 * r + 2] * t;
    pts[3 * i] = px;
    pts[3 * i + 1] = py;
    pts[3 * i + 2] = pz;
    if (valid) valid[i] = any_mask(ms, px, py, pz) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Device-side, order-preserving compaction of the valid points (the boolean-mask gather of implicit_surface.py:121-126,
// 174-179, 370-375 without its host synchronisation), including the "no valid point -> first 10" rescue (Q7).
//   pass 1: per 1024-element block, number of set flags          pass 2 (one block): exclusive scan + total + rescue
//   pass 3: each block writes its indices at block_offset + rank (ballot / popcount inside the wave)
// ---------------------------------------------------------------------------------------------------------------
#define CP_BLOCK 1024
__global__ __launch_bounds__(256) void compact_count_k(const uint8_t* __restrict__ valid, int64_t n, int32_t* __restrict__ block_cnt) {
    __shared__ int red[4];
    int64_t base = (int64_t)blockIdx.x * CP_BLOCK;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int64_t i = base + k * 256 + threadIdx.x;
        c += (i < n && valid[i]) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void compact_scan_k(int32_t* __restrict__ block_cnt, int n_blocks, int64_t n, int64_t* __restrict__ idx,
                                                       int32_t* __restrict__ count) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (n_blocks + 1023) / 1024;
    int local = 0;
    for (int k = 0; k < per; ++k) {
        int b = t * per + k;
        if (b < n_blocks) local += block_cnt[b];
    }
    part[t] = local;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive scan over the 1024 partials
        int v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - local;                           // exclusive prefix of this thread's first block
    for (int k = 0; k < per; ++k) {
        int b = t * per + k;
        if (b < n_blocks) {
            int c = block_cnt[b];
            block_cnt[b] = run;
            run += c;
        }
    }
    if (t == 1023) {
        int total = part[1023];
        if (total < 1) {                                 // implicit_surface.py:123-124: nothing valid -> the first 10 points
            total = (int)min((int64_t)10, n);
            for (int k = 0; k < total; ++k) idx[k] = k;
        }
        count[0] = total;
    }
}

__global__ __launch_bounds__(256) void compact_write_k(const uint8_t* __restrict__ valid, int64_t n, const int32_t* __restrict__ block_off,
                                                       int64_t* __restrict__ idx) {
    __shared__ int wave_cnt[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t base = (int64_t)blockIdx.x * CP_BLOCK;
    bool f[4];
    int rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int64_t i = base + k * 256 + threadIdx.x;
        f[k] = i < n && valid[i];
        unsigned long long m = __ballot(f[k]);
        rank[k] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[k][wave] = __popcll(m);
    }
    __syncthreads();
    int off = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int before = 0;
        for (int kk = 0; kk < k; ++kk) before += wave_cnt[kk][0] + wave_cnt[kk][1] + wave_cnt[kk][2] + wave_cnt[kk][3];
        for (int w = 0; w < wave; ++w) before += wave_cnt[k][w];
        if (f[k]) idx[off + before + rank[k]] = base + k * 256 + threadIdx.x;
    }
}

extern "C" int gens_compact_valid(const uint8_t* valid, int64_t n, int64_t* idx, int32_t* count, int32_t* scratch, void* stream) {
    GENS_CHECK_ARG(n >= 0 && count && (n == 0 || (valid && idx && scratch)), GENS_EINVAL, "gens_compact_valid: null pointer");
    GENS_CHECK_ARG(n < ((int64_t)1 << 31), GENS_ELIMIT, "gens_compact_valid: at most 2^31-1 points per call");
    hipStream_t s = (hipStream_t)stream;
    int n_blocks = (int)gens_blocks(n, CP_BLOCK);
    if (n_blocks == 0) n_blocks = 1;
    if (n > 0) compact_count_k<<<n_blocks, 256, 0, s>>>(valid, n, scratch);
    else if (hipMemsetAsync(scratch, 0, sizeof(int32_t), s) != hipSuccess) return gens_launch_status("gens_compact_valid");
    compact_scan_k<<<1, 1024, 0, s>>>(scratch, n > 0 ? n_blocks : 0, n, idx, count);
    if (n > 0) compact_write_k<<<n_blocks, 256, 0, s>>>(valid, n, scratch, idx);
    return gens_launch_status("gens_compact_valid");
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels) {
    GENS_CHECK_ARG(data && dims, GENS_EINVAL, "%s: null level table", who);
    GENS_CHECK_ARG(n_levels > 0 && n_levels <= GENS_MAX_LEVELS, GENS_ELIMIT, "%s: n_levels=%d not in 1..%d", who, n_levels,
                   GENS_MAX_LEVELS);
    ls->n = n_levels;
    ls->bits = 0;
    for (int l = 0; l < GENS_MAX_LEVELS; ++l) {
        ls->data[l] = nullptr;
        ls->grad[l] = nullptr;
        ls->aux[l] = nullptr;
        ls->dx[l] = ls->dy[l] = ls->dz[l] = 1;
    }
    for (int l = 0; l < n_levels; ++l) {
        GENS_CHECK_ARG(data[l], GENS_EINVAL, "%s: level %d is null", who, l);
        GENS_CHECK_ARG(dims[3 * l] > 0 && dims[3 * l + 1] > 0 && dims[3 * l + 2] > 0, GENS_EINVAL, "%s: level %d has a zero dim", who, l);
        ls->data[l] = data[l];
        ls->dx[l] = dims[3 * l];
        ls->dy[l] = dims[3 * l + 1];
        ls->dz[l] = dims[3 * l + 2];
    }
    return 0;
}

#define DISPATCH_LAYOUT(layout, KERNEL, grid, stream, ...)                                                     \
    do {                                                                                                       \
        if ((layout) == GENS_LAYOUT_PACKED)                                                                    \
            KERNEL<GENS_LAYOUT_PACKED><<<(grid), 256, 0, (hipStream_t)(stream)>>>(__VA_ARGS__);                \
        else                                                                                                   \
            KERNEL<GENS_LAYOUT_PLANAR><<<(grid), 256, 0, (hipStream_t)(stream)>>>(__VA_ARGS__);                \
    } while (0)

extern "C" int gens_lookup_volume_fwd(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                                      int64_t n, float* out, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_lookup_volume_fwd", &vs, vols, dims, n_levels)) return e;
    GENS_CHECK_ARG(layout == 0 || layout == 1, GENS_EINVAL, "gens_lookup_volume_fwd: bad layout %d", layout);
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && out)), GENS_EINVAL, "gens_lookup_volume_fwd: null pts/out");
    if (n == 0) return 0;
    const int remap = getenv("GENS_NO_XCD_REMAP") == nullptr;
    bool paired = layout == GENS_LAYOUT_PACKED && getenv("GENS_K2_NO_PAIRS") == nullptr;      // (texel indices in 32 bits there)
    for (int l = 0; l < n_levels; ++l) paired = paired && (int64_t)vs.dx[l] * vs.dy[l] * vs.dz[l] < (1ll << 31);
    paired = paired && n * n_levels < (1ll << 29);
    if (paired)
        lookup_fwd_paired_k<<<gens_blocks(n * n_levels, 256), 256, 0, (hipStream_t)stream>>>(
            vs, pts, (uint32_t)(n * n_levels), (uint32_t)(((1ull << 32) + (uint32_t)n_levels - 1u) / (uint32_t)n_levels), (float4*)out, remap);
    else
        DISPATCH_LAYOUT(layout, lookup_fwd_k, gens_blocks(n * n_levels, 256), stream, vs, pts, n, (float4*)out, remap);
    return gens_launch_status("gens_lookup_volume_fwd");
}

extern "C" int gens_lookup_volume_bwd(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                                      const float* g_out, int64_t n, float* const* g_vols, float* g_pts, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_lookup_volume_bwd", &vs, vols, dims, n_levels)) return e;
    GENS_CHECK_ARG(layout == 0 || layout == 1, GENS_EINVAL, "gens_lookup_volume_bwd: bad layout %d", layout);
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_out)), GENS_EINVAL, "gens_lookup_volume_bwd: null pts/g_out");
    GENS_CHECK_ARG(g_vols || g_pts, GENS_EINVAL, "gens_lookup_volume_bwd: no output requested");
    if (n == 0) return 0;
    const bool lane_per_point = getenv("GENS_K2_SCATTER_PER_POINT") != nullptr;       // (switch: the scatter inside the per-point kernel, for A/B runs)
    bool scatter = false;
    if (g_vols && lane_per_point)
        for (int l = 0; l < n_levels; ++l) vs.grad[l] = g_vols[l];
    if (g_pts || (g_vols && lane_per_point))
        DISPATCH_LAYOUT(layout, lookup_bwd_k, gens_blocks(n, 256), stream, vs, pts, (const float4*)g_out, n, g_pts);
    if (g_vols && !lane_per_point) {
        for (int l = 0; l < n_levels; ++l) {
            vs.grad[l] = g_vols[l];
            scatter = scatter || g_vols[l];
        }
        if (scatter) launch_lookup_scatter<false>(layout, vs, pts, g_out, nullptr, n, stream);
    }
    return gens_launch_status("gens_lookup_volume_bwd");
}

extern "C" int gens_lookup_volume_bwd2(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                                       const float* g_out, const float* gg_pts, const float* const* gg_vols, int64_t n,
                                       float* gg_out, float* const* g_vols2, float* g_pts2, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_lookup_volume_bwd2", &vs, vols, dims, n_levels)) return e;
    GENS_CHECK_ARG(layout == 0 || layout == 1, GENS_EINVAL, "gens_lookup_volume_bwd2: bad layout %d", layout);
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_out && gg_pts && gg_out && g_pts2)), GENS_EINVAL,
                   "gens_lookup_volume_bwd2: null pointer");
    if (n == 0) return 0;
    const bool lane_per_point = getenv("GENS_K2_SCATTER_PER_POINT") != nullptr;
    bool scatter = false;
    for (int l = 0; l < n_levels; ++l) {
        if (g_vols2 && lane_per_point) vs.grad[l] = g_vols2[l];
        if (gg_vols) vs.aux[l] = gg_vols[l];
    }
    DISPATCH_LAYOUT(layout, lookup_bwd2_k, gens_blocks(n, 256), stream, vs, pts, (const float4*)g_out, gg_pts, n,
                    (float4*)gg_out, g_pts2);
    if (g_vols2 && !lane_per_point) {
        for (int l = 0; l < n_levels; ++l) {
            vs.grad[l] = g_vols2[l];
            scatter = scatter || g_vols2[l];
        }
        if (scatter) launch_lookup_scatter<true>(layout, vs, pts, g_out, gg_pts, n, stream);
    }
    return gens_launch_status("gens_lookup_volume_bwd2");
}

extern "C" int gens_lookup_volume_bwd_bricks(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                                             const float* g_out, int64_t n, float* const* g_vols, float* g_pts, void* scratch, int64_t scratch_bytes,
                                             void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_lookup_volume_bwd_bricks", &vs, vols, dims, n_levels)) return e;
    GENS_CHECK_ARG(layout == 0 || layout == 1, GENS_EINVAL, "gens_lookup_volume_bwd_bricks: bad layout %d", layout);
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_out)), GENS_EINVAL, "gens_lookup_volume_bwd_bricks: null pts/g_out");
    GENS_CHECK_ARG(g_vols || g_pts, GENS_EINVAL, "gens_lookup_volume_bwd_bricks: no output requested");
    if (n == 0) return 0;
    BrickPlan P;
    if (int e = bricks_sort("gens_lookup_volume_bwd_bricks", vs, pts, n, scratch, scratch_bytes, stream, &P)) return e;
    // the per-point kernel walks the points in BRICK order too: thread k's neighbours are its neighbours in the volume (the same sums per point)
    if (g_pts) DISPATCH_LAYOUT(layout, lookup_bwd_k, gens_blocks(n, 256), stream, vs, pts, (const float4*)g_out, n, g_pts, (const uint32_t*)P.order);
    bool scatter = false;
    for (int l = 0; g_vols && l < n_levels; ++l) {
        vs.grad[l] = g_vols[l];
        scatter = scatter || g_vols[l];
    }
    if (scatter) scatter_bricks<false>(layout, vs, P, pts, g_out, nullptr, n, stream);
    return gens_launch_status("gens_lookup_volume_bwd_bricks");
}

extern "C" int gens_lookup_volume_bwd2_bricks(const float* const* vols, const int* dims, int n_levels, int layout, const float* pts,
                                              const float* g_out, const float* gg_pts, const float* const* gg_vols, int64_t n,
                                              float* gg_out, float* const* g_vols2, float* g_pts2, void* scratch, int64_t scratch_bytes, void* stream) {
    LevelSet vs;
    if (int e = gens_fill_levels("gens_lookup_volume_bwd2_bricks", &vs, vols, dims, n_levels)) return e;
    GENS_CHECK_ARG(layout == 0 || layout == 1, GENS_EINVAL, "gens_lookup_volume_bwd2_bricks: bad layout %d", layout);
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_out && gg_pts && gg_out && g_pts2)), GENS_EINVAL, "gens_lookup_volume_bwd2_bricks: null pointer");
    if (n == 0) return 0;
    for (int l = 0; l < n_levels; ++l)
        if (gg_vols) vs.aux[l] = gg_vols[l];
    BrickPlan P;
    if (int e = bricks_sort("gens_lookup_volume_bwd2_bricks", vs, pts, n, scratch, scratch_bytes, stream, &P)) return e;
    DISPATCH_LAYOUT(layout, lookup_bwd2_k, gens_blocks(n, 256), stream, vs, pts, (const float4*)g_out, gg_pts, n, (float4*)gg_out, g_pts2, (const uint32_t*)P.order);
    bool scatter = false;
    for (int l = 0; g_vols2 && l < n_levels; ++l) {
        vs.grad[l] = g_vols2[l];
        scatter = scatter || g_vols2[l];
    }
    if (scatter) scatter_bricks<true>(layout, vs, P, pts, g_out, gg_pts, n, stream);
    return gens_launch_status("gens_lookup_volume_bwd2_bricks");
}

extern "C" int gens_lookup_mask_nearest(const float* const* masks, const int* dims, int n_levels, const float* pts, int64_t n,
                                        uint8_t* valid, float* vals, void* stream) {
    LevelSet ms;
    if (int e = gens_fill_levels("gens_lookup_mask_nearest", &ms, masks, dims, n_levels)) return e;
    GENS_CHECK_ARG(n >= 0 && (n == 0 || pts) && (valid || vals), GENS_EINVAL, "gens_lookup_mask_nearest: null pointer");
    if (n == 0) return 0;
    mask_nearest_k<<<gens_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(ms, pts, n, valid, vals);
    return gens_launch_status("gens_lookup_mask_nearest");
}

// mask (n floats, > 0 = set) -> n bits, 32 voxels per word in C order (one ballot per wavefront)
__global__ __launch_bounds__(256) void pack_mask_bits_k(const float* __restrict__ mask, int64_t n, uint32_t* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(i < n && mask[i] > 0.0f);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && i < n) bits[i >> 5] = (uint32_t)b;
    if (lane == 32 && i < n) bits[i >> 5] = (uint32_t)(b >> 32);
}

extern "C" int gens_pack_mask_bits(const float* mask, int64_t n, uint32_t* bits, void* stream) {
    GENS_CHECK_ARG(mask && bits && n > 0, GENS_EINVAL, "gens_pack_mask_bits: bad argument");
    pack_mask_bits_k<<<gens_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(mask, n, bits);
    return gens_launch_status("gens_pack_mask_bits");
}

extern "C" int gens_ray_points(const float* rays_o, const float* rays_d, const float* z, int64_t n_rays, int n_samples, int mid,
                               float sample_dist, const float* const* masks, const int* dims, int n_levels, int mask_bits,
                               float* pts, uint8_t* valid, void* stream) {
    LevelSet ms;
    ms.n = 0;
    ms.bits = 0;
    if (valid) {
        if (int e = gens_fill_levels("gens_ray_points", &ms, masks, dims, n_levels)) return e;
        ms.bits = mask_bits ? 1 : 0;
    }
    GENS_CHECK_ARG(n_rays >= 0 && n_samples > 0 && (n_rays == 0 || (rays_o && rays_d && z && pts)), GENS_EINVAL,
                   "gens_ray_points: bad argument");
    if (n_rays == 0) return 0;
    int64_t total = n_rays * n_samples;
    ray_points_k<<<gens_blocks(total, 256), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, z, total, n_samples, mid, sample_dist,
                                                                          ms, pts, valid);
    return gens_launch_status("gens_ray_points");
}
