// K20: the reference's sampler boundary in FULL generality -- grid_sample_2d / grid_sample_3d with padding_mode in {zeros, border},
// align_corners in {true, false}, any batch and channel count, value, first backward and the backward of the backward:
//
//   forward     cuda_gridsample.py:7-14 -> F.grid_sample (aten::grid_sampler_2d / _3d, bilinear)
//   backward    aten::grid_sampler_2d_backward / _3d_backward through cuda_gridsample.py:38-50, 94-108
//   backward^2  gridsample_grad2.grad2_2d / grad2_3d (gridsample_cuda.cpp:26-56; kernels gridsample_cuda.cu:27-210, 212-533)
//
// The hot path never comes here: lookup_volume's calls (batch 1, four-channel levels, zeros padding, align_corners=True, 3-D) run on K2
// (k2_lookup.hip), which serves all pyramid levels in one launch from packed texels.  K20 is what makes the drop-in for
// cuda_gridsample.py COMPLETE: everything that file exports, in the tensor layouts it takes (contiguous NCHW / NCDHW, grid (N, ..., DIM)
// with the last axis (x, y[, z]) indexing (W, H[, D])).
//
// One thread per output location, the channels in a loop (the corner indices, weights and their derivatives are computed once per
// location); gradients with respect to the input are float atomics into a buffer the caller zeroed.  Written from the mathematics:
// with the source index s_a = unnormalize(g_a) per axis a (clipped to [0, size-1] under border padding, which also zeroes d s_a / d g_a
// outside), corner k = (k_0, ..) has weight w_k = prod_a u_a(k_a), u_a(0) = 1 - t_a, u_a(1) = t_a, t_a = s_a - floor(s_a); then
//   out[c]     = sum_k V[c][k] w_k                                 (corners outside the input contribute nothing)
//   dV[c][k]  += gO[c] w_k ,   dg_a = m_a sum_c gO[c] sum_k V[c][k] dw_k/ds_a           (m_a = d s_a / d g_a)
// and for the cotangents (ggV on dV, ggG on dg) of the first backward, with e_a = ggG_a m_a and T_k = sum_a e_a dw_k/ds_a:
//   ggO[c]     = sum_k ggV[c][k] w_k + V[c][k] T_k
//   dV'[c][k] += gO[c] T_k
//   dg'_a      = m_a sum_c gO[c] sum_k ( ggV[c][k] dw_k/ds_a + V[c][k] sum_{b != a} e_b d2w_k/ds_a ds_b )
// (w_k is linear in every s_a, so d2w_k/ds_a^2 = 0; the clip has no second derivative either).
#include "common.h"

#define GS_ZEROS 0
#define GS_BORDER 1

struct GsGeom {
    int n, c;
    int size[3];          // input extent per GRID axis: size[0] = W (x), size[1] = H (y), size[2] = D (z)
    int64_t stride[3];    // input stride (floats) per grid axis: x -> 1, y -> W, z -> H W
    int64_t chan;         // floats per channel plane
    int64_t n_out;        // output locations per batch item
    int padding, align;
};

// source index of a normalised coordinate and d(source index) / d(coordinate) (ATen: grid_sampler_compute_source_index_set_grad)
__device__ __forceinline__ float gs_source_index(float g, int size, int padding, int align, float& mult) {
    float s;
    if (align) {
        mult = (float)(size - 1) / 2.0f;
        s = (g + 1.0f) / 2.0f * (float)(size - 1);
    } else {
        mult = (float)size / 2.0f;
        s = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
    }
    if (padding == GS_BORDER) {               // clip_coordinates_set_grad: the clipped coordinate no longer moves with g
        if (s <= 0.0f) { s = 0.0f; mult = 0.0f; }
        else if (s >= (float)(size - 1)) { s = (float)(size - 1); mult = 0.0f; }
    }
    return s;
}

template <int DIM>
struct GsCell {
    int i0[DIM];
    float u0[DIM], u1[DIM], mult[DIM];
    bool in0[DIM], in1[DIM];
};

template <int DIM>
__device__ __forceinline__ GsCell<DIM> gs_cell(const GsGeom& g, const float* __restrict__ grid_pt) {
    GsCell<DIM> cell;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        const float s = gs_source_index(grid_pt[a], g.size[a], g.padding, g.align, cell.mult[a]);
        float f = floorf(s);
        f = fminf(fmaxf(f, -2.0f), (float)g.size[a] + 1.0f);      // far-away / infinite coordinates: every corner out of bounds, no int overflow
        cell.i0[a] = (int)f;
        cell.u0[a] = (f + 1.0f) - s;                              // (ix_se - ix): ATen's operand order
        cell.u1[a] = s - f;
        cell.in0[a] = cell.i0[a] >= 0 && cell.i0[a] < g.size[a];
        cell.in1[a] = cell.i0[a] + 1 >= 0 && cell.i0[a] + 1 < g.size[a];
        if (!(s == s)) { cell.in0[a] = cell.in1[a] = false; cell.u0[a] = cell.u1[a] = 0.0f; }     // NaN coordinate
    }
    return cell;
}

// corner k of the cell: in-bounds flag, linear offset inside a channel plane, weight and its first derivatives per axis
template <int DIM>
struct GsCorner {
    bool ok;
    int64_t off;
    float w, dw[DIM];
};
template <int DIM>
__device__ __forceinline__ GsCorner<DIM> gs_corner(const GsGeom& g, const GsCell<DIM>& cell, int k) {
    GsCorner<DIM> c;
    c.ok = true;
    c.off = 0;
    c.w = 1.0f;
    float u[DIM], sgn[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        const int bit = (k >> a) & 1;
        c.ok = c.ok && (bit ? cell.in1[a] : cell.in0[a]);
        c.off += (int64_t)(cell.i0[a] + bit) * g.stride[a];
        u[a] = bit ? cell.u1[a] : cell.u0[a];
        sgn[a] = bit ? 1.0f : -1.0f;
        c.w *= u[a];
    }
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        float d = sgn[a];
#pragma unroll
        for (int b = 0; b < DIM; ++b)
            if (b != a) d *= u[b];
        c.dw[a] = d;
    }
    return c;
}
// d2 w_k / ds_a ds_b (a != b)
template <int DIM>
__device__ __forceinline__ float gs_d2w(const GsCell<DIM>& cell, int k, int a, int b) {
    float d = (((k >> a) & 1) ? 1.0f : -1.0f) * (((k >> b) & 1) ? 1.0f : -1.0f);
#pragma unroll
    for (int e = 0; e < DIM; ++e)
        if (e != a && e != b) d *= ((k >> e) & 1) ? cell.u1[e] : cell.u0[e];
    return d;
}

template <int DIM>
__global__ __launch_bounds__(256) void grid_sample_fwd_k(GsGeom g, const float* __restrict__ input, const float* __restrict__ grid, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)g.n * g.n_out) return;
    const int64_t b = idx / g.n_out, p = idx - b * g.n_out;
    const GsCell<DIM> cell = gs_cell<DIM>(g, grid + idx * DIM);
    GsCorner<DIM> cr[1 << DIM];
#pragma unroll
    for (int k = 0; k < (1 << DIM); ++k) cr[k] = gs_corner<DIM>(g, cell, k);
    const float* in_b = input + b * g.c * g.chan;
    float* out_b = out + b * g.c * g.n_out + p;
    for (int ch = 0; ch < g.c; ++ch) {
        const float* plane = in_b + ch * g.chan;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < (1 << DIM); ++k)
            if (cr[k].ok) acc += plane[cr[k].off] * cr[k].w;
        out_b[ch * g.n_out] = acc;
    }
}

template <int DIM>
__global__ __launch_bounds__(256) void grid_sample_bwd_k(GsGeom g, const float* __restrict__ g_out, const float* __restrict__ input,
                                                         const float* __restrict__ grid, float* __restrict__ g_input, float* __restrict__ g_grid) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)g.n * g.n_out) return;
    const int64_t b = idx / g.n_out, p = idx - b * g.n_out;
    const GsCell<DIM> cell = gs_cell<DIM>(g, grid + idx * DIM);
    GsCorner<DIM> cr[1 << DIM];
#pragma unroll
    for (int k = 0; k < (1 << DIM); ++k) cr[k] = gs_corner<DIM>(g, cell, k);
    float gs[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) gs[a] = 0.0f;
    for (int ch = 0; ch < g.c; ++ch) {
        const int64_t plane = (b * g.c + ch) * g.chan;
        const float go = g_out[(b * g.c + ch) * g.n_out + p];
#pragma unroll
        for (int k = 0; k < (1 << DIM); ++k) {
            if (!cr[k].ok) continue;
            if (g_input) atomicAdd(g_input + plane + cr[k].off, go * cr[k].w);
            if (g_grid) {
                const float v = input[plane + cr[k].off] * go;
#pragma unroll
                for (int a = 0; a < DIM; ++a) gs[a] += v * cr[k].dw[a];
            }
        }
    }
    if (g_grid)
#pragma unroll
        for (int a = 0; a < DIM; ++a) g_grid[idx * DIM + a] = gs[a] * cell.mult[a];
}

template <int DIM>
__global__ __launch_bounds__(256) void grid_sample_bwd2_k(GsGeom g, const float* __restrict__ gg_input, const float* __restrict__ gg_grid,
                                                          const float* __restrict__ g_out, const float* __restrict__ input,
                                                          const float* __restrict__ grid, float* __restrict__ gg_out, float* __restrict__ g_input,
                                                          float* __restrict__ g_grid) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)g.n * g.n_out) return;
    const int64_t b = idx / g.n_out, p = idx - b * g.n_out;
    const GsCell<DIM> cell = gs_cell<DIM>(g, grid + idx * DIM);
    GsCorner<DIM> cr[1 << DIM];
    float e[DIM], T[1 << DIM], gs[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        e[a] = gg_grid[idx * DIM + a] * cell.mult[a];
        gs[a] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < (1 << DIM); ++k) {
        cr[k] = gs_corner<DIM>(g, cell, k);
        float t = 0.0f;
#pragma unroll
        for (int a = 0; a < DIM; ++a) t += e[a] * cr[k].dw[a];
        T[k] = t;
    }
    for (int ch = 0; ch < g.c; ++ch) {
        const int64_t plane = (b * g.c + ch) * g.chan;
        const float go = g_out[(b * g.c + ch) * g.n_out + p];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < (1 << DIM); ++k) {
            if (!cr[k].ok) continue;
            const float v = input[plane + cr[k].off];
            acc += v * T[k];
            if (g_input) atomicAdd(g_input + plane + cr[k].off, go * T[k]);
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                float mixed = 0.0f;
#pragma unroll
                for (int bb = 0; bb < DIM; ++bb)
                    if (bb != a) mixed += e[bb] * gs_d2w<DIM>(cell, k, a, bb);
                gs[a] += go * v * mixed;
            }
            if (gg_input) {
                const float g2 = gg_input[plane + cr[k].off];
                acc += g2 * cr[k].w;
#pragma unroll
                for (int a = 0; a < DIM; ++a) gs[a] += go * g2 * cr[k].dw[a];
            }
        }
        gg_out[(b * g.c + ch) * g.n_out + p] = acc;
    }
#pragma unroll
    for (int a = 0; a < DIM; ++a) g_grid[idx * DIM + a] = gs[a] * cell.mult[a];
}

static int gs_geom(const char* who, int ndim, int n, int c, const int* in_size, int64_t n_out, int padding, int align, GsGeom& g) {
    GENS_CHECK_ARG(ndim == 2 || ndim == 3, GENS_EINVAL, "%s: ndim = %d (2 or 3)", who, ndim);
    GENS_CHECK_ARG(in_size, GENS_EINVAL, "%s: null size table", who);
    GENS_CHECK_ARG(n >= 0 && c >= 0 && n_out >= 0, GENS_EINVAL, "%s: negative extent", who);
    GENS_CHECK_ARG(padding == GS_ZEROS || padding == GS_BORDER, GENS_EINVAL, "%s: padding_mode %d (0 = zeros, 1 = border; the reference asserts the same two, cuda_gridsample.py:8,13)", who, padding);
    for (int a = 0; a < ndim; ++a) GENS_CHECK_ARG(in_size[a] > 0, GENS_EINVAL, "%s: input extent %d is %d", who, a, in_size[a]);
    g.n = n;
    g.c = c;
    // in_size is given as the tensor's spatial shape, slowest first: (H, W) or (D, H, W); grid axis 0 (x) is the FASTEST tensor axis
    int64_t stride = 1;
    for (int a = 0; a < 3; ++a) { g.size[a] = 1; g.stride[a] = 0; }
    for (int a = 0; a < ndim; ++a) {
        g.size[a] = in_size[ndim - 1 - a];
        g.stride[a] = stride;
        stride *= g.size[a];
    }
    g.chan = stride;
    g.n_out = n_out;
    g.padding = padding;
    g.align = align ? 1 : 0;
    return 0;
}

#define GS_LAUNCH(KERNEL, ndim, total, stream, ...)                                                                  \
    do {                                                                                                             \
        if ((ndim) == 2) KERNEL<2><<<gens_blocks((total), 256), 256, 0, (hipStream_t)(stream)>>>(__VA_ARGS__);      \
        else KERNEL<3><<<gens_blocks((total), 256), 256, 0, (hipStream_t)(stream)>>>(__VA_ARGS__);                  \
    } while (0)

extern "C" int gens_grid_sample_fwd(const float* input, const float* grid, int ndim, int n, int c, const int* in_size, int64_t n_out,
                                    int padding_mode, int align_corners, float* out, void* stream) {
    GsGeom g;
    if (int e = gs_geom("gens_grid_sample_fwd", ndim, n, c, in_size, n_out, padding_mode, align_corners, g)) return e;
    const int64_t total = (int64_t)n * n_out;
    if (total == 0 || c == 0) return 0;
    GENS_CHECK_ARG(input && grid && out, GENS_EINVAL, "gens_grid_sample_fwd: null pointer");
    GS_LAUNCH(grid_sample_fwd_k, ndim, total, stream, g, input, grid, out);
    return gens_launch_status("gens_grid_sample_fwd");
}

extern "C" int gens_grid_sample_bwd(const float* grad_out, const float* input, const float* grid, int ndim, int n, int c, const int* in_size,
                                    int64_t n_out, int padding_mode, int align_corners, float* grad_input, float* grad_grid, void* stream) {
    GsGeom g;
    if (int e = gs_geom("gens_grid_sample_bwd", ndim, n, c, in_size, n_out, padding_mode, align_corners, g)) return e;
    GENS_CHECK_ARG(grad_input || grad_grid, GENS_EINVAL, "gens_grid_sample_bwd: no output requested");
    const int64_t total = (int64_t)n * n_out;
    if (total == 0) return 0;
    GENS_CHECK_ARG(grid && (c == 0 || (grad_out && input)), GENS_EINVAL, "gens_grid_sample_bwd: null pointer");
    GS_LAUNCH(grid_sample_bwd_k, ndim, total, stream, g, grad_out, input, grid, grad_input, grad_grid);
    return gens_launch_status("gens_grid_sample_bwd");
}

extern "C" int gens_grid_sample_bwd2(const float* gg_input, const float* gg_grid, const float* grad_out, const float* input, const float* grid,
                                     int ndim, int n, int c, const int* in_size, int64_t n_out, int padding_mode, int align_corners,
                                     float* gg_out, float* grad_input, float* grad_grid, void* stream) {
    GsGeom g;
    if (int e = gs_geom("gens_grid_sample_bwd2", ndim, n, c, in_size, n_out, padding_mode, align_corners, g)) return e;
    const int64_t total = (int64_t)n * n_out;
    if (total == 0) return 0;
    GENS_CHECK_ARG(gg_grid && grid && grad_grid && (c == 0 || (grad_out && input && gg_out)), GENS_EINVAL, "gens_grid_sample_bwd2: null pointer");
    GS_LAUNCH(grid_sample_bwd2_k, ndim, total, stream, g, gg_input, gg_grid, grad_out, input, grid, gg_out, grad_input, grad_grid);
    return gens_launch_status("gens_grid_sample_bwd2");
}
