// K4: lookup_feature + compute_angle (/root/reference/models/modules/projector.py:278-349): re-project N sample points
// into the S source views at every pyramid level, bilinearly read features (+RGB at level 0), build the in-frustum
// mask and the IBRNet ray-direction-difference feature -- in ONE launch (the reference issues, per level, an
// inverse, two matmuls, a divide, a grid_sample, a permute, and a final cat).
//
// One thread per (point, source view); pyramid levels are a register loop, every tap is one 16-B texel load from
// the NHWC-packed maps.  The (N, S, 3+4L) row a thread produces is first parked in LDS and then written by the whole
// block as one contiguous, coalesced span (a row is 92 B for L=5 -- direct per-lane stores would touch a new
// 128-B line each).
//
// Quirks kept: no epsilon in the perspective divide and the half-open pixel test (Q3); coordinates are normalised
// with (W-1)/2 but READ with align_corners=False, i.e. pixel = ((g+1)*W-1)/2 (Q6).
#include <stdlib.h>

#include "common.h"

#include "k4_common.h"

#define K4_BLOCK 256
#define K4_MAX_ROW (3 + 4 * GENS_MAX_LEVELS)

template <int NLEV>   // number of pyramid levels when known at compile time (the loads of all levels are then in flight together), 0 = fs.n
__global__ __launch_bounds__(K4_BLOCK) void lookup_feature_fwd_k(MapSet fs, const float4* __restrict__ imgs,
                                                                 const float* __restrict__ w2c, const float* __restrict__ intr,
                                                                 const float* __restrict__ c2w, int nv, const float* __restrict__ pts,
                                                                 int64_t n, float* __restrict__ out, float4* __restrict__ ray_diff,
                                                                 uint8_t* __restrict__ vis, int plain_copy, int xcd_remap, int paired, uint32_t magic) {
    extern __shared__ __attribute__((aligned(16))) float row_lds[];  // K4_BLOCK rows of `row` floats, row stride padded to an odd count
    const int S = nv - 1;
    const int n_lev = NLEV ? NLEV : fs.n;
    const int row = 3 + 4 * n_lev;
    const int stride = row | 1;
    // Workgroups are dealt to the eight XCDs round robin: with block b = its launch index, neighbouring points (consecutive samples of a ray,
    // neighbouring rays) read their shared texel lines through eight different L2s.  xcd_remap (probe switch GENS_K4_XCD_REMAP): XCD x takes
    // the x-th CONTIGUOUS eighth of the blocks instead (b % 8 = XCD is a placement hint, not a contract: any mapping is correct).  Measured
    // (scripts/probe/gather_ab.py, 3.8 M points x 4 views): 686 us against 664 without -- this kernel is bound by its 1.7 GB of row stores
    // and its arithmetic, not by texel traffic, and the contiguous order serialises the store streams of an XCD; OFF.  The same remap in K2's
    // forward (texel-bound) gains 3 % and is on.
    const uint32_t nb = gridDim.x;
    uint32_t blk = blockIdx.x;
    if (xcd_remap) {
        const uint32_t per = nb >> 3;                      // blocks [0, 8 per) are remapped, the remainder keeps its place
        if (blk < 8u * per) blk = (blk & 7u) * per + (blk >> 3);
    }
    const int64_t gid = (int64_t)blk * K4_BLOCK + threadIdx.x;
    const int64_t total = n * S;
    float* mine = row_lds + threadIdx.x * stride;
    // Lanes work in pairs (sample_texel_pair, common.h: two lanes to a 128-byte line in every load), so a lane past the end still serves its
    // partner: it works on the last item and writes nothing to global memory.
    const bool active = gid < total;
    const int64_t g = active ? gid : total - 1;
    const int odd = threadIdx.x & 1;
    {
        int sv;                     // source view index in [1, nv)
        int64_t i;
        if (magic) {                // fewer than 2^29 items: item / S as one multiply-high by magic = ceil(2^32 / S)
            const uint32_t g32 = (uint32_t)g, i32 = __umulhi(g32, magic);
            sv = (int)(g32 - i32 * (uint32_t)S) + 1;
            i = i32;
        } else {
            sv = (int)(g % S) + 1;
            i = g / S;
        }
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        bool inside = true;
        const SrcBase pb = project_src_base(w2c + 16 * sv, intr + 16 * sv, x, y, z);     // once per (point, view): see k4_common.h
        auto level = [&](int l) {
            int h = fs.h[l], w = fs.w[l];
            SrcProj p = project_src_level(pb, exp2f(-(float)l), h, w, fs.cw[l], fs.ch[l], fs.rcw[l], fs.rch[l]);
            inside = inside && p.inside;
            Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
            float4 f;
            PairTaps q;
            if (paired) {
                q = pair_taps(sv * h * w, h, w, t, odd);
                f = sample_texel_pair(fs.data[l], q, t, odd);
            } else {
                f = sample_texel(fs.data[l] + (int64_t)sv * h * w, h, w, 1, 0, t);
            }
            mine[3 + 4 * l] = f.x;
            mine[4 + 4 * l] = f.y;
            mine[5 + 4 * l] = f.z;
            mine[6 + 4 * l] = f.w;
            if (l == 0) {
                float4 c = paired ? sample_texel_pair(imgs, q, t, odd) : sample_texel(imgs + (int64_t)sv * h * w, h, w, 1, 0, t);
                mine[0] = c.x;
                mine[1] = c.y;
                mine[2] = c.z;
            }
        };
        if constexpr (NLEV > 0) {
#pragma unroll
            for (int l = 0; l < NLEV; ++l) level(l);
        } else {
            for (int l = 0; l < fs.n; ++l) level(l);
        }
        // compute_angle (projector.py:278-291)
        float rx = c2w[3] - x, ry = c2w[7] - y, rz = c2w[11] - z;
        float rn = sqrtf(rx * rx + ry * ry + rz * rz) + 1e-6f;
        rx /= rn; ry /= rn; rz /= rn;
        const float* cs = c2w + 16 * sv;
        float sx = cs[3] - x, sy = cs[7] - y, sz = cs[11] - z;
        float sn = sqrtf(sx * sx + sy * sy + sz * sz) + 1e-6f;
        sx /= sn; sy /= sn; sz /= sn;
        float dx = rx - sx, dy = ry - sy, dz = rz - sz;
        float dn = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
        if (active) {
            vis[gid] = inside ? 1 : 0;
            ray_diff[gid] = make_float4(dx / dn, dy / dn, dz / dn, rx * sx + ry * sy + rz * sz);
        }
    }
    __syncthreads();
    // cooperative, coalesced write of this block's rows.  3 + 4 L is odd, so the padded stride IS the row length: the block's rows are one
    // contiguous span in LDS and in the output (which starts 16-byte aligned: 256 rows are a multiple of 16 bytes) -- copied as float4 with
    // streaming stores (the texels, not these rows, are what should stay in L2); the element-wise loop this replaces spent a division and a
    // remainder by `row` per float (23 trips per thread at five levels: as many instructions as the look-up itself).
    int64_t first = (int64_t)blk * K4_BLOCK;
    int rows_here = (int)min((int64_t)K4_BLOCK, total - first);
    int span = rows_here * row;
    float* dst = out + first * row;
    if (plain_copy) {
        for (int e = threadIdx.x; e < span; e += K4_BLOCK) dst[e] = row_lds[(e / row) * stride + (e % row)];
        return;
    }
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int quads = span >> 2;
    for (int e = threadIdx.x; e < quads; e += K4_BLOCK) __builtin_nontemporal_store(((const f4v*)row_lds)[e], (f4v*)dst + e);
    for (int e = 4 * quads + threadIdx.x; e < span; e += K4_BLOCK) dst[e] = row_lds[e];
}

// Backward of the look-up: the bilinear taps of every (point, source view, level) scattered into the maps' gradients.  Bound by the float
// atomics at L2, which are served per REQUEST, not per lane: an instruction whose lanes hit consecutive floats runs at 190 - 280 G atomics/s,
// one whose lanes each hit their own texel at 66 G/s (scripts/probe/atomic_scope_probe.py).  So SIXTEEN lanes share a (point, view) pair --
// lane = (tap, channel): the two taps of a row are 32 contiguous bytes -- and one atomic instruction per level carries four pairs.
__global__ __launch_bounds__(256) void lookup_feature_bwd_k(MapSet fs, float* __restrict__ g_imgs, const float* __restrict__ w2c,
                                                            const float* __restrict__ intr, int nv, const float* __restrict__ pts,
                                                            const float* __restrict__ g_out, const int64_t* __restrict__ index, int64_t n_max,
                                                            const int32_t* __restrict__ n_dev) {
    const int S = nv - 1;
    const int row = 3 + 4 * fs.n;
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int64_t n = n_dev ? min(n_max, (int64_t)n_dev[0]) : n_max;
    if (gid >= n * S) return;
    const int sub = threadIdx.x & 15, tap = sub >> 2, ch = sub & 3, dy = tap >> 1, dx = tap & 1;
    int sv = (int)(gid % S) + 1;
    int64_t i = gid / S;                                   // compact row of g_out; the point itself is pts[index[i]]
    const int64_t src = index ? index[i] : i;
    float x = pts[3 * src], y = pts[3 * src + 1], z = pts[3 * src + 2];
    const float* g = g_out + gid * row;
    for (int l = 0; l < fs.n; ++l) {
        int h = fs.h[l], w = fs.w[l];
        SrcProj p = project_src(w2c + 16 * sv, intr + 16 * sv, exp2f(-(float)l), h, w, fs.cw[l], fs.ch[l], fs.rcw[l], fs.rch[l], x, y, z);
        Taps2 t = bilinear_taps(p.ix, p.iy, h, w);
        const bool ok = tap == 0 ? t.ok00 : tap == 1 ? t.ok01 : tap == 2 ? t.ok10 : t.ok11;
        if (!ok) continue;
        const float wt = tap == 0 ? t.w00 : tap == 1 ? t.w01 : tap == 2 ? t.w10 : t.w11;
        const int64_t off = (((int64_t)sv * h + t.y0 + dy) * w + t.x0 + dx) * 4 + ch;
        if (fs.grad[l]) {
            const float v = g[3 + 4 * l + ch] * wt;
            if (v != 0.0f) atomicAdd(fs.grad[l] + off, v);                       // (masked samples carry exact zeros: nothing to add)
        }
        if (l == 0 && g_imgs && ch < 3) {
            const float v = g[ch] * wt;
            if (v != 0.0f) atomicAdd(g_imgs + off, v);
        }
    }
}

int gens_fill_maps(const char* who, MapSet* ms, const float* const* feats, const int* hw, int n_levels) {
    GENS_CHECK_ARG(hw, GENS_EINVAL, "%s: null hw table", who);
    GENS_CHECK_ARG(n_levels > 0 && n_levels <= GENS_MAX_LEVELS, GENS_ELIMIT, "%s: n_levels=%d not in 1..%d", who, n_levels,
                   GENS_MAX_LEVELS);
    ms->n = n_levels;
    for (int l = 0; l < GENS_MAX_LEVELS; ++l) {
        ms->data[l] = nullptr;
        ms->grad[l] = nullptr;
        ms->h[l] = ms->w[l] = 2;
        ms->cw[l] = ms->ch[l] = 0.5f;
        ms->rcw[l] = ms->rch[l] = 2.0f;
    }
    for (int l = 0; l < n_levels; ++l) {
        GENS_CHECK_ARG(hw[2 * l] > 1 && hw[2 * l + 1] > 1, GENS_EINVAL, "%s: level %d map smaller than 2x2", who, l);
        if (feats) {
            GENS_CHECK_ARG(feats[l], GENS_EINVAL, "%s: level %d is null", who, l);
            ms->data[l] = (const float4*)feats[l];
        }
        ms->h[l] = hw[2 * l];
        ms->w[l] = hw[2 * l + 1];
        ms->cw[l] = (float)(ms->w[l] - 1) / 2.0f;
        ms->ch[l] = (float)(ms->h[l] - 1) / 2.0f;
        ms->rcw[l] = 1.0f / ms->cw[l];
        ms->rch[l] = 1.0f / ms->ch[l];
    }
    return 0;
}

extern "C" int gens_lookup_feature_fwd(const float* const* feats, const int* hw, int n_levels, const float* imgs, const float* w2c,
                                       const float* intr, const float* c2w, int nv, const float* pts, int64_t n, float* out,
                                       float* ray_diff, uint8_t* vis, void* stream) {
    MapSet fs;
    GENS_CHECK_ARG(feats, GENS_EINVAL, "gens_lookup_feature_fwd: null feature table");
    if (int e = gens_fill_maps("gens_lookup_feature_fwd", &fs, feats, hw, n_levels)) return e;
    GENS_CHECK_ARG(nv >= 2 && nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "gens_lookup_feature_fwd: nv=%d not in 2..%d", nv, GENS_MAX_VIEWS);
    GENS_CHECK_ARG(imgs && w2c && intr && c2w, GENS_EINVAL, "gens_lookup_feature_fwd: null camera / image pointer");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && out && ray_diff && vis)), GENS_EINVAL, "gens_lookup_feature_fwd: null pts / output");
    if (n == 0) return 0;
    int row = 3 + 4 * n_levels;
    // (texel indices in 32 bits in the paired read; GENS_K4_NO_PAIRS: the lane-per-item read, for A/B runs)
    const int paired = getenv("GENS_K4_NO_PAIRS") == nullptr && (int64_t)nv * hw[0] * hw[1] < (1ll << 31);
    size_t lds = (size_t)K4_BLOCK * (row | 1) * sizeof(float);
    const dim3 grid = gens_blocks(n * (nv - 1), K4_BLOCK);
    // item / S as one multiply-high by m = ceil(2^32 / S) is exact while item * (m S - 2^32) < 2^32 (S up to 15: m S - 2^32 reaches S - 1, so a fixed
    // 2^29 bound on the items is not enough -- S = 15 goes wrong from item 306 783 389); otherwise the kernel divides in 64 bits (magic = 0)
    uint32_t magic = 0u;                                                                                                                     // (S = 1: no magic)
    if (nv > 2) {
        const uint64_t s = (uint64_t)(nv - 1), m = ((1ull << 32) + s - 1) / s, excess = m * s - (1ull << 32), items = (uint64_t)n * s;
        if (items < (1ull << 31) && items * excess < (1ull << 32)) magic = (uint32_t)m;
    }
    const int plain = getenv("GENS_K4_PLAIN_COPY") != nullptr, remap = getenv("GENS_K4_XCD_REMAP") != nullptr;
#define K4_LAUNCH(NLEV) lookup_feature_fwd_k<NLEV><<<grid, K4_BLOCK, lds, (hipStream_t)stream>>>(fs, (const float4*)imgs, w2c, intr, c2w, nv, pts, n, out, (float4*)ray_diff, vis, plain, remap, paired, magic)
    if (n_levels == 5 && !getenv("GENS_K4_NO_UNROLL")) K4_LAUNCH(5);
    else if (n_levels == 3 && !getenv("GENS_K4_NO_UNROLL")) K4_LAUNCH(3);
    else K4_LAUNCH(0);
#undef K4_LAUNCH
    return gens_launch_status("gens_lookup_feature_fwd");
}

static int lookup_feature_bwd_run(const int* hw, int n_levels, const float* w2c, const float* intr, int nv, const float* pts, const float* g_out,
                                  const int64_t* index, int64_t n, const int32_t* n_device, float* const* g_feats, float* g_imgs, void* stream);

extern "C" int gens_lookup_feature_bwd(const int* hw, int n_levels, const float* w2c, const float* intr, int nv, const float* pts,
                                       const float* g_out, int64_t n, float* const* g_feats, float* g_imgs, void* stream) {
    return lookup_feature_bwd_run(hw, n_levels, w2c, intr, nv, pts, g_out, nullptr, n, nullptr, g_feats, g_imgs, stream);
}

// g_out (n, S, 3 + 4 L) stays COMPACT (row i = the i-th selected point); the point itself is pts[index[i]], only min(n, *n_device) rows exist
extern "C" int gens_lookup_feature_bwd_idx(const int* hw, int n_levels, const float* w2c, const float* intr, int nv, const float* pts,
                                           const float* g_out, const int64_t* index, int64_t n, const int32_t* n_device, float* const* g_feats,
                                           float* g_imgs, void* stream) {
    return lookup_feature_bwd_run(hw, n_levels, w2c, intr, nv, pts, g_out, index, n, n_device, g_feats, g_imgs, stream);
}

static int lookup_feature_bwd_run(const int* hw, int n_levels, const float* w2c, const float* intr, int nv, const float* pts, const float* g_out,
                                  const int64_t* index, int64_t n, const int32_t* n_device, float* const* g_feats, float* g_imgs, void* stream) {
    MapSet fs;
    if (int e = gens_fill_maps("gens_lookup_feature_bwd", &fs, nullptr, hw, n_levels)) return e;
    GENS_CHECK_ARG(nv >= 2 && nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "gens_lookup_feature_bwd: nv=%d not in 2..%d", nv, GENS_MAX_VIEWS);
    GENS_CHECK_ARG(w2c && intr, GENS_EINVAL, "gens_lookup_feature_bwd: null camera pointer");
    GENS_CHECK_ARG(g_feats || g_imgs, GENS_EINVAL, "gens_lookup_feature_bwd: no output requested");
    GENS_CHECK_ARG(n >= 0 && (n == 0 || (pts && g_out)), GENS_EINVAL, "gens_lookup_feature_bwd: null pts / g_out");
    if (n == 0) return 0;
    if (g_feats)
        for (int l = 0; l < n_levels; ++l) fs.grad[l] = g_feats[l];
    lookup_feature_bwd_k<<<gens_blocks(n * (nv - 1) * 16, 256), 256, 0, (hipStream_t)stream>>>(fs, g_imgs, w2c, intr, nv, pts, g_out, index, n, n_device);
    return gens_launch_status("gens_lookup_feature_bwd");
}
