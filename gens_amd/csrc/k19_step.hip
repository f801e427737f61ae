// K19: the step-boundary kernels of a TRAINING step.  A 512-ray step of the reference is ~2 400 PyTorch launches; after K17 / K18 took the
// networks, ~300 of the ~530 launches left were scalar-sized torch glue around the hot-path kernels (camera inverses through a batched LU,
// one layout pass per map, fills / index_put around the masked evaluation, the per-level total-variation epilogue, ...), each shorter than
// its own launch.  The entry points below do that work in one launch each, on the device, without host synchronisation.
//   gens_scene_setup     torch.inverse(c2ws), the per-level intrinsics, inverse(c2ws[0,:3,:3]), inverse(intrs)[0,:3,:3]
//                        (volume.py:24-25,34; projector.py:322,364; implicit_surface.py:242,245)
//   gens_pack_maps       gens_pack_nchw for all maps of a scene (images + feature pyramid);  gens_unpack_maps its adjoint
//   gens_compact_points  the masked evaluation's index list (implicit_surface.py:121-124,174-177,484-497) for ray samples, random points and
//                        pseudo points together, with the default values of the unselected rows written in the same launch
//   gens_tv_levels_*     tv_regularization (implicit_surface.py:135-150) of all levels: partial sums, then one finishing workgroup
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// scene set-up
// ---------------------------------------------------------------------------------------------------------------
// Inverse of an N x N row-major float32 matrix: Gauss-Jordan with partial pivoting in float64, rounded once to float32 (within half an
// ulp of the exact inverse for the well-conditioned camera matrices of the path; torch.inverse's float32 LU is a few ulp away from it on
// either side).  Returns false for a singular matrix (an exactly zero pivot column, as LAPACK's info > 0) and writes NaN.
template <int N>
__device__ bool invert_f64(const float* __restrict__ a, int lda, float* __restrict__ out, int ldo) {
    double m[N][2 * N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) {
            m[r][c] = (double)a[r * lda + c];
            m[r][N + c] = r == c ? 1.0 : 0.0;
        }
    bool ok = true;
#pragma unroll
    for (int col = 0; col < N; ++col) {
        int piv = col;
        double best = fabs(m[col][col]);
#pragma unroll
        for (int r = col + 1; r < N; ++r)
            if (fabs(m[r][col]) > best) { best = fabs(m[r][col]); piv = r; }
        if (!(best > 0.0)) { ok = false; break; }
#pragma unroll
        for (int r = 0; r < N; ++r)
            if (r == piv && piv != col) {
#pragma unroll
                for (int c = 0; c < 2 * N; ++c) { const double t = m[col][c]; m[col][c] = m[r][c]; m[r][c] = t; }
            }
        const double inv = 1.0 / m[col][col];
#pragma unroll
        for (int c = 0; c < 2 * N; ++c) m[col][c] *= inv;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (r == col) continue;
            const double f = m[r][col];
#pragma unroll
            for (int c = 0; c < 2 * N; ++c) m[r][c] -= f * m[col][c];
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) out[r * ldo + c] = ok ? (float)m[r][N + c] : __builtin_nanf("");
    return ok;
}

// cams layout (floats): [w2c nv x 16][ks GENS_MAX_LEVELS x nv x 16][rot_inv 9 (+3 pad)][kinv_ref 9 (+3 pad)][status 1 (int32 bits)]
extern "C" int64_t gens_scene_cams_floats(int nv) { return (int64_t)nv * 16 * (1 + GENS_MAX_LEVELS) + 12 + 12 + 4; }

__global__ __launch_bounds__(64) void scene_setup_k(const float* __restrict__ c2w, const float* __restrict__ intr, int nv, float* __restrict__ cams) {
    const int t = threadIdx.x;
    float* w2c = cams;
    float* ks = cams + (int64_t)nv * 16;
    float* rot = ks + (int64_t)GENS_MAX_LEVELS * nv * 16;
    float* kinv = rot + 12;
    int* status = (int*)(kinv + 12);
    if (t == 0) *status = 0;
    __syncthreads();
    bool ok = true;
    if (t < nv) {
        ok = invert_f64<4>(c2w + 16 * t, 4, w2c + 16 * t, 4);
        float s = 1.0f;
        for (int l = 0; l < GENS_MAX_LEVELS; ++l) {               // rows 0-1 times 0.5^l, the product the reference forms per level (Q2)
            float* k = ks + ((int64_t)l * nv + t) * 16;
            for (int e = 0; e < 16; ++e) k[e] = e < 8 ? intr[16 * t + e] * s : intr[16 * t + e];
            s *= 0.5f;
        }
    } else if (t == nv) {
        ok = invert_f64<3>(c2w, 4, rot, 3);                        // inverse(c2ws[0, :3, :3])   (implicit_surface.py:242,245)
    } else if (t == nv + 1) {
        float full[16];
        ok = invert_f64<4>(intr, 4, full, 4);                      // inverse(intrinsics)[0, :3, :3]   (projector.py:364)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) kinv[3 * r + c] = full[4 * r + c];
    }
    if (!ok) atomicOr(status, 1);
}

extern "C" int gens_scene_setup(const float* c2ws, const float* intrs, int nv, float* cams, void* stream) {
    GENS_CHECK_ARG(c2ws && intrs && cams, GENS_EINVAL, "gens_scene_setup: null pointer");
    GENS_CHECK_ARG(nv >= 1 && nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "gens_scene_setup: nv=%d not in 1..%d", nv, GENS_MAX_VIEWS);
    scene_setup_k<<<1, 64, 0, (hipStream_t)stream>>>(c2ws, intrs, nv, cams);
    return gens_launch_status("gens_scene_setup");
}

// ---------------------------------------------------------------------------------------------------------------
// all maps of a scene to texels in one launch (and back)
// ---------------------------------------------------------------------------------------------------------------
#define GENS_MAX_MAPS 8
struct MapPack {
    const float* src[GENS_MAX_MAPS];
    float* dst[GENS_MAX_MAPS];
    int c[GENS_MAX_MAPS];
    int64_t hw[GENS_MAX_MAPS];
    int64_t first[GENS_MAX_MAPS + 1];      // first work item of map k (pack: texels n * hw * q4; unpack: floats n * c * hw)
    int n_maps;
};

__global__ __launch_bounds__(256) void pack_maps_k(MapPack P) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= P.first[P.n_maps]) return;
    int k = 0;
#pragma unroll
    for (int j = 1; j < GENS_MAX_MAPS; ++j) k += (j < P.n_maps && gid >= P.first[j]) ? 1 : 0;
    const int64_t i = gid - P.first[k];
    const int c = P.c[k], q4 = (c + 3) / 4;
    const int64_t hw = P.hw[k];
    const int q = (int)(i % q4);
    const int64_t pix = (i / q4) % hw, img = i / (q4 * hw);
    const float* s = P.src[k] + (img * c + (int64_t)q * 4) * hw + pix;
    float4 v;
    v.x = (q * 4 + 0 < c) ? s[0] : 0.0f;
    v.y = (q * 4 + 1 < c) ? s[hw] : 0.0f;
    v.z = (q * 4 + 2 < c) ? s[2 * hw] : 0.0f;
    v.w = (q * 4 + 3 < c) ? s[3 * hw] : 0.0f;
    ((float4*)P.dst[k])[i] = v;
}

__global__ __launch_bounds__(256) void unpack_maps_k(MapPack P) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= P.first[P.n_maps]) return;
    int k = 0;
#pragma unroll
    for (int j = 1; j < GENS_MAX_MAPS; ++j) k += (j < P.n_maps && gid >= P.first[j]) ? 1 : 0;
    const int64_t i = gid - P.first[k];
    const int c = P.c[k], cpad = 4 * ((c + 3) / 4);
    const int64_t hw = P.hw[k];
    const int64_t pix = i % hw;
    const int ch = (int)((i / hw) % c);
    const int64_t img = i / (hw * c);
    P.dst[k][i] = P.src[k][(img * hw + pix) * cpad + ch];
}

static int fill_map_pack(const char* who, MapPack* P, const float* const* src, float* const* dst, const int* nchw, int n_maps, bool unpack) {
    GENS_CHECK_ARG(src && dst && nchw, GENS_EINVAL, "%s: null table", who);
    GENS_CHECK_ARG(n_maps >= 1 && n_maps <= GENS_MAX_MAPS, GENS_ELIMIT, "%s: %d maps, at most %d", who, n_maps, GENS_MAX_MAPS);
    P->n_maps = n_maps;
    P->first[0] = 0;
    for (int k = 0; k < n_maps; ++k) {
        const int n = nchw[4 * k], c = nchw[4 * k + 1], h = nchw[4 * k + 2], w = nchw[4 * k + 3];
        GENS_CHECK_ARG(src[k] && dst[k] && n > 0 && c > 0 && h > 0 && w > 0, GENS_EINVAL, "%s: bad map %d", who, k);
        GENS_CHECK_ARG(((uintptr_t)(unpack ? (const void*)src[k] : (const void*)dst[k]) & 15) == 0, GENS_EINVAL, "%s: texels of map %d are not 16-byte aligned", who, k);
        P->src[k] = src[k];
        P->dst[k] = dst[k];
        P->c[k] = c;
        P->hw[k] = (int64_t)h * w;
        P->first[k + 1] = P->first[k] + (unpack ? (int64_t)n * c * h * w : (int64_t)n * h * w * ((c + 3) / 4));
    }
    for (int k = n_maps; k < GENS_MAX_MAPS; ++k) { P->src[k] = nullptr; P->dst[k] = nullptr; P->c[k] = 1; P->hw[k] = 1; P->first[k + 1] = P->first[n_maps]; }
    return 0;
}

extern "C" int gens_pack_maps(const float* const* src, float* const* dst, const int* nchw, int n_maps, void* stream) {
    MapPack P;
    if (int e = fill_map_pack("gens_pack_maps", &P, src, dst, nchw, n_maps, false)) return e;
    pack_maps_k<<<gens_blocks(P.first[n_maps], 256), 256, 0, (hipStream_t)stream>>>(P);
    return gens_launch_status("gens_pack_maps");
}

extern "C" int gens_unpack_maps(const float* const* src, float* const* dst, const int* nchw, int n_maps, void* stream) {
    MapPack P;
    if (int e = fill_map_pack("gens_unpack_maps", &P, src, dst, nchw, n_maps, true)) return e;
    unpack_maps_k<<<gens_blocks(P.first[n_maps], 256), 256, 0, (hipStream_t)stream>>>(P);
    return gens_launch_status("gens_unpack_maps");
}

// ---------------------------------------------------------------------------------------------------------------
// the index list of a step's masked evaluation, in ONE launch of one workgroup
// ---------------------------------------------------------------------------------------------------------------
// rows [0, n0): ray samples, flags from gens_ray_points (no flag set -> the first min(10, n0) rows, Q7);
// rows [n0, n0 + n1): always selected (the 1 024 random points, implicit_surface.py:256-257);
// rows [n0 + n1, n): pseudo points with their own flags (:484-497; none set -> counts[2] = 0 and the caller raises).
// idx: the selected rows in increasing order; counts = {all selected, selected ray samples, selected pseudo points}.
// The same launch writes what the reference's dense tensors hold for the rows the networks never see (Q8: sdf 100 for ray samples -- 0 for
// pseudo points, implicit_surface.py:497 --, gradient / smooth / colour 0, no visible source view), and the two scalars of a render_core
// call that are reductions or functions of a parameter: max(z_vals) (:301) and inv_s = clip(exp(10 variance), 1e-6, 1e6) (:206).
struct StepFill {
    float *y, *g, *s;          // (n), (n, 3), (n, 3) or NULL
    float* rgb;                // (n0, 3) or NULL
    uint8_t* vis;              // (n0, n_src) or NULL
    int n_src;
    const float* z;            // (nz) ray depths or NULL
    int64_t nz;
    const float* variance;     // (1) or NULL
    float* scalars;            // [z_max, inv_s, 1 / inv_s, inv_s inside the clip range ? 1 : 0]
};
// Two launches: per-workgroup counts (+ partial maxima of z), then -- every workgroup adding up the counts before its own -- the ordered
// write.  (One workgroup doing all of it took 127 us for a step's 68 k rows: longer than the compositing it serves.)
#define CP_BLOCK 256
#define CP_PER 8                       // rows per thread
#define CP_ROWS (CP_BLOCK * CP_PER)    // rows per workgroup
#define CP_MAX_BLOCKS 8192             // n < 2^24

__device__ __forceinline__ uint32_t cp_flags(const uint8_t* __restrict__ valid, int64_t first, int64_t n) {
    uint32_t bits = 0;
    if (first + CP_PER <= n && (((uintptr_t)(valid + first)) & 7) == 0) {
        const uint2 q = *(const uint2*)(valid + first);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bits |= ((q.x >> (8 * k)) & 0xffu) ? (1u << k) : 0u;
            bits |= ((q.y >> (8 * k)) & 0xffu) ? (1u << (4 + k)) : 0u;
        }
    } else {
        for (int k = 0; k < CP_PER && first + k < n; ++k) bits |= valid[first + k] ? (1u << k) : 0u;
    }
    return bits;
}

__global__ __launch_bounds__(CP_BLOCK) void compact_points_count_k(const uint8_t* __restrict__ valid, int64_t n0, int64_t n1, int64_t n,
                                                                   int32_t* __restrict__ blk, const float* __restrict__ z, int64_t nz,
                                                                   float* __restrict__ zpart) {
    __shared__ int red[3][CP_BLOCK / 64];
    __shared__ float zred[CP_BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t first = (int64_t)blockIdx.x * CP_ROWS + (int64_t)t * CP_PER;
    const uint32_t bits = first < n ? cp_flags(valid, first, n) : 0u;
    int c_ray = 0, c_other = 0, c_pseudo = 0;
    for (int k = 0; k < CP_PER; ++k) {
        const int64_t i = first + k;
        const int v = (i < n && ((bits >> k) & 1u)) ? 1 : 0;
        if (i < n0) c_ray += v;
        else if (i < n0 + n1) c_other += i < n ? 1 : 0;
        else { c_other += v; c_pseudo += v; }
    }
    const float f0 = wave_sum((float)c_ray), f1 = wave_sum((float)c_other), f2 = wave_sum((float)c_pseudo);
    float zm = -3.402823466e38f;
    if (z) {
        for (int64_t i = (int64_t)blockIdx.x * CP_BLOCK + t; i < nz; i += (int64_t)gridDim.x * CP_BLOCK) zm = fmaxf(zm, z[i]);
        zm = wave_max(zm);
    }
    if (lane == 0) { red[0][wv] = (int)f0; red[1][wv] = (int)f1; red[2][wv] = (int)f2; zred[wv] = zm; }
    __syncthreads();
    if (t == 0) {
        int a = 0, b = 0, c = 0;
        for (int k = 0; k < CP_BLOCK / 64; ++k) { a += red[0][k]; b += red[1][k]; c += red[2][k]; zm = fmaxf(zm, zred[k]); }
        blk[3 * blockIdx.x] = a; blk[3 * blockIdx.x + 1] = b; blk[3 * blockIdx.x + 2] = c;
        if (z) zpart[blockIdx.x] = zm;
    }
}

__global__ __launch_bounds__(CP_BLOCK) void compact_points_write_k(const uint8_t* __restrict__ valid, int64_t n0, int64_t n1, int64_t n,
                                                                   const int32_t* __restrict__ blk, const float* __restrict__ zpart,
                                                                   int64_t* __restrict__ idx, int32_t* __restrict__ counts, StepFill F) {
    __shared__ int red[4][CP_BLOCK / 64];
    __shared__ float zred[CP_BLOCK / 64];
    __shared__ int wave_tot[CP_BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nb = gridDim.x;
    // the counts of all workgroups: rays selected in all, selected before this one (both branches of the rescue rule), pseudo points
    int tot_ray = 0, before_ray = 0, before_other = 0, tot_other = 0, tot_pseudo = 0;
    float zm = -3.402823466e38f;
    for (int b = t; b < nb; b += CP_BLOCK) {
        const int a = blk[3 * b], o = blk[3 * b + 1];
        tot_ray += a; tot_other += o; tot_pseudo += blk[3 * b + 2];
        if (b < (int)blockIdx.x) { before_ray += a; before_other += o; }
        if (F.z) zm = fmaxf(zm, zpart[b]);
    }
    const float s0 = wave_sum((float)tot_ray), s1 = wave_sum((float)before_ray), s2 = wave_sum((float)before_other), s3 = wave_sum((float)tot_other);
    const float s4 = wave_sum((float)tot_pseudo);
    zm = wave_max(zm);
    if (lane == 0) { red[0][wv] = (int)s0; red[1][wv] = (int)s1; red[2][wv] = (int)s2; red[3][wv] = (int)s3; zred[wv] = zm; wave_tot[wv] = (int)s4; }
    __syncthreads();
    tot_ray = before_ray = before_other = tot_other = tot_pseudo = 0;
    for (int k = 0; k < CP_BLOCK / 64; ++k) {
        tot_ray += red[0][k]; before_ray += red[1][k]; before_other += red[2][k]; tot_other += red[3][k]; tot_pseudo += wave_tot[k];
        zm = fmaxf(zm, zred[k]);
    }
    __syncthreads();
    const bool rescue = tot_ray == 0;
    const int64_t n_rescue = min((int64_t)10, n0);
    const int64_t block_first = (int64_t)blockIdx.x * CP_ROWS;
    const int64_t rescued_before = min(n_rescue, block_first);                      // rows < min(10, n0) that lie before this workgroup
    const int64_t base = before_other + (rescue ? rescued_before : (int64_t)before_ray);
    const int64_t first = block_first + (int64_t)t * CP_PER;
    const uint32_t bits = first < n ? cp_flags(valid, first, n) : 0u;
    uint32_t sel = 0;
    int mine = 0;
    for (int k = 0; k < CP_PER; ++k) {
        const int64_t i = first + k;
        const bool v = (bits >> k) & 1u;
        const bool s = i < n && (i < n0 ? (rescue ? i < n_rescue : v) : (i < n0 + n1 ? true : v));
        sel |= s ? (1u << k) : 0u;
        mine += s ? 1 : 0;
    }
    const float inc = wave_scan_add((float)mine, lane);
    if (lane == 63) wave_tot[wv] = (int)inc;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < wv; ++k) wbase += wave_tot[k];
    int64_t pos = base + wbase + (int)inc - mine;
    for (int k = 0; k < CP_PER; ++k) {
        const int64_t i = first + k;
        if (i < n) {
            if ((sel >> k) & 1u) {
                idx[pos++] = i;
            } else {
                if (F.y) F.y[i] = i < n0 ? 100.0f : 0.0f;
                if (F.g) { F.g[3 * i] = 0.0f; F.g[3 * i + 1] = 0.0f; F.g[3 * i + 2] = 0.0f; }
                if (F.s) { F.s[3 * i] = 0.0f; F.s[3 * i + 1] = 0.0f; F.s[3 * i + 2] = 0.0f; }
                if (i < n0) {
                    if (F.rgb) { F.rgb[3 * i] = 0.0f; F.rgb[3 * i + 1] = 0.0f; F.rgb[3 * i + 2] = 0.0f; }
                    if (F.vis) for (int v = 0; v < F.n_src; ++v) F.vis[i * F.n_src + v] = 0;
                }
            }
        }
    }
    if (blockIdx.x == 0 && t == 0) {
        const int n_ray_sel = rescue ? (int)n_rescue : tot_ray;
        counts[0] = n_ray_sel + tot_other;
        counts[1] = n_ray_sel;
        counts[2] = tot_pseudo;
        if (F.scalars) {
            if (F.z) F.scalars[0] = zm;
            if (F.variance) {
                const float raw = expf(F.variance[0] * 10.0f);                 // SingleVarianceNetwork.forward (variance_network.py:11), then :206
                const float inv_s = fminf(fmaxf(raw, 1e-6f), 1e6f);
                F.scalars[1] = inv_s;
                F.scalars[2] = 1.0f / inv_s;
                F.scalars[3] = (raw >= 1e-6f && raw <= 1e6f) ? 1.0f : 0.0f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// views taken out of several per-scene maps in ONE launch (fine-tuning: `self.features[i][view_ids]`, gens.py:151-153, and the same selection of the
// maps' texel / warp layouts): dst[k][j] = src[k][index[j]] for every map k, a map's view being `floats[k]` contiguous floats (a multiple of 4,
// 16-byte aligned).  torch.index_select per map was eleven launches at ~1 TB/s.
// ---------------------------------------------------------------------------------------------------------------
#define GENS_MAX_SELECT 16
struct ViewSelect {
    const float4* src[GENS_MAX_SELECT];
    float4* dst[GENS_MAX_SELECT];
    int64_t quads[GENS_MAX_SELECT];          // float4s per view
    int n_views[GENS_MAX_SELECT];            // views the source holds (index range check)
    int n_maps, n_sel;
};
__global__ __launch_bounds__(256) void select_views_k(ViewSelect S, const int64_t* __restrict__ index) {
    const int k = blockIdx.y;
    if (k >= S.n_maps) return;
    const int64_t q = S.quads[k], total = q * S.n_sel;
    const float4* __restrict__ src = S.src[k];
    float4* __restrict__ dst = S.dst[k];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t j = i / q, r = i - j * q;
        int64_t v = index[j];
        v = v < 0 ? v + S.n_views[k] : v;                                             // (torch's negative indices)
        if (v < 0 || v >= S.n_views[k]) continue;                                     // (an index torch would have refused: the view is left as it is)
        dst[i] = src[v * q + r];
    }
}

extern "C" int gens_select_views(const float* const* src, float* const* dst, const int* floats_per_view, const int* views_in_src, int n_maps,
                                 const int64_t* index, int n_sel, void* stream) {
    GENS_CHECK_ARG(src && dst && floats_per_view && views_in_src && index, GENS_EINVAL, "gens_select_views: null pointer");
    GENS_CHECK_ARG(n_maps >= 1 && n_maps <= GENS_MAX_SELECT && n_sel >= 0, GENS_ELIMIT, "gens_select_views: %d maps (1..%d), %d views", n_maps, GENS_MAX_SELECT, n_sel);
    if (n_sel == 0) return 0;
    ViewSelect S = {};
    S.n_maps = n_maps;
    S.n_sel = n_sel;
    int64_t most = 0;
    for (int k = 0; k < n_maps; ++k) {
        GENS_CHECK_ARG(src[k] && dst[k] && floats_per_view[k] > 0 && (floats_per_view[k] & 3) == 0 && views_in_src[k] > 0, GENS_EINVAL,
                       "gens_select_views: map %d: null, empty or a view that is not a multiple of four floats", k);
        GENS_CHECK_ARG((((uintptr_t)src[k] | (uintptr_t)dst[k]) & 15) == 0, GENS_EINVAL, "gens_select_views: map %d is not 16-byte aligned", k);
        S.src[k] = (const float4*)src[k];
        S.dst[k] = (float4*)dst[k];
        S.quads[k] = floats_per_view[k] >> 2;
        S.n_views[k] = views_in_src[k];
        most = std::max<int64_t>(most, S.quads[k] * n_sel);
    }
    const dim3 grid((unsigned)std::min<int64_t>((most + 255) / 256, 4096), (unsigned)n_maps);
    select_views_k<<<grid, 256, 0, (hipStream_t)stream>>>(S, index);
    return gens_launch_status("gens_select_views");
}

extern "C" int64_t gens_compact_points_scratch(int64_t n) { return 4 * ((n + CP_ROWS - 1) / CP_ROWS + 1); }      // int32 / float words

extern "C" int gens_compact_points(const uint8_t* valid, int64_t n_rays_pts, int64_t n_always, int64_t n, int64_t* idx, int32_t* counts,
                                   float* y_fill, float* g_fill, float* s_fill, float* rgb_fill, uint8_t* vis_fill, int n_src, const float* z,
                                   int64_t nz, const float* variance, float* scalars, int32_t* scratch, void* stream) {
    GENS_CHECK_ARG(valid && idx && counts && scratch, GENS_EINVAL, "gens_compact_points: null pointer");
    GENS_CHECK_ARG((!z && !variance) || scalars, GENS_EINVAL, "gens_compact_points: z / variance given without a scalars output");
    GENS_CHECK_ARG(!vis_fill || (n_src >= 1 && n_src < GENS_MAX_VIEWS), GENS_ELIMIT, "gens_compact_points: n_src=%d", n_src);
    StepFill F = {y_fill, g_fill, s_fill, rgb_fill, vis_fill, n_src, z, z ? nz : 0, variance, scalars};
    GENS_CHECK_ARG(n_rays_pts >= 0 && n_always >= 0 && n_rays_pts + n_always <= n && n < ((int64_t)1 << 24), GENS_EINVAL,
                   "gens_compact_points: bad segment sizes (%lld, %lld of %lld; fewer than 2^24 rows)", (long long)n_rays_pts, (long long)n_always, (long long)n);
    const unsigned nb = n > 0 ? gens_blocks(n, CP_ROWS) : 1u;
    float* zpart = (float*)(scratch + 3 * (int64_t)nb);
    hipStream_t st = (hipStream_t)stream;
    compact_points_count_k<<<nb, CP_BLOCK, 0, st>>>(valid, n_rays_pts, n_always, n, scratch, z, F.nz, zpart);
    compact_points_write_k<<<nb, CP_BLOCK, 0, st>>>(valid, n_rays_pts, n_always, n, scratch, zpart, idx, counts, F);
    return gens_launch_status("gens_compact_points");
}

// ---------------------------------------------------------------------------------------------------------------
// total-variation regulariser of all levels (implicit_surface.py:135-150; Q13)
// ---------------------------------------------------------------------------------------------------------------
#define TVL_BLOCK 256
struct TvLevels {
    const float* vol[GENS_MAX_LEVELS];
    const float* mask[GENS_MAX_LEVELS];
    float* g_vol[GENS_MAX_LEVELS];
    int X[GENS_MAX_LEVELS], Y[GENS_MAX_LEVELS], Z[GENS_MAX_LEVELS];
    int first_block[GENS_MAX_LEVELS + 1];
    int n;
};

// one thread owns FOUR consecutive z of one (x, y) row (k9_misc.hip::tv_fwd4_k, same arithmetic and order per voxel)
__device__ __forceinline__ float4 tv_fwd4_thread(const float* __restrict__ vol, const float* __restrict__ mask, int X, int Y, int Z, uint32_t t) {
    const uint32_t n = (uint32_t)X * Y * Z, q = n >> 2;
    float4 acc = f4_zero();
    if (t >= q) return acc;
    const uint32_t i = t << 2, zq = (uint32_t)Z >> 2;
    const uint32_t kz = (t % zq) << 2, row = t / zq, jy = row % (uint32_t)Y, ix = row / (uint32_t)Y;
    const uint32_t sx = (uint32_t)Y * Z, sy = (uint32_t)Z;
    const bool hx = ix + 1 < (uint32_t)X, hy = jy + 1 < (uint32_t)Y, hz = kz + 4 < (uint32_t)Z;
    const float4 m = *(const float4*)(mask + i);
    const float4 mxv = hx ? *(const float4*)(mask + i + sx) : f4_zero();
    const float4 myv = hy ? *(const float4*)(mask + i + sy) : f4_zero();
    const float mzn = hz ? mask[i + 4] : 0.0f;
    const float mm[4] = {m.x, m.y, m.z, m.w}, mxa[4] = {mxv.x, mxv.y, mxv.z, mxv.w}, mya[4] = {myv.x, myv.y, myv.z, myv.w};
    const float mza[4] = {m.y, m.z, m.w, mzn};
    bool bx[4], by[4], bz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bx[k] = hx && (mm[k] * mxa[k] > 0.0f);
        by[k] = hy && (mm[k] * mya[k] > 0.0f);
        bz[k] = (k < 3 || hz) && (mm[k] * mza[k] > 0.0f);
        acc.w += bx[k] ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* v = vol + (size_t)c * n + i;
        const float4 v0 = *(const float4*)v;
        const float4 vx = hx ? *(const float4*)(v + sx) : f4_zero();
        const float4 vy = hy ? *(const float4*)(v + sy) : f4_zero();
        const float vzn = hz ? v[4] : 0.0f;
        const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, ax[4] = {vx.x, vx.y, vx.z, vx.w}, ay[4] = {vy.x, vy.y, vy.z, vy.w};
        const float az[4] = {v0.y, v0.z, v0.w, vzn};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (bx[k]) { const float d = ax[k] - a0[k]; acc.x += d * d; }
            if (by[k]) { const float d = ay[k] - a0[k]; acc.y += d * d; }
            if (bz[k]) { const float d = az[k] - a0[k]; acc.z += d * d; }
        }
    }
    return acc;
}

__global__ __launch_bounds__(TVL_BLOCK) void tv_levels_fwd_k(TvLevels L, float4* __restrict__ partial) {
    __shared__ float4 red[TVL_BLOCK / 64];
    int l = 0;
#pragma unroll
    for (int j = 1; j < GENS_MAX_LEVELS; ++j) l += (j < L.n && (int)blockIdx.x >= L.first_block[j]) ? 1 : 0;
    const uint32_t t = (blockIdx.x - L.first_block[l]) * TVL_BLOCK + threadIdx.x;
    float4 acc = tv_fwd4_thread(L.vol[l], L.mask[l], L.X[l], L.Y[l], L.Z[l], t);
    acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y); acc.z = wave_sum(acc.z); acc.w = wave_sum(acc.w);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 r = red[0];
        for (int k = 1; k < TVL_BLOCK / 64; ++k) { r.x += red[k].x; r.y += red[k].y; r.z += red[k].z; r.w += red[k].w; }
        partial[blockIdx.x] = r;
    }
}

// One workgroup adds the per-block partials of every level in float64 in a fixed order (deterministic) and writes
//   out[0]         tv_reg = sum_l 0.5^l sqrt((sx + sy + sz) / (count_x + 1e-8))            (Q13: all three axes over mx's count)
//   out[1 + l]     the level's backward coefficient 0.5^l / (2 tv_l den_l)   (d tv_reg / d (squared-difference sum of level l))
__global__ __launch_bounds__(256) void tv_levels_finish_k(TvLevels L, const float4* __restrict__ partial, float* __restrict__ out) {
    __shared__ double red[4][4];
    double total = 0.0, scale = 1.0;
    for (int l = 0; l < L.n; ++l) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = L.first_block[l] + threadIdx.x; b < L.first_block[l + 1]; b += 256) {
            const float4 p = partial[b];
            s[0] += (double)p.x; s[1] += (double)p.y; s[2] += (double)p.z; s[3] += (double)p.w;
        }
        for (int k = 0; k < 4; ++k) {
            double v = s[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t[4];
            for (int k = 0; k < 4; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
            const double den = t[3] + 1e-8;
            const float tv = (float)sqrt((t[0] + t[1] + t[2]) / den);          // the reference's value is float32
            total += scale * (double)tv;
            out[1 + l] = (float)(scale / (2.0 * (double)tv * (double)(float)den));
        }
        __syncthreads();
        scale *= 0.5;
    }
    if (threadIdx.x == 0) out[0] = (float)total;
}

// d tv_reg / d volume for all levels: g_vol[l] = g * coef[l] * d(sum of squared differences)/d vol   (overwrites g_vol)
__device__ __forceinline__ void tv_bwd4_thread(const float* __restrict__ vol, const float* __restrict__ mask, int X, int Y, int Z, uint32_t t, float coef,
                                               float* __restrict__ g_vol) {
    const uint32_t n = (uint32_t)X * Y * Z, q = n >> 2;
    if (t >= q) return;
    const uint32_t i = t << 2, zq = (uint32_t)Z >> 2;
    const uint32_t kz = (t % zq) << 2, row = t / zq, jy = row % (uint32_t)Y, ix = row / (uint32_t)Y;
    const uint32_t sx = (uint32_t)Y * Z, sy = (uint32_t)Z;
    const bool hxp = ix + 1 < (uint32_t)X, hxn = ix > 0, hyp = jy + 1 < (uint32_t)Y, hyn = jy > 0, hzp = kz + 4 < (uint32_t)Z, hzn = kz > 0;
    const float4 m = *(const float4*)(mask + i);
    const float4 mxp = hxp ? *(const float4*)(mask + i + sx) : f4_zero(), mxn = hxn ? *(const float4*)(mask + i - sx) : f4_zero();
    const float4 myp = hyp ? *(const float4*)(mask + i + sy) : f4_zero(), myn = hyn ? *(const float4*)(mask + i - sy) : f4_zero();
    const float mzp = hzp ? mask[i + 4] : 0.0f, mzn = hzn ? mask[i - 1] : 0.0f;
    const float mm[4] = {m.x, m.y, m.z, m.w};
    const float axp[4] = {mxp.x, mxp.y, mxp.z, mxp.w}, axn[4] = {mxn.x, mxn.y, mxn.z, mxn.w};
    const float ayp[4] = {myp.x, myp.y, myp.z, myp.w}, ayn[4] = {myn.x, myn.y, myn.z, myn.w};
    const float azp[4] = {m.y, m.z, m.w, mzp}, azn[4] = {mzn, m.x, m.y, m.z};
    bool px[4], nx[4], py[4], ny[4], pz[4], nz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = hxp && (mm[k] * axp[k] > 0.0f);
        nx[k] = hxn && (mm[k] * axn[k] > 0.0f);
        py[k] = hyp && (mm[k] * ayp[k] > 0.0f);
        ny[k] = hyn && (mm[k] * ayn[k] > 0.0f);
        pz[k] = (k < 3 || hzp) && (mm[k] * azp[k] > 0.0f);
        nz[k] = (k > 0 || hzn) && (mm[k] * azn[k] > 0.0f);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* v = vol + (size_t)c * n + i;
        const float4 v0 = *(const float4*)v;
        const float4 vxp = hxp ? *(const float4*)(v + sx) : f4_zero(), vxn = hxn ? *(const float4*)(v - sx) : f4_zero();
        const float4 vyp = hyp ? *(const float4*)(v + sy) : f4_zero(), vyn = hyn ? *(const float4*)(v - sy) : f4_zero();
        const float vzp = hzp ? v[4] : 0.0f, vzn = hzn ? v[-1] : 0.0f;
        const float a0[4] = {v0.x, v0.y, v0.z, v0.w};
        const float bxp[4] = {vxp.x, vxp.y, vxp.z, vxp.w}, bxn[4] = {vxn.x, vxn.y, vxn.z, vxn.w};
        const float byp[4] = {vyp.x, vyp.y, vyp.z, vyp.w}, byn[4] = {vyn.x, vyn.y, vyn.z, vyn.w};
        const float bzp[4] = {v0.y, v0.z, v0.w, vzp}, bzn[4] = {vzn, v0.x, v0.y, v0.z};
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = 0.0f;
            if (px[k]) s -= 2.0f * (bxp[k] - a0[k]);
            if (nx[k]) s += 2.0f * (a0[k] - bxn[k]);
            if (py[k]) s -= 2.0f * (byp[k] - a0[k]);
            if (ny[k]) s += 2.0f * (a0[k] - byn[k]);
            if (pz[k]) s -= 2.0f * (bzp[k] - a0[k]);
            if (nz[k]) s += 2.0f * (a0[k] - bzn[k]);
            g[k] = coef * s;
        }
        *(float4*)(g_vol + (size_t)c * n + i) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

__global__ __launch_bounds__(TVL_BLOCK) void tv_levels_bwd_k(TvLevels L, const float* __restrict__ coefs, const float* __restrict__ g) {
    int l = 0;
#pragma unroll
    for (int j = 1; j < GENS_MAX_LEVELS; ++j) l += (j < L.n && (int)blockIdx.x >= L.first_block[j]) ? 1 : 0;
    const uint32_t t = (blockIdx.x - L.first_block[l]) * TVL_BLOCK + threadIdx.x;
    tv_bwd4_thread(L.vol[l], L.mask[l], L.X[l], L.Y[l], L.Z[l], t, g[0] * coefs[1 + l], L.g_vol[l]);
}

static int fill_tv_levels(const char* who, TvLevels* L, const float* const* vols, const float* const* masks, float* const* g_vols, const int* dims,
                          int n_levels) {
    GENS_CHECK_ARG(vols && masks && dims, GENS_EINVAL, "%s: null table", who);
    GENS_CHECK_ARG(n_levels >= 1 && n_levels <= GENS_MAX_LEVELS, GENS_ELIMIT, "%s: n_levels=%d", who, n_levels);
    L->n = n_levels;
    L->first_block[0] = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int x = dims[3 * l], y = dims[3 * l + 1], z = dims[3 * l + 2];
        GENS_CHECK_ARG(vols[l] && masks[l] && x > 0 && y > 0 && z > 0, GENS_EINVAL, "%s: bad level %d", who, l);
        GENS_CHECK_ARG((z & 3) == 0 && (int64_t)x * y * z < ((int64_t)1 << 31), GENS_ELIMIT,
                       "%s: level %d (%d x %d x %d): Z must be a multiple of 4 and the level smaller than 2^31 voxels (use gens_tv_fwd / gens_tv_bwd)", who, l, x, y, z);
        GENS_CHECK_ARG((((uintptr_t)vols[l] | (uintptr_t)masks[l] | (uintptr_t)(g_vols ? g_vols[l] : nullptr)) & 15) == 0, GENS_EINVAL,
                       "%s: level %d is not 16-byte aligned (use gens_tv_fwd / gens_tv_bwd)", who, l);
        L->vol[l] = vols[l];
        L->mask[l] = masks[l];
        L->g_vol[l] = g_vols ? g_vols[l] : nullptr;
        L->X[l] = x; L->Y[l] = y; L->Z[l] = z;
        L->first_block[l + 1] = L->first_block[l] + (int)gens_blocks((int64_t)x * y * z / 4, TVL_BLOCK);
    }
    for (int l = n_levels; l < GENS_MAX_LEVELS; ++l) { L->vol[l] = L->mask[l] = nullptr; L->g_vol[l] = nullptr; L->X[l] = L->Y[l] = L->Z[l] = 0; L->first_block[l + 1] = L->first_block[n_levels]; }
    return 0;
}

extern "C" int gens_tv_levels_blocks(const int* dims, int n_levels) {
    if (!dims) return 0;
    int b = 0;
    for (int l = 0; l < n_levels && l < GENS_MAX_LEVELS; ++l) b += (int)gens_blocks((int64_t)dims[3 * l] * dims[3 * l + 1] * dims[3 * l + 2] / 4, TVL_BLOCK);
    return b;
}

extern "C" int gens_tv_levels_fwd(const float* const* vols, const float* const* masks, const int* dims, int n_levels, float* partial, float* out,
                                  void* stream) {
    TvLevels L;
    if (int e = fill_tv_levels("gens_tv_levels_fwd", &L, vols, masks, nullptr, dims, n_levels)) return e;
    GENS_CHECK_ARG(partial && out && ((uintptr_t)partial & 15) == 0, GENS_EINVAL, "gens_tv_levels_fwd: null / misaligned partial or out");
    tv_levels_fwd_k<<<L.first_block[n_levels], TVL_BLOCK, 0, (hipStream_t)stream>>>(L, (float4*)partial);
    tv_levels_finish_k<<<1, 256, 0, (hipStream_t)stream>>>(L, (const float4*)partial, out);
    return gens_launch_status("gens_tv_levels_fwd");
}

extern "C" int gens_tv_levels_bwd(const float* const* vols, const float* const* masks, const int* dims, int n_levels, const float* out,
                                  const float* g, float* const* g_vols, void* stream) {
    TvLevels L;
    GENS_CHECK_ARG(g_vols && out && g, GENS_EINVAL, "gens_tv_levels_bwd: null pointer");
    if (int e = fill_tv_levels("gens_tv_levels_bwd", &L, vols, masks, g_vols, dims, n_levels)) return e;
    for (int l = 0; l < n_levels; ++l) GENS_CHECK_ARG(g_vols[l], GENS_EINVAL, "gens_tv_levels_bwd: null gradient buffer of level %d", l);
    tv_levels_bwd_k<<<L.first_block[n_levels], TVL_BLOCK, 0, (hipStream_t)stream>>>(L, out, g);
    return gens_launch_status("gens_tv_levels_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// surface_patch_warp fused (projector.py:353-437 as called from implicit_surface.py:301-328): the surface point of a ray, the
// plane-induced homographies into the source views, the 11 x 11 patch grids and the bilinear reads of the warp features -- ~60 PyTorch
// launches forward and as many backward -- in one launch each.  One wavefront per ray; lanes over the (view, pixel) samples.
//   p = o + d z (z = z_cross, the only differentiable input: the normal is used detached, :306-310)
//   n = normalise(g0) R_ref                         x_cam = p R_ref + t_ref            disp = n . x_cam
//   H_v = K_v (R_v^T R_ref + (R_v^T (c_ref - c_v)) n^T / (disp + 1e-10)) K_ref^-1
//   (u0, v0) = K_ref x_cam, dehomogenised with + 1e-8; pixel (ox, oy): q = H_v (u0 + ox, v0 + oy, 1), grid = q.xy / (q.z + 1e-8)
//   read with align_corners=True after the reference's normalise / un-normalise round trip (kept for its rounding).
// Backward: every quantity carries ONE tangent (d / dz); g_z[ray] = sum over views, pixels, channels of g_sampled * d sampled / dz.
// ---------------------------------------------------------------------------------------------------------------
#define PW_MAX_Q4 4
struct PatchWarpArgs {
    const float *rays_o, *rays_d, *z, *g0;     // (B,3) (B,3) (B) (B,3)
    const float *c2w, *intr, *kinv_ref;        // (nv,4,4) (nv,4,4) (3,3)
    const float4* tex;                         // (nv, H, W, q4) texels
    int nv, h, w, c, q4, patch;
    int64_t n_rays;
    float *ref, *sampled;                      // (1, B, P, C), (S, B, P, C)
    const float* g_sampled;                    // backward: (S, B, P, C)
    float* g_z;                                // backward: (B)
};

struct PwRay {
    float u0, v0, du0, dv0;       // reference pixel and its tangent
    float nrm[3], disp, ddisp;
    bool ok;
};

__device__ __forceinline__ PwRay pw_ray(const PatchWarpArgs& A, int64_t r) {
    PwRay R;
    const float* o = A.rays_o + 3 * r;
    const float* d = A.rays_d + 3 * r;
    const float z = A.z[r];
    const float p[3] = {o[0] + d[0] * z, o[1] + d[1] * z, o[2] + d[2] * z};
    const float* cr = A.c2w;                                   // reference view: rows of [R | c]
    float g[3] = {A.g0[3 * r], A.g0[3 * r + 1], A.g0[3 * r + 2]};
    float gn = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    gn = gn <= 0.0f ? 1e-8f : gn;                              // (:308-309)
    g[0] /= gn; g[1] /= gn; g[2] /= gn;
    float xc[3], dx[3], t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        R.nrm[j] = g[0] * cr[j] + g[1] * cr[4 + j] + g[2] * cr[8 + j];                     // g R_ref (row vector)
        t[j] = -(cr[j] * cr[3] + cr[4 + j] * cr[7] + cr[8 + j] * cr[11]);                  // -(R_ref^T c_ref)
        xc[j] = (p[0] * cr[j] + p[1] * cr[4 + j] + p[2] * cr[8 + j]) + t[j];
        dx[j] = d[0] * cr[j] + d[1] * cr[4 + j] + d[2] * cr[8 + j];
    }
    const float* k = A.intr;
    float pr[3], dp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pr[i] = xc[0] * k[4 * i] + xc[1] * k[4 * i + 1] + xc[2] * k[4 * i + 2];
        dp[i] = dx[0] * k[4 * i] + dx[1] * k[4 * i + 1] + dx[2] * k[4 * i + 2];
    }
    R.disp = R.nrm[0] * xc[0] + R.nrm[1] * xc[1] + R.nrm[2] * xc[2];
    R.ddisp = R.nrm[0] * dx[0] + R.nrm[1] * dx[1] + R.nrm[2] * dx[2];
    const float den = pr[2] + 1e-8f;
    R.u0 = pr[0] / den;
    R.v0 = pr[1] / den;
    R.du0 = (dp[0] * den - pr[0] * dp[2]) / (den * den);
    R.dv0 = (dp[1] * den - pr[1] * dp[2]) / (den * den);
    R.ok = true;
    return R;
}

// H_v (and, if DUAL, its tangent) for source view sv >= 1
template <bool DUAL>
__device__ __forceinline__ void pw_homography(const PatchWarpArgs& A, const PwRay& R, int sv, float H[9], float dH[9]) {
    const float* cr = A.c2w;
    const float* cs = A.c2w + 16 * sv;
    float rel[9], tv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) rel[3 * i + j] = cs[i] * cr[j] + cs[4 + i] * cr[4 + j] + cs[8 + i] * cr[8 + j];      // R_v^T R_ref
        tv[i] = cs[i] * (cr[3] - cs[3]) + cs[4 + i] * (cr[7] - cs[7]) + cs[8 + i] * (cr[11] - cs[11]);                    // R_v^T (c_ref - c_v)
    }
    const float den = R.disp + 1e-10f;
    float hom[9], dhom[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float outer = tv[i] * R.nrm[j];
            hom[3 * i + j] = rel[3 * i + j] + outer / den;
            if (DUAL) dhom[3 * i + j] = -outer * R.ddisp / (den * den);
        }
    const float* ks = A.intr + 16 * sv;
    const float* ki = A.kinv_ref;
    float tmp[9], dtmp[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            tmp[3 * i + j] = ks[4 * i] * hom[j] + ks[4 * i + 1] * hom[3 + j] + ks[4 * i + 2] * hom[6 + j];
            if (DUAL) dtmp[3 * i + j] = ks[4 * i] * dhom[j] + ks[4 * i + 1] * dhom[3 + j] + ks[4 * i + 2] * dhom[6 + j];
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            H[3 * i + j] = tmp[3 * i] * ki[j] + tmp[3 * i + 1] * ki[3 + j] + tmp[3 * i + 2] * ki[6 + j];
            if (DUAL) dH[3 * i + j] = dtmp[3 * i] * ki[j] + dtmp[3 * i + 1] * ki[3 + j] + dtmp[3 * i + 2] * ki[6 + j];
        }
}

#define PW_RAYS_PER_BLOCK 4
__global__ __launch_bounds__(64 * PW_RAYS_PER_BLOCK) void patch_warp_fwd_k(PatchWarpArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * PW_RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= A.n_rays) return;
    const int S = A.nv - 1, P = A.patch * A.patch, half = A.patch / 2;
    const PwRay R = pw_ray(A, r);
    const float wm = (float)(A.w - 1), hm = (float)(A.h - 1);
    for (int i = lane; i < (S + 1) * P; i += 64) {
        const int v = i / P, px = i % P;                                     // v = 0: the reference view itself
        const float u = R.u0 + (float)(px % A.patch - half), vv = R.v0 + (float)(px / A.patch - half);
        float gx = u, gy = vv;
        if (v > 0) {
            float H[9], dH[9];
            pw_homography<false>(A, R, v, H, dH);
            const float q0 = H[0] * u + H[1] * vv + H[2], q1 = H[3] * u + H[4] * vv + H[5], q2 = H[6] * u + H[7] * vv + H[8];
            gx = q0 / (q2 + 1e-8f);
            gy = q1 / (q2 + 1e-8f);
        }
        const float ix = ((2.0f * gx / wm - 1.0f) + 1.0f) / 2.0f * wm, iy = ((2.0f * gy / hm - 1.0f) + 1.0f) / 2.0f * hm;
        const Taps2 t = bilinear_taps(ix, iy, A.h, A.w);
        const float4* img = A.tex + (int64_t)v * A.h * A.w * A.q4;
        float* o = (v == 0 ? A.ref : A.sampled + ((int64_t)(v - 1) * A.n_rays) * P * A.c) + ((int64_t)r * P + px) * A.c;
        for (int q = 0; q < A.q4; ++q) {
            const float4 val = sample_texel(img, A.h, A.w, A.q4, q, t);
            o[4 * q] = val.x;
            if (4 * q + 1 < A.c) o[4 * q + 1] = val.y;
            if (4 * q + 2 < A.c) o[4 * q + 2] = val.z;
            if (4 * q + 3 < A.c) o[4 * q + 3] = val.w;
        }
    }
}

__global__ __launch_bounds__(64 * PW_RAYS_PER_BLOCK) void patch_warp_bwd_k(PatchWarpArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * PW_RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= A.n_rays) return;
    const int S = A.nv - 1, P = A.patch * A.patch, half = A.patch / 2;
    const PwRay R = pw_ray(A, r);
    const float wm = (float)(A.w - 1), hm = (float)(A.h - 1);
    float acc = 0.0f;
    for (int i = lane; i < S * P; i += 64) {
        const int v = i / P + 1, px = i % P;
        const float u = R.u0 + (float)(px % A.patch - half), vv = R.v0 + (float)(px / A.patch - half);
        float H[9], dH[9];
        pw_homography<true>(A, R, v, H, dH);
        const float q0 = H[0] * u + H[1] * vv + H[2], q1 = H[3] * u + H[4] * vv + H[5], q2 = H[6] * u + H[7] * vv + H[8];
        const float dq0 = (dH[0] * u + dH[1] * vv + dH[2]) + (H[0] * R.du0 + H[1] * R.dv0);
        const float dq1 = (dH[3] * u + dH[4] * vv + dH[5]) + (H[3] * R.du0 + H[4] * R.dv0);
        const float dq2 = (dH[6] * u + dH[7] * vv + dH[8]) + (H[6] * R.du0 + H[7] * R.dv0);
        const float den = q2 + 1e-8f;
        const float gx = q0 / den, gy = q1 / den;
        const float dgx = (dq0 * den - q0 * dq2) / (den * den), dgy = (dq1 * den - q1 * dq2) / (den * den);
        const float ix = ((2.0f * gx / wm - 1.0f) + 1.0f) / 2.0f * wm, iy = ((2.0f * gy / hm - 1.0f) + 1.0f) / 2.0f * hm;
        const Taps2 t = bilinear_taps(ix, iy, A.h, A.w);
        if (!(isfinite(ix) && isfinite(iy))) continue;
        const float fx = (float)t.x0, fy = (float)t.y0;
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
        const float4* img = A.tex + (int64_t)v * A.h * A.w * A.q4;
        const int64_t base = ((int64_t)t.y0 * A.w + t.x0) * A.q4;
        const float* g = A.g_sampled + (((int64_t)(v - 1) * A.n_rays + r) * P + px) * A.c;
        float sx = 0.0f, sy = 0.0f;
        for (int q = 0; q < A.q4; ++q) {
            float gq[4] = {g[4 * q], 0.f, 0.f, 0.f};
            if (4 * q + 1 < A.c) gq[1] = g[4 * q + 1];
            if (4 * q + 2 < A.c) gq[2] = g[4 * q + 2];
            if (4 * q + 3 < A.c) gq[3] = g[4 * q + 3];
#define PW_DOT(V) ((V).x * gq[0] + (V).y * gq[1] + (V).z * gq[2] + (V).w * gq[3])
            const float d00 = t.ok00 ? PW_DOT(img[base + q]) : 0.0f;
            const float d01 = t.ok01 ? PW_DOT(img[base + A.q4 + q]) : 0.0f;
            const float d10 = t.ok10 ? PW_DOT(img[base + (int64_t)A.w * A.q4 + q]) : 0.0f;
            const float d11 = t.ok11 ? PW_DOT(img[base + (int64_t)A.w * A.q4 + A.q4 + q]) : 0.0f;
#undef PW_DOT
            sx += (d01 - d00) * wy0 + (d11 - d10) * wy1;
            sy += (d10 - d00) * wx0 + (d11 - d01) * wx1;
        }
        const float term = sx * dgx + sy * dgy;
        if (term == term) acc += term;                                   // (a degenerate homography: no gradient, like a masked ray)
    }
    acc = wave_sum(acc);
    if (lane == 0) A.g_z[r] = acc;
}

static int check_patch_warp(const char* who, const PatchWarpArgs& A) {
    GENS_CHECK_ARG(A.rays_o && A.rays_d && A.z && A.g0 && A.c2w && A.intr && A.kinv_ref && A.tex, GENS_EINVAL, "%s: null input pointer", who);
    GENS_CHECK_ARG(A.nv >= 2 && A.nv <= GENS_MAX_VIEWS, GENS_ELIMIT, "%s: nv=%d not in 2..%d", who, A.nv, GENS_MAX_VIEWS);
    GENS_CHECK_ARG(A.h > 1 && A.w > 1 && A.c >= 1 && A.c <= 4 * PW_MAX_Q4 && A.patch >= 1 && A.patch <= 15 && (A.patch & 1), GENS_ELIMIT,
                   "%s: image %d x %d, %d channels (<= %d), odd patch size %d (<= 15)", who, A.h, A.w, A.c, 4 * PW_MAX_Q4, A.patch);
    GENS_CHECK_ARG(((uintptr_t)A.tex & 15) == 0, GENS_EINVAL, "%s: texels must be 16-byte aligned", who);
    return 0;
}

extern "C" int gens_patch_warp_fwd(const float* rays_o, const float* rays_d, const float* z, const float* g0, int64_t n_rays, const float* c2ws,
                                   const float* intrs, const float* kinv_ref, int nv, const float* tex, int h, int w, int c, int patch, float* ref,
                                   float* sampled, void* stream) {
    PatchWarpArgs A = {rays_o, rays_d, z, g0, c2ws, intrs, kinv_ref, (const float4*)tex, nv, h, w, c, (c + 3) / 4, patch, n_rays, ref, sampled, nullptr, nullptr};
    if (int e = check_patch_warp("gens_patch_warp_fwd", A)) return e;
    GENS_CHECK_ARG(n_rays >= 0 && (n_rays == 0 || (ref && sampled)), GENS_EINVAL, "gens_patch_warp_fwd: null output");
    if (n_rays == 0) return 0;
    patch_warp_fwd_k<<<gens_blocks(n_rays, PW_RAYS_PER_BLOCK), 64 * PW_RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(A);
    return gens_launch_status("gens_patch_warp_fwd");
}

extern "C" int gens_patch_warp_bwd(const float* rays_o, const float* rays_d, const float* z, const float* g0, int64_t n_rays, const float* c2ws,
                                   const float* intrs, const float* kinv_ref, int nv, const float* tex, int h, int w, int c, int patch,
                                   const float* g_sampled, float* g_z, void* stream) {
    PatchWarpArgs A = {rays_o, rays_d, z, g0, c2ws, intrs, kinv_ref, (const float4*)tex, nv, h, w, c, (c + 3) / 4, patch, n_rays, nullptr, nullptr, g_sampled, g_z};
    if (int e = check_patch_warp("gens_patch_warp_bwd", A)) return e;
    GENS_CHECK_ARG(n_rays >= 0 && (n_rays == 0 || (g_sampled && g_z)), GENS_EINVAL, "gens_patch_warp_bwd: null gradient pointer");
    if (n_rays == 0) return 0;
    patch_warp_bwd_k<<<gens_blocks(n_rays, PW_RAYS_PER_BLOCK), 64 * PW_RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(A);
    return gens_launch_status("gens_patch_warp_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// Loss.forward of a training step (models/losses/loss.py:24-93) in one launch, its backward in one launch.
//   color  = sum |color_fine - target| * valid / (sum valid + 1e-5)             (:25-27)
//   eik    = gradient_error;  smooth = smooth_error;  tv = tv_reg               (:29, 33, 35; means of scalars)
//   sparse = mean exp(-|sparse_sdf| * scale)                                    (:31)
//   mfc    = 0.5 * sum ncc * m / (sum m + 1e-8),  m = valid * mid_inside_sphere (:37-39)
//   pseudo_sdf = mean |pseudo_sdf|  (0 when absent)                             (:41-44)
//   pseudo_depth / depth = sum |render_depth - t| [t > 0] / (sum [t > 0] + 1e-8)  (0 when the target is absent; :46-54)
//   loss = color w_c + eik w_igr + sparse w_sp + mfc w_mfc + smooth w_sm + tv w_tv + pseudo_sdf w_ps + pseudo_depth w_pd   (:64-71)
// out (16): [loss, color, eik, sparse, mfc, smooth, tv, depth, pseudo_sdf, pseudo_depth, den_color, den_mfc, den_pseudo_depth, den_depth, 0, 0]
// ---------------------------------------------------------------------------------------------------------------
typedef gens_loss_args LossArgs;      // (layout in include/gens_hip.h)

#define LS_THREADS 1024
__global__ __launch_bounds__(LS_THREADS) void loss_fwd_k(LossArgs A) {
    __shared__ float red[10][LS_THREADS / 64];
    float s[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x;
    for (int64_t r = t; r < A.b; r += LS_THREADS) {
        const float v = A.valid[r] ? 1.0f : 0.0f;
        s[0] += (fabsf(A.color[3 * r] - A.target[3 * r]) + fabsf(A.color[3 * r + 1] - A.target[3 * r + 1]) + fabsf(A.color[3 * r + 2] - A.target[3 * r + 2])) * v;
        s[1] += v;
        const float m = v * A.mid_in[r];
        s[2] += A.ncc[r] * m;
        s[3] += m;
        if (A.pseudo_depth_t) {
            const float tt = A.pseudo_depth_t[r], on = tt > 0.0f ? 1.0f : 0.0f;
            s[4] += fabsf(A.depth[r] - tt) * on;
            s[5] += on;
        }
        if (A.depth_t) {
            const float tt = A.depth_t[r], on = tt > 0.0f ? 1.0f : 0.0f;
            s[6] += fabsf(A.depth[r] - tt) * on;
            s[7] += on;
        }
    }
    for (int64_t i = t; i < A.n_sparse; i += LS_THREADS) s[8] += expf(-fabsf(A.sparse[i]) * A.sparse_scale);
    if (A.pseudo)
        for (int64_t i = t; i < A.n_pseudo; i += LS_THREADS) s[9] += fabsf(A.pseudo[i]);
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float w = wave_sum(s[k]);
        if ((t & 63) == 0) red[k][t >> 6] = w;
    }
    __syncthreads();
    if (t == 0) {
        float tot[10];
        for (int k = 0; k < 10; ++k) {
            float a = 0.0f;
            for (int w = 0; w < LS_THREADS / 64; ++w) a += red[k][w];
            tot[k] = a;
        }
        const float color = tot[0] / (tot[1] + 1e-5f);
        const float mfc = 0.5f * (tot[2] / (tot[3] + 1e-8f));
        const float pdepth = A.pseudo_depth_t ? tot[4] / (tot[5] + 1e-8f) : 0.0f;
        const float depth = A.depth_t ? tot[6] / (tot[7] + 1e-8f) : 0.0f;
        const float sparse = A.n_sparse > 0 ? tot[8] / (float)A.n_sparse : 0.0f;
        const float psdf = (A.pseudo && A.n_pseudo > 0) ? tot[9] / (float)A.n_pseudo : 0.0f;
        const float eik = A.ge[0], smooth = A.se[0], tv = A.tv[0];
        float loss = color * A.w_color;
        loss += eik * A.w_igr;
        loss += sparse * A.w_sparse;
        loss += mfc * A.w_mfc;
        loss += smooth * A.w_smooth;
        loss += tv * A.w_tv;
        loss += psdf * A.w_pseudo_sdf;
        loss += pdepth * A.w_pseudo_depth;
        float* o = A.out;
        o[0] = loss; o[1] = color; o[2] = eik; o[3] = sparse; o[4] = mfc; o[5] = smooth; o[6] = tv; o[7] = depth; o[8] = psdf; o[9] = pdepth;
        o[10] = tot[1] + 1e-5f; o[11] = tot[3] + 1e-8f; o[12] = tot[5] + 1e-8f; o[13] = tot[7] + 1e-8f; o[14] = 0.0f; o[15] = 0.0f;
    }
}

__device__ __forceinline__ float sgn_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

__global__ __launch_bounds__(256) void loss_bwd_k(LossArgs A) {
    const float g = A.g[0];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && A.g_scalars) { A.g_scalars[0] = g * A.w_igr; A.g_scalars[1] = g * A.w_smooth; A.g_scalars[2] = g * A.w_tv; }
    if (i < A.b) {
        const float v = A.valid[i] ? 1.0f : 0.0f;
        if (A.g_color) {
            const float k = g * A.w_color * v / A.out[10];
#pragma unroll
            for (int a = 0; a < 3; ++a) A.g_color[3 * i + a] = k * sgn_(A.color[3 * i + a] - A.target[3 * i + a]);
        }
        if (A.g_ncc) A.g_ncc[i] = g * A.w_mfc * 0.5f * v * A.mid_in[i] / A.out[11];
        if (A.g_depth) {
            float gd = 0.0f;
            if (A.pseudo_depth_t) {
                const float tt = A.pseudo_depth_t[i];
                if (tt > 0.0f) gd = g * A.w_pseudo_depth * sgn_(A.depth[i] - tt) / A.out[12];
            }
            A.g_depth[i] = gd;
        }
    }
    if (i < A.n_sparse && A.g_sparse) {
        const float x = A.sparse[i];
        A.g_sparse[i] = g * A.w_sparse * (-A.sparse_scale * sgn_(x) * expf(-fabsf(x) * A.sparse_scale)) / (float)A.n_sparse;
    }
    if (A.pseudo && i < A.n_pseudo && A.g_pseudo) A.g_pseudo[i] = g * A.w_pseudo_sdf * sgn_(A.pseudo[i]) / (float)A.n_pseudo;
}

static int check_loss(const char* who, const LossArgs* A) {
    GENS_CHECK_ARG(A, GENS_EINVAL, "%s: null argument block", who);
    GENS_CHECK_ARG(A->color && A->target && A->valid && A->sparse && A->ncc && A->mid_in && A->ge && A->se && A->tv && A->out, GENS_EINVAL,
                   "%s: null input pointer", who);
    GENS_CHECK_ARG(A->b >= 0 && A->n_sparse >= 0 && A->n_pseudo >= 0, GENS_EINVAL, "%s: negative size", who);
    GENS_CHECK_ARG(!(A->pseudo_depth_t || A->depth_t) || A->depth, GENS_EINVAL, "%s: a depth target needs render_depth", who);
    return 0;
}

extern "C" int gens_loss_fwd(const gens_loss_args* args, void* stream) {
    const LossArgs* A = args;
    if (int e = check_loss("gens_loss_fwd", A)) return e;
    loss_fwd_k<<<1, LS_THREADS, 0, (hipStream_t)stream>>>(*A);
    return gens_launch_status("gens_loss_fwd");
}

extern "C" int gens_loss_bwd(const gens_loss_args* args, void* stream) {
    const LossArgs* A = args;
    if (int e = check_loss("gens_loss_bwd", A)) return e;
    GENS_CHECK_ARG(A->g, GENS_EINVAL, "gens_loss_bwd: null cotangent");
    const int64_t work = max(max(A->b, A->n_sparse), max(A->n_pseudo, (int64_t)1));
    loss_bwd_k<<<gens_blocks(work, 256), 256, 0, (hipStream_t)stream>>>(*A);
    return gens_launch_status("gens_loss_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// coarse depths of a render (implicit_surface.py:356-363): z = near + (far - near) * linspace(0, 1, n)[j] (+ (t_rand - 0.5) * 2 / n), the
// reference's float32 operations in its order (this file is compiled with -ffp-contract=off), one launch instead of eight.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coarse_z_k(const float* __restrict__ near, const float* __restrict__ far, int per_ray, const float* __restrict__ steps,
                                                  const float* __restrict__ t_rand, int64_t b, int n, float* __restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= b * n) return;
    const int64_t r = i / n;
    const int j = (int)(i - r * n);
    const float nr = near[per_ray ? r : 0], fr = far[per_ray ? r : 0];
    float v = nr + (fr - nr) * steps[j];
    if (t_rand) v = v + (t_rand[r] - 0.5f) * 2.0f / (float)n;
    z[i] = v;
}

extern "C" int gens_coarse_z(const float* near, const float* far, int per_ray, const float* steps, const float* t_rand, int64_t n_rays, int n, float* z,
                             void* stream) {
    GENS_CHECK_ARG(near && far && steps && z && n_rays >= 0 && n >= 1, GENS_EINVAL, "gens_coarse_z: bad argument");
    if (n_rays == 0) return 0;
    coarse_z_k<<<gens_blocks(n_rays * n, 256), 256, 0, (hipStream_t)stream>>>(near, far, per_ray, steps, t_rand, n_rays, n, z);
    return gens_launch_status("gens_coarse_z");
}

// ---------------------------------------------------------------------------------------------------------------
// the 23 parameter gradients of a BlendingNetwork from the batched products of gens_blend_train_bwd's operand rows, in one launch:
// cc = the eleven [dW_l | db_l] blocks (rows even(out_l), leading dimension even(in_l + 1)) concatenated; d loss / d s = sign(s) sum s_part.
// ---------------------------------------------------------------------------------------------------------------
struct BlendWgrad {
    const float* cc;
    const float* s_part;
    int n_part;
    const float* s;
    float* out[23];
    int outs[11], ins[11], off[11], first[12];      // off: block offset in cc; first: first output element of layer l (weights then bias)
};

__global__ __launch_bounds__(256) void blend_wgrad_k(BlendWgrad A) {
    if (blockIdx.x == gridDim.x - 1) {              // the last workgroup: the anti-alias temperature, sign(s) * sum of the partials
        __shared__ float red[4];
        float s = 0.0f;
        for (int k = threadIdx.x; k < A.n_part; k += 256) s += A.s_part[k];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float v = A.s[0];
            A.out[22][0] = (v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f)) * ((red[0] + red[1]) + (red[2] + red[3]));
        }
        return;
    }
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= A.first[11]) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 11; ++k) l += gid >= A.first[k] ? 1 : 0;
    const int e = gid - A.first[l], o_n = A.outs[l], i_n = A.ins[l], ld = (i_n + 2) / 2 * 2;
    if (e < o_n * i_n) A.out[2 * l][e] = A.cc[A.off[l] + (e / i_n) * ld + (e % i_n)];
    else A.out[2 * l + 1][e - o_n * i_n] = A.cc[A.off[l] + (e - o_n * i_n) * ld + i_n];
}

extern "C" int gens_blend_train_wgrad(const float* cc, const float* s_part, int n_part, const float* s, int n_feat, float* const* grads, void* stream) {
    GENS_CHECK_ARG(cc && s_part && s && grads && n_part >= 0, GENS_EINVAL, "gens_blend_train_wgrad: null pointer");
    const int f = n_feat;
    const int ins[11] = {4, 16, 3 * f, 64, 32, 32, 32, 32, 37, 16, 8}, outs[11] = {16, f, 64, 32, 32, 33, 32, 1, 16, 8, 1};
    BlendWgrad A;
    A.cc = cc; A.s_part = s_part; A.n_part = n_part; A.s = s;
    int off = 0, first = 0;
    for (int l = 0; l < 11; ++l) {
        GENS_CHECK_ARG(grads[2 * l] && grads[2 * l + 1], GENS_EINVAL, "gens_blend_train_wgrad: gradient buffer %d is null", l);
        A.out[2 * l] = grads[2 * l];
        A.out[2 * l + 1] = grads[2 * l + 1];
        A.outs[l] = outs[l]; A.ins[l] = ins[l];
        A.off[l] = off;
        A.first[l] = first;
        off += ((outs[l] + 1) / 2 * 2) * ((ins[l] + 2) / 2 * 2);
        first += outs[l] * (ins[l] + 1);
    }
    GENS_CHECK_ARG(grads[22], GENS_EINVAL, "gens_blend_train_wgrad: gradient buffer of s is null");
    A.out[22] = grads[22];
    A.first[11] = first;
    blend_wgrad_k<<<gens_blocks(first, 256) + 1, 256, 0, (hipStream_t)stream>>>(A);
    return gens_launch_status("gens_blend_train_wgrad");
}
