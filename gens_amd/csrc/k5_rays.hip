// K5-K8: the per-ray kernels of the NeuS-style renderer, one 64-lane wavefront per ray.
//
//   K5 up_sample        implicit_surface.py:60-109      K6 sample_pdf(det=True)   implicit_surface.py:14-44
//   K7 cat_z_vals       implicit_surface.py:111-133     K8 render_core composite  implicit_surface.py:160-168, 202-303
//
// A ray never has more than 128 samples (64 coarse + 4 x 16), so a wavefront keeps the whole ray in registers: lane l
// owns samples l and l+64.  The along-ray transmittance (torch.cumprod) and the CDF (torch.cumsum) are wavefront
// shuffle scans, searchsorted is a binary search over a 128-entry LDS row, torch.sort of [z | z_new] is a rank
// computation (both inputs are already sorted), and every reduction over samples (colour, normal, depth, eikonal,
// smoothness, visibility count, first sign change) is a wavefront reduction -- the reference spends ~40 (K5), ~10
// (K6/K7) and ~120 (K8) elementwise launches on (B,128) tensors for the same work.
#include "common.h"

#define RAYS_PER_BLOCK 4
#define MAX_SAMPLES 128

// value of a two-slot per-lane array at sample index idx (wave-uniform or per-lane idx)
__device__ __forceinline__ float pick2(float v0, float v1, int idx) {
    float a = __shfl(v0, idx & 63, 64), b = __shfl(v1, idx & 63, 64);
    return idx < 64 ? a : b;
}
// element j+1 of the two-slot array, seen from the owner of element j
__device__ __forceinline__ void next2(float v0, float v1, int lane, float& n0, float& n1) {
    (void)lane;
    n0 = lane_next(v0, lane_value(v1, 0));
    n1 = lane_next(v1, v1);
}
// element j-1 (zero for j = 0)
__device__ __forceinline__ void prev2(float v0, float v1, int lane, float& p0, float& p1) {
    (void)lane;
    p0 = lane_prev(v0, 0.0f);
    p1 = lane_prev(v1, lane_value(v0, 63));
}
// exclusive prefix product over the 128 slots
__device__ __forceinline__ void excl_cumprod2(float f0, float f1, int lane, float& t0, float& t1) {
    float p0 = wave_scan_mul(f0, lane);
    float p1 = wave_scan_mul(f1, lane);
    t0 = lane_prev(p0, 1.0f);
    t1 = lane_prev(p1, 1.0f) * lane_value(p0, 63);
}
// number of entries <= u in a sorted LDS row (torch.searchsorted(..., right=True))
__device__ __forceinline__ int upper_bound_lds(const float* row, int n, float u) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (row[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------------------------
// K5 + K6
// ---------------------------------------------------------------------------------------------------------------
// up_sample + sample_pdf for one ray held in registers (lane l owns samples l and l + 64): everything of implicit_surface.py:60-109, 14-44 after
// the samples, their SDF, radii and mask decisions are known.  Shared by gens_upsample and by the fused merge + up-sample kernel, so that the two
// produce the same bits.  s_cdf / s_z: this wave's LDS rows (s_z receives the z values; it may already hold them).
__device__ __forceinline__ void upsample_core(bool active, int lane, int64_t r, int n, int n_new, float inv_s, const LevelSet& ms, float ox, float oy,
                                              float oz, float dx, float dy, float dz, const float zj[2], const float sj[2], const float rad[2],
                                              const float vm[2], float* __restrict__ s_cdf_row, float* __restrict__ s_z_row, float* __restrict__ z_new,
                                              float* __restrict__ pts_new, uint8_t* __restrict__ valid_new) {
    float zn[2], sn[2], rn[2], vn[2];
    next2(zj[0], zj[1], lane, zn[0], zn[1]);
    next2(sj[0], sj[1], lane, sn[0], sn[1]);
    next2(rad[0], rad[1], lane, rn[0], rn[1]);
    next2(vm[0], vm[1], lane, vn[0], vn[1]);
    float cosv[2], dzv[2], midv[2], ins[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bool seg = (lane + 64 * s) < n - 1;
        dzv[s] = zn[s] - zj[s];
        midv[s] = (sj[s] + sn[s]) * 0.5f;
        cosv[s] = seg ? (sn[s] - sj[s]) / (dzv[s] + 1e-5f) : 0.0f;
        ins[s] = (seg && ((rad[s] < 1.0f) || (rn[s] < 1.0f)) && (vm[s] * vn[s] > 0.0f)) ? 1.0f : 0.0f;
    }
    float pc[2];
    prev2(cosv[0], cosv[1], lane, pc[0], pc[1]);
    float alpha[2], fac[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bool seg = (lane + 64 * s) < n - 1;
        float c = fminf(pc[s], cosv[s]);
        c = fminf(fmaxf(c, -1e3f), 0.0f) * ins[s];
        float e_prev = midv[s] - c * dzv[s] * 0.5f;
        float e_next = midv[s] + c * dzv[s] * 0.5f;
        float c_prev = sigmoidf_(e_prev * inv_s), c_next = sigmoidf_(e_next * inv_s);
        alpha[s] = seg ? (c_prev - c_next + 1e-5f) / (c_prev + 1e-5f) : 0.0f;
        fac[s] = seg ? 1.0f - alpha[s] + 1e-7f : 1.0f;
    }
    float tr[2];
    excl_cumprod2(fac[0], fac[1], lane, tr[0], tr[1]);
    // sample_pdf: weights + 1e-5, normalise, cumulative sum
    float wp[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) wp[s] = ((lane + 64 * s) < n - 1) ? alpha[s] * tr[s] + 1e-5f : 0.0f;
    float total = wave_sum(wp[0] + wp[1]);
    float c0 = wave_scan_add(wp[0] / total, lane);
    float c1 = wave_scan_add(wp[1] / total, lane) + lane_value(c0, 63);
    if (active) {
        if (lane == 0) s_cdf_row[0] = 0.0f;
        if (lane < n - 1) s_cdf_row[lane + 1] = c0;
        if (lane + 64 < n - 1) s_cdf_row[lane + 65] = c1;
        if (lane < n) s_z_row[lane] = zj[0];
        if (lane + 64 < n) s_z_row[lane + 64] = zj[1];
    }
    __syncthreads();
    if (active && lane < n_new) {
        float u = linspace_at(0.5f / (float)n_new, 1.0f - 0.5f / (float)n_new, n_new, lane);
        int cnt = upper_bound_lds(s_cdf_row, n, u);
        int lo = max(cnt - 1, 0), hi = min(cnt, n - 1);
        float c_lo = s_cdf_row[lo], c_hi = s_cdf_row[hi];
        float b_lo = s_z_row[lo], b_hi = s_z_row[hi];
        float den = c_hi - c_lo;
        if (den < 1e-5f) den = 1.0f;
        float t = (u - c_lo) / den;
        float zs = b_lo + t * (b_hi - b_lo);
        int64_t o = r * n_new + lane;
        z_new[o] = zs;
        float px = ox + dx * zs, py = oy + dy * zs, pz = oz + dz * zs;
        if (pts_new) {
            pts_new[3 * o] = px;
            pts_new[3 * o + 1] = py;
            pts_new[3 * o + 2] = pz;
        }
        if (valid_new) valid_new[o] = any_mask(ms, px, py, pz) ? 1 : 0;
    }
}


__global__ __launch_bounds__(64 * RAYS_PER_BLOCK) void upsample_k(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                                  const float* __restrict__ z, const float* __restrict__ sdf,
                                                                  int64_t n_rays, int n, int n_new, float inv_s, LevelSet ms,
                                                                  const uint8_t* __restrict__ valid_in, float* __restrict__ z_new,
                                                                  float* __restrict__ pts_new, uint8_t* __restrict__ valid_new) {
    __shared__ float s_cdf[RAYS_PER_BLOCK][MAX_SAMPLES];
    __shared__ float s_z[RAYS_PER_BLOCK][MAX_SAMPLES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wave;
    const bool active = r < n_rays;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    if (active) {
        ox = rays_o[3 * r]; oy = rays_o[3 * r + 1]; oz = rays_o[3 * r + 2];
        dx = rays_d[3 * r]; dy = rays_d[3 * r + 1]; dz = rays_d[3 * r + 2];
    }
    float zj[2], sj[2], rad[2], vm[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        bool have = active && j < n;
        zj[s] = have ? z[r * n + j] : 0.0f;
        sj[s] = have ? sdf[r * n + j] : 0.0f;
        float px = ox + dx * zj[s], py = oy + dy * zj[s], pz = oz + dz * zj[s];
        rad[s] = sqrtf(px * px + py * py + pz * pz);
        // validity of the existing samples: carried from the launches that created them (valid_in) or looked up again
        vm[s] = (have && (valid_in ? valid_in[r * n + j] != 0 : any_mask(ms, px, py, pz))) ? 1.0f : 0.0f;
    }
    upsample_core(active, lane, r, n, n_new, inv_s, ms, ox, oy, oz, dx, dy, dz, zj, sj, rad, vm, s_cdf[wave], s_z[wave], z_new, pts_new, valid_new);
}

// ---------------------------------------------------------------------------------------------------------------
// K7: sorted merge by rank (ties keep the older sample first)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * RAYS_PER_BLOCK) void merge_k(const float* __restrict__ z, const float* __restrict__ sdf,
                                                               const float* __restrict__ z_new, const float* __restrict__ sdf_new,
                                                               const uint8_t* __restrict__ valid, const uint8_t* __restrict__ valid_new,
                                                               int64_t n_rays, int n, int n_new, float* __restrict__ z_out,
                                                               float* __restrict__ sdf_out, uint8_t* __restrict__ valid_out) {
    __shared__ float s_z[RAYS_PER_BLOCK][MAX_SAMPLES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wave;
    const bool active = r < n_rays;
    const int m = n + n_new;
    float zj[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        zj[s] = (active && j < n) ? z[r * n + j] : 0.0f;
        if (active && j < n) s_z[wave][j] = zj[s];
    }
    float zk = (active && lane < n_new) ? z_new[r * n_new + lane] : 0.0f;
    __syncthreads();
    if (!active) return;
    int less[2] = {0, 0}, before = 0;
    for (int k = 0; k < n_new; ++k) {
        float v = lane_value(zk, k);
        less[0] += (v < zj[0]) ? 1 : 0;
        less[1] += (v < zj[1]) ? 1 : 0;
        before += (v < zk || (v == zk && k < lane)) ? 1 : 0;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        if (j < n) {
            int64_t o = r * m + j + less[s];
            z_out[o] = zj[s];
            if (sdf_out) sdf_out[o] = sdf[r * n + j];
            if (valid_out) valid_out[o] = valid[r * n + j];
        }
    }
    if (lane < n_new) {
        int64_t o = r * m + upper_bound_lds(s_z[wave], n, zk) + before;
        z_out[o] = zk;
        if (sdf_out) sdf_out[o] = sdf_new[r * n_new + lane];
        if (valid_out) valid_out[o] = valid_new[r * n_new + lane];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K7 + K5/K6 (+ K3) fused: one launch per sampling round.  cat_z_vals of round i (implicit_surface.py:111-133) and up_sample + sample_pdf of
// round i + 1 (:60-109) work on the same ray: the merged samples stay in the wave's LDS rows and registers instead of going to HBM and back
// (the merge's (n + 16) x 9 bytes per ray are still written -- the next round reads them -- but not re-read by a second kernel), and a round is
// one launch instead of two.  MODE 0: merge + up-sample -> merged (z, sdf, valid) and the next 16 samples with their points and mask decisions.
// MODE 1 (the last round, :129-131 merges z only): merge + render_core's section mid-points and their mask decisions (:163-168, 170-173; what
// gens_ray_points(mid) computes).  Same arithmetic as merge_k / upsample_k / ray_points_k: the results are theirs bit for bit (tests).
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64 * RAYS_PER_BLOCK) void merge_upsample_k(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                                        const float* __restrict__ z, const float* __restrict__ sdf,
                                                                        const uint8_t* __restrict__ valid, const float* __restrict__ z_add,
                                                                        const float* __restrict__ sdf_add, const uint8_t* __restrict__ valid_add,
                                                                        int64_t n_rays, int n, int n_add, int n_new, float inv_s, float sample_dist,
                                                                        LevelSet ms, float* __restrict__ z_out, float* __restrict__ sdf_out,
                                                                        uint8_t* __restrict__ valid_out, float* __restrict__ z_new,
                                                                        float* __restrict__ pts_out, uint8_t* __restrict__ valid_new) {
    __shared__ float s_old[RAYS_PER_BLOCK][MAX_SAMPLES];      // the old samples' z (the new ones' upper bounds are searched in it)
    __shared__ float m_z[RAYS_PER_BLOCK][MAX_SAMPLES];        // merged z: also the bin edges of the up-sampling
    __shared__ float m_s[RAYS_PER_BLOCK][MAX_SAMPLES];
    __shared__ float s_cdf[RAYS_PER_BLOCK][MAX_SAMPLES];
    __shared__ uint8_t m_v[RAYS_PER_BLOCK][MAX_SAMPLES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wave;
    const bool active = r < n_rays;
    const int m = n + n_add;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    if (active) {
        ox = rays_o[3 * r]; oy = rays_o[3 * r + 1]; oz = rays_o[3 * r + 2];
        dx = rays_d[3 * r]; dy = rays_d[3 * r + 1]; dz = rays_d[3 * r + 2];
    }
    // ---- merge by rank (merge_k)
    float zo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        zo[s] = (active && j < n) ? z[r * n + j] : 0.0f;
        if (active && j < n) s_old[wave][j] = zo[s];
    }
    const float zk = (active && lane < n_add) ? z_add[r * n_add + lane] : 0.0f;
    __syncthreads();
    if (active) {
        int less[2] = {0, 0}, before = 0;
        for (int k = 0; k < n_add; ++k) {
            const float v = lane_value(zk, k);
            less[0] += (v < zo[0]) ? 1 : 0;
            less[1] += (v < zo[1]) ? 1 : 0;
            before += (v < zk || (v == zk && k < lane)) ? 1 : 0;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int j = lane + 64 * s;
            if (j < n) {
                const int pos = j + less[s];
                const int64_t o = r * m + pos;
                m_z[wave][pos] = zo[s];
                z_out[o] = zo[s];
                if (MODE == 0) {
                    const float sv = sdf[r * n + j];
                    const uint8_t vv = valid[r * n + j];
                    m_s[wave][pos] = sv;
                    m_v[wave][pos] = vv;
                    sdf_out[o] = sv;
                    valid_out[o] = vv;
                }
            }
        }
        if (lane < n_add) {
            const int pos = upper_bound_lds(s_old[wave], n, zk) + before;
            const int64_t o = r * m + pos;
            m_z[wave][pos] = zk;
            z_out[o] = zk;
            if (MODE == 0) {
                const float sv = sdf_add[r * n_add + lane];
                const uint8_t vv = valid_add[r * n_add + lane];
                m_s[wave][pos] = sv;
                m_v[wave][pos] = vv;
                sdf_out[o] = sv;
                valid_out[o] = vv;
            }
        }
    }
    __syncthreads();
    // ---- the merged ray, lane l owns samples l and l + 64
    float zj[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        zj[s] = (active && j < m) ? m_z[wave][j] : 0.0f;
    }
    if (MODE == 0) {
        float sj[2], rad[2], vm[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int j = lane + 64 * s;
            const bool have = active && j < m;
            sj[s] = have ? m_s[wave][j] : 0.0f;
            const float px = ox + dx * zj[s], py = oy + dy * zj[s], pz = oz + dz * zj[s];
            rad[s] = sqrtf(px * px + py * py + pz * pz);
            vm[s] = (have && m_v[wave][j] != 0) ? 1.0f : 0.0f;
        }
        upsample_core(active, lane, r, m, n_new, inv_s, ms, ox, oy, oz, dx, dy, dz, zj, sj, rad, vm, s_cdf[wave], m_z[wave], z_new, pts_out, valid_new);
    } else {
        // section mid-points (ray_points_k with mid = 1): t = z + dist / 2, dist = the gap to the next sample or sample_dist behind the last (Q10)
        float zn[2];
        next2(zj[0], zj[1], lane, zn[0], zn[1]);
        if (!active) return;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int j = lane + 64 * s;
            if (j >= m) continue;
            const float dist = (j + 1 < m) ? zn[s] - zj[s] : sample_dist;
            const float t = zj[s] + dist * 0.5f;
            const float px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;
            const int64_t o = r * m + j;
            pts_out[3 * o] = px;
            pts_out[3 * o + 1] = py;
            pts_out[3 * o + 2] = pz;
            valid_new[o] = any_mask(ms, px, py, pz) ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K8 forward
// ---------------------------------------------------------------------------------------------------------------
struct RaySample {  // everything the forward and backward share, per slot
    float z, dist, mid, vm, inside, relax, sdf, gx, gy, gz, gn, tc, ic, ep, en, cp, cn, q, alpha, fac;
};

__device__ __forceinline__ void ray_setup(const gens_composite_in& in, int64_t r, int lane, bool active, float inv_s, float d[3],
                                          float o[3], RaySample smp[2]) {
    const int n = in.n;
    if (active) {
        for (int a = 0; a < 3; ++a) { o[a] = in.rays_o[3 * r + a]; d[a] = in.rays_d[3 * r + a]; }
    } else {
        for (int a = 0; a < 3; ++a) { o[a] = 0.0f; d[a] = 0.0f; }
    }
    float zj[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        zj[s] = (active && j < n) ? in.z[r * n + j] : 0.0f;
    }
    float zn[2];
    next2(zj[0], zj[1], lane, zn[0], zn[1]);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        bool have = active && j < n;
        RaySample& p = smp[s];
        p.z = zj[s];
        p.dist = (j + 1 < n) ? zn[s] - zj[s] : in.sample_dist;                       // (Q10)
        p.mid = p.z + p.dist * 0.5f;
        float px = o[0] + d[0] * p.mid, py = o[1] + d[1] * p.mid, pz = o[2] + d[2] * p.mid;
        float norm = sqrtf(px * px + py * py + pz * pz);
        int64_t e = r * n + j;
        p.vm = (have && in.voxel_mask[e]) ? 1.0f : 0.0f;
        p.inside = (norm < 1.0f ? 1.0f : 0.0f) * p.vm;
        p.relax = (norm < 1.2f ? 1.0f : 0.0f) * p.vm;
        p.sdf = have ? in.sdf[e] : 0.0f;
        p.gx = have ? in.grad[3 * e] : 0.0f;
        p.gy = have ? in.grad[3 * e + 1] : 0.0f;
        p.gz = have ? in.grad[3 * e + 2] : 0.0f;
        p.gn = sqrtf(p.gx * p.gx + p.gy * p.gy + p.gz * p.gz);
        p.tc = d[0] * p.gx + d[1] * p.gy + d[2] * p.gz;
        const float a = in.cos_anneal_dev ? *in.cos_anneal_dev : in.cos_anneal;   // (uniform: a scalar load)
        p.ic = -(fmaxf(-p.tc * 0.5f + 0.5f, 0.0f) * (1.0f - a) + fmaxf(-p.tc, 0.0f) * a);
        p.ic *= p.vm;
        float icc = fminf(fmaxf(p.ic, -10.0f), 10.0f);
        p.en = p.sdf + icc * p.dist * 0.5f;
        p.ep = p.sdf - icc * p.dist * 0.5f;
        p.cp = sigmoidf_(p.ep * inv_s);
        p.cn = sigmoidf_(p.en * inv_s);
        p.q = (p.cp - p.cn + 1e-5f) / (p.cp + 1e-5f);
        p.alpha = have ? fminf(fmaxf(p.q, 0.0f), 1.0f) * p.vm : 0.0f;                // (Q9)
        p.fac = have ? 1.0f - p.alpha + 1e-7f : 1.0f;
    }
}

__global__ __launch_bounds__(64 * RAYS_PER_BLOCK) void composite_fwd_k(gens_composite_in in, gens_composite_out out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wave;
    if (r >= in.n_rays) return;  // no block-level sync in this kernel: whole waves may leave
    const int n = in.n;
    const float inv_s = in.inv_s[0];
    float d[3], o[3];
    RaySample smp[2];
    ray_setup(in, r, lane, true, inv_s, d, o, smp);
    float tr[2];
    excl_cumprod2(smp[0].fac, smp[1].fac, lane, tr[0], tr[1]);
    float acc[16];
    for (int a = 0; a < 16; ++a) acc[a] = 0.0f;
    float wmax = 0.0f, wv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        const RaySample& p = smp[s];
        float w = p.alpha * tr[s];
        wv[s] = w;
        if (j < n) {
            int64_t e = r * n + j;
            out.weights[e] = w;
            out.inside[e] = p.inside;
            acc[0] += in.color[3 * e] * w;
            acc[1] += in.color[3 * e + 1] * w;
            acc[2] += in.color[3 * e + 2] * w;
            acc[3] += p.gx * w;
            acc[4] += p.gy * w;
            acc[5] += p.gz * w;
            acc[6] += p.mid * w;
            acc[7] += w;
            acc[8] += p.relax * (p.gn - 1.0f) * (p.gn - 1.0f);
            acc[9] += p.relax;
            if (in.smooth) {
                float k = w * p.inside;
                acc[10] += in.smooth[3 * e] * k;
                acc[11] += in.smooth[3 * e + 1] * k;
                acc[12] += in.smooth[3 * e + 2] * k;
            }
            if (in.src_vis) {
                int c = 0;
                for (int v = 0; v < in.n_src; ++v) c += in.src_vis[e * in.n_src + v] ? 1 : 0;
                acc[13] += (c > 1) ? 1.0f : 0.0f;                                       // (Q12)
            }
            wmax = fmaxf(wmax, w);
        }
    }
    for (int a = 0; a < 14; ++a) acc[a] = wave_sum(acc[a]);
    wmax = wave_max(wmax);
    // first masked sign change (implicit_surface.py:262-275): argmax(sign * reversed_index * pair_ok)
    float sn[2], vn[2];
    next2(smp[0].sdf, smp[1].sdf, lane, sn[0], sn[1]);
    next2(smp[0].vm, smp[1].vm, lane, vn[0], vn[1]);
    bool f0 = (lane < n - 1) && (smp[0].sdf * sn[0] <= 0.0f) && (smp[0].vm * vn[0] > 0.0f);
    bool f1 = (lane + 64 < n - 1) && (smp[1].sdf * sn[1] <= 0.0f) && (smp[1].vm * vn[1] > 0.0f);
    unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
    bool any = (b0 | b1) != 0ull;
    int i0 = b0 ? __ffsll((long long)b0) - 1 : (b1 ? 63 + __ffsll((long long)b1) : 0);
    int i1 = i0 + 1;
    float in0 = pick2(smp[0].inside, smp[1].inside, i0), in1 = pick2(smp[0].inside, smp[1].inside, i1);
    float g0x = pick2(smp[0].gx, smp[1].gx, i0), g0y = pick2(smp[0].gy, smp[1].gy, i0), g0z = pick2(smp[0].gz, smp[1].gz, i0);
    float g1x = pick2(smp[0].gx, smp[1].gx, i1), g1y = pick2(smp[0].gy, smp[1].gy, i1), g1z = pick2(smp[0].gz, smp[1].gz, i1);
    float n0 = pick2(smp[0].gn, smp[1].gn, i0), n1 = pick2(smp[0].gn, smp[1].gn, i1);
    float s0 = pick2(smp[0].sdf, smp[1].sdf, i0), s1 = pick2(smp[0].sdf, smp[1].sdf, i1);
    float z0 = pick2(smp[0].mid, smp[1].mid, i0), z1 = pick2(smp[0].mid, smp[1].mid, i1);
    if (lane == 0) {
        const float* R = in.rot_dev ? in.rot_dev : in.rot;
        float camz = R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
        out.color[3 * r] = acc[0];
        out.color[3 * r + 1] = acc[1];
        out.color[3 * r + 2] = acc[2];
        out.normal[3 * r] = R[0] * acc[3] + R[1] * acc[4] + R[2] * acc[5];
        out.normal[3 * r + 1] = R[3] * acc[3] + R[4] * acc[4] + R[5] * acc[5];
        out.normal[3 * r + 2] = R[6] * acc[3] + R[7] * acc[4] + R[8] * acc[5];
        out.depth[r] = acc[6] * camz;                                                   // (Q11)
        out.wsum[r] = acc[7];
        out.wmax[r] = wmax;
        out.eik_num[r] = acc[8];
        out.eik_den[r] = acc[9];
        out.smooth_vec[3 * r] = acc[10];
        out.smooth_vec[3 * r + 1] = acc[11];
        out.smooth_vec[3 * r + 2] = acc[12];
        out.valid[r] = acc[13] > 8.0f ? 1 : 0;
        float mid_in = (0.5f * (in0 + in1) > 0.5f) ? 1.0f : 0.0f;
        mid_in *= any ? 1.0f : 0.0f;
        float cosd = (g0x * g1x + g0y * g1y + g0z * g1z) / (n0 * n1 + 1e-8f);
        mid_in *= (cosd > 0.5f) ? 1.0f : 0.0f;
        float zc = (s0 * z1 - s1 * z0) / (s0 - s1 + 1e-10f);
        out.mid_in[r] = mid_in;
        out.sdf_depth[r] = zc * camz * mid_in;
        if (zc < 0.0f) zc = 0.0f;
        if (zc > in.z_max[0]) zc = 0.0f;
        out.z_cross[r] = zc;
        out.cross_idx[r] = i0;
        if (out.pts_cross) {                                    // pts_sdf0 = o + d z (:304), for the surface-point gradient launch
            out.pts_cross[3 * r] = o[0] + d[0] * zc;
            out.pts_cross[3 * r + 1] = o[1] + d[1] * zc;
            out.pts_cross[3 * r + 2] = o[2] + d[2] * zc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K8 backward
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * RAYS_PER_BLOCK) void composite_bwd_k(gens_composite_in in, gens_composite_grad g) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wave;
    if (r >= in.n_rays) return;
    const int n = in.n;
    const float inv_s = in.inv_s[0];
    float d[3], o[3];
    RaySample smp[2];
    ray_setup(in, r, lane, true, inv_s, d, o, smp);
    float tr[2];
    excl_cumprod2(smp[0].fac, smp[1].fac, lane, tr[0], tr[1]);
    const float* R = in.rot_dev ? in.rot_dev : in.rot;
    float camz = R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
    float gc[3] = {0, 0, 0}, gnr[3] = {0, 0, 0}, gsv[3] = {0, 0, 0};
    if (g.g_color) for (int a = 0; a < 3; ++a) gc[a] = g.g_color[3 * r + a];
    if (g.g_normal) {  // normal = R * nsum  ->  d/dnsum = R^T g
        float t0 = g.g_normal[3 * r], t1 = g.g_normal[3 * r + 1], t2 = g.g_normal[3 * r + 2];
        gnr[0] = R[0] * t0 + R[3] * t1 + R[6] * t2;
        gnr[1] = R[1] * t0 + R[4] * t1 + R[7] * t2;
        gnr[2] = R[2] * t0 + R[5] * t1 + R[8] * t2;
    }
    if (g.g_smooth_vec) for (int a = 0; a < 3; ++a) gsv[a] = g.g_smooth_vec[3 * r + a];
    float gd = g.g_depth ? g.g_depth[r] * camz : 0.0f;
    float gws = g.g_wsum ? g.g_wsum[r] : 0.0f;
    float geik = g.g_eik_num ? g.g_eik_num[r] : 0.0f;
    // cotangents of the two per-batch scalars gens_composite_finish_fwd formed from the per-ray outputs (implicit_surface.py:248-253):
    //   gradient_error = sum eik_num / (sum eik_den + 1e-5);  smooth_error = mean over rays of |smooth_vec| (norm's subgradient 0 at 0)
    if (g.g_gradient_error) geik += g.g_gradient_error[0] / (g.finish[1] + 1e-5f);
    if (g.g_smooth_error && g.smooth_vec) {
        const float v0 = g.smooth_vec[3 * r], v1 = g.smooth_vec[3 * r + 1], v2 = g.smooth_vec[3 * r + 2];
        const float nv = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
        const float k = nv > 0.0f ? g.g_smooth_error[0] / (nv * (float)in.n_rays) : 0.0f;
        gsv[0] += k * v0; gsv[1] += k * v1; gsv[2] += k * v2;
    }

    float gw[2], w[2], gww[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        const RaySample& p = smp[s];
        w[s] = p.alpha * tr[s];
        gw[s] = 0.0f;
        if (j < n) {
            int64_t e = r * n + j;
            gw[s] = gc[0] * in.color[3 * e] + gc[1] * in.color[3 * e + 1] + gc[2] * in.color[3 * e + 2] + gnr[0] * p.gx +
                    gnr[1] * p.gy + gnr[2] * p.gz + gd * p.mid + gws + (g.g_weights ? g.g_weights[e] : 0.0f);
        }
        gww[s] = gw[s] * w[s];
    }
    // S_j = sum_{k>j} gw_k w_k
    float inc0 = wave_scan_add(gww[0], lane), inc1 = wave_scan_add(gww[1], lane);
    float tot0 = lane_value(inc0, 63), tot1 = lane_value(inc1, 63);
    float suf[2] = {tot0 + tot1 - inc0, tot1 - inc1};
    float gs_acc = 0.0f;
    const int i0 = g.cross_idx ? g.cross_idx[r] : 0;
    float gz0 = 0.0f, gz1 = 0.0f;
    if (g.g_z_cross) {
        float s0 = pick2(smp[0].sdf, smp[1].sdf, i0), s1 = pick2(smp[0].sdf, smp[1].sdf, i0 + 1);
        float z0 = pick2(smp[0].mid, smp[1].mid, i0), z1 = pick2(smp[0].mid, smp[1].mid, i0 + 1);
        float den = s0 - s1 + 1e-10f, num = s0 * z1 - s1 * z0;
        float zc = num / den;
        float up = (zc < 0.0f || zc > in.z_max[0]) ? 0.0f : g.g_z_cross[r];
        gz0 = up * (z1 * den - num) / (den * den);
        gz1 = up * (-z0 * den + num) / (den * den);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int j = lane + 64 * s;
        if (j >= n) continue;
        const RaySample& p = smp[s];
        int64_t e = r * n + j;
        float ga = gw[s] * tr[s] - suf[s] / p.fac;
        float gq = (p.q >= 0.0f && p.q <= 1.0f) ? ga * p.vm : 0.0f;
        float gcp = gq * p.cn / ((p.cp + 1e-5f) * (p.cp + 1e-5f));
        float gcn = -gq / (p.cp + 1e-5f);
        float dcp = p.cp * (1.0f - p.cp), dcn = p.cn * (1.0f - p.cn);
        float gep = gcp * inv_s * dcp, gen = gcn * inv_s * dcn;
        gs_acc += gcp * p.ep * dcp + gcn * p.en * dcn;
        float gsdf = gep + gen;
        float gicc = (gen - gep) * p.dist * 0.5f;
        float gic = (p.ic >= -10.0f && p.ic <= 10.0f) ? gicc * p.vm : 0.0f;
        const float a = in.cos_anneal_dev ? *in.cos_anneal_dev : in.cos_anneal;   // (uniform: a scalar load)
        float dic = ((-p.tc * 0.5f + 0.5f) > 0.0f ? 0.5f * (1.0f - a) : 0.0f) + ((-p.tc) > 0.0f ? a : 0.0f);
        float gtc = gic * dic;
        float ek = (p.gn > 0.0f) ? geik * p.relax * 2.0f * (p.gn - 1.0f) / p.gn : 0.0f;
        if (j == i0) gsdf += gz0;
        if (j == i0 + 1) gsdf += gz1;
        g.g_sdf[e] = gsdf;
        g.g_grad[3 * e] = gtc * d[0] + w[s] * gnr[0] + ek * p.gx;
        g.g_grad[3 * e + 1] = gtc * d[1] + w[s] * gnr[1] + ek * p.gy;
        g.g_grad[3 * e + 2] = gtc * d[2] + w[s] * gnr[2] + ek * p.gz;
        g.g_col[3 * e] = w[s] * gc[0];
        g.g_col[3 * e + 1] = w[s] * gc[1];
        g.g_col[3 * e + 2] = w[s] * gc[2];
        if (g.g_smooth) {
            float k = w[s] * p.inside;
            g.g_smooth[3 * e] = gsv[0] * k;
            g.g_smooth[3 * e + 1] = gsv[1] * k;
            g.g_smooth[3 * e + 2] = gsv[2] * k;
        }
    }
    gs_acc = wave_sum(gs_acc);
    if (lane == 0 && g.g_inv_s) g.g_inv_s[r] = gs_acc;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int gens_fill_levels(const char* who, LevelSet* ls, const float* const* data, const int* dims, int n_levels);

extern "C" int gens_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t n_rays, int n,
                             int n_new, float inv_s, const float* const* masks, const int* dims, int n_levels, int mask_bits,
                             const uint8_t* valid_in, float* z_new, float* pts_new, uint8_t* valid_new, void* stream) {
    LevelSet ms;
    if (int e = gens_fill_levels("gens_upsample", &ms, masks, dims, n_levels)) return e;
    ms.bits = mask_bits ? 1 : 0;
    GENS_CHECK_ARG(n >= 2 && n <= MAX_SAMPLES && n_new >= 1 && n_new <= 64, GENS_ELIMIT, "gens_upsample: n=%d (2..128) n_new=%d (1..64)", n, n_new);
    GENS_CHECK_ARG(n_rays >= 0 && (n_rays == 0 || (rays_o && rays_d && z && sdf && z_new)), GENS_EINVAL, "gens_upsample: null pointer");
    if (n_rays == 0) return 0;
    upsample_k<<<gens_blocks(n_rays, RAYS_PER_BLOCK), 64 * RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, z, sdf, n_rays, n, n_new, inv_s, ms, valid_in, z_new, pts_new, valid_new);
    return gens_launch_status("gens_upsample");
}

extern "C" int gens_merge_samples(const float* z, const float* sdf, const float* z_new, const float* sdf_new, const uint8_t* valid,
                                  const uint8_t* valid_new, int64_t n_rays, int n, int n_new, float* z_out, float* sdf_out,
                                  uint8_t* valid_out, void* stream) {
    GENS_CHECK_ARG(n >= 1 && n_new >= 1 && n_new <= 64 && n + n_new <= MAX_SAMPLES, GENS_ELIMIT,
                   "gens_merge_samples: n=%d n_new=%d (n_new <= 64, n+n_new <= 128)", n, n_new);
    GENS_CHECK_ARG(n_rays >= 0 && (n_rays == 0 || (z && z_new && z_out)), GENS_EINVAL, "gens_merge_samples: null pointer");
    GENS_CHECK_ARG(!sdf_out || (sdf && sdf_new), GENS_EINVAL, "gens_merge_samples: sdf_out needs sdf and sdf_new");
    GENS_CHECK_ARG(!valid_out || (valid && valid_new), GENS_EINVAL, "gens_merge_samples: valid_out needs valid and valid_new");
    if (n_rays == 0) return 0;
    merge_k<<<gens_blocks(n_rays, RAYS_PER_BLOCK), 64 * RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(z, sdf, z_new, sdf_new, valid, valid_new,
                                                                                              n_rays, n, n_new, z_out, sdf_out, valid_out);
    return gens_launch_status("gens_merge_samples");
}

// merge (z, sdf, valid) with the previous round's (z_add, sdf_add, valid_add) and, in the same launch,
//   mode 0: up-sample the merged ray: -> merged arrays (n + n_add) and z_new (n_new), pts_out (n_rays n_new, 3), valid_new;
//   mode 1: section mid-points of the merged ray (sdf / valid arrays unused, may be NULL): -> z_out, pts_out (n_rays (n + n_add), 3), valid_new.
extern "C" int gens_merge_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const uint8_t* valid,
                                   const float* z_add, const float* sdf_add, const uint8_t* valid_add, int64_t n_rays, int n, int n_add, int n_new,
                                   float inv_s, float sample_dist, const float* const* masks, const int* dims, int n_levels, int mask_bits, int mode,
                                   float* z_out, float* sdf_out, uint8_t* valid_out, float* z_new, float* pts_out, uint8_t* valid_new, void* stream) {
    LevelSet ms;
    if (int e = gens_fill_levels("gens_merge_upsample", &ms, masks, dims, n_levels)) return e;
    ms.bits = mask_bits ? 1 : 0;
    GENS_CHECK_ARG(mode == 0 || mode == 1, GENS_EINVAL, "gens_merge_upsample: mode %d (0 = merge + up-sample, 1 = merge + mid-points)", mode);
    GENS_CHECK_ARG(n >= 1 && n_add >= 1 && n_add <= 64 && n + n_add <= MAX_SAMPLES && n + n_add >= 2, GENS_ELIMIT,
                   "gens_merge_upsample: n=%d n_add=%d (n_add <= 64, 2 <= n + n_add <= 128)", n, n_add);
    GENS_CHECK_ARG(mode == 1 || (n_new >= 1 && n_new <= 64), GENS_ELIMIT, "gens_merge_upsample: n_new=%d (1..64)", n_new);
    GENS_CHECK_ARG(n_rays >= 0 && (n_rays == 0 || (rays_o && rays_d && z && z_add && z_out && pts_out && valid_new)), GENS_EINVAL,
                   "gens_merge_upsample: null pointer");
    GENS_CHECK_ARG(mode == 1 || n_rays == 0 || (sdf && valid && sdf_add && valid_add && sdf_out && valid_out && z_new), GENS_EINVAL,
                   "gens_merge_upsample: mode 0 needs the sdf / valid arrays and z_new");
    if (n_rays == 0) return 0;
    const dim3 grid(gens_blocks(n_rays, RAYS_PER_BLOCK));
    if (mode == 0)
        merge_upsample_k<0><<<grid, 64 * RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(rays_o, rays_d, z, sdf, valid, z_add, sdf_add, valid_add, n_rays, n, n_add,
                                                                                 n_new, inv_s, sample_dist, ms, z_out, sdf_out, valid_out, z_new, pts_out, valid_new);
    else
        merge_upsample_k<1><<<grid, 64 * RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(rays_o, rays_d, z, sdf, valid, z_add, sdf_add, valid_add, n_rays, n, n_add,
                                                                                 n_new, inv_s, sample_dist, ms, z_out, sdf_out, valid_out, z_new, pts_out, valid_new);
    return gens_launch_status("gens_merge_upsample");
}

static int check_composite_in(const char* who, const gens_composite_in* in) {
    GENS_CHECK_ARG(in, GENS_EINVAL, "%s: null input block", who);
    GENS_CHECK_ARG(in->n >= 2 && in->n <= MAX_SAMPLES, GENS_ELIMIT, "%s: n=%d not in 2..128", who, in->n);
    GENS_CHECK_ARG(in->n_rays >= 0, GENS_EINVAL, "%s: negative ray count", who);
    GENS_CHECK_ARG(in->n_rays == 0 || (in->rays_o && in->rays_d && in->z && in->sdf && in->grad && in->color && in->voxel_mask &&
                                       in->inv_s && in->z_max),
                   GENS_EINVAL, "%s: null input pointer", who);
    GENS_CHECK_ARG(!in->src_vis || (in->n_src >= 1 && in->n_src < GENS_MAX_VIEWS), GENS_ELIMIT, "%s: n_src=%d", who, in->n_src);
    return 0;
}

extern "C" int gens_composite_fwd(const gens_composite_in* in, const gens_composite_out* out, void* stream) {
    if (int e = check_composite_in("gens_composite_fwd", in)) return e;
    GENS_CHECK_ARG(out && out->color && out->normal && out->depth && out->wsum && out->wmax && out->mid_in && out->sdf_depth &&
                       out->z_cross && out->eik_num && out->eik_den && out->smooth_vec && out->valid && out->cross_idx &&
                       out->weights && out->inside,
                   GENS_EINVAL, "gens_composite_fwd: null output pointer");
    if (in->n_rays == 0) return 0;
    composite_fwd_k<<<gens_blocks(in->n_rays, RAYS_PER_BLOCK), 64 * RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(*in, *out);
    return gens_launch_status("gens_composite_fwd");
}

// gradient_error and smooth_error of a batch from the per-ray outputs (one workgroup, fixed summation order):
//   finish = {sum eik_num, sum eik_den, gradient_error, smooth_error}
__global__ __launch_bounds__(256) void composite_finish_fwd_k(const float* __restrict__ eik_num, const float* __restrict__ eik_den,
                                                              const float* __restrict__ smooth_vec, int64_t b, float* __restrict__ finish) {
    __shared__ float red[3][4];
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (int64_t r = threadIdx.x; r < b; r += 256) {
        s0 += eik_num[r];
        s1 += eik_den[r];
        if (smooth_vec) {
            const float v0 = smooth_vec[3 * r], v1 = smooth_vec[3 * r + 1], v2 = smooth_vec[3 * r + 2];
            s2 += sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; red[2][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float num = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), den = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float sm = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        finish[0] = num;
        finish[1] = den;
        finish[2] = num / (den + 1e-5f);
        finish[3] = b > 0 ? sm / (float)b : 0.0f;
    }
}

extern "C" int gens_composite_finish_fwd(const float* eik_num, const float* eik_den, const float* smooth_vec, int64_t n_rays, float* finish,
                                         void* stream) {
    GENS_CHECK_ARG(eik_num && eik_den && finish && n_rays >= 0, GENS_EINVAL, "gens_composite_finish_fwd: null pointer");
    composite_finish_fwd_k<<<1, 256, 0, (hipStream_t)stream>>>(eik_num, eik_den, smooth_vec, n_rays, finish);
    return gens_launch_status("gens_composite_finish_fwd");
}

// after gens_composite_bwd: d loss / d variance from the per-ray partials of d loss / d inv_s (inv_s = clip(exp(10 variance), 1e-6, 1e6):
// scalars = gens_compact_points' {z_max, inv_s, 1 / inv_s, inside the clip range}), and zeros into the rows [n_ray, n_all) of the dense
// gradient arrays that the compositing does not touch (the random / pseudo points of the step's evaluation set)
__global__ __launch_bounds__(256) void composite_finish_bwd_k(const float* __restrict__ g_inv_s, int64_t b, const float* __restrict__ scalars,
                                                              float* __restrict__ g_variance, float* __restrict__ g_sdf, float* __restrict__ g_grad,
                                                              float* __restrict__ g_smooth, int64_t n_ray, int64_t n_all) {
    __shared__ float red[4];
    if (blockIdx.x == 0) {
        float s = 0.0f;
        for (int64_t r = threadIdx.x; r < b; r += 256) s += g_inv_s[r];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0 && g_variance) g_variance[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scalars[1] * scalars[3] * 10.0f;
    }
    const int64_t tail = n_all - n_ray;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tail; i += (int64_t)gridDim.x * 256) {
        const int64_t row = n_ray + i;
        if (g_sdf) g_sdf[row] = 0.0f;
        if (g_grad) { g_grad[3 * row] = 0.0f; g_grad[3 * row + 1] = 0.0f; g_grad[3 * row + 2] = 0.0f; }
        if (g_smooth) { g_smooth[3 * row] = 0.0f; g_smooth[3 * row + 1] = 0.0f; g_smooth[3 * row + 2] = 0.0f; }
    }
}

extern "C" int gens_composite_finish_bwd(const float* g_inv_s, int64_t n_rays, const float* scalars, float* g_variance, float* g_sdf, float* g_grad,
                                         float* g_smooth, int64_t n_ray_pts, int64_t n_all, void* stream) {
    GENS_CHECK_ARG(g_inv_s && scalars && n_rays >= 0 && n_all >= n_ray_pts, GENS_EINVAL, "gens_composite_finish_bwd: bad argument");
    const int64_t tail = n_all - n_ray_pts;
    const unsigned grid = tail > 0 ? (unsigned)min((int64_t)64, (tail + 255) / 256) : 1u;
    composite_finish_bwd_k<<<grid, 256, 0, (hipStream_t)stream>>>(g_inv_s, n_rays, scalars, g_variance, g_sdf, g_grad, g_smooth, n_ray_pts, n_all);
    return gens_launch_status("gens_composite_finish_bwd");
}

extern "C" int gens_composite_bwd(const gens_composite_in* in, const gens_composite_grad* g, void* stream) {
    if (int e = check_composite_in("gens_composite_bwd", in)) return e;
    GENS_CHECK_ARG(g && g->g_sdf && g->g_grad && g->g_col, GENS_EINVAL, "gens_composite_bwd: null gradient output");
    GENS_CHECK_ARG(!g->g_z_cross || g->cross_idx, GENS_EINVAL, "gens_composite_bwd: g_z_cross needs cross_idx");
    if (in->n_rays == 0) return 0;
    composite_bwd_k<<<gens_blocks(in->n_rays, RAYS_PER_BLOCK), 64 * RAYS_PER_BLOCK, 0, (hipStream_t)stream>>>(*in, *g);
    return gens_launch_status("gens_composite_bwd");
}
