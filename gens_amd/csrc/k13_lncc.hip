// K13: the photometric patch loss statistic compute_LNCC
// (/root/reference/models/losses/ncc.py:7-55, called from loss.py:36 on render_core's ref_gray_val / sampled_gray_val).
//
// The reference builds five (B*S, C, 11, 11) tensors and runs five grouped all-ones 11x11 convolutions only to read
// their centre tap -- i.e. five sums over the P = 121 patch samples per (ray, source view, channel).  Here one
// wavefront owns one ray: lane (s, c) streams its 121 (ref, src) pairs once and keeps the five sums in registers, the
// channel mean / two-smallest-over-sources selection happen in the wave's LDS row.  Backward recomputes the sums (the
// inputs are 58 kB per ray and L2-resident) instead of storing them.  HBM-bound: 4 B x P x C x (S + 1) per ray in,
// 4 B out (forward); the same in + as much out (backward).
#include "common.h"

#define LN_MAX_SC 64      // source views x channels per ray must fit one wavefront
#define LN_RAYS 4         // wavefronts (rays) per workgroup

struct LnccStat {
    float cross, ref_var, src_var, den, cc, ncc;
    float ref_sum, src_sum;
};

// ncc.py:29-47 for one (ray, source, channel); sums accumulated over p in index order
__device__ __forceinline__ LnccStat lncc_stat(const float* __restrict__ r, const float* __restrict__ q, int p_count, int c_count) {
    float rs = 0.f, qs = 0.f, rr = 0.f, qq = 0.f, rq = 0.f;
    for (int p = 0; p < p_count; ++p) {
        const float a = r[p * c_count], b = q[p * c_count];
        rs += a;
        qs += b;
        rr += a * a;
        qq += b * b;
        rq += a * b;
    }
    const float np = (float)p_count;
    const float u_ref = rs / np, u_src = qs / np;
    LnccStat st;
    st.ref_sum = rs;
    st.src_sum = qs;
    st.cross = rq - u_src * rs - u_ref * qs + u_ref * u_src * np;
    st.ref_var = rr - 2.0f * u_ref * rs + u_ref * u_ref * np;
    st.src_var = qq - 2.0f * u_src * qs + u_src * u_src * np;
    st.den = st.ref_var * st.src_var + 1e-5f;
    st.cc = st.cross * st.cross / st.den;
    st.ncc = fminf(fmaxf(1.0f - st.cc, 0.0f), 2.0f);
    return st;
}

__global__ __launch_bounds__(64 * LN_RAYS) void lncc_fwd_k(const float* __restrict__ ref, const float* __restrict__ src, int64_t n_rays, int s_count,
                                                            int p_count, int c_count, float* __restrict__ ncc_out, int32_t* __restrict__ sel_out) {
    __shared__ float s_ncc[LN_RAYS][LN_MAX_SC];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * LN_RAYS + wave;
    const bool active = b < n_rays;
    const int s = lane / c_count, c = lane % c_count;
    if (active && s < s_count) {
        const float* r = ref + b * p_count * c_count + c;
        const float* q = src + ((int64_t)s * n_rays + b) * p_count * c_count + c;
        s_ncc[wave][lane] = lncc_stat(r, q, p_count, c_count).ncc;
    }
    __syncthreads();
    if (!active || lane != 0) return;
    // mean over channels (ncc.py:53), then the two smallest source views (first index wins a tie) and their mean (:54-55)
    float best0 = 3.4e38f, best1 = 3.4e38f;
    int i0 = 0, i1 = 0;
    for (int v = 0; v < s_count; ++v) {
        float m = 0.0f;
        for (int k = 0; k < c_count; ++k) m += s_ncc[wave][v * c_count + k];
        m /= (float)c_count;
        if (m < best0) { best1 = best0; i1 = i0; best0 = m; i0 = v; }
        else if (m < best1) { best1 = m; i1 = v; }
    }
    if (s_count == 1) { best1 = best0; i1 = i0; }
    ncc_out[b] = (best0 + best1) / 2.0f;
    sel_out[2 * b] = i0;
    sel_out[2 * b + 1] = i1;
}

__global__ __launch_bounds__(64 * LN_RAYS) void lncc_bwd_k(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ g_ncc,
                                                            const int32_t* __restrict__ sel, int64_t n_rays, int s_count, int p_count, int c_count,
                                                            float* __restrict__ g_ref, float* __restrict__ g_src) {
    __shared__ float s_co[LN_RAYS][LN_MAX_SC][3];   // per (s, c): d/d(sum r), d/d(sum r^2), d/d(sum r*s)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * LN_RAYS + wave;
    const bool active = b < n_rays;
    const int s = lane / c_count, c = lane % c_count;
    if (active && s < s_count) {
        const float* r = ref + b * p_count * c_count + c;
        const float* q = src + ((int64_t)s * n_rays + b) * p_count * c_count + c;
        const LnccStat st = lncc_stat(r, q, p_count, c_count);
        const float np = (float)p_count;
        // d loss / d ncc_c(s, c): mean of the two selected sources (1/2 each; a source selected twice only when S = 1), mean over channels
        float g = 0.0f;
        if (s == sel[2 * b]) g += 0.5f;
        if (s == sel[2 * b + 1]) g += 0.5f;
        g *= g_ncc[b] / (float)c_count;
        const float x = 1.0f - st.cc;
        if (!(x >= 0.0f && x <= 2.0f)) g = 0.0f;                         // clamp passes the gradient on [0, 2]
        const float g_cc = -g;
        const float g_cross = g_cc * 2.0f * st.cross / st.den;
        const float g_den = -g_cc * st.cc / st.den;
        const float g_rvar = g_den * st.src_var, g_svar = g_den * st.ref_var;
        // cross = sum(rs) - sum(r) sum(s) / P ; var = sum(x^2) - sum(x)^2 / P   (ncc.py:38-45 expanded)
        const float g_rs = -g_cross * st.src_sum / np - g_rvar * 2.0f * st.ref_sum / np;
        const float g_qs = -g_cross * st.ref_sum / np - g_svar * 2.0f * st.src_sum / np;
        s_co[wave][lane][0] = g_rs;
        s_co[wave][lane][1] = g_rvar;
        s_co[wave][lane][2] = g_cross;
        float* gq = g_src + ((int64_t)s * n_rays + b) * p_count * c_count + c;
        for (int p = 0; p < p_count; ++p) gq[p * c_count] = g_qs + 2.0f * q[p * c_count] * g_svar + r[p * c_count] * g_cross;
    }
    __syncthreads();
    if (active) {     // d/d ref: the same ref patch meets every source view; the P*C elements are spread over the 64 lanes (coalesced)
        const int pc = p_count * c_count;
        const float* r = ref + b * pc;
        float* gr = g_ref + b * pc;
        for (int i = lane; i < pc; i += 64) {
            const int ch = i % c_count;
            const float a = r[i];
            float acc = 0.0f;
            for (int v = 0; v < s_count; ++v) {
                const float* co = s_co[wave][v * c_count + ch];
                acc += co[0] + 2.0f * a * co[1] + src[((int64_t)v * n_rays + b) * pc + i] * co[2];
            }
            gr[i] = acc;
        }
    }
}

static int lncc_check(const char* who, const void* ref, const void* src, int64_t n_rays, int s, int p, int c) {
    GENS_CHECK_ARG(n_rays >= 0 && s >= 1 && p >= 1 && c >= 1, GENS_EINVAL, "%s: bad shape rays=%lld S=%d P=%d C=%d", who, (long long)n_rays, s, p, c);
    GENS_CHECK_ARG(s <= LN_MAX_SC && c <= LN_MAX_SC && s * c <= LN_MAX_SC, GENS_ELIMIT, "%s: S = %d, C = %d: S*C exceeds %d (one wavefront per ray)", who, s, c, LN_MAX_SC);
    GENS_CHECK_ARG(n_rays == 0 || (ref && src), GENS_EINVAL, "%s: null input", who);
    return 0;
}

extern "C" int gens_lncc_fwd(const float* ref, const float* src, int64_t n_rays, int n_src, int n_patch, int n_ch, float* ncc, int32_t* sel,
                             void* stream) {
    if (int e = lncc_check("gens_lncc_fwd", ref, src, n_rays, n_src, n_patch, n_ch)) return e;
    GENS_CHECK_ARG(n_rays == 0 || (ncc && sel), GENS_EINVAL, "gens_lncc_fwd: null output");
    if (n_rays == 0) return 0;
    lncc_fwd_k<<<gens_blocks(n_rays, LN_RAYS), 64 * LN_RAYS, 0, (hipStream_t)stream>>>(ref, src, n_rays, n_src, n_patch, n_ch, ncc, sel);
    return gens_launch_status("gens_lncc_fwd");
}

extern "C" int gens_lncc_bwd(const float* ref, const float* src, const float* g_ncc, const int32_t* sel, int64_t n_rays, int n_src, int n_patch,
                             int n_ch, float* g_ref, float* g_src, void* stream) {
    if (int e = lncc_check("gens_lncc_bwd", ref, src, n_rays, n_src, n_patch, n_ch)) return e;
    GENS_CHECK_ARG(n_rays == 0 || (g_ncc && sel && g_ref && g_src), GENS_EINVAL, "gens_lncc_bwd: null pointer");
    if (n_rays == 0) return 0;
    lncc_bwd_k<<<gens_blocks(n_rays, LN_RAYS), 64 * LN_RAYS, 0, (hipStream_t)stream>>>(ref, src, g_ncc, sel, n_rays, n_src, n_patch, n_ch, g_ref, g_src);
    return gens_launch_status("gens_lncc_bwd");
}
