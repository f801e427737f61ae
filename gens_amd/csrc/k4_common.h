// Shared by K4 (lookup_feature) and K7 (fused feature look-up + blending network): source-view projection of a point.
#pragma once
#include "common.h"

struct MapSet {
    const float4* data[GENS_MAX_LEVELS];
    float* grad[GENS_MAX_LEVELS];
    int h[GENS_MAX_LEVELS], w[GENS_MAX_LEVELS];
    float cw[GENS_MAX_LEVELS], ch[GENS_MAX_LEVELS];     // (w-1)/2, (h-1)/2 and their correctly rounded reciprocals (gens_fill_maps):
    float rcw[GENS_MAX_LEVELS], rch[GENS_MAX_LEVELS];   // a / c = fma(fma(-q, c, a), r, q), q = a * r  -- bit-equal to the IEEE division
    int n;
};

struct SrcProj {
    float ix, iy;   // read position (align_corners=False un-normalisation)
    bool inside;    // mask term of this level
};
__device__ __forceinline__ float div_rn_(float a, float b, float y) {   // a / b given y = RN(1 / b): the correctly rounded quotient
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-q, b, a), y, q);
}
__device__ __forceinline__ SrcProj project_src(const float* __restrict__ w2c, const float* __restrict__ k, float s, int h,
                                               int w, float cw, float ch, float rcw, float rch, float x, float y, float z) {
    float cx = w2c[0] * x + w2c[1] * y + w2c[2] * z + w2c[3];
    float cy = w2c[4] * x + w2c[5] * y + w2c[6] * z + w2c[7];
    float cz = w2c[8] * x + w2c[9] * y + w2c[10] * z + w2c[11];
    float u = (k[0] * s) * cx + (k[1] * s) * cy + (k[2] * s) * cz;
    float v = (k[4] * s) * cx + (k[5] * s) * cy + (k[6] * s) * cz;
    float d = k[8] * cx + k[9] * cy + k[10] * cz;
    const float yd = 1.0f / d;                                   // one IEEE reciprocal serves both quotients (exact, see div_rn_)
    float px = div_rn_(u, d, yd), py = div_rn_(v, d, yd);
    float nx = div_rn_(px, cw, rcw) - 1.0f, ny = div_rn_(py, ch, rch) - 1.0f;
    SrcProj p;
    p.inside = (d > 0.0f) && (px >= 0.0f) && (px < (float)w) && (py >= 0.0f) && (py < (float)h);
    p.ix = ((nx + 1.0f) * (float)w - 1.0f) / 2.0f;
    p.iy = ((ny + 1.0f) * (float)h - 1.0f) / 2.0f;
    return p;
}

// The same projection split in two for callers that read SEVERAL levels of one (point, view): the level scale s = 2^-l multiplies rows
// 0-1 of the intrinsics, and a power-of-two factor passes through every product, sum and correctly rounded quotient above unchanged
// (u_l = s u_0 exactly, hence px_l = s px_0), so the matrix products, the reciprocal and the two quotients are done ONCE per (point,
// view) and a level costs only its own normalisation.  Bit-identical to project_src (tests/test_hip_blend.py compares the fused
// kernel, which uses this form, with K4, which uses the other).
struct SrcBase {
    float px, py, d;    // level-0 pixel coordinates and depth
};
__device__ __forceinline__ SrcBase project_src_base(const float* __restrict__ w2c, const float* __restrict__ k, float x, float y, float z) {
    float cx = w2c[0] * x + w2c[1] * y + w2c[2] * z + w2c[3];
    float cy = w2c[4] * x + w2c[5] * y + w2c[6] * z + w2c[7];
    float cz = w2c[8] * x + w2c[9] * y + w2c[10] * z + w2c[11];
    float u = k[0] * cx + k[1] * cy + k[2] * cz;
    float v = k[4] * cx + k[5] * cy + k[6] * cz;
    SrcBase b;
    b.d = k[8] * cx + k[9] * cy + k[10] * cz;
    const float yd = 1.0f / b.d;
    b.px = div_rn_(u, b.d, yd);
    b.py = div_rn_(v, b.d, yd);
    return b;
}
__device__ __forceinline__ SrcProj project_src_level(const SrcBase& b, float s, int h, int w, float cw, float ch, float rcw, float rch) {
    const float px = b.px * s, py = b.py * s;
    float nx = div_rn_(px, cw, rcw) - 1.0f, ny = div_rn_(py, ch, rch) - 1.0f;
    SrcProj p;
    p.inside = (b.d > 0.0f) && (px >= 0.0f) && (px < (float)w) && (py >= 0.0f) && (py < (float)h);
    p.ix = ((nx + 1.0f) * (float)w - 1.0f) / 2.0f;
    p.iy = ((ny + 1.0f) * (float)h - 1.0f) / 2.0f;
    return p;
}
