// Shared by K4 (lookup_feature) and K7 (fused feature look-up + blending network): source-view projection of a point.
#pragma once
#include "common.h"

struct MapSet {
    const float4* data[GENS_MAX_LEVELS];
    float* grad[GENS_MAX_LEVELS];
    int h[GENS_MAX_LEVELS], w[GENS_MAX_LEVELS];
    int n;
};

struct SrcProj {
    float ix, iy;   // read position (align_corners=False un-normalisation)
    bool inside;    // mask term of this level
};
__device__ __forceinline__ SrcProj project_src(const float* __restrict__ w2c, const float* __restrict__ k, float s, int h,
                                               int w, float x, float y, float z) {
    float cx = w2c[0] * x + w2c[1] * y + w2c[2] * z + w2c[3];
    float cy = w2c[4] * x + w2c[5] * y + w2c[6] * z + w2c[7];
    float cz = w2c[8] * x + w2c[9] * y + w2c[10] * z + w2c[11];
    float u = (k[0] * s) * cx + (k[1] * s) * cy + (k[2] * s) * cz;
    float v = (k[4] * s) * cx + (k[5] * s) * cy + (k[6] * s) * cz;
    float d = k[8] * cx + k[9] * cy + k[10] * cz;
    float px = u / d, py = v / d;
    float nx = px / ((float)(w - 1) / 2.0f) - 1.0f, ny = py / ((float)(h - 1) / 2.0f) - 1.0f;
    SrcProj p;
    p.inside = (d > 0.0f) && (px >= 0.0f) && (px < (float)w) && (py >= 0.0f) && (py < (float)h);
    p.ix = ((nx + 1.0f) * (float)w - 1.0f) / 2.0f;
    p.iy = ((ny + 1.0f) * (float)h - 1.0f) / 2.0f;
    return p;
}

