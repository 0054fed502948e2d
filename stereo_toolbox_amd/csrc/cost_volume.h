// Internal interface between the two cost-volume translation units (not part of the C-ABI).
#pragma once
// Second-generation builder (cost_volume_mfma.hip).  Returns -1 when it does not serve the configuration
// (the caller then uses the first-generation kernels of cost_volume.hip), else the C-ABI status code.
int stx_cv_fwd_mfma(const float* Lg, const float* Rg, int Cg, int G, const float* Lc, const float* Rc, int Cc,
                    const float* scale, float* vol, int B, int H, int W, int D, int mask_left, void* stream);
// Second-generation backward (cost_volume_bwd_mfma.hip), same convention.
int stx_cv_bwd_mfma(const float* gvol, const float* Lg, const float* Rg, int Cg, int G, int Cc, float* gLg, float* gRg,
                    float* gLc, float* gRc, int B, int H, int W, int D, int mask_left, void* stream);
