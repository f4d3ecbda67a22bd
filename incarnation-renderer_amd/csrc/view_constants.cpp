// view_constants.cpp -- host-side per-frame constants of the raster path (no GPU): the FViewConstants upload of
// Renderer::Render (reference: source/demo-dll/src/renderer.cpp:1630-1665) plus the matrices that individual passes
// derive on their own (render-jobs/light-culling.inl:111-112, render-jobs/sky-lighting.inl:124, renderer.cpp:2007-2025).
#include "host_math.h"
#include "../../include/incr.h"

using namespace incr;

namespace {

// renderer.cpp:53-66
float halton(uint64_t sampleIndex, uint32_t base) {
    float result = 0.f, f = 1.f;
    while (sampleIndex > 0) {
        f = f / (float)base;
        result += f * (float)(sampleIndex % base);
        sampleIndex = sampleIndex / base;
    }
    return result;
}

// SimpleMath Matrix::CreateTranslation(x, y, 0)
M44 translation(float x, float y) { M44 t = m44_identity(); t.m[12] = x; t.m[13] = y; return t; }

void store(incr_matrix* dst, const M44& m) { memcpy(dst->m, m.m, 64); }

} // namespace

extern "C" {

void incr_pixel_jitter(uint32_t frame_index, uint32_t res_x, uint32_t res_y, float out_jitter[2]) {
    const uint32_t i = frame_index % 16u;
    const float hx = halton(i, 2), hy = halton(i, 3);
    out_jitter[0] = 2.f * (hx - 0.5f) / (float)res_x;
    out_jitter[1] = 2.f * (hy - 0.5f) / (float)res_y;
}

int incr_make_view_constants(const incr_view* view, const incr_view* culling_view, const float jitter[2],
                             const incr_matrix* prev_view_proj, float near_plane, uint32_t res_x, uint32_t res_y,
                             incr_view_constants* out) {
    if (!view || !out) return INCR_ERR_INVALID_ARGUMENT;
    const incr_view* cull = culling_view ? culling_view : view;
    const float jx = jitter ? jitter[0] : 0.f, jy = jitter ? jitter[1] : 0.f;
    const M44 V = m44_from(view->view_transform), P = m44_from(view->projection_transform);
    const M44 Vc = m44_from(cull->view_transform), Pc = m44_from(cull->projection_transform);
    const M44 T = translation(jx, jy);
    const M44 PT = m44_mul(P, T);
    const M44 VPT = m44_mul(V, PT);
    const M44 VP = m44_mul(V, P);
    memset(out, 0, sizeof *out);
    store(&out->view_transform, V);
    store(&out->proj_transform, PT);
    store(&out->view_proj_transform, VPT);
    store(&out->inv_view_proj_transform, m44_inverse(VPT));
    store(&out->inv_proj_transform, m44_inverse(PT));
    store(&out->cull_view_proj_transform, m44_mul(Vc, PT));
    store(&out->prev_view_proj_transform, prev_view_proj ? m44_from(*prev_view_proj) : m44_identity());
    store(&out->lightcull_proj_transform, m44_mul(Pc, T));
    store(&out->lightcull_inv_view_proj, m44_inverse(m44_mul(m44_mul(Vc, Pc), T)));
    store(&out->sky_inv_view_proj, m44_inverse(m44_mul(VP, T)));
    store(&out->unjittered_view_proj, VP);
    store(&out->taa_inv_view_proj, m44_inverse(VP));
    M44 parallaxView = V;                                      // environmentmap.inl:127-128: Translation(Vector3::Zero)
    parallaxView.m[12] = parallaxView.m[13] = parallaxView.m[14] = 0.f;
    store(&out->envmap_inv_parallax_view_proj, m44_inverse(m44_mul(parallaxView, P)));
    for (int i = 0; i < 3; ++i) { out->eye_pos[i] = view->position[i]; out->camera_up[i] = view->up[i]; }
    out->near_plane = near_plane;
    out->res_x = res_x; out->res_y = res_y;
    out->jitter[0] = jx; out->jitter[1] = jy;
    return INCR_OK;
}

} // extern "C"
