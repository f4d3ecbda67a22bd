"""Path-trace parity with every pixel accounted for (round 4: shared by all HIP-vs-oracle colour comparisons).

The colour contract is 2^-7 relative per channel on the RGBA16F target.  A pixel outside it must be EXPLAINED by the
hit-chain instrumentation (incr_pathtrace_desc::debug_hit_chain on both sides): either its path took another triangle / one of its shadow
queries flipped (device vs glibc ulps in sin / cos / sqrt move a ray across an edge), or it is a path of at least three vertices over the
same triangles (ulp differences compounded through BRDF-sampled bounces).  A path of one or two vertices over the oracle's triangles
outside the tolerance is a shading error and fails.  The share of outliers is capped per test at what was observed plus a margin."""
import os

import numpy as np

NOT_TRACED = 0xfffffffe


def outliers(g_target, o_target):
    gt = g_target.view(np.float16).astype(np.float32)
    ot = o_target.view(np.float16).astype(np.float32)
    same = (g_target == o_target).all(axis=-1)
    with np.errstate(invalid="ignore"):
        rel = np.abs(gt - ot) / np.maximum(np.abs(ot), 1e-3)
    return ~(same | (rel <= 2.0 ** -7).all(axis=-1))


def classify(g_target, o_target, g_chain, o_chain):
    """outliers of the colour tolerance, pixels whose path or shadow-query outcomes differ, and the number of vertices of every path"""
    bad = outliers(g_target, o_target)
    differs = (g_chain != o_chain).any(axis=(-1, -2))
    vertices = (g_chain[..., 0] != NOT_TRACED).sum(axis=-1)
    return bad, differs, vertices


def stats_tuple(s):
    return (s.camera_rays, s.bounce_rays, s.sky_shadow_rays, s.light_shadow_rays, s.shaded_hits)


def accounted_parity(rt, pt, osc, view, W, H, sample, cap, label, shard=None, cfg=None, gpu_envmap=None, oracle_envmap=None, min_outliers_allowed=0):
    """Render sample `sample` on the GPU path tracer `pt` and on the oracle scene `osc` with hit chains; assert the accounted contract.
    cap: largest share of pixels that may lie outside 2^-7 (all of them explained).  Returns (gpu result, oracle result, bad mask)."""
    import ctypes as C
    from conftest import mod
    A = mod("ctypes_api")
    kw = {} if cfg is None else {"cfg": cfg}
    pt.ctx.clear(pt.stats, C.sizeof(A.RayStats))            # the counters accumulate over calls
    pt.render(view, sample, hit_chain=True, envmap=gpu_envmap, **kw)
    g = pt.read(); gc = pt.read_chain()
    d = rt.make_pathtrace_desc(view, W, H, sample, shard, envmap=oracle_envmap, **kw)
    o = osc.pathtrace(d, num_threads=os.cpu_count() or 1, want_chain=True)
    bad, differs, vertices = classify(g["target"], o["target"], gc, o["chain"])
    assert g["stats"].camera_rays == o["stats"].camera_rays
    # ray counts: equal, or off by what the pixels with another path can account for (<= 3 bounce rays + 4 x 66 shadow rays each)
    for a, b in zip(stats_tuple(g["stats"]), stats_tuple(o["stats"])):
        assert abs(a - b) <= 70 * int(differs.sum()), (label, stats_tuple(g["stats"]), stats_tuple(o["stats"]), int(differs.sum()))
    n = bad.size
    print("%s %dx%d sample %d: %d of %d pixels outside 2^-7 (%.5f %%): %d with another chain, %d same chain with >= 3 vertices; %d pixels with another chain in all"
          % (label, W, H, sample, int(bad.sum()), n, 100 * bad.mean(), int((bad & differs).sum()), int((bad & ~differs & (vertices >= 3)).sum()), int(differs.sum())))
    # A third explanation, for paths of any length (round 5: first seen on a two-vertex path of the helmet under an HDRI probe): the outlying channel is a DIM one -- its error is below
    # 2^-7 of the pixel's brightest channel.  F_Schlick's (1 - VoH)^5 of a near-mirror half vector is a difference of nearly equal numbers: on a saturated metal (one channel of F0 near 1,
    # the others near 0) the dim channels are that term alone and carry the ulps of VoH at a few per cent, 250 x below the channel that makes the pixel.  Counted, printed, capped at 5e-5.
    gt = g["target"].view(np.float16).astype(np.float32)[..., :3]; ot = o["target"].view(np.float16).astype(np.float32)[..., :3]
    with np.errstate(invalid="ignore"):
        dim_only = (np.abs(gt - ot) <= 2.0 ** -7 * np.maximum(np.abs(ot).max(axis=-1, keepdims=True), 1e-3)).all(axis=-1)
    dim = bad & ~differs & (vertices <= 2) & dim_only
    if dim.any():
        print("  %d short path(s) off only in a channel far below the pixel's brightest: %s" % (int(dim.sum()), [(int(x), int(y), gt[y, x].tolist(), ot[y, x].tolist()) for y, x in list(zip(*np.nonzero(dim)))[:4]]))
    assert int(dim.sum()) <= max(int(np.floor(5e-5 * bad.size + 0.5)), 1), (label, int(dim.sum()))
    unexplained = bad & ~differs & (vertices <= 2) & ~dim_only
    for y, x in list(zip(*np.nonzero(unexplained)))[:8]:          # what a failure looks like, for the log
        print("  unexplained pixel (%d, %d): gpu %s oracle %s vertices %d chain %s" % (x, y, g["target"][y, x].view(np.float16), o["target"][y, x].view(np.float16), int(vertices[y, x]), gc[y, x].tolist()))
    assert not unexplained.any(), "%s: %d short paths over the oracle's triangles are outside the colour tolerance" % (label, int(unexplained.sum()))
    allowed = max(int(np.floor(cap * n + 0.5)), min_outliers_allowed)
    assert int(bad.sum()) <= allowed, (label, int(bad.sum()), allowed)
    assert int(differs.sum()) <= max(4 * allowed, 2), (label, int(differs.sum()))
    return g, o, bad
