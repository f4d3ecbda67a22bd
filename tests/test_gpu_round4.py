"""Round 4, GPU side: the RCCL branch of bench.py at world size 1, raster ids at the bench configuration's real size, fidelity items."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "1", "--res", "640x360", "--texture-size", "64",
           "--no-cpu-baseline", "--checksums", "--spp-per-gather", "4"] + list(extra)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_force_collectives_runs_the_rccl_branch_bit_identically():
    """`bench.py --gpus 1 --force-collectives`: the transport branch of the N > 1 path (_Collectives(staged=False)) on the one GPU there is --
    init_process_group("nccl", device_id), all_gather_into_tensor of the R11G11B10 history shard and incr_unshard after every 4th sample and
    after the last, the TAA band gather on the library-owned surface through __cuda_array_interface__, all_reduce of the clock and the ray
    counts, destroy_process_group.  The gathered history and the TAA accumulation surface equal the plain run's bit for bit: RCCL's stream
    is ordered against the caller's stream and the library's pipeline streams (fork / join inside incr_pathtrace) without a host sync."""
    plain = _bench()
    coll = _bench("--force-collectives")
    assert "force_collectives" in coll["config"] and "RCCL" in coll["config"]["sharding"] and "row bands" in coll["raster"]["sharding"]
    assert plain["config"]["sharding"] == "none"
    assert coll["config"]["history_crc32"] == plain["config"]["history_crc32"]
    assert coll["raster"]["taa_accumulation_crc32"] == plain["raster"]["taa_accumulation_crc32"]
    assert coll["config"]["rays_per_step"] == plain["config"]["rays_per_step"]
    assert coll["n_gpus"] == 1 and coll["value"] > 0


# ------------------------------------------------------------------------------------------------ raster ids at config 3's real size
def test_raster_4k_bench_configuration_bands_vs_oracle(gpu_ctx, rt, R, sio, synth, ob, A):
    """BASELINE config 3 at its real size -- full Sponza-stat (262,267 triangles, 1024^2 textures, the x16 anisotropic sampler, 65 lights,
    the sky's SH9 / prefiltered cube / env-BRDF LUT from the IBL precompute) at 3840x2160 -- against oracle_raster.h: the GPU renders the whole
    frame pass by pass, the oracle the same passes on three scissor bands of 16 rows (upper arcade, nave, floor; every raster pass of the
    oracle honours incr_view_constants::scissor_y0/y1).  Visibility ids (the north star's one exact requirement: visibility-pass.hlsl:96-114,
    encoding.hlsli:4-28), depth and the three G-buffers must be bit-identical on the bands, the lit colour within colour_close, the draw lists
    and the light grid identical."""
    from conftest import mod
    from test_gpu_raster import colour_close
    W, H = 3840, 2160
    scene = synth.sponza_stat(texture_size=1024)
    R.generate_meshlets(scene)
    assert sum(p["index_count"] // 3 for prims in scene.mesh_primitives for p in prims) == 262267 and len(scene.global_lights) == 65
    view = sio.FView().reset(scene, aspect=W / H)
    gscene = gpu_ctx.create_scene(scene)
    osc = ob.OracleScene(scene)
    ibl = mod("ibl")
    LW = LH = 512
    try:
        sky = ibl.SkyIbl(gpu_ctx, gscene)
        lut = ibl.generate_env_brdf(gpu_ctx, LW, LH)
        rr = R.RasterRenderer(gpu_ctx, gscene, W, H, A.FConfig)
        rr.ibl = R.IblInputs.from_sky(sky, lut, (LW, LH))
        vc = rr.render(view); gpu_ctx.synchronize()
        g = {n: rr.surf.download(n) for n in ("vis", "depth", "gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra", "color", "counts", "default_list", "double_sided_list",
                                              "light_grid", "light_lists", "light_count")}
        prod = sky.download()
        lut_host = gpu_ctx.download(lut, (LH, LW), np.uint32)
        env_host, sh_host = np.ascontiguousarray(prod["envmap"]), np.ascontiguousarray(prod["sh9"])
        ibl_host = R.IblInputs(sh_host.ctypes.data, env_host.ctypes.data, sky.info.cubemap_res, sky.info.num_mips, lut_host.ctypes.data, (LW, LH))
        hs = ob.HostSurfaces(W, H, scene.num_meshlets, 16 * 9 * 24, 64)
        assert (g["vis"] != 0xFFFFF000).mean() > 0.95
        for y0, y1 in ((320, 336), (1080, 1096), (1840, 1856)):
            for name, d in R.build_descs(None, scene, R.with_scissor(vc, y0, y1), hs.ptr, 0, A.FConfig, ibl=ibl_host):
                if name != "taa_resolve":
                    ob.run_pass(osc, name, d)
            o = hs.arr
            band = slice(y0, y1)
            np.testing.assert_array_equal(g["counts"], o["counts"])
            for name, n in (("default_list", o["counts"][0]), ("double_sided_list", o["counts"][1])):
                np.testing.assert_array_equal(g[name][:n], o[name][:n])
            np.testing.assert_array_equal(g["light_grid"], o["light_grid"])
            np.testing.assert_array_equal(g["vis"][band], o["vis"][band], err_msg="visibility ids rows %d-%d" % (y0, y1))
            np.testing.assert_array_equal(g["depth"][band].view(np.uint32), o["depth"][band].view(np.uint32), err_msg="depth rows %d-%d" % (y0, y1))
            for name in ("gbuffer_basecolor", "gbuffer_normals", "gbuffer_mra"):
                np.testing.assert_array_equal(g[name][band], o[name][band], err_msg="%s rows %d-%d" % (name, y0, y1))
            frac = colour_close(ob, g["color"][band], o["color"][band], "lit colour rows %d-%d" % (y0, y1))
            assert frac > 0.995, frac
            ids = np.unique(g["vis"][band] >> 7)
            print("rows %d-%d: %d distinct meshlets, %.1f %% covered, colour bit-identical on %.4f of the pixels" % (y0, y1, len(ids), 100 * (g["vis"][band] != 0xFFFFF000).mean(), frac))
            assert len(ids) > 20
        rr.close(); sky.close(); gpu_ctx.free(lut)
    finally:
        osc.close(); gscene.close()
