// host_san.cpp -- AddressSanitizer / UBSan harness for the host-only C++ of libincr_hip.so (scene flattening, BVH
// build, light grid, meshletizer, view constants).  GPU sanitizers are not available on the MI355X pool, so the CPU
// build is where memory errors in the host code get caught (tests/test_host_sanitizers.py drives this through ctypes
// under LD_PRELOAD=libasan).  Built by tools/sanitize/Makefile with g++ -fsanitize=address,undefined.
#include <cstdio>
#include <cstring>

#include "../../incarnation-renderer_amd/csrc/scene_build.h"

using namespace incr;

extern "C" int san_build_scene(const incr_scene_desc* desc, uint64_t out_counts[4], char* why, uint32_t why_bytes) {
    HostScene hs; HostBvh bvh; LightGrid grid;
    copy_scene_desc(*desc, hs);
    const char* msg = "";
    if (!validate_scene(hs, &msg)) { snprintf(why, why_bytes, "%s", msg); return -1; }
    build_world_bvh(hs, bvh);
    if (const char* err = check_bvh(bvh)) { snprintf(why, why_bytes, "%s", err); return -2; }
    build_light_grid(hs, bvh, grid);
    // hittable triangles: hidden meshes stay in the tree as NaN records (scene_build.cpp)
    uint64_t hittable = 0; for (const TriIsect& t : bvh.tris) if (t.v0x == t.v0x) ++hittable;
    std::vector<MeshInstanceInfo> info; refresh_mesh_instances(hs, bvh, info);      // the update path's host half, under the sanitizers too
    out_counts[0] = hittable; out_counts[1] = bvh.nodes8.size(); out_counts[2] = grid.cellOffsets.size(); out_counts[3] = grid.cellLights.size();
    return 0;
}
