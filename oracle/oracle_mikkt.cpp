// oracle_mikkt.cpp -- TEST INFRASTRUCTURE (CPU oracle): tangent generation for primitives without a TANGENT accessor.
//
// The reference calls MikkTSpace for them: MeshUtils::FixupMeshes (source/demo-dll/src/mesh-utils.cpp:560-638) runs
// genTangSpaceDefault per primitive whose material has a normal texture and whose attributes lack TANGENT, with callbacks that
// read POSITION / NORMAL / TEXCOORD_0 through the index buffer (:28-103) and write float4(tangent, sign) to the INDEXED vertex
// (:105-112) -- so a vertex shared by several faces keeps what the LAST face (in face order) wrote.
//
// The library itself is an empty submodule of the reference tree (ext/MikkTSpace; .gitmodules names github.com/mmikk/MikkTSpace,
// pinned commit unknown), so this file restates its PUBLISHED algorithm -- mikktspace.c, Morten S. Mikkelsen, zlib licence;
// "Simulation of Wrinkled Surfaces Revisited", 2008 -- step by step for the case the reference exercises: triangles only,
// genTangSpaceDefault (angular threshold 180 degrees), m_setTSpaceBasic.  Function names below are those of mikktspace.c.
// Parity is unpinned (no fixture of the reference exists); tests/test_mikktspace.py holds it against the tangents the Damaged
// Helmet asset ships (exported with MikkTSpace) and against the product's own implementation (csrc/mikktspace_gen.cpp).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct SVec3 { float x, y, z; };
static bool veq(SVec3 a, SVec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static SVec3 vadd(SVec3 a, SVec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static SVec3 vsub(SVec3 a, SVec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static SVec3 vscale(float s, SVec3 v) { return {s * v.x, s * v.y, s * v.z}; }
static float LengthSquared(SVec3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
static float Length(SVec3 v) { return sqrtf(LengthSquared(v)); }
static SVec3 Normalize(SVec3 v) { return vscale(1 / Length(v), v); }
static float vdot(SVec3 a, SVec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static bool NotZero(float f) { return fabsf(f) > FLT_MIN; }
static bool VNotZero(SVec3 v) { return NotZero(v.x) || NotZero(v.y) || NotZero(v.z); }

enum { MARK_DEGENERATE = 1, QUAD_ONE_DEGEN_TRI = 2, GROUP_WITH_ANY = 4, ORIENT_PRESERVING = 8 };

struct SGroup { int iNrFaces; int* pFaceIndices; int iVertexRepresentitive; bool bOrientPreservering; };
struct STriInfo {
    int FaceNeighbors[3]; SGroup* AssignedGroup[3];
    SVec3 vOs, vOt; float fMagS, fMagT;
    int iOrgFaceNumber, iFlag, iTSpacesOffs; unsigned char vert_num[4];
};
struct STSpace { SVec3 vOs; float fMagS; SVec3 vOt; float fMagT; int iCounter; bool bOrient; };

// the SMikkTSpaceContext of mesh-utils.cpp:28-103: attribute reads go through the index buffer
struct Context {
    const float* pos; const float* nrm; const float* uv; const uint32_t* idx; uint32_t nVerts, nTris;
    uint32_t vertex_of(int index) const { return idx[(size_t)(index >> 2) * 3 + (index & 3)]; }      // IndexToData: face = index >> 2, vert = index & 3
    SVec3 GetPosition(int index) const { const float* p = pos + (size_t)vertex_of(index) * 3; return {p[0], p[1], p[2]}; }
    SVec3 GetNormal(int index) const { const float* p = nrm + (size_t)vertex_of(index) * 3; return {p[0], p[1], p[2]}; }
    SVec3 GetTexCoord(int index) const { const float* p = uv + (size_t)vertex_of(index) * 2; return {p[0], p[1], 1.0f}; }
};
static int MakeIndex(int iFace, int iVert) { return (iFace << 2) | (iVert & 0x3); }

// ---- GenerateSharedVerticesIndexList: corners with identical position, normal and texture coordinate share one index ----
struct STmpVert { float vert[3]; int index; };
static const int g_iCells = 2048;
static int FindGridCell(float fMin, float fMax, float fVal) {
    const float fIndex = g_iCells * ((fVal - fMin) / (fMax - fMin));
    const int iIndex = (int)fIndex;
    return iIndex < g_iCells ? (iIndex >= 0 ? iIndex : 0) : (g_iCells - 1);
}
static void MergeVertsFast(int piTriList_in_and_out[], STmpVert pTmpVert[], const Context& C, int iL_in, int iR_in) {
    float fvMin[3], fvMax[3];
    for (int c = 0; c < 3; c++) { fvMin[c] = pTmpVert[iL_in].vert[c]; fvMax[c] = fvMin[c]; }
    for (int l = iL_in + 1; l <= iR_in; l++)
        for (int c = 0; c < 3; c++) { if (fvMin[c] > pTmpVert[l].vert[c]) fvMin[c] = pTmpVert[l].vert[c]; if (fvMax[c] < pTmpVert[l].vert[c]) fvMax[c] = pTmpVert[l].vert[c]; }
    const float dx = fvMax[0] - fvMin[0], dy = fvMax[1] - fvMin[1], dz = fvMax[2] - fvMin[2];
    int channel = 0;
    if (dy > dx && dy > dz) channel = 1; else if (dz > dx) channel = 2;
    const float fSep = 0.5f * (fvMax[channel] + fvMin[channel]);
    if (fSep >= fvMax[channel] || fSep <= fvMin[channel]) {       // the range cannot be split further: weld by complete comparison
        for (int l = iL_in; l <= iR_in; l++) {
            const int i = pTmpVert[l].index, index = piTriList_in_and_out[i];
            const SVec3 vP = C.GetPosition(index), vN = C.GetNormal(index), vT = C.GetTexCoord(index);
            bool bNotFound = true; int l2 = iL_in, i2rec = -1;
            while (l2 < l && bNotFound) {
                const int i2 = pTmpVert[l2].index, index2 = piTriList_in_and_out[i2];
                const SVec3 vP2 = C.GetPosition(index2), vN2 = C.GetNormal(index2), vT2 = C.GetTexCoord(index2);
                i2rec = i2;
                if (vP.x == vP2.x && vP.y == vP2.y && vP.z == vP2.z && vN.x == vN2.x && vN.y == vN2.y && vN.z == vN2.z && vT.x == vT2.x && vT.y == vT2.y && vT.z == vT2.z) bNotFound = false;
                else ++l2;
            }
            if (!bNotFound) piTriList_in_and_out[i] = piTriList_in_and_out[i2rec];
        }
    } else {
        int iL = iL_in, iR = iR_in;
        while (iL < iR) {
            bool bReadyLeftSwap = false, bReadyRightSwap = false;
            while (!bReadyLeftSwap && iL < iR) { bReadyLeftSwap = !(pTmpVert[iL].vert[channel] < fSep); if (!bReadyLeftSwap) ++iL; }
            while (!bReadyRightSwap && iL < iR) { bReadyRightSwap = pTmpVert[iR].vert[channel] < fSep; if (!bReadyRightSwap) --iR; }
            if (bReadyLeftSwap && bReadyRightSwap) { const STmpVert t = pTmpVert[iL]; pTmpVert[iL] = pTmpVert[iR]; pTmpVert[iR] = t; ++iL; --iR; }
        }
        if (iL == iR) { const bool bReadyRightSwap = pTmpVert[iR].vert[channel] < fSep; if (bReadyRightSwap) ++iL; else --iR; }
        if (iL_in < iR) MergeVertsFast(piTriList_in_and_out, pTmpVert, C, iL_in, iR);
        if (iL < iR_in) MergeVertsFast(piTriList_in_and_out, pTmpVert, C, iL, iR_in);
    }
}
static void GenerateSharedVerticesIndexList(int piTriList_in_and_out[], const Context& C, int iNrTrianglesIn) {
    if (iNrTrianglesIn <= 0) return;
    SVec3 vMin = C.GetPosition(0), vMax = vMin;
    for (int i = 1; i < iNrTrianglesIn * 3; i++) {
        const SVec3 vP = C.GetPosition(piTriList_in_and_out[i]);
        if (vMin.x > vP.x) vMin.x = vP.x; else if (vMax.x < vP.x) vMax.x = vP.x;
        if (vMin.y > vP.y) vMin.y = vP.y; else if (vMax.y < vP.y) vMax.y = vP.y;
        if (vMin.z > vP.z) vMin.z = vP.z; else if (vMax.z < vP.z) vMax.z = vP.z;
    }
    const SVec3 vDim = vsub(vMax, vMin);
    int iChannel = 0; float fMin = vMin.x, fMax = vMax.x;
    if (vDim.y > vDim.x && vDim.y > vDim.z) { iChannel = 1; fMin = vMin.y; fMax = vMax.y; }
    else if (vDim.z > vDim.x) { iChannel = 2; fMin = vMin.z; fMax = vMax.z; }
    std::vector<int> piHashTable((size_t)iNrTrianglesIn * 3), piHashCount(g_iCells, 0), piHashOffsets(g_iCells, 0), piHashCount2(g_iCells, 0);
    auto channel_of = [&](SVec3 v) { return iChannel == 0 ? v.x : (iChannel == 1 ? v.y : v.z); };
    for (int i = 0; i < iNrTrianglesIn * 3; i++) ++piHashCount[FindGridCell(fMin, fMax, channel_of(C.GetPosition(piTriList_in_and_out[i])))];
    for (int k = 1; k < g_iCells; k++) piHashOffsets[k] = piHashOffsets[k - 1] + piHashCount[k - 1];
    for (int i = 0; i < iNrTrianglesIn * 3; i++) {
        const int iCell = FindGridCell(fMin, fMax, channel_of(C.GetPosition(piTriList_in_and_out[i])));
        piHashTable[(size_t)piHashOffsets[iCell] + piHashCount2[iCell]] = i;
        ++piHashCount2[iCell];
    }
    int iMaxCount = piHashCount[0];
    for (int k = 1; k < g_iCells; k++) if (iMaxCount < piHashCount[k]) iMaxCount = piHashCount[k];
    std::vector<STmpVert> pTmpVert((size_t)(iMaxCount > 0 ? iMaxCount : 1));
    for (int k = 0; k < g_iCells; k++) {
        const int iEntries = piHashCount[k];
        if (iEntries < 2) continue;
        const int* pTable = &piHashTable[(size_t)piHashOffsets[k]];
        for (int e = 0; e < iEntries; e++) {
            const int i = pTable[e];
            const SVec3 vP = C.GetPosition(piTriList_in_and_out[i]);
            pTmpVert[(size_t)e].vert[0] = vP.x; pTmpVert[(size_t)e].vert[1] = vP.y; pTmpVert[(size_t)e].vert[2] = vP.z; pTmpVert[(size_t)e].index = i;
        }
        MergeVertsFast(piTriList_in_and_out, pTmpVert.data(), C, 0, iEntries - 1);
    }
}

// ---- DegenPrologue (triangles only): degenerate triangles go behind the good ones ----
static void DegenPrologue(STriInfo pTriInfos[], int piTriList_out[], int iNrTrianglesIn, int iTotTris) {
    int iNextGoodTriangleSearchIndex = 1, t = 0;
    bool bStillFindingGoodOnes = true;
    while (t < iNrTrianglesIn && bStillFindingGoodOnes) {
        const bool bIsGood = (pTriInfos[t].iFlag & MARK_DEGENERATE) == 0;
        if (bIsGood) { if (iNextGoodTriangleSearchIndex < (t + 2)) iNextGoodTriangleSearchIndex = t + 2; }
        else {
            bool bJustADegenerate = true;
            while (bJustADegenerate && iNextGoodTriangleSearchIndex < iTotTris) {
                const bool bIsGood2 = (pTriInfos[iNextGoodTriangleSearchIndex].iFlag & MARK_DEGENERATE) == 0;
                if (bIsGood2) bJustADegenerate = false; else ++iNextGoodTriangleSearchIndex;
            }
            const int t0 = t, t1 = iNextGoodTriangleSearchIndex;
            ++iNextGoodTriangleSearchIndex;
            if (!bJustADegenerate) {
                for (int i = 0; i < 3; i++) { const int index = piTriList_out[t0 * 3 + i]; piTriList_out[t0 * 3 + i] = piTriList_out[t1 * 3 + i]; piTriList_out[t1 * 3 + i] = index; }
                const STriInfo tri_info = pTriInfos[t0]; pTriInfos[t0] = pTriInfos[t1]; pTriInfos[t1] = tri_info;
            } else bStillFindingGoodOnes = false;
        }
        if (bStillFindingGoodOnes) ++t;
    }
}

// ---- InitTriInfo: per-triangle first-order derivatives + neighbours ----
struct SEdge { int i0, i1, f; };
static void GetEdge(int* i0_out, int* i1_out, int* edgenum_out, const int indices[], int i0_in, int i1_in) {
    *edgenum_out = -1;
    if (indices[0] == i0_in || indices[0] == i1_in) {
        if (indices[1] == i0_in || indices[1] == i1_in) { edgenum_out[0] = 0; i0_out[0] = indices[0]; i1_out[0] = indices[1]; }      // first edge
        else { edgenum_out[0] = 2; i0_out[0] = indices[2]; i1_out[0] = indices[0]; }                                                    // third edge
    } else { edgenum_out[0] = 1; i0_out[0] = indices[1]; i1_out[0] = indices[2]; }                                                       // second edge
}
static void BuildNeighborsFast(STriInfo pTriInfos[], std::vector<SEdge>& pEdges, const int piTriListIn[], int iNrTrianglesIn) {
    for (int f = 0; f < iNrTrianglesIn; f++)
        for (int i = 0; i < 3; i++) {
            const int i0 = piTriListIn[f * 3 + i], i1 = piTriListIn[f * 3 + (i < 2 ? i + 1 : 0)];
            pEdges[(size_t)f * 3 + i] = {i0 < i1 ? i0 : i1, !(i0 < i1) ? i0 : i1, f};
        }
    // mikktspace.c sorts by i0, then ranges of equal i0 by i1, then ranges of equal (i0, i1) by f: the lexicographic order
    std::sort(pEdges.begin(), pEdges.end(), [](const SEdge& a, const SEdge& b) { return a.i0 != b.i0 ? a.i0 < b.i0 : (a.i1 != b.i1 ? a.i1 < b.i1 : a.f < b.f); });
    const int iEntries = iNrTrianglesIn * 3;
    for (int i = 0; i < iEntries; i++) {
        const int i0 = pEdges[(size_t)i].i0, i1 = pEdges[(size_t)i].i1, f = pEdges[(size_t)i].f;
        int i0_A, i1_A, edgenum_A, edgenum_B = 0;
        GetEdge(&i0_A, &i1_A, &edgenum_A, &piTriListIn[f * 3], i0, i1);
        const bool bUnassigned_A = pTriInfos[f].FaceNeighbors[edgenum_A] == -1;
        if (bUnassigned_A) {
            int j = i + 1; bool bNotFound = true;
            while (j < iEntries && i0 == pEdges[(size_t)j].i0 && i1 == pEdges[(size_t)j].i1 && bNotFound) {
                const int t = pEdges[(size_t)j].f;
                int i0_B, i1_B;
                GetEdge(&i1_B, &i0_B, &edgenum_B, &piTriListIn[t * 3], pEdges[(size_t)j].i0, pEdges[(size_t)j].i1);      // flip i0_B and i1_B
                const bool bUnassigned_B = pTriInfos[t].FaceNeighbors[edgenum_B] == -1;
                if (i0_A == i0_B && i1_A == i1_B && bUnassigned_B) bNotFound = false; else ++j;
            }
            if (!bNotFound) { const int t = pEdges[(size_t)j].f; pTriInfos[f].FaceNeighbors[edgenum_A] = t; pTriInfos[t].FaceNeighbors[edgenum_B] = f; }
        }
    }
}
static void InitTriInfo(STriInfo pTriInfos[], const int piTriListIn[], const Context& C, int iNrTrianglesIn) {
    for (int f = 0; f < iNrTrianglesIn; f++)
        for (int i = 0; i < 3; i++) {
            pTriInfos[f].FaceNeighbors[i] = -1; pTriInfos[f].AssignedGroup[i] = nullptr;
            pTriInfos[f].vOs = {0, 0, 0}; pTriInfos[f].vOt = {0, 0, 0}; pTriInfos[f].fMagS = 0; pTriInfos[f].fMagT = 0;
            pTriInfos[f].iFlag |= GROUP_WITH_ANY;      // assumed bad
        }
    for (int f = 0; f < iNrTrianglesIn; f++) {
        const SVec3 v1 = C.GetPosition(piTriListIn[f * 3 + 0]), v2 = C.GetPosition(piTriListIn[f * 3 + 1]), v3 = C.GetPosition(piTriListIn[f * 3 + 2]);
        const SVec3 t1 = C.GetTexCoord(piTriListIn[f * 3 + 0]), t2 = C.GetTexCoord(piTriListIn[f * 3 + 1]), t3 = C.GetTexCoord(piTriListIn[f * 3 + 2]);
        const float t21x = t2.x - t1.x, t21y = t2.y - t1.y, t31x = t3.x - t1.x, t31y = t3.y - t1.y;
        const SVec3 d1 = vsub(v2, v1), d2 = vsub(v3, v1);
        const float fSignedAreaSTx2 = t21x * t31y - t21y * t31x;
        SVec3 vOs = vsub(vscale(t31y, d1), vscale(t21y, d2));      // eq 18
        SVec3 vOt = vadd(vscale(-t31x, d1), vscale(t21x, d2));     // eq 19
        pTriInfos[f].iFlag |= (fSignedAreaSTx2 > 0 ? ORIENT_PRESERVING : 0);
        if (NotZero(fSignedAreaSTx2)) {
            const float fAbsArea = fabsf(fSignedAreaSTx2), fLenOs = Length(vOs), fLenOt = Length(vOt);
            const float fS = (pTriInfos[f].iFlag & ORIENT_PRESERVING) == 0 ? (-1.0f) : 1.0f;
            if (NotZero(fLenOs)) pTriInfos[f].vOs = vscale(fS / fLenOs, vOs);
            if (NotZero(fLenOt)) pTriInfos[f].vOt = vscale(fS / fLenOt, vOt);
            pTriInfos[f].fMagS = fLenOs / fAbsArea;      // evaluate magnitudes prior to normalization of vOs and vOt
            pTriInfos[f].fMagT = fLenOt / fAbsArea;
            if (NotZero(pTriInfos[f].fMagS) && NotZero(pTriInfos[f].fMagT)) pTriInfos[f].iFlag &= (~GROUP_WITH_ANY);      // if this is a good triangle
        }
    }
    std::vector<SEdge> pEdges((size_t)iNrTrianglesIn * 3);
    BuildNeighborsFast(pTriInfos, pEdges, piTriListIn, iNrTrianglesIn);
}

// ---- Build4RuleGroups: the faces around a vertex that share its tangent space ----
static void AddTriToGroup(SGroup* pGroup, int iTriIndex) { pGroup->pFaceIndices[pGroup->iNrFaces] = iTriIndex; ++pGroup->iNrFaces; }
static bool AssignRecur(const int piTriListIn[], STriInfo psTriInfos[], int iMyTriIndex, SGroup* pGroup) {
    STriInfo* pMyTriInfo = &psTriInfos[iMyTriIndex];
    const int iVertRep = pGroup->iVertexRepresentitive;
    const int* pVerts = &piTriListIn[3 * iMyTriIndex];
    int i = -1;
    if (pVerts[0] == iVertRep) i = 0; else if (pVerts[1] == iVertRep) i = 1; else if (pVerts[2] == iVertRep) i = 2;
    if (i < 0) return false;      // (cannot happen: the neighbour shares the edge that contains the vertex)
    if (pMyTriInfo->AssignedGroup[i] == pGroup) return true;
    else if (pMyTriInfo->AssignedGroup[i] != nullptr) return false;
    if ((pMyTriInfo->iFlag & GROUP_WITH_ANY) != 0) {
        // first to group with a group-with-anything triangle determines its orientation (the one order dependency of the algorithm)
        if (pMyTriInfo->AssignedGroup[0] == nullptr && pMyTriInfo->AssignedGroup[1] == nullptr && pMyTriInfo->AssignedGroup[2] == nullptr) {
            pMyTriInfo->iFlag &= (~ORIENT_PRESERVING);
            pMyTriInfo->iFlag |= (pGroup->bOrientPreservering ? ORIENT_PRESERVING : 0);
        }
    }
    { const bool bOrient = (pMyTriInfo->iFlag & ORIENT_PRESERVING) != 0; if (bOrient != pGroup->bOrientPreservering) return false; }
    AddTriToGroup(pGroup, iMyTriIndex);
    pMyTriInfo->AssignedGroup[i] = pGroup;
    { const int neigh_indexL = pMyTriInfo->FaceNeighbors[i], neigh_indexR = pMyTriInfo->FaceNeighbors[i > 0 ? (i - 1) : 2];
      if (neigh_indexL >= 0) AssignRecur(piTriListIn, psTriInfos, neigh_indexL, pGroup);
      if (neigh_indexR >= 0) AssignRecur(piTriListIn, psTriInfos, neigh_indexR, pGroup); }
    return true;
}
static int Build4RuleGroups(STriInfo pTriInfos[], SGroup pGroups[], int piGroupTrianglesBuffer[], const int piTriListIn[], int iNrTrianglesIn) {
    int iNrActiveGroups = 0, iOffset = 0;
    for (int f = 0; f < iNrTrianglesIn; f++)
        for (int i = 0; i < 3; i++) {
            if ((pTriInfos[f].iFlag & GROUP_WITH_ANY) == 0 && pTriInfos[f].AssignedGroup[i] == nullptr) {
                const int vert_index = piTriListIn[f * 3 + i];
                pTriInfos[f].AssignedGroup[i] = &pGroups[iNrActiveGroups];
                SGroup* g = pTriInfos[f].AssignedGroup[i];
                g->iVertexRepresentitive = vert_index; g->bOrientPreservering = (pTriInfos[f].iFlag & ORIENT_PRESERVING) != 0;
                g->iNrFaces = 0; g->pFaceIndices = &piGroupTrianglesBuffer[iOffset];
                ++iNrActiveGroups;
                AddTriToGroup(g, f);
                const int neigh_indexL = pTriInfos[f].FaceNeighbors[i], neigh_indexR = pTriInfos[f].FaceNeighbors[i > 0 ? (i - 1) : 2];
                if (neigh_indexL >= 0) AssignRecur(piTriListIn, pTriInfos, neigh_indexL, g);
                if (neigh_indexR >= 0) AssignRecur(piTriListIn, pTriInfos, neigh_indexR, g);
                iOffset += g->iNrFaces;
            }
        }
    return iNrActiveGroups;
}

// ---- GenerateTSpaces ----
static STSpace EvalTspace(const int face_indices[], int iFaces, const int piTriListIn[], const STriInfo pTriInfos[], const Context& C, int iVertexRepresentitive) {
    STSpace res; res.vOs = {0, 0, 0}; res.vOt = {0, 0, 0}; res.fMagS = 0; res.fMagT = 0; res.iCounter = 0; res.bOrient = false;
    float fAngleSum = 0;
    for (int face = 0; face < iFaces; face++) {
        const int f = face_indices[face];
        if ((pTriInfos[f].iFlag & GROUP_WITH_ANY) == 0) {      // only valid triangles get to add their contribution
            int i = -1;
            if (piTriListIn[3 * f + 0] == iVertexRepresentitive) i = 0; else if (piTriListIn[3 * f + 1] == iVertexRepresentitive) i = 1; else if (piTriListIn[3 * f + 2] == iVertexRepresentitive) i = 2;
            if (i < 0) continue;
            const int index = piTriListIn[3 * f + i];
            const SVec3 n = C.GetNormal(index);
            SVec3 vOs = vsub(pTriInfos[f].vOs, vscale(vdot(n, pTriInfos[f].vOs), n));      // project
            SVec3 vOt = vsub(pTriInfos[f].vOt, vscale(vdot(n, pTriInfos[f].vOt), n));
            if (VNotZero(vOs)) vOs = Normalize(vOs);
            if (VNotZero(vOt)) vOt = Normalize(vOt);
            const int i2 = piTriListIn[3 * f + (i < 2 ? (i + 1) : 0)], i1 = piTriListIn[3 * f + i], i0 = piTriListIn[3 * f + (i > 0 ? (i - 1) : 2)];
            const SVec3 p0 = C.GetPosition(i0), p1 = C.GetPosition(i1), p2 = C.GetPosition(i2);
            SVec3 v1 = vsub(p0, p1), v2 = vsub(p2, p1);
            v1 = vsub(v1, vscale(vdot(n, v1), n)); if (VNotZero(v1)) v1 = Normalize(v1);      // project
            v2 = vsub(v2, vscale(vdot(n, v2), n)); if (VNotZero(v2)) v2 = Normalize(v2);
            float fCos = vdot(v1, v2); fCos = fCos > 1 ? 1 : (fCos < (-1) ? (-1) : fCos);      // weight contribution by the angle between the two edge vectors
            const float fAngle = (float)acos(fCos);
            const float fMagS = pTriInfos[f].fMagS, fMagT = pTriInfos[f].fMagT;
            res.vOs = vadd(res.vOs, vscale(fAngle, vOs)); res.vOt = vadd(res.vOt, vscale(fAngle, vOt));
            res.fMagS += (fAngle * fMagS); res.fMagT += (fAngle * fMagT);
            fAngleSum += fAngle;
        }
    }
    if (VNotZero(res.vOs)) res.vOs = Normalize(res.vOs);
    if (VNotZero(res.vOt)) res.vOt = Normalize(res.vOt);
    if (fAngleSum > 0) { res.fMagS /= fAngleSum; res.fMagT /= fAngleSum; }
    return res;
}
static STSpace AvgTSpace(const STSpace* pTS0, const STSpace* pTS1) {
    STSpace ts_res;
    if (pTS0->fMagS == pTS1->fMagS && pTS0->fMagT == pTS1->fMagT && veq(pTS0->vOs, pTS1->vOs) && veq(pTS0->vOt, pTS1->vOt)) {
        ts_res = *pTS0; ts_res.fMagS = pTS0->fMagS; ts_res.fMagT = pTS0->fMagT; ts_res.vOs = pTS0->vOs; ts_res.vOt = pTS0->vOt;
    } else {
        ts_res = *pTS0;
        ts_res.fMagS = 0.5f * (pTS0->fMagS + pTS1->fMagS); ts_res.fMagT = 0.5f * (pTS0->fMagT + pTS1->fMagT);
        ts_res.vOs = vadd(pTS0->vOs, pTS1->vOs); ts_res.vOt = vadd(pTS0->vOt, pTS1->vOt);
        if (VNotZero(ts_res.vOs)) ts_res.vOs = Normalize(ts_res.vOs);
        if (VNotZero(ts_res.vOt)) ts_res.vOt = Normalize(ts_res.vOt);
    }
    return ts_res;
}
static void GenerateTSpaces(STSpace psTspace[], const STriInfo pTriInfos[], const SGroup pGroups[], int iNrActiveGroups, const int piTriListIn[], float fThresCos, const Context& C) {
    int iMaxNrFaces = 0;
    for (int g = 0; g < iNrActiveGroups; g++) if (iMaxNrFaces < pGroups[g].iNrFaces) iMaxNrFaces = pGroups[g].iNrFaces;
    if (iMaxNrFaces == 0) return;
    std::vector<STSpace> pSubGroupTspace((size_t)iMaxNrFaces);
    std::vector<std::vector<int>> pUniSubGroups((size_t)iMaxNrFaces);
    std::vector<int> pTmpMembers((size_t)iMaxNrFaces);
    for (int g = 0; g < iNrActiveGroups; g++) {
        const SGroup* pGroup = &pGroups[g];
        int iUniqueSubGroups = 0;
        for (int i = 0; i < pGroup->iNrFaces; i++) {
            const int f = pGroup->pFaceIndices[i];
            int index = -1;
            if (pTriInfos[f].AssignedGroup[0] == pGroup) index = 0; else if (pTriInfos[f].AssignedGroup[1] == pGroup) index = 1; else if (pTriInfos[f].AssignedGroup[2] == pGroup) index = 2;
            if (index < 0) continue;
            const int iVertIndex = piTriListIn[f * 3 + index];
            const SVec3 n = C.GetNormal(iVertIndex);      // is normalized already
            SVec3 vOs = vsub(pTriInfos[f].vOs, vscale(vdot(n, pTriInfos[f].vOs), n));
            SVec3 vOt = vsub(pTriInfos[f].vOt, vscale(vdot(n, pTriInfos[f].vOt), n));
            if (VNotZero(vOs)) vOs = Normalize(vOs);
            if (VNotZero(vOt)) vOt = Normalize(vOt);
            const int iOF_1 = pTriInfos[f].iOrgFaceNumber;
            int iMembers = 0;
            for (int j = 0; j < pGroup->iNrFaces; j++) {
                const int t = pGroup->pFaceIndices[j];
                const int iOF_2 = pTriInfos[t].iOrgFaceNumber;
                SVec3 vOs2 = vsub(pTriInfos[t].vOs, vscale(vdot(n, pTriInfos[t].vOs), n));
                SVec3 vOt2 = vsub(pTriInfos[t].vOt, vscale(vdot(n, pTriInfos[t].vOt), n));
                if (VNotZero(vOs2)) vOs2 = Normalize(vOs2);
                if (VNotZero(vOt2)) vOt2 = Normalize(vOt2);
                const bool bAny = ((pTriInfos[f].iFlag | pTriInfos[t].iFlag) & GROUP_WITH_ANY) != 0;
                const bool bSameOrgFace = iOF_1 == iOF_2;
                const float fCosS = vdot(vOs, vOs2), fCosT = vdot(vOt, vOt2);
                if (bAny || bSameOrgFace || (fCosS > fThresCos && fCosT > fThresCos)) pTmpMembers[(size_t)iMembers++] = t;
            }
            if (iMembers > 1) std::sort(pTmpMembers.begin(), pTmpMembers.begin() + iMembers);      // QuickSort of mikktspace.c: ascending triangle numbers
            bool bFound = false; int l = 0;
            while (l < iUniqueSubGroups && !bFound) {
                const std::vector<int>& u = pUniSubGroups[(size_t)l];
                bFound = (int)u.size() == iMembers && std::equal(u.begin(), u.end(), pTmpMembers.begin());
                if (!bFound) ++l;
            }
            if (!bFound) {
                pUniSubGroups[(size_t)iUniqueSubGroups].assign(pTmpMembers.begin(), pTmpMembers.begin() + iMembers);
                pSubGroupTspace[(size_t)iUniqueSubGroups] = EvalTspace(pTmpMembers.data(), iMembers, piTriListIn, pTriInfos, C, pGroup->iVertexRepresentitive);
                ++iUniqueSubGroups;
            }
            const int iOffs = pTriInfos[f].iTSpacesOffs, iVert = pTriInfos[f].vert_num[index];
            STSpace* pTS_out = &psTspace[iOffs + iVert];
            if (pTS_out->iCounter == 1) { *pTS_out = AvgTSpace(pTS_out, &pSubGroupTspace[(size_t)l]); pTS_out->iCounter = 2; pTS_out->bOrient = pGroup->bOrientPreservering; }
            else { *pTS_out = pSubGroupTspace[(size_t)l]; pTS_out->iCounter = 1; pTS_out->bOrient = pGroup->bOrientPreservering; }
        }
    }
}

// ---- DegenEpilogue (triangles only): a degenerate triangle's corners take the tangent space of a good triangle on the same vertex ----
static void DegenEpilogue(STSpace psTspace[], const STriInfo pTriInfos[], const int piTriListIn[], int iNrTrianglesIn, int iTotTris) {
    for (int t = iNrTrianglesIn; t < iTotTris; t++) {
        const bool bSkip = (pTriInfos[t].iFlag & QUAD_ONE_DEGEN_TRI) != 0;
        if (bSkip) continue;
        for (int i = 0; i < 3; i++) {
            const int index1 = piTriListIn[t * 3 + i];
            bool bNotFound = true; int j = 0;
            while (bNotFound && j < (3 * iNrTrianglesIn)) { const int index2 = piTriListIn[j]; if (index1 == index2) bNotFound = false; else ++j; }
            if (!bNotFound) {
                const int iTri = j / 3, iVert = j % 3;
                const int iSrcVert = pTriInfos[iTri].vert_num[iVert], iSrcOffs = pTriInfos[iTri].iTSpacesOffs;
                const int iDstVert = pTriInfos[t].vert_num[i], iDstOffs = pTriInfos[t].iTSpacesOffs;
                psTspace[iDstOffs + iDstVert] = psTspace[iSrcOffs + iSrcVert];
            }
        }
    }
}

} // namespace

// genTangSpaceDefault for one triangle primitive + the reference's SetTSpaceBasic (mesh-utils.cpp:105-112).
// pos: nVerts x 3, nrm: nVerts x 3, uv: nVerts x 2, idx: nTris x 3; out_tangents: nVerts x 4 (zero where no face references the vertex).
extern "C" int orc_generate_tangents(const float* pos, const float* nrm, const float* uv, const uint32_t* idx, uint32_t nVerts, uint32_t nTris, float* out_tangents) {
    if (!pos || !nrm || !uv || !idx || !out_tangents) return -1;
    if (nTris > 0x1fffffffu) return -1;
    for (size_t i = 0; i < (size_t)nTris * 3; ++i) if (idx[i] >= nVerts) return -2;
    memset(out_tangents, 0, (size_t)nVerts * 16);
    const Context C = {pos, nrm, uv, idx, nVerts, nTris};
    const int iNrFaces = (int)nTris;
    int iNrTrianglesIn = iNrFaces;
    if (iNrTrianglesIn <= 0) return 0;
    const float fThresCos = (float)cos((180.0f * (float)3.14159265358979323846) / 180.0f);      // genTangSpaceDefault: fAngularThreshold = 180
    std::vector<int> piTriListIn((size_t)3 * iNrTrianglesIn);
    std::vector<STriInfo> pTriInfos((size_t)iNrTrianglesIn);
    memset(pTriInfos.data(), 0, pTriInfos.size() * sizeof(STriInfo));
    // GenerateInitialVerticesIndexList
    int iTSpacesOffs = 0;
    for (int f = 0; f < iNrFaces; f++) {
        pTriInfos[(size_t)f].iOrgFaceNumber = f; pTriInfos[(size_t)f].iTSpacesOffs = iTSpacesOffs;
        unsigned char* pVerts = pTriInfos[(size_t)f].vert_num; pVerts[0] = 0; pVerts[1] = 1; pVerts[2] = 2;
        piTriListIn[(size_t)f * 3 + 0] = MakeIndex(f, 0); piTriListIn[(size_t)f * 3 + 1] = MakeIndex(f, 1); piTriListIn[(size_t)f * 3 + 2] = MakeIndex(f, 2);
        iTSpacesOffs += 3;
    }
    const int iNrTSPaces = iTSpacesOffs;
    for (int t = 0; t < iNrTrianglesIn; t++) pTriInfos[(size_t)t].iFlag = 0;
    GenerateSharedVerticesIndexList(piTriListIn.data(), C, iNrTrianglesIn);
    const int iTotTris = iNrTrianglesIn;
    int iDegenTriangles = 0;
    for (int t = 0; t < iTotTris; t++) {
        const SVec3 p0 = C.GetPosition(piTriListIn[(size_t)t * 3 + 0]), p1 = C.GetPosition(piTriListIn[(size_t)t * 3 + 1]), p2 = C.GetPosition(piTriListIn[(size_t)t * 3 + 2]);
        if (veq(p0, p1) || veq(p0, p2) || veq(p1, p2)) { pTriInfos[(size_t)t].iFlag |= MARK_DEGENERATE; ++iDegenTriangles; }
    }
    iNrTrianglesIn = iTotTris - iDegenTriangles;
    DegenPrologue(pTriInfos.data(), piTriListIn.data(), iNrTrianglesIn, iTotTris);
    InitTriInfo(pTriInfos.data(), piTriListIn.data(), C, iNrTrianglesIn);
    const int iNrMaxGroups = iNrTrianglesIn * 3;
    std::vector<SGroup> pGroups((size_t)(iNrMaxGroups > 0 ? iNrMaxGroups : 1));
    std::vector<int> piGroupTrianglesBuffer((size_t)(iNrTrianglesIn > 0 ? iNrTrianglesIn * 3 : 1));
    const int iNrActiveGroups = Build4RuleGroups(pTriInfos.data(), pGroups.data(), piGroupTrianglesBuffer.data(), piTriListIn.data(), iNrTrianglesIn);
    std::vector<STSpace> psTspace((size_t)iNrTSPaces);
    for (int t = 0; t < iNrTSPaces; t++) { psTspace[(size_t)t].vOs = {1.0f, 0.0f, 0.0f}; psTspace[(size_t)t].fMagS = 1.0f; psTspace[(size_t)t].vOt = {0.0f, 1.0f, 0.0f}; psTspace[(size_t)t].fMagT = 1.0f; psTspace[(size_t)t].iCounter = 0; psTspace[(size_t)t].bOrient = false; }
    GenerateTSpaces(psTspace.data(), pTriInfos.data(), pGroups.data(), iNrActiveGroups, piTriListIn.data(), fThresCos, C);
    DegenEpilogue(psTspace.data(), pTriInfos.data(), piTriListIn.data(), iNrTrianglesIn, iTotTris);
    // output: m_setTSpaceBasic(tangent, bOrient ? 1 : -1, face, vert) in face order -> the reference writes the INDEXED vertex (last writer wins)
    int index = 0;
    for (int f = 0; f < iNrFaces; f++)
        for (int i = 0; i < 3; i++) {
            const STSpace* pTSpace = &psTspace[(size_t)index];
            float* dst = out_tangents + (size_t)idx[(size_t)f * 3 + i] * 4;
            dst[0] = pTSpace->vOs.x; dst[1] = pTSpace->vOs.y; dst[2] = pTSpace->vOs.z; dst[3] = pTSpace->bOrient ? 1.0f : (-1.0f);
            ++index;
        }
    return 0;
}
