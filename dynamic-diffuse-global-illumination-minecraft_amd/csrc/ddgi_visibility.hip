// ddgi_visibility.hip — k_light_visibility: per (voxel, face) and light, is a light feeler that starts just off this
// face of the voxel CERTAIN to reach the light, CERTAIN to be blocked, or neither?
//
// A light feeler (get_direct_lighting, probe_pass.comp:186-207 -> intersect_scene -> grid_march) only decides a
// boolean: does the march land in an occupied voxel before it reaches the light sphere.  For most surface
// points that is decided by the geometry with room to spare, and the trace kernel (wf_event) then skips the
// feeler's march, its queue trip and its event.  The table is CONSERVATIVE — a class is only given when the
// reference's float march provably comes to that outcome for EVERY start point in the voxel's interior — so
// results are bit-identical with and without it (tests: trace with/without table agree; both equal the oracle).
//
// Voxel id n covers (n-1, n] per axis (voxel id = ceil(p)).  A feeler starts 1e-3 off the face of the
// block that was hit, inside the empty voxel on the other side of that face.  Start points o of table entry
// (voxel, face): between 4e-4 and 1.6e-3 off that face, and at least kShrink inside the voxel on the two other axes
// (wf_event only uses the entry for an origin 5e-4 .. 1.5e-3 off the face and 5e-4 inside laterally).  Rays run from o to the light centre L; the march
// positions p_k = fma(dn, t_k, o) stay within ~3e-5 of that segment for t_k <= t_light (binary32 rounding at
// |p| < 2^10 and a unit direction good to 1e-7), so every voxel the march looks up intersects the kEps-tube
// around the bundle B = hull(shrunk voxel, L).  Cross-sections of B are axis-aligned boxes
//     X(s) = (1-s) * voxel + s * L,   s in [0, 1],
// so along the dominant axis a of (L - centre) the part of B inside voxel layer i is covered by the bounding box
// of X(s_in) and X(s_out), the cross-sections where the bundle enters and leaves the layer (+- kEps).
//   LIT     every voxel (other than the start voxel) in every layer's bounding box, from the start layer to
//           the light's, is empty: no march position can be in an occupied voxel.
//   SHADOW  some layer strictly between (>= 3 layers before the light's: t_hit < t_light with a margin far above
//           the error of the sphere quadratic) has its bounding box fully occupied, and is reached within
//           grid_march's 125 iterations: an iteration ends 1e-4 past the NEXT voxel boundary on its way, so it
//           crosses at least one boundary, and a ray of the bundle crosses at most (layers + lateral voxel offsets
//           of the bounding box) boundaries before it is inside the layer; kVisMaxCrossings leaves 15 iterations
//           for the rare step that starts exactly on a boundary.  The march lands in an occupied voxel there or
//           earlier — and the outcome "blocked" does not depend on where — PROVIDED it cannot step over the layer:
//           it can, when the bundle travels in the POSITIVE direction of the dominant axis.  grid_march takes the next
//           boundary from fract(p): a position that lands EXACTLY on an integer plane while travelling in the positive
//           direction belongs to the voxel below the plane (id = ceil(p)) and its next boundary along that axis is a whole
//           voxel further on — if that boundary comes first, the layer above the plane is never looked up, however long
//           the ray's stretch inside it.  (One landing in a million is exact; a C3 update has half a billion landings.)
//           The landing after such a step is at most 1e-4 past the layer's far plane, and not on a plane: so for a
//           positive dominant direction the voxels of the NEXT layer that the bundle touches within kVisSkipDepth of
//           that plane must be occupied as well.  In the negative direction a position on a plane belongs to the voxel it
//           is about to cross and nothing is stepped over.
//   LISTED  neither, but every layer was examined and the bundle's bounding boxes hold at most kVisListMax occupied voxels: they
//           are written out (relative to the start voxel) and wf_event tests the feeler's OWN ray against them — a ray that clears
//           them all with a margin cannot land in an occupied voxel: everything else the march can look up is empty
//           (ddgi_trace_wf.hip: listed_feeler_outcome).  Three quarters of the feelers that used to be marched start in such a
//           patch.  (The converse — "the ray runs through the inside of a listed voxel, so the march lands in it" — is false for
//           the same reason one full layer is not enough for SHADOW; tried, and one feeler of a C3 update's 33 million differed.)
//   UNKNOWN otherwise: the feeler is marched.
#include "ddgi_device.h"

namespace ddgi {

constexpr double kVisShrink = 4.0e-4;  // start points: the voxel shrunk by this much laterally (wf_event requires 5e-4)
constexpr double kVisFaceNear = 4.0e-4, kVisFaceFar = 1.6e-3;  // ... and this far off the face that was hit (wf_event: 5e-4 .. 1.5e-3)
constexpr double kVisEps = 1.0e-4;     // tube around the exact bundle that contains every march position
constexpr int kVisMaxCrossings = 110;  // boundary crossings to a blocking layer (grid_march: 125 iterations)
constexpr double kVisMinRange = 4.0;   // lights closer than this along the dominant axis: not classified
constexpr double kVisMaxRange = 400.0;
constexpr int kVisLanes = 16;          // lanes that share one voxel's layers
constexpr double kVisSkipDepth = 1.0e-3;  // SHADOW, positive direction: how deep into the next layer a march that stepped over the full one can land

DDGI_D bool vis_occupied(const SceneK& S, const uint32_t* __restrict__ bits, int x, int y, int z)
{
    const int idx = cell_index(S, x, y, z);  // clamped: outside the box the world is the extrusion of the border layer
    return ((bits[(idx >> 5) - (S.bias32 >> 5)] >> (idx & 31)) & 1u) != 0u;
}

// list: the (voxel, face) pairs that can hold a feeler origin — an empty voxel and a face whose other side is occupied (the
// origin is 1e-3 off the face that was hit); entry = voxel * 8 + face, built once per scene on the host (ddgi_engine.cpp:
// plan_trace).  Every other table entry keeps class 0.
// kLds: the occupancy bitmap is copied to LDS first (a thread makes several hundred dependent lookups)
template <bool kLds>
__global__ __launch_bounds__(256) void k_light_visibility(const SceneK S, const double lx, const double ly, const double lz, const int32_t* __restrict__ list,
                                                          const int n_list, uint8_t* __restrict__ out, uint32_t* __restrict__ out_occ)
{
    extern __shared__ uint32_t vis_lds[];
    const uint32_t* __restrict__ bits = S.bits;
    if (kLds)
    {
        for (int i = threadIdx.x; i < S.nwords; i += blockDim.x) vis_lds[i] = S.bits[i];
        __syncthreads();
        bits = vis_lds;
    }
    // kVisLanes lanes per voxel: the layers between the voxel and the light are independent tests, dealt round robin to the
    // lanes of a group and joined with an OR / AND across the group (a single lane walking ~60 layers was 150 us per update)
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / kVisLanes, sub = threadIdx.x % kVisLanes;
    const bool live = t < n_list;  // (no early return: the group shuffles below need every lane)
    const int entry = list[live ? t : n_list - 1];  // voxel * 8 + face; face = 2 axis + (the solid neighbour is on the + side)
    const int r = entry >> 3, face = entry & 7, face_axis = face >> 1;
    const int ny = S.nxy / S.nx;
    int v[3] = {S.lo[0] + r % S.nx, S.lo[1] + (r / S.nx) % ny, S.lo[2] + r / S.nxy};
    uint8_t cls = kVisUnknown;
    uint32_t found[kVisListMax];  // this lane's occupied voxels (its layers), packed offsets from the start voxel
    int n_found = 0;
    bool complete = false;        // this lane has examined every one of its layers up to the light's
    const bool relevant = true;
    const double L[3] = {lx, ly, lz};
    double lo0[3], hi0[3];
    int a = 0;
    double best = -1.0;
    for (int k = 0; k < 3; ++k)
    {
        lo0[k] = static_cast<double>(v[k] - 1) + kVisShrink;
        hi0[k] = static_cast<double>(v[k]) - kVisShrink;
        if (k == face_axis)  // a thin slab 4e-4 .. 1.6e-3 off the face
        {
            if (face & 1) lo0[k] = static_cast<double>(v[k]) - kVisFaceFar, hi0[k] = static_cast<double>(v[k]) - kVisFaceNear;
            else lo0[k] = static_cast<double>(v[k] - 1) + kVisFaceNear, hi0[k] = static_cast<double>(v[k] - 1) + kVisFaceFar;
        }
        const double d = fabs(L[k] - (static_cast<double>(v[k]) - 0.5));
        if (d > best) best = d, a = k;
    }
    if (relevant && best >= kVisMinRange && best <= kVisMaxRange)
    {
        const int b = (a + 1) % 3, c = (a + 2) % 3;
        const int sgn = L[a] > hi0[a] ? 1 : -1;
        const int n_a = v[a], m_a = static_cast<int>(ceil(L[a]));
        bool all_empty = true, blocked = false;
        complete = true;
        // voxel-id range (along b and c) of the bounding box of the bundle's cross-sections inside layer i = (i-1, i], from the plane
        // the bundle enters it through to `depth` behind that plane (1: the whole layer)
        auto layer_box = [&](int i, double depth, int (&w0)[2], int (&w1)[2]) {
            // s-range over which the cross-section X(s) reaches into that slab (widened by kVisEps)
            double s_in, s_out;
            if (sgn > 0)
            {
                s_in = (static_cast<double>(i - 1) - kVisEps - hi0[a]) / (L[a] - hi0[a]);
                s_out = (static_cast<double>(i - 1) + depth + kVisEps - lo0[a]) / (L[a] - lo0[a]);
            }
            else
            {
                s_in = (lo0[a] - (static_cast<double>(i) + kVisEps)) / (lo0[a] - L[a]);
                s_out = (hi0[a] - (static_cast<double>(i) - depth - kVisEps)) / (hi0[a] - L[a]);
            }
            s_in = fmin(fmax(s_in, 0.0), 1.0), s_out = fmin(fmax(s_out, 0.0), 1.0);
            const int ax2[2] = {b, c};
            for (int q = 0; q < 2; ++q)
            {
                const int k = ax2[q];
                const double lo_in = (1.0 - s_in) * lo0[k] + s_in * L[k], lo_out = (1.0 - s_out) * lo0[k] + s_out * L[k];
                const double hi_in = (1.0 - s_in) * hi0[k] + s_in * L[k], hi_out = (1.0 - s_out) * hi0[k] + s_out * L[k];
                w0[q] = static_cast<int>(ceil(fmin(lo_in, lo_out) - kVisEps));
                w1[q] = static_cast<int>(ceil(fmax(hi_in, hi_out) + kVisEps));
            }
        };
        for (int i = n_a + sgn * sub; sgn > 0 ? i <= m_a : i >= m_a; i += sgn * kVisLanes)
        {
            int w0[2], w1[2];
            layer_box(i, 1.0, w0, w1);
            bool layer_full = true;
            for (int wb = w0[0]; wb <= w1[0]; ++wb)
                for (int wc = w0[1]; wc <= w1[1]; ++wc)
                {
                    int w[3];
                    w[a] = i, w[b] = wb, w[c] = wc;
                    if (w[0] == v[0] && w[1] == v[1] && w[2] == v[2])
                    {
                        layer_full = false;  // the start voxel itself: empty, and never looked up by the march
                        continue;
                    }
                    const bool occ = vis_occupied(S, bits, w[0], w[1], w[2]);
                    all_empty = all_empty && !occ;
                    layer_full = layer_full && occ;
                    if (occ)
                    {
                        const uint32_t packed = vis_pack_offset(w[0] - v[0], w[1] - v[1], w[2] - v[2]);
                        if (n_found == 0) found[0] = packed;
                        else if (n_found == 1) found[1] = packed;
                        else if (n_found == 2) found[2] = packed;
                        else if (n_found == 3) found[3] = packed;
                        ++n_found;
                    }
                }
            const int from_start = sgn * (i - n_a), to_light = sgn * (m_a - i);
            const int lat_b = max(abs(w0[0] - v[b]), abs(w1[0] - v[b])), lat_c = max(abs(w0[1] - v[c]), abs(w1[1] - v[c]));
            const int crossings = from_start + lat_b + lat_c + 2;
            if (layer_full && from_start >= 1 && crossings <= kVisMaxCrossings && to_light >= 4)
            {
                // ... and, in the positive direction, where a march that stepped over it lands (see SHADOW above)
                int x0[2], x1[2];
                bool next_full = true;
                if (sgn > 0) layer_box(i + 1, kVisSkipDepth, x0, x1);
                else x0[0] = x0[1] = 0, x1[0] = x1[1] = -1;  // (nothing to examine)
                for (int wb = x0[0]; wb <= x1[0] && next_full; ++wb)
                    for (int wc = x0[1]; wc <= x1[1] && next_full; ++wc)
                    {
                        int w[3];
                        w[a] = i + sgn, w[b] = wb, w[c] = wc;
                        next_full = vis_occupied(S, bits, w[0], w[1], w[2]);
                    }
                if (next_full) blocked = true;
            }
            if (blocked || (!all_empty && crossings > kVisMaxCrossings))  // this lane's layers: decided, or can no longer decide
            {
                complete = false;
                break;
            }
        }
        cls = blocked ? kVisShadow : (all_empty ? kVisLit : kVisUnknown);
    }
    // join the group: any lane blocked -> SHADOW; every lane all-empty -> LIT; else UNKNOWN.  (A voxel out of range has
    // cls == kVisUnknown in every lane.)
    unsigned any_shadow = cls == kVisShadow ? 1u : 0u, all_lit = cls == kVisLit ? 1u : 0u;
#pragma unroll
    for (int m = 1; m < kVisLanes; m <<= 1)
    {
        any_shadow |= static_cast<unsigned>(__shfl_xor(static_cast<int>(any_shadow), m));
        all_lit &= static_cast<unsigned>(__shfl_xor(static_cast<int>(all_lit), m));
    }
    // the group's occupied voxels: how many, and where this lane's go in the entry's list
    int incl = n_found;
    unsigned all_complete = complete ? 1u : 0u;
#pragma unroll
    for (int m = 1; m < kVisLanes; m <<= 1)
    {
        const int up = __shfl_up(incl, m, kVisLanes);
        if (sub >= m) incl += up;
        all_complete &= static_cast<unsigned>(__shfl_xor(static_cast<int>(all_complete), m));
    }
    const int total = __shfl(incl, kVisLanes - 1, kVisLanes), excl = incl - n_found;
    const bool listed = out_occ != nullptr && !any_shadow && !all_lit && all_complete && total >= 1 && total <= kVisListMax;
    if (live && listed)
    {
        uint32_t* dst = out_occ + static_cast<size_t>(entry) * kVisListMax;
        if (n_found >= 1) dst[excl] = found[0];
        if (n_found >= 2) dst[excl + 1] = found[1];
        if (n_found >= 3) dst[excl + 2] = found[2];
        if (n_found >= 4) dst[excl + 3] = found[3];
        if (sub < kVisListMax && sub >= total) dst[sub] = kVisListEnd;  // (every slot has exactly one writer)
    }
    if (live && sub == 0) out[entry] = any_shadow ? kVisShadow : (all_lit ? kVisLit : (listed ? kVisListed : kVisUnknown));
}

hipError_t launch_light_visibility(const SceneK& scene, const float light_pos[3], const int32_t* list, int n_list, uint8_t* out, uint32_t* out_occ, hipStream_t stream)
{
    if (n_list <= 0) return hipSuccess;
    const size_t lds = static_cast<size_t>(scene.nwords) * sizeof(uint32_t);
    const int per_block = 256 / kVisLanes;  // voxels per 256-thread block
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(k_light_visibility<true>, dim3((n_list + per_block - 1) / per_block), dim3(256), lds, stream, scene, static_cast<double>(light_pos[0]), static_cast<double>(light_pos[1]),
                           static_cast<double>(light_pos[2]), list, n_list, out, out_occ);
    else
        hipLaunchKernelGGL(k_light_visibility<false>, dim3((n_list + per_block - 1) / per_block), dim3(256), 0, stream, scene, static_cast<double>(light_pos[0]), static_cast<double>(light_pos[1]),
                           static_cast<double>(light_pos[2]), list, n_list, out, out_occ);
    return hipGetLastError();
}

}  // namespace ddgi
