// host/bvh.cpp — order-exact CPU BVH builder (see bvh.hpp).
//
// What the reference algorithm amounts to (reference pathtracer/bvh.cpp:141-336):
// per node, the range is stably sorted by a 12-way bucket index along the
// longest axis of the centroid bounds (its pre/post split writes the buckets
// back in bucket order whatever bestSplit is, :291-315) and is then cut at the
// array midpoint (:325-326).  The "SAH" cost (:235-271) only decides whether a
// node of <= 24 triangles becomes a leaf, and it is computed from the bounding
// boxes of the triangles currently at global positions 0 and i+1.  All of that
// is kept; float/double mixing of the bucket index (:210-215) too.
#include "bvh.hpp"

#include <cstring>

namespace hpt {

static inline float fmin_ref(float a, float b) { return (b < a) ? b : a; }   // std::min(a,b)
static inline float fmax_ref(float a, float b) { return (a < b) ? b : a; }   // std::max(a,b)

VectorTriangle::VectorTriangle(Vec3f a, Vec3f b, Vec3f c) : v1(a), v2(b), v3(c) {   // common.hpp:94-106
    for (int k = 0; k < 3; k++) {
        bbox.minbounds[k] = fmin_ref(fmin_ref(v1[k], v2[k]), v3[k]);
        bbox.maxbounds[k] = fmax_ref(fmax_ref(v1[k], v2[k]), v3[k]);
        centroid[k] = (float)((bbox.minbounds[k] + bbox.maxbounds[k]) * 0.5);
    }
}

static inline float surface_area(const float mn[3], const float mx[3]) {   // common.hpp:52-56
    float v0 = mx[0] - mn[0], v1 = mx[1] - mn[1], v2 = mx[2] - mn[2];
    return 2.0f * (v0 * v1 + v0 * v2 + v1 * v2);
}

int BoundingVolumeHierarchy::build(int start, int end, int level) {
    const int k = (int)bvh_bbox.size();
    bvh_bbox.emplace_back();
    bvh_info.push_back(hpt_bvh_info{0, 0, 0, 0});
    if (level > depth) depth = level;

    Item *it = items.data();
    float bmn[3], bmx[3];
    std::memcpy(bmn, it[start].mn, sizeof bmn);
    std::memcpy(bmx, it[start].mx, sizeof bmx);
    for (int i = start + 1; i < end; i++)
        for (int a = 0; a < 3; a++) { bmn[a] = fmin_ref(bmn[a], it[i].mn[a]); bmx[a] = fmax_ref(bmx[a], it[i].mx[a]); }

    const int n = end - start;
    auto make_leaf = [&]() {
        bvh_bbox[k] = hpt_bvh_bbox{bmn[0], bmn[1], bmn[2], bmx[0], bmx[1], bmx[2]};
        bvh_info[k].offset = start;          // leaves are created left to right, so the running count is `start`
        bvh_info[k].nPrimitives = n;
        return k;
    };
    if (n == 1) return make_leaf();

    float cmn[3], cmx[3];
    std::memcpy(cmn, it[start].c, sizeof cmn);
    std::memcpy(cmx, it[start].c, sizeof cmx);
    for (int i = start + 1; i < end; i++)
        for (int a = 0; a < 3; a++) { cmn[a] = fmin_ref(cmn[a], it[i].c[a]); cmx[a] = fmax_ref(cmx[a], it[i].c[a]); }
    int axis = 0;                                                           // common.hpp:39-50
    if ((cmx[1] - cmn[1]) > (cmx[0] - cmn[0])) axis = 1;
    if ((cmx[2] - cmn[2]) > (cmx[axis] - cmn[axis])) axis = 2;
    const int mid = (start + end) / 2;

    bool reorder = false;
    if (cmx[axis] == cmn[axis]) {                                           // bvh.cpp:174-193
        if (n <= MAX_TRIANGLES_LEAF) return make_leaf();
    } else {
        int count[NUM_BUCKETS] = {0};
        const double span = cmx[axis] - cmn[axis];
        for (int i = start; i < end; i++) {                                 // bvh.cpp:212-220
            int b = (int)((float)NUM_BUCKETS * (it[i].c[axis] - cmn[axis]) / span);
            if (b == NUM_BUCKETS) b--;
            bucket[i] = (unsigned char)b;
            count[b]++;
        }
        if (n <= MAX_TRIANGLES_LEAF) {                                      // the cost only matters here (bvh.cpp:285)
            float best = 0;
            for (int i = 0; i < NUM_BUCKETS - 1; i++) {                     // bvh.cpp:235-282
                int c0 = 0, c1 = 0;
                for (int j = 0; j <= i; j++) c0 += count[j];
                for (int j = i + 1; j < NUM_BUCKETS; j++) c1 += count[j];
                float cost = 0.125f + (c0 * surface_area(it[0].mn, it[0].mx) + c1 * surface_area(it[i + 1].mn, it[i + 1].mx)) /
                                          surface_area(bmn, bmx);
                if (i == 0 || cost < best) best = cost;
            }
            if (!(best < n)) return make_leaf();
        }
        reorder = true;
        int pos[NUM_BUCKETS], run = 0;                                      // stable counting sort == bvh.cpp:291-315
        for (int b = 0; b < NUM_BUCKETS; b++) { pos[b] = run; run += count[b]; }
        Item *tmp = scratch.data();
        for (int i = start; i < end; i++) tmp[pos[bucket[i]]++] = it[i];
        std::memcpy(it + start, tmp, sizeof(Item) * (size_t)n);
    }
    (void)reorder;
    const int c0 = build(start, mid, level + 1);
    const int c1 = build(mid, end, level + 1);
    const hpt_bvh_bbox &a = bvh_bbox[c0], &b = bvh_bbox[c1];               // combine(children), bvh.cpp:327-328
    bvh_bbox[k] = hpt_bvh_bbox{fmin_ref(a.minboundsx, b.minboundsx), fmin_ref(a.minboundsy, b.minboundsy),
                               fmin_ref(a.minboundsz, b.minboundsz), fmax_ref(a.maxboundsx, b.maxboundsx),
                               fmax_ref(a.maxboundsy, b.maxboundsy), fmax_ref(a.maxboundsz, b.maxboundsz)};
    bvh_info[k].secondChildOffset = c1;
    bvh_info[k].axis = axis;
    return k;
}

bool BoundingVolumeHierarchy::buildTree() {
    built = false;
    bvh_bbox.clear(); bvh_info.clear(); order.clear(); depth = 0;
    const int n = (int)vtriangles.size();
    if (n < NUM_BUCKETS) return false;
    items.resize(n); scratch.resize(n); bucket.resize(n);
    for (int i = 0; i < n; i++) {
        const VectorTriangle &t = vtriangles[i];
        for (int a = 0; a < 3; a++) { items[i].mn[a] = t.bbox.minbounds[a]; items[i].mx[a] = t.bbox.maxbounds[a]; items[i].c[a] = t.centroid[a]; }
        items[i].src = i;
    }
    bvh_bbox.reserve(2 * (size_t)n); bvh_info.reserve(2 * (size_t)n);
    build(0, n, 0);
    order.resize(n);
    std::vector<VectorTriangle> sorted(n);
    for (int i = 0; i < n; i++) { order[i] = items[i].src; sorted[i] = vtriangles[items[i].src]; }
    vtriangles.swap(sorted);
    items.clear(); items.shrink_to_fit(); scratch.clear(); scratch.shrink_to_fit(); bucket.clear(); bucket.shrink_to_fit();
    built = true;
    return true;
}

}  // namespace hpt
