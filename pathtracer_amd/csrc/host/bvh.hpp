// host/bvh.hpp — CPU BVH builder of the host side.
//
// Keeps the reference's public shape, class BoundingVolumeHierarchy with
// buildTree / get_triangles / get_bvh_bbox / get_bvh_info / get_n_nodes
// (reference pathtracer/bvh.hpp:59-73), and produces node and triangle ORDER
// identical to the reference builder (bvh.cpp:141-390) including its quirks,
// because hit order on ties depends on it.  The
// implementation is its own: flat arrays, counting sort per node, no per-node
// allocations, and parallel over host threads (HPT_BUILD_THREADS, default
// min(cores, 16)) with results identical to the sequential recursion.
#pragma once
#include <cstdint>
#include <thread>
#include <vector>

#include "../../../include/hpt/types.h"

namespace hpt {

struct Vec3f {
    float n[3];
    Vec3f() : n{0, 0, 0} {}
    Vec3f(float x, float y, float z) : n{x, y, z} {}
    float &operator[](int i) { return n[i]; }
    const float &operator[](int i) const { return n[i]; }
};
inline Vec3f operator-(const Vec3f &a, const Vec3f &b) { return Vec3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }

struct BoundingBox {
    Vec3f minbounds, maxbounds;
};

// reference common.hpp:83-113
struct VectorTriangle {
    Vec3f v1, v2, v3, centroid;
    BoundingBox bbox;
    VectorTriangle() {}
    VectorTriangle(Vec3f a, Vec3f b, Vec3f c);
};

// host threads of the builder: min(cores, 16), HPT_BUILD_THREADS overrides
int build_threads();

// f(thread, begin, end) over [0, n) in contiguous chunks, one per thread (chunk order = thread order)
template <class F>
inline void parallel_chunks(int threads, long n, F f) {
    if (threads <= 1 || n < 2) { f(0, 0L, n); return; }
    std::vector<std::thread> th;
    th.reserve(threads - 1);
    for (int t = 1; t < threads; t++) th.emplace_back([=]() { f(t, n * t / threads, n * (t + 1) / threads); });
    f(0, 0L, n / threads);
    for (auto &x : th) x.join();
}

const int MAX_TRIANGLES_LEAF = 24;   // bvh.hpp:8
const int NUM_BUCKETS = 12;          // bvh.hpp:9

class BoundingVolumeHierarchy {
public:
    BoundingVolumeHierarchy() {}
    explicit BoundingVolumeHierarchy(std::vector<VectorTriangle> vt) : vtriangles(std::move(vt)) {}

    // Returns false (and builds nothing) for fewer than NUM_BUCKETS triangles: the
    // reference reads vtriangles[0..11] unconditionally (bvh.cpp:242-261).
    bool buildTree();
    const std::vector<VectorTriangle> &get_triangles() const { return vtriangles; }   // leaf order after buildTree
    const std::vector<int32_t> &get_order() const { return order; }                   // input index of the k-th leaf-order triangle
    hpt_bvh_bbox *get_bvh_bbox() { return built ? bvh_bbox.data() : nullptr; }
    hpt_bvh_info *get_bvh_info() { return built ? bvh_info.data() : nullptr; }
    int get_n_nodes() const { return (int)bvh_bbox.size(); }
    int get_depth() const { return depth; }

private:
    struct Item { float mn[3], mx[3], c[3]; int32_t src; };
    struct Subtree {                       // nodes of one subtree in depth-first order, local indices
        std::vector<hpt_bvh_bbox> bbox;
        std::vector<hpt_bvh_info> info;
        int depth = 0;
    };
    int build(Subtree &out, int start, int end, int level);
    bool centroid_axis(int start, int end, int threads, int &axis, float &lo, double &span) const;
    int partition_large(int start, int end, int threads);
    std::vector<VectorTriangle> vtriangles;
    std::vector<Item> items, scratch;
    std::vector<unsigned char> bucket;
    std::vector<int32_t> order;
    std::vector<hpt_bvh_bbox> bvh_bbox;
    std::vector<hpt_bvh_info> bvh_info;
    int depth = 0;
    bool built = false;
};

}  // namespace hpt
