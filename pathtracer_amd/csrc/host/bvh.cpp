// host/bvh.cpp — order-exact CPU BVH builder (see bvh.hpp), parallel over host threads.
//
// What the reference algorithm amounts to (reference pathtracer/bvh.cpp:141-336):
// per node, the range is stably sorted by a 12-way bucket index along the
// longest axis of the centroid bounds (its pre/post split writes the buckets
// back in bucket order whatever bestSplit is, :291-315) and is then cut at the
// array midpoint (:325-326).  The "SAH" cost (:235-271) only decides whether a
// node of <= 24 triangles becomes a leaf, and it is computed from the bounding
// boxes of the triangles currently at global positions 0 and i+1.  All of that
// is kept; float/double mixing of the bucket index (:210-215) too.
//
// Parallel form, same arrays bit for bit as the depth-first recursion:
//  * a node of more than 24 triangles never reads anything outside its own range, so the top of the tree
//    is partitioned level by level (threads share one big node, or take one node each), down to subtrees
//    of `task_size` triangles;
//  * each such subtree is built by one thread into its own node arrays (local indices);
//  * nodes of <= 24 triangles read global positions 0..11.  In depth-first order every node that can move
//    those positions is on the leftmost chain or inside the leftmost subtree, which finishes before any
//    other subtree starts: so the top phase runs first, then the subtree that starts at position 0 alone,
//    then all other subtrees in parallel (they see positions 0..11 final, as the recursion would);
//  * pre-order node numbers come from subtree sizes (prefix sums); parent boxes are combined bottom-up.
#include "bvh.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <thread>

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

static inline hpt_bvh_bbox combine(const hpt_bvh_bbox &a, const hpt_bvh_bbox &b) {   // bvh.cpp:327-328
    return hpt_bvh_bbox{fmin_ref(a.minboundsx, b.minboundsx), fmin_ref(a.minboundsy, b.minboundsy),
                        fmin_ref(a.minboundsz, b.minboundsz), fmax_ref(a.maxboundsx, b.maxboundsx),
                        fmax_ref(a.maxboundsy, b.maxboundsy), fmax_ref(a.maxboundsz, b.maxboundsz)};
}

int build_threads() {
    int threads = (int)std::thread::hardware_concurrency();
    if (threads > 16) threads = 16;
    if (const char *e = getenv("HPT_BUILD_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 256) threads = v; }
    return threads < 1 ? 1 : threads;
}

// f(i) for i in [0, n), handed out one at a time
template <class F>
static void parallel_items(int threads, int n, F f) {
    if (threads <= 1 || n < 2) { for (int i = 0; i < n; i++) f(i); return; }
    std::atomic<int> next{0};
    auto run = [&]() { for (int i; (i = next.fetch_add(1)) < n;) f(i); };
    std::vector<std::thread> th;
    int extra = (threads < n ? threads : n) - 1;
    th.reserve(extra);
    for (int t = 0; t < extra; t++) th.emplace_back(run);
    run();
    for (auto &x : th) x.join();
}

// Longest axis of the centroid bounds (common.hpp:39-50) of [start, end); returns false when the centroids
// do not spread along it (bvh.cpp:174-193: no reorder then).
bool BoundingVolumeHierarchy::centroid_axis(int start, int end, int threads, int &axis, float &lo, double &span) const {
    const Item *it = items.data();
    const int T = threads > 1 ? threads : 1;
    std::vector<float> part((size_t)T * 6);
    std::vector<char> used((size_t)T, 0);
    parallel_chunks(T, end - start, [&](int t, long b, long e) {
        if (b >= e) return;
        float mn[3], mx[3];
        std::memcpy(mn, it[start + b].c, sizeof mn);
        std::memcpy(mx, it[start + b].c, sizeof mx);
        for (long i = start + b + 1; i < start + e; i++)
            for (int a = 0; a < 3; a++) { mn[a] = fmin_ref(mn[a], it[i].c[a]); mx[a] = fmax_ref(mx[a], it[i].c[a]); }
        float *p = part.data() + (size_t)t * 6;
        std::memcpy(p, mn, sizeof mn); std::memcpy(p + 3, mx, sizeof mx);
        used[t] = 1;
    });
    // min/max are exact, so folding the chunk results in chunk order gives the sequential result
    float cmn[3] = {0, 0, 0}, cmx[3] = {0, 0, 0};
    bool have = false;
    for (int t = 0; t < T; t++) {
        if (!used[t]) continue;
        const float *p = part.data() + (size_t)t * 6;
        if (!have) { std::memcpy(cmn, p, sizeof cmn); std::memcpy(cmx, p + 3, sizeof cmx); have = true; continue; }
        for (int a = 0; a < 3; a++) { cmn[a] = fmin_ref(cmn[a], p[a]); cmx[a] = fmax_ref(cmx[a], p[3 + a]); }
    }
    axis = 0;
    if ((cmx[1] - cmn[1]) > (cmx[0] - cmn[0])) axis = 1;
    if ((cmx[2] - cmn[2]) > (cmx[axis] - cmn[axis])) axis = 2;
    lo = cmn[axis];
    span = cmx[axis] - cmn[axis];
    return !(cmx[axis] == cmn[axis]);
}

// The partition step of a node of more than MAX_TRIANGLES_LEAF triangles: stable 12-bucket counting sort of
// [start, end) along the longest centroid axis (== bvh.cpp:196-220, 291-315).  `threads` share the node.
int BoundingVolumeHierarchy::partition_large(int start, int end, int threads) {
    int axis; float lo; double span;
    if (!centroid_axis(start, end, threads, axis, lo, span)) return axis;
    Item *it = items.data();
    Item *tmp = scratch.data();
    const int T = threads > 1 ? threads : 1;
    std::vector<int> count((size_t)T * NUM_BUCKETS, 0);
    parallel_chunks(T, end - start, [&](int t, long b, long e) {
        int *c = count.data() + (size_t)t * NUM_BUCKETS;
        for (long i = start + b; i < start + e; i++) {                       // bvh.cpp:212-220
            int k = (int)((float)NUM_BUCKETS * (it[i].c[axis] - lo) / span);
            if (k == NUM_BUCKETS) k--;
            bucket[i] = (unsigned char)k;
            c[k]++;
        }
    });
    // stable: bucket-major, then chunk order, then position inside the chunk
    std::vector<int> pos((size_t)T * NUM_BUCKETS);
    int run = start;
    for (int k = 0; k < NUM_BUCKETS; k++)
        for (int t = 0; t < T; t++) { pos[(size_t)t * NUM_BUCKETS + k] = run; run += count[(size_t)t * NUM_BUCKETS + k]; }
    parallel_chunks(T, end - start, [&](int t, long b, long e) {
        int *p = pos.data() + (size_t)t * NUM_BUCKETS;
        for (long i = start + b; i < start + e; i++) tmp[p[bucket[i]]++] = it[i];
    });
    parallel_chunks(T, end - start, [&](int, long b, long e) {
        if (e > b) std::memcpy(it + start + b, tmp + start + b, sizeof(Item) * (size_t)(e - b));
    });
    return axis;
}

// Depth-first build of [start, end) by one thread into `out` (local node indices, root = out's next node).
int BoundingVolumeHierarchy::build(Subtree &out, int start, int end, int level) {
    const int k = (int)out.bbox.size();
    out.bbox.emplace_back();
    out.info.push_back(hpt_bvh_info{0, 0, 0, 0});
    if (level > out.depth) out.depth = level;

    Item *it = items.data();
    float bmn[3], bmx[3];
    std::memcpy(bmn, it[start].mn, sizeof bmn);
    std::memcpy(bmx, it[start].mx, sizeof bmx);
    for (int i = start + 1; i < end; i++)
        for (int a = 0; a < 3; a++) { bmn[a] = fmin_ref(bmn[a], it[i].mn[a]); bmx[a] = fmax_ref(bmx[a], it[i].mx[a]); }

    const int n = end - start;
    auto make_leaf = [&]() {
        out.bbox[k] = hpt_bvh_bbox{bmn[0], bmn[1], bmn[2], bmx[0], bmx[1], bmx[2]};
        out.info[k].offset = start;          // leaves are created left to right, so the running count is `start`
        out.info[k].nPrimitives = n;
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
        int pos[NUM_BUCKETS], run = start;                                  // stable counting sort == bvh.cpp:291-315
        for (int b = 0; b < NUM_BUCKETS; b++) { pos[b] = run; run += count[b]; }
        Item *tmp = scratch.data();
        for (int i = start; i < end; i++) tmp[pos[bucket[i]]++] = it[i];
        std::memcpy(it + start, tmp + start, sizeof(Item) * (size_t)n);
    }
    const int c0 = build(out, start, mid, level + 1);
    const int c1 = build(out, mid, end, level + 1);
    out.bbox[k] = combine(out.bbox[c0], out.bbox[c1]);
    out.info[k].secondChildOffset = c1;
    out.info[k].axis = axis;
    return k;
}

bool BoundingVolumeHierarchy::buildTree() {
    built = false;
    bvh_bbox.clear(); bvh_info.clear(); order.clear(); depth = 0;
    const int n = (int)vtriangles.size();
    if (n < NUM_BUCKETS) return false;

    int threads = build_threads();
    // subtrees of at most task_size triangles are built by one thread each; > MAX_TRIANGLES_LEAF so that no
    // node of the top phase can become a leaf or read positions 0..11
    int task_size = n / (threads * 8);
    if (task_size < 8192) task_size = 8192;
    if (const char *e = getenv("HPT_BUILD_TASK_SIZE")) { int v = atoi(e); if (v > MAX_TRIANGLES_LEAF) task_size = v; }
    if (threads == 1) task_size = n;

    const bool verbose = getenv("HPT_BUILD_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hpt bvh] %-10s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    items.resize(n); scratch.resize(n); bucket.resize(n);
    parallel_chunks(threads, n, [&](int, long b, long e) {
        for (long i = b; i < e; i++) {
            const VectorTriangle &t = vtriangles[i];
            for (int a = 0; a < 3; a++) { items[i].mn[a] = t.bbox.minbounds[a]; items[i].mx[a] = t.bbox.maxbounds[a]; items[i].c[a] = t.centroid[a]; }
            items[i].src = (int32_t)i;
        }
    });

    lap("items");
    // ---- top phase: nodes above the task cut, level by level
    struct Top { int start, end, level, axis, child[2], size, index; };     // child >= 0: top node; < 0: task ~child
    struct Task { int start, end, level; Subtree sub; int index; };
    std::vector<Top> top;
    std::vector<Task> tasks;
    std::vector<int> frontier;
    if (n > task_size) { top.push_back(Top{0, n, 0, 0, {0, 0}, 0, 0}); frontier.push_back(0); }
    else tasks.push_back(Task{0, n, 0, Subtree(), 0});
    while (!frontier.empty()) {
        if ((int)frontier.size() >= threads)
            parallel_items(threads, (int)frontier.size(), [&](int i) { Top &t = top[frontier[i]]; t.axis = partition_large(t.start, t.end, 1); });
        else
            for (int f : frontier) top[f].axis = partition_large(top[f].start, top[f].end, threads);
        std::vector<int> next;
        for (int f : frontier) {
            const int s = top[f].start, e = top[f].end, mid = (s + e) / 2, lvl = top[f].level + 1;
            const int rs[2] = {s, mid}, re[2] = {mid, e};
            for (int c = 0; c < 2; c++) {
                if (re[c] - rs[c] > task_size) {
                    top[f].child[c] = (int)top.size();
                    next.push_back((int)top.size());
                    top.push_back(Top{rs[c], re[c], lvl, 0, {0, 0}, 0, 0});
                } else {
                    top[f].child[c] = ~(int)tasks.size();
                    tasks.push_back(Task{rs[c], re[c], lvl, Subtree(), 0});
                }
            }
        }
        frontier.swap(next);
    }

    lap("top");
    // ---- subtrees: the one holding positions 0..11 first and alone, then the rest in parallel
    auto run_task = [&](Task &t) {
        t.sub.bbox.reserve(2 * (size_t)(t.end - t.start) / 8 + 16);
        t.sub.info.reserve(2 * (size_t)(t.end - t.start) / 8 + 16);
        build(t.sub, t.start, t.end, t.level);
    };
    int first = -1;
    for (size_t i = 0; i < tasks.size(); i++) if (tasks[i].start == 0) first = (int)i;
    run_task(tasks[first]);
    parallel_items(threads, (int)tasks.size(), [&](int i) { if (i != first) run_task(tasks[i]); });

    lap("subtrees");
    // ---- pre-order numbering from subtree sizes, then assembly
    for (int i = (int)top.size() - 1; i >= 0; i--) {
        int sz = 1;
        for (int c = 0; c < 2; c++) sz += top[i].child[c] >= 0 ? top[top[i].child[c]].size : (int)tasks[~top[i].child[c]].sub.bbox.size();
        top[i].size = sz;
    }
    size_t total = top.empty() ? tasks[0].sub.bbox.size() : (size_t)top[0].size;
    bvh_bbox.resize(total); bvh_info.resize(total);
    for (const Task &t : tasks) if (t.sub.depth > depth) depth = t.sub.depth;
    if (!top.empty()) top[0].index = 0; else tasks[0].index = 0;
    for (size_t i = 0; i < top.size(); i++) {                               // parents are created before children
        int at = top[i].index + 1;
        for (int c = 0; c < 2; c++) {
            const int ch = top[i].child[c];
            if (ch >= 0) { top[ch].index = at; at += top[ch].size; }
            else { tasks[~ch].index = at; at += (int)tasks[~ch].sub.bbox.size(); }
        }
    }
    parallel_items(threads, (int)tasks.size(), [&](int i) {
        Task &t = tasks[i];
        const size_t m = t.sub.bbox.size();
        std::memcpy(bvh_bbox.data() + t.index, t.sub.bbox.data(), sizeof(hpt_bvh_bbox) * m);
        for (size_t j = 0; j < m; j++) {
            hpt_bvh_info v = t.sub.info[j];
            if (v.nPrimitives == 0) v.secondChildOffset += t.index;         // interior: local -> global index
            bvh_info[t.index + j] = v;
        }
        t.sub = Subtree();
    });
    // boxes and links of the top nodes, children before parents
    for (int i = (int)top.size() - 1; i >= 0; i--) {
        const int k = top[i].index;
        const int c1 = top[i].child[1] >= 0 ? top[top[i].child[1]].index : tasks[~top[i].child[1]].index;
        bvh_bbox[k] = combine(bvh_bbox[k + 1], bvh_bbox[c1]);
        bvh_info[k] = hpt_bvh_info{0, 0, 0, 0};
        bvh_info[k].secondChildOffset = c1;
        bvh_info[k].axis = top[i].axis;
        if (top[i].level > depth) depth = top[i].level;
    }

    lap("assembly");
    order.resize(n);
    std::vector<VectorTriangle> sorted(n);
    parallel_chunks(threads, n, [&](int, long b, long e) {
        for (long i = b; i < e; i++) { order[i] = items[i].src; sorted[i] = vtriangles[items[i].src]; }
    });
    vtriangles.swap(sorted);
    items.clear(); items.shrink_to_fit(); scratch.clear(); scratch.shrink_to_fit(); bucket.clear(); bucket.shrink_to_fit();
    lap("reorder");
    built = true;
    return true;
}

}  // namespace hpt
