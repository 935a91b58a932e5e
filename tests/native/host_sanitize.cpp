// Driver for tests/test_host.py::test_host_code_under_sanitizers: the parallel BVH builder and the parallel
// OBJ reader on a generated scene, built with -fsanitize=thread and -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bvh.hpp"
#include "obj_reader.hpp"

using namespace hpt;

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int n = std::atoi(argv[1]);
    const char *obj = argv[2];
    srand(3);
    auto r = []() { return (float)rand() / (float)RAND_MAX * 200.0f - 100.0f; };
    std::vector<VectorTriangle> vt;
    FILE *f = std::fopen(obj, "w");
    if (!f) return 2;
    for (int i = 0; i < n; i++) {
        Vec3f a(r(), r(), r()), b(a[0] + r() * 0.02f, a[1] + r() * 0.02f, a[2] + r() * 0.02f), c(a[0] + r() * 0.02f, a[1] + r() * 0.02f, a[2] + r() * 0.02f);
        vt.push_back(VectorTriangle(a, b, c));
        std::fprintf(f, "v %.9g %.9g %.9g\nv %.9g %.9g %.9g\nv %.9g %.9g %.9g\nf -3 -2 -1\n", a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2]);
    }
    std::fclose(f);
    BoundingVolumeHierarchy bvh(vt);
    if (!bvh.buildTree()) return 1;
    std::vector<Vec3f> v;
    if (!load_obj(obj, v)) return 1;
    if ((int)v.size() != 3 * n) return 1;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            if (v[3 * (size_t)i][k] != vt[i].v1[k]) return 1;
    std::printf("ok %d nodes, depth %d, %zu vertices\n", bvh.get_n_nodes(), bvh.get_depth(), v.size());
    return 0;
}
