// host/obj_reader.cpp — OBJ reader (see obj_reader.hpp): memory-mapped file, parsed in parallel.
// The reference tokenises with fscanf("%s") and prints every vertex
// (reference pathtracer/obj_reader.cpp:20-75); this one maps the file, prints nothing, and yields the same
// vertex list on files of `v x y z` and three-index `f a b c` lines, faces before or after the vertices they
// name (the reference resolves indices after the whole file is read, obj_reader.cpp:77-81).  Two deliberate
// differences: an `f` line with more than three indices is fan-triangulated here, while the
// reference takes its first three indices and drops the rest of the line (obj_reader.cpp:45: one triangle per quad);
// `a/b/c` tokens and negative indices, which the reference rejects or misreads, are accepted.  Numbers go through
// strtof/strtol, as fscanf's conversions do.
//
// Three parallel passes over line-aligned chunks (threads = build_threads()):
//   1. count `v` lines and output vertices (3 per fan triangle) per chunk  -> prefix sums
//   2. parse the `v` lines into the vertex array at the chunk's offset
//   3. parse the `f` lines; a positive index names a vertex of the whole file, a negative one counts back from
//      the vertices defined before the face's line (the OBJ rule), which the chunk's running vertex count gives.
#include "obj_reader.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace hpt {
namespace {

inline bool blank(char c) { return c == ' ' || c == '\t'; }
inline bool eol(char c) { return c == '\n' || c == '\r' || c == '\0'; }

// One line [p, e): e points at '\n' or at the end of the data.  The data is followed by at least one readable
// '\0' (see load_obj), so strtof/strtol never run off the end; they cannot cross a '\n' either.
struct Line { const char *p, *e; };

inline bool is_kind(const Line &l, char k, const char *&body) {
    const char *p = l.p;
    while (p < l.e && blank(*p)) p++;
    if (p + 1 < l.e && p[0] == k && blank(p[1])) { body = p + 1; return true; }
    return false;
}

// vertices a face line emits (3 per fan triangle), or -1 when it has fewer than 3 index tokens
long face_output(const char *q, const char *e) {
    long n = 0;
    for (;;) {
        while (q < e && blank(*q)) q++;
        if (q >= e || eol(*q) || *q == '#') break;
        n++;
        while (q < e && !blank(*q) && !eol(*q)) q++;
    }
    return n < 3 ? -1 : 3 * (n - 2);
}

template <class F>
void for_lines(const char *b, const char *e, F f) {
    while (b < e) {
        const char *nl = (const char *)std::memchr(b, '\n', (size_t)(e - b));
        const char *le = nl ? nl : e;
        f(Line{b, le});
        b = nl ? nl + 1 : e;
    }
}

}  // namespace

bool load_obj(const char *path, std::vector<Vec3f> &out_vertices) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
    const size_t size = (size_t)st.st_size;
    if (size == 0) { ::close(fd); return true; }
    // A mapping is zero-filled up to the page boundary; when the file ends exactly on one there is no '\0'
    // behind the last number, so such files (and tiny ones) are read into a terminated buffer instead.
    const size_t page = (size_t)::sysconf(_SC_PAGESIZE);
    std::string owned;
    const char *data = nullptr;
    void *map = MAP_FAILED;
    if (size % page != 0 && size > (1u << 16)) {
        map = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map != MAP_FAILED) data = (const char *)map;
    }
    if (!data) {
        owned.resize(size);
        size_t got = 0;
        while (got < size) {
            ssize_t r = ::read(fd, &owned[got], size - got);
            if (r <= 0) { ::close(fd); return false; }
            got += (size_t)r;
        }
        data = owned.c_str();                               // c_str(): '\0' behind the last byte
    }
    ::close(fd);

    int T = build_threads();
    if (size < (1u << 20)) T = 1;
    // chunk t = [cut[t], cut[t+1]), cut on line starts
    std::vector<size_t> cut((size_t)T + 1);
    cut[0] = 0; cut[T] = size;
    for (int t = 1; t < T; t++) {
        size_t c = size * (size_t)t / (size_t)T;
        if (c < cut[t - 1]) c = cut[t - 1];
        const char *nl = (const char *)std::memchr(data + c, '\n', size - c);
        cut[t] = nl ? (size_t)(nl - data) + 1 : size;
    }

    // pass 1: counts
    std::vector<long> nv((size_t)T, 0), nout((size_t)T, 0);
    std::atomic<bool> ok{true};
    parallel_chunks(T, T, [&](int, long tb, long te) {
        for (long t = tb; t < te; t++) {
            long v = 0, o = 0;
            for_lines(data + cut[t], data + cut[t + 1], [&](const Line &l) {
                const char *body;
                if (is_kind(l, 'v', body)) v++;
                else if (is_kind(l, 'f', body)) {
                    long k = face_output(body, l.e);
                    if (k < 0) ok = false; else o += k;
                }
            });
            nv[t] = v; nout[t] = o;
        }
    });
    if (!ok) { if (map != MAP_FAILED) ::munmap(map, size); return false; }
    std::vector<long> v0((size_t)T + 1, 0), o0((size_t)T + 1, 0);
    for (int t = 0; t < T; t++) { v0[t + 1] = v0[t] + nv[t]; o0[t + 1] = o0[t] + nout[t]; }

    // pass 2: vertices
    std::vector<Vec3f> verts((size_t)v0[T]);
    parallel_chunks(T, T, [&](int, long tb, long te) {
        for (long t = tb; t < te; t++) {
            long at = v0[t];
            for_lines(data + cut[t], data + cut[t + 1], [&](const Line &l) {
                const char *body;
                if (!is_kind(l, 'v', body)) return;
                Vec3f v;
                const char *q = body;
                for (int k = 0; k < 3; k++) {               // a missing coordinate reads as 0 (strtof would skip the '\n')
                    while (q < l.e && blank(*q)) q++;
                    if (q >= l.e || eol(*q)) break;
                    char *end;
                    v[k] = std::strtof(q, &end);
                    if (end == q) break;
                    q = end;
                }
                verts[(size_t)at++] = v;
            });
        }
    });

    // pass 3: faces
    const size_t base = out_vertices.size();
    out_vertices.resize(base + (size_t)o0[T]);
    parallel_chunks(T, T, [&](int, long tb, long te) {
        std::vector<long> face;
        for (long t = tb; t < te; t++) {
            long seen = v0[t];                              // vertices defined before the current line
            const long total = v0[T];
            size_t at = base + (size_t)o0[t];
            for_lines(data + cut[t], data + cut[t + 1], [&](const Line &l) {
                const char *body;
                if (is_kind(l, 'v', body)) { seen++; return; }
                if (!is_kind(l, 'f', body)) return;
                face.clear();
                const char *q = body;
                for (;;) {
                    while (q < l.e && blank(*q)) q++;
                    if (q >= l.e || eol(*q) || *q == '#') break;
                    char *end;
                    long idx = std::strtol(q, &end, 10);
                    if (end == q) { ok = false; break; }
                    const char *e2 = end;
                    while (e2 < l.e && !blank(*e2) && !eol(*e2)) e2++;          // skip /vt/vn
                    q = e2;
                    if (idx < 0) { idx = seen + 1 + idx; if (idx < 1) { ok = false; break; } }
                    if (idx < 1 || idx > total) { ok = false; break; }
                    face.push_back(idx - 1);
                }
                if (!ok) return;
                for (size_t k = 2; k < face.size(); k++) {
                    out_vertices[at++] = verts[(size_t)face[0]];
                    out_vertices[at++] = verts[(size_t)face[k - 1]];
                    out_vertices[at++] = verts[(size_t)face[k]];
                }
            });
        }
    });
    if (map != MAP_FAILED) ::munmap(map, size);
    if (!ok) { out_vertices.resize(base); return false; }
    return true;
}

}  // namespace hpt
