// persist.hip -- flat binary index file: the replacement for the reference's EDN dump
// (src/hnsw/helper/index_io.clj:10-80, one pr-str of every node: 492.9 MB for 31k vectors, README.md:22).
// Layout (little endian): 64-byte header, then the sections that the flags announce, each a plain array that
// can be mmap'ed by a JVM (FileChannel.map) or uploaded with one DMA:
//   base      n * dim        float32   (unpadded rows)
//   graph     levels[n] int32, l0_adj[n * M0] int32, up_off[n + 1] int64, up_adj[up_blocks * M] int32
//   ivf       centroids[nlist * dim] float32, list_off[nlist + 1] int64, list_ids[n] int32
// The String-id table stays with the caller (the Python mirror writes it next to this file).
#include <stdio.h>
#include <string.h>

#include <memory>

#include "engine.hpp"

namespace hg {

struct FileHeader {
    char magic[8];  // "HNSWGPU1"
    int32_t version;
    int32_t metric;
    int64_t n;
    int32_t dim;
    int32_t flags;  // bit 0: graph, bit 1: ivf
    int32_t M, M0, entry, max_level;
    int64_t up_blocks;
    int32_t nlist;
    int32_t reserved;
};
static_assert(sizeof(FileHeader) == 64, "header is 64 bytes");

struct FileCloser {
    void operator()(FILE *f) const {
        if (f) fclose(f);
    }
};

template <class T>
static int put(FILE *f, const T *p, size_t cnt) {
    HG_REQUIRE(cnt == 0 || fwrite(p, sizeof(T), cnt, f) == cnt, HNSWGPU_EINVAL, "short write");
    return 0;
}
template <class T>
static int get(FILE *f, std::vector<T> &v, size_t cnt) {
    v.resize(cnt);
    HG_REQUIRE(cnt == 0 || fread(v.data(), sizeof(T), cnt, f) == cnt, HNSWGPU_EINVAL, "index file is truncated");
    return 0;
}

}  // namespace hg

using namespace hg;

extern "C" {

int hnswgpu_save(hnswgpu_index *idx, const char *path) {
    HG_REQUIRE(idx && path, HNSWGPU_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    std::unique_ptr<FILE, FileCloser> f(fopen(path, "wb"));
    HG_REQUIRE(f, HNSWGPU_EINVAL, "cannot open %s for writing", path);
    FileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "HNSWGPU1", 8);
    h.version = 1;
    h.metric = idx->metric;
    h.n = idx->n;
    h.dim = idx->dim;
    h.flags = (idx->has_graph ? 1 : 0) | (idx->nlist > 0 ? 2 : 0);
    h.M = idx->M;
    h.M0 = idx->M0;
    h.entry = idx->entry;
    h.max_level = idx->max_level;
    h.up_blocks = idx->up_blocks;
    h.nlist = idx->nlist;
    HG_TRY(put(f.get(), &h, 1));
    if (idx->n > 0) {
        std::vector<float> base(static_cast<size_t>(idx->n) * idx->dim);
        HG_TRY(begin_call(idx, idx->stream));
        HG_HIP(hipMemcpy2DAsync(base.data(), sizeof(float) * idx->dim, idx->d_base, sizeof(float) * idx->ld,
                                sizeof(float) * idx->dim, idx->n, hipMemcpyDeviceToHost, idx->stream));
        HG_HIP(hipStreamSynchronize(idx->stream));
        HG_TRY(put(f.get(), base.data(), base.size()));
    }
    if (idx->has_graph) {
        HG_TRY(put(f.get(), idx->h_levels.data(), idx->h_levels.size()));
        HG_TRY(put(f.get(), idx->h_l0.data(), idx->h_l0.size()));
        HG_TRY(put(f.get(), idx->h_upoff.data(), idx->h_upoff.size()));
        HG_TRY(put(f.get(), idx->h_upadj.data(), idx->h_upadj.size()));
    }
    if (idx->nlist > 0) {
        HG_TRY(put(f.get(), idx->h_cent.data(), idx->h_cent.size()));
        HG_TRY(put(f.get(), idx->h_listoff.data(), idx->h_listoff.size()));
        HG_TRY(put(f.get(), idx->h_listids.data(), idx->h_listids.size()));
    }
    HG_REQUIRE(fflush(f.get()) == 0, HNSWGPU_EINVAL, "flush failed");
    return 0;
}

int hnswgpu_load(const char *path, int32_t device, hnswgpu_index **out) {
    HG_REQUIRE(path && out, HNSWGPU_EINVAL, "null argument");
    *out = nullptr;
    std::unique_ptr<FILE, FileCloser> f(fopen(path, "rb"));
    HG_REQUIRE(f, HNSWGPU_EINVAL, "cannot open %s", path);
    FileHeader h;
    HG_REQUIRE(fread(&h, sizeof(h), 1, f.get()) == 1, HNSWGPU_EINVAL, "index file is truncated");
    HG_REQUIRE(memcmp(h.magic, "HNSWGPU1", 8) == 0 && h.version == 1, HNSWGPU_EINVAL, "%s is not an HNSWGPU1 index file", path);
    HG_REQUIRE(h.n >= 0 && h.dim >= 1 && h.up_blocks >= 0 && h.nlist >= 0 && h.M >= 0 && h.M0 >= 0, HNSWGPU_EINVAL,
               "corrupt header");
    std::vector<float> base;
    HG_TRY(get(f.get(), base, static_cast<size_t>(h.n) * h.dim));
    hnswgpu_index *idx = nullptr;
    HG_TRY(hnswgpu_create(base.data(), h.n, h.dim, h.metric, device, &idx));
    int rc = [&]() -> int {
        if (h.flags & 1) {
            std::vector<int32_t> levels, l0, up;
            std::vector<int64_t> upoff;
            HG_TRY(get(f.get(), levels, static_cast<size_t>(h.n)));
            HG_TRY(get(f.get(), l0, static_cast<size_t>(h.n) * h.M0));
            HG_TRY(get(f.get(), upoff, static_cast<size_t>(h.n) + 1));
            HG_TRY(get(f.get(), up, static_cast<size_t>(h.up_blocks) * h.M));
            // set_graph re-validates every edge: a damaged file is rejected, never traversed
            HG_TRY(hnswgpu_set_graph(idx, levels.data(), l0.data(), h.M0, upoff.data(), up.data(), h.M, h.entry,
                                     h.max_level));
        }
        if (h.flags & 2) {
            std::vector<float> cent;
            std::vector<int64_t> off;
            std::vector<int32_t> ids;
            HG_TRY(get(f.get(), cent, static_cast<size_t>(h.nlist) * h.dim));
            HG_TRY(get(f.get(), off, static_cast<size_t>(h.nlist) + 1));
            HG_TRY(get(f.get(), ids, static_cast<size_t>(h.n)));
            HG_TRY(hnswgpu_set_ivf(idx, cent.data(), h.nlist, off.data(), ids.data()));
        }
        return 0;
    }();
    if (rc != 0) {
        hnswgpu_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

}  // extern "C"
