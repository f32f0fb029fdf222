// persist.hip -- flat binary index file: the replacement for the reference's EDN dump
// (src/hnsw/helper/index_io.clj:10-80, one pr-str of every node: 492.9 MB for 31k vectors, README.md:22).
// Layout (little endian): 64-byte header, then the sections that the flags announce, each a plain array that
// can be mmap'ed by a JVM (FileChannel.map) or uploaded with one DMA:
//   base      n * dim        float32   (unpadded rows)
//   graph     levels[n] int32, l0_adj[n * M0] int32, up_off[n + 1] int64, up_adj[up_blocks * M] int32
//   ivf       centroids[nlist * dim] float32, list_off[nlist + 1] int64, list_ids[n] int32
// The String-id table stays with the caller (the Python mirror writes it next to this file).
#include <stdio.h>
#include <unistd.h>
#include <string.h>

#include <exception>
#include <memory>
#include <new>
#include <string>

#include <atomic>

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

// Body of hnswgpu_save, writing to an already opened temporary file.
static int save_to(hnswgpu_index *idx, FILE *f) {
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
    HG_TRY(put(f, &h, 1));
    if (idx->n > 0) {
        std::vector<float> base(static_cast<size_t>(idx->n) * idx->dim);
        HG_TRY(begin_call(idx, idx->stream));
        HG_HIP(hipMemcpy2DAsync(base.data(), sizeof(float) * idx->dim, idx->d_base, sizeof(float) * idx->ld,
                                sizeof(float) * idx->dim, idx->n, hipMemcpyDeviceToHost, idx->stream));
        HG_HIP(hipStreamSynchronize(idx->stream));
        HG_TRY(put(f, base.data(), base.size()));
    }
    if (idx->has_graph) {
        HG_TRY(put(f, idx->h_levels.data(), idx->h_levels.size()));
        HG_TRY(put(f, idx->h_l0.data(), idx->h_l0.size()));
        HG_TRY(put(f, idx->h_upoff.data(), idx->h_upoff.size()));
        HG_TRY(put(f, idx->h_upadj.data(), idx->h_upadj.size()));
    }
    if (idx->nlist > 0) {
        HG_TRY(put(f, idx->h_cent.data(), idx->h_cent.size()));
        HG_TRY(put(f, idx->h_listoff.data(), idx->h_listoff.size()));
        HG_TRY(put(f, idx->h_listids.data(), idx->h_listids.size()));
    }
    HG_REQUIRE(fflush(f) == 0, HNSWGPU_EINVAL, "flush failed");
    return 0;
}

// bytes a well-formed file with this header has; -1 if the header's sizes are out of range
static int64_t expected_file_size(const FileHeader &h) {
    if (h.n < 0 || h.n >= 2147483647LL || h.dim < 1 || h.dim > 3072) return -1;
    if (h.M < 0 || h.M > kMaxDeg || h.M0 < 0 || h.M0 > kMaxDeg) return -1;
    if (h.nlist < 0 || h.up_blocks < 0 || h.up_blocks > h.n * 31) return -1;   // a node has at most 30 upper levels
    if ((h.flags & 2) && (h.nlist < 1)) return -1;
    if ((h.flags & 1) && (h.M < 1 || h.M0 < 1)) return -1;
    int64_t sz = sizeof(FileHeader) + h.n * h.dim * 4;   // < 2^31 * 3072 * 4: no overflow in int64
    if (h.flags & 1) sz += h.n * 4 + h.n * h.M0 * 4 + (h.n + 1) * 8 + h.up_blocks * h.M * 4;
    if (h.flags & 2) sz += static_cast<int64_t>(h.nlist) * h.dim * 4 + (static_cast<int64_t>(h.nlist) + 1) * 8 + h.n * 4;
    return sz;
}

}  // namespace hg

using namespace hg;

extern "C" {

int hnswgpu_save(hnswgpu_index *idx, const char *path) {
    HG_REQUIRE(idx && path, HNSWGPU_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    // a shard of a larger IVF index (hnswgpu_set_ivf_shard) numbers its candidates by the WHOLE index's list lengths, which
    // the file format does not carry: loaded back it would silently be an ordinary index with another tie order
    HG_REQUIRE(idx->h_glistlen.empty(), HNSWGPU_ESTATE,
               "this handle holds a shard of a larger IVF index (hnswgpu_set_ivf_shard): save the whole index instead");
    HG_HIP(hipSetDevice(idx->device));
    // written beside the target -- under a name of this process's and this call's own: two savers of one path never share
    // a temporary -- flushed to the disk, and renamed over the target once complete and closed: a reader never sees half a
    // file, a crash never leaves a renamed file with missing contents, a failed save leaves the previous file untouched
    static std::atomic<unsigned> save_ctr{0};
    std::string tmp;
    try {
        tmp = std::string(path) + "." + std::to_string(static_cast<long long>(getpid())) + "." +
              std::to_string(save_ctr.fetch_add(1)) + ".tmp";
    } catch (...) {
        set_error("host allocation failed");
        return HNSWGPU_ENOMEM;
    }
    FILE *f = fopen(tmp.c_str(), "wb");
    HG_REQUIRE(f, HNSWGPU_EINVAL, "cannot open %s for writing", tmp.c_str());
    int rc;
    try {
        rc = save_to(idx, f);
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed while saving the index");
        rc = HNSWGPU_ENOMEM;
    }
    if (rc == 0 && (fflush(f) != 0 || fsync(fileno(f)) != 0)) {
        set_error("flushing %s to the disk failed", tmp.c_str());
        rc = HNSWGPU_EINVAL;
    }
    if (fclose(f) != 0 && rc == 0) {
        set_error("closing %s failed (disk full?)", tmp.c_str());
        rc = HNSWGPU_EINVAL;
    }
    if (rc == 0 && rename(tmp.c_str(), path) != 0) {
        set_error("cannot rename %s to %s", tmp.c_str(), path);
        rc = HNSWGPU_EINVAL;
    }
    if (rc != 0) (void)remove(tmp.c_str());
    return rc;
}

static int load_from(FILE *fp, const char *path, int32_t device, hnswgpu_index **out) {
    FileHeader h;
    HG_REQUIRE(fread(&h, sizeof(h), 1, fp) == 1, HNSWGPU_EINVAL, "index file is truncated");
    HG_REQUIRE(memcmp(h.magic, "HNSWGPU1", 8) == 0 && h.version == 1, HNSWGPU_EINVAL, "%s is not an HNSWGPU1 index file", path);
    // Nothing in the body is read before the header has been checked against the documented limits AND against the
    // size of the file: a damaged or hostile header can then neither overflow a size computation nor make a vector
    // allocate more than the file holds.
    const int64_t want = expected_file_size(h);
    HG_REQUIRE(want >= 0 && h.metric >= 0 && h.metric <= 2, HNSWGPU_EINVAL, "corrupt header");
    HG_REQUIRE(fseek(fp, 0, SEEK_END) == 0, HNSWGPU_EINVAL, "cannot seek in %s", path);
    const int64_t have = static_cast<int64_t>(ftell(fp));
    HG_REQUIRE(have == want, HNSWGPU_EINVAL, "index file is %s: %lld bytes, its header implies %lld",
               have < want ? "truncated" : "corrupt", (long long)have, (long long)want);
    HG_REQUIRE(fseek(fp, sizeof(FileHeader), SEEK_SET) == 0, HNSWGPU_EINVAL, "cannot seek in %s", path);
    std::vector<float> base;
    HG_TRY(get(fp, base, static_cast<size_t>(h.n) * h.dim));
    hnswgpu_index *idx = nullptr;
    HG_TRY(hnswgpu_create(base.data(), h.n, h.dim, h.metric, device, &idx));
    std::vector<float>().swap(base);
    auto body = [&]() -> int {
        if (h.flags & 1) {
            std::vector<int32_t> levels, l0, up;
            std::vector<int64_t> upoff;
            HG_TRY(get(fp, levels, static_cast<size_t>(h.n)));
            HG_TRY(get(fp, l0, static_cast<size_t>(h.n) * h.M0));
            HG_TRY(get(fp, upoff, static_cast<size_t>(h.n) + 1));
            HG_TRY(get(fp, up, static_cast<size_t>(h.up_blocks) * h.M));
            // set_graph indexes up_adj with up_off: the two must describe the same array before it may look
            HG_REQUIRE(upoff[static_cast<size_t>(h.n)] == h.up_blocks, HNSWGPU_EINVAL,
                       "corrupt index file: up_off ends at %lld, the header announces %lld upper-level blocks",
                       (long long)upoff[static_cast<size_t>(h.n)], (long long)h.up_blocks);
            // set_graph re-validates every level, offset and edge: a damaged file is rejected, never traversed
            HG_TRY(hnswgpu_set_graph(idx, levels.data(), l0.data(), h.M0, upoff.data(), up.data(), h.M, h.entry,
                                     h.max_level));
        }
        if (h.flags & 2) {
            std::vector<float> cent;
            std::vector<int64_t> off;
            std::vector<int32_t> ids;
            HG_TRY(get(fp, cent, static_cast<size_t>(h.nlist) * h.dim));
            HG_TRY(get(fp, off, static_cast<size_t>(h.nlist) + 1));
            HG_TRY(get(fp, ids, static_cast<size_t>(h.n)));
            HG_TRY(hnswgpu_set_ivf(idx, cent.data(), h.nlist, off.data(), ids.data()));
        }
        return 0;
    };
    int rc;
    try {
        rc = body();
    } catch (...) {  // the handle (and its device memory) must not outlive a failed load
        hnswgpu_destroy(idx);
        throw;
    }
    if (rc != 0) {
        hnswgpu_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

int hnswgpu_load(const char *path, int32_t device, hnswgpu_index **out) {
    HG_REQUIRE(path && out, HNSWGPU_EINVAL, "null argument");
    *out = nullptr;
    std::unique_ptr<FILE, FileCloser> f(fopen(path, "rb"));
    HG_REQUIRE(f, HNSWGPU_EINVAL, "cannot open %s", path);
    try {
        return load_from(f.get(), path, device, out);
    } catch (const std::bad_alloc &) {  // no exception crosses the C boundary
        set_error("host allocation failed while loading %s", path);
        return HNSWGPU_ENOMEM;
    } catch (const std::exception &e) {
        set_error("loading %s failed: %s", path, e.what());
        return HNSWGPU_EINVAL;
    }
}

}  // extern "C"
