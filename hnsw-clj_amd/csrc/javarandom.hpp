// javarandom.hpp -- java.util.Random as the JDK documents it (48-bit LCG).  The reference seeds
// k-means++ with (Random. 42) (src/hnsw/ann/partition/ivf_flat.clj:36) and draws HNSW levels from
// (Random.) (src/hnsw/ultra_fast.clj:139-147); using the same generator keeps a JVM user's seeds
// meaningful on this engine.
#pragma once
#include <stdint.h>

namespace hg {

class JavaRandom {
   public:
    explicit JavaRandom(int64_t seed) : seed_((static_cast<uint64_t>(seed) ^ 0x5DEECE66DULL) & kMask) {}
    int32_t next(int bits) {
        seed_ = (seed_ * 0x5DEECE66DULL + 0xBULL) & kMask;
        return static_cast<int32_t>(static_cast<int64_t>(seed_ >> (48 - bits)));
    }
    int32_t next_int(int32_t bound) {
        int32_t r = next(31);
        int32_t m = bound - 1;
        if ((bound & m) == 0) return static_cast<int32_t>((static_cast<int64_t>(bound) * r) >> 31);
        for (int32_t u = r;
             static_cast<int32_t>(static_cast<uint32_t>(u) - static_cast<uint32_t>(r = u % bound) +
                                  static_cast<uint32_t>(m)) < 0;
             u = next(31)) {
        }
        return r;
    }
    double next_double() {
        int64_t hi = next(26), lo = next(27);
        return static_cast<double>((hi << 27) + lo) * 0x1.0p-53;
    }

   private:
    static constexpr uint64_t kMask = (1ULL << 48) - 1;
    uint64_t seed_;
};

}  // namespace hg
