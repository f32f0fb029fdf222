// tuning.hpp -- the library's tuning / test switches: one process-wide table of 64-bit values behind hnswgpu_set_tuning
// (include/hnswgpu.h lists the keys).  No getenv on any search path: the six environment variables the header documents
// are read ONCE, when the library is loaded (engine.hip: TuneInit); everything else is set through the call.  An unset key
// reads as the default its use site names.
#pragma once
#include <atomic>
#include <cstdint>

#include "../../include/hnswgpu.h"

namespace hg {

constexpr int64_t kTuneUnset = INT64_MIN;
extern std::atomic<int64_t> g_tune[HNSWGPU_TUNE_COUNT];

inline int64_t tune(int key, int64_t dflt) {
    const int64_t v = g_tune[key].load(std::memory_order_relaxed);
    return v == kTuneUnset ? dflt : v;
}

}  // namespace hg
