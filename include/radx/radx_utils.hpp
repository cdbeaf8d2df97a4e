#pragma once
// radx_utils.hpp — the one helper of the reference's utility header the sort path uses
// (reference: include/radx/radx_utils.hpp:10-14).  Pipelines, SPIR-V readers, barriers and
// submit helpers have no HIP counterpart: stream order replaces them.
#include "radx_core.hpp"

namespace radx {
    // Number of tiles of `tile` elements needed to cover sz elements (ceil division, 0 for sz <= 0).
    template <class T = uint64_t>
    static inline T tiled(const T& sz, const T& tile) {
        if (sz <= T(0)) return T(0);
        return static_cast<T>(sz / tile + (sz % tile != T(0) ? 1 : 0));
    }
    template <class T>
    static inline std::size_t strided(const std::size_t& count) { return sizeof(T) * count; }
}
