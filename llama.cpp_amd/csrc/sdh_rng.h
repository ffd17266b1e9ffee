// Host-side random draws of the scan, bit-identical to the standard library forms the reference builder uses.
#pragma once
#include <cmath>
#include <cstdint>
#include <random>

namespace sdh {

// n draws of std::normal_distribution<float>(0, 1) from a fresh distribution object over g — what eval_x and
// the FFN proxy do per pass (tools/seeddelta-build/sd_eval.cpp:1084-1087, sd_ffn_proxy.cpp:77-80) — bit for
// bit, leaving g in the same state. libstdc++ (bits/random.tcc) implements it as the Marsaglia polar method
// over generate_canonical<float, 24>, one 32-bit draw per canonical value; the library template evaluates two
// long-double logarithms per canonical value when its constants are not folded (this compiler: 65 ns per
// normal), the same arithmetic written out takes 12 ns. tests/test_scan_host.py compares the two through
// sdh_test_normals on every build.
inline void fast_normals(std::mt19937 & g, int64_t n, float * out) {
    auto canon = [&]() -> float {
        float r = (float) (uint32_t) g() / 4294967296.0f;
        if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
        return r;
    };
    int64_t i = 0;
    while (i < n) {
        float x, y, r2;
        do {
            x = (float) ((double) (2.0f * canon()) - 1.0);
            y = (float) ((double) (2.0f * canon()) - 1.0);
            r2 = x * x + y * y;
        } while (r2 > 1.0f || r2 == 0.0f);
        const float mult = std::sqrt(-2 * std::log(r2) / r2);
        out[i++] = (y * mult) * 1.0f + 0.0f;
        if (i < n) out[i++] = (x * mult) * 1.0f + 0.0f;      // the distribution's saved value
    }
}

} // namespace sdh
