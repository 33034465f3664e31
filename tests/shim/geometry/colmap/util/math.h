// TEST SHIM of colmap::Percentile (src/geometry/colmap/util/math.h)
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
namespace colmap {
template <typename T> T Percentile(const std::vector<T> &elems, const double p) {
    std::vector<T> s = elems;
    std::sort(s.begin(), s.end());
    const int idx = static_cast<int>(std::round(p / 100 * (s.size() - 1)));
    return s[std::max(0, std::min<int>(static_cast<int>(s.size()) - 1, idx))];
}
}
