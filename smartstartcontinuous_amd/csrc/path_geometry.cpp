// path_geometry.cpp -- host-side plan geometry of the navigator: path_shortcutter (smartstart/utilities/numerical.py:
// 189-246) in native code.  It runs once per SmartStart plan on the host (a 300-state path has ~45 000 candidate
// shortcuts); in numpy it was 1.5-6.5 ms per plan -- with eight plans per selection the serial part of the vectorised
// SmartStart loop (8 of 16 ms per chunk at 65 536 envs).  No GPU work here; the arithmetic is the reference's fp64
// expression evaluated in the same order (no contraction into FMAs), so every shortcut decision is the numpy one.
#include <stdint.h>
#include <math.h>

#include <vector>

#include "ssc_host.h"

#pragma clang fp contract(off)

namespace {

struct Interval { int32_t start, end; };

// numerical.py:116-124: sqrt(sum_k ((x_k - y_k) / radii_k)^2), summed in index order like np.sum over a short last axis
inline double ell_distance(const double *x, const double *y, const double *radii, int d) {
    double s = 0.0;
    for (int k = 0; k < d; ++k) {
        const double t = (x[k] - y[k]) / radii[k];
        s += t * t;
    }
    return sqrt(s);
}

}  // namespace

extern "C" int ssc_path_shortcut(const double *path, int32_t n, int32_t d, const double *radii, double theta, uint8_t *keep,
                                 int32_t *n_kept) {
    SSC_REQUIRE(n >= 0 && d >= 1 && d <= SSC_MAX_STATE, "ssc_path_shortcut: n = %d, d = %d", n, d);
    SSC_REQUIRE((n == 0 || (path && keep)) && radii, "ssc_path_shortcut: NULL pointer");
    for (int k = 0; k < d; ++k) SSC_REQUIRE(radii[k] > 0.0, "ssc_path_shortcut: radii must be positive (numerical.py:112-113)");
    for (int i = 0; i < n; ++i) keep[i] = 1;
    // candidate shortcuts (i, j >= i + 2) with d(path_i, path_j) <= theta, ordered by end j, equal ends by start i: what
    // sorted(np.where(np.triu(dist <= theta, k=2)) pairs, key=end) gives (row-major pairs, stable sort)  (:226-241, :194)
    std::vector<Interval> acts;
    for (int j = 2; j < n; ++j)
        for (int i = 0; i + 2 <= j; ++i)
            if (ell_distance(path + (int64_t)i * d, path + (int64_t)j * d, radii, d) <= theta) acts.push_back({i, j});
    if (!acts.empty()) {
        // length_weighted_activities_solver (:189-222) with sub_extra = 1: one table row per distinct end time; the first
        // interval seeds the table WITHOUT sub_extra; `inc >= best` lets a later interval replace an equal earlier one
        const int sub_extra = 1;
        std::vector<int32_t> ends{0, acts[0].end}, best{0, acts[0].end - acts[0].start}, back{0, 0};
        std::vector<int32_t> take{-1, 0};                       // index into acts, -1: none
        for (size_t a = 1; a < acts.size(); ++a) {
            const Interval &iv = acts[a];
            // bisect_right(ends, start) - 1: last row whose end <= start
            int lo = 0, hi = (int)ends.size();
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (iv.start < ends[mid]) hi = mid; else lo = mid + 1;
            }
            const int j = lo - 1;
            const int32_t inc = best[j] + (iv.end - iv.start - sub_extra);
            if (iv.end == ends.back()) {
                if (inc >= best.back()) { best.back() = inc; take.back() = (int32_t)a; back.back() = j; }
            } else {
                const int32_t prev = best.back();
                ends.push_back(iv.end);
                if (inc >= prev) { best.push_back(inc); take.push_back((int32_t)a); back.push_back(j); }
                else { best.push_back(prev); take.push_back(-1); back.push_back((int32_t)ends.size() - 2); }
            }
        }
        for (int i = (int)ends.size() - 1;;) {
            if (take[i] >= 0)
                for (int k = acts[take[i]].start + 1; k < acts[take[i]].end; ++k) keep[k] = 0;   // drop the interior (:242-245)
            if (back[i] == i) break;
            i = back[i];
        }
    }
    if (n_kept) {
        int32_t c = 0;
        for (int i = 0; i < n; ++i) c += keep[i];
        *n_kept = c;
    }
    return SSC_OK;
}
