// Host-only arithmetic of the GEMM sampling hook (gemm.hip, uniir_gemm_timing_on): which sampled launches were taken while another
// stream's GEMMs shared the device.  No HIP calls in here -- the event times arrive as plain floats -- so the rule has a CPU test
// (tests/test_host.py through uniir_gemm_timing_filter).  Round 5 lost a whole GPU test record to this rule (a fixed 3-ms merge gap
// swallowed every sample of a millisecond-long step and the read returned nothing): it now scales with the samples' own duration and
// can never return an empty set.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#define GT_MERGE_DURS 2.0f     // windows closer than this many MEAN sampled-GEMM durations are one window (the LayerNorm / attention
                               // kernels between two GEMMs of the other tower share the device just the same) ...
#define GT_MERGE_CAP_MS 3.0f   // ... but never more than this (round 5's constant; the headline step's mean sample is ~1.3 ms)
#define GT_MIN_KEEP 8          // fewer survivors than this (or than all of them, if fewer were taken) is not a statistic: keep all

// windows: nwin x (begin, end) ms on one time axis, any order; samples: n x (begin, duration) ms on the same axis.
// merge_ms < 0: GT_MERGE_DURS x the mean sample duration, capped.  keep[i] = 1 for the samples that count.
// Returns the number kept; *fallback = 1 when the rule left fewer than min(GT_MIN_KEEP, n) and every sample was kept instead.
static inline int gt_filter_samples(const float* windows, int nwin, const float* samples, int n, float merge_ms, uint8_t* keep,
                                    int* fallback, int* left_out) {
    if (fallback) *fallback = 0;
    if (left_out) *left_out = 0;
    if (n <= 0) return 0;
    for (int i = 0; i < n; ++i) keep[i] = 1;
    if (nwin <= 0) return n;
    if (merge_ms < 0.f) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += samples[2 * i + 1];
        merge_ms = std::min(GT_MERGE_CAP_MS, (float)(GT_MERGE_DURS * s / n));
    }
    std::vector<std::pair<float, float>> w(nwin);
    for (int i = 0; i < nwin; ++i) w[i] = {windows[2 * i], windows[2 * i + 1]};
    std::sort(w.begin(), w.end());
    int nw = 0;
    for (int i = 0; i < nwin; ++i) {
        if (nw > 0 && w[i].first <= w[nw - 1].second + merge_ms) w[nw - 1].second = std::max(w[nw - 1].second, w[i].second);
        else w[nw++] = w[i];
    }
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const float a = samples[2 * i], b = a + samples[2 * i + 1];
        // first merged window that ends after the sample begins
        int lo = 0, hi = nw;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (w[mid].second > a) hi = mid; else lo = mid + 1;
        }
        const bool hit = lo < nw && w[lo].first < b;
        keep[i] = hit ? 0 : 1;
        kept += !hit;
    }
    if (kept < std::min(GT_MIN_KEEP, n)) {
        for (int i = 0; i < n; ++i) keep[i] = 1;
        if (fallback) *fallback = 1;
        if (left_out) *left_out = n - kept;
        return n;
    }
    if (left_out) *left_out = n - kept;
    return kept;
}
