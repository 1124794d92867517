// viamd_amd/csrc/vmd_eval_post.cpp - VIAMD's consumer-side post-processing of evaluated properties, behind the C ABI: compute_histogram,
// compute_histogram_masked, downsample_histogram, scale_histogram (/root/reference/src/main.cpp:139-261).  Pinned to the reference's own
// code bit for bit (tests/test_ref_pin.py, tests/native/ref_callsites.cpp): this is the formula the 1e-5 tolerance of g(r) goes through.
#include "vmd_eval_internal.h"

// ------------------------------------------------------------------------------------------------ consumer post-processing
// what VIAMD does with a distribution before plotting it (/root/reference/src/main.cpp:232-250)
extern "C" void vmd_downsample_histogram(float* dst_bins, int num_dst_bins, const float* src_bins, const float* src_weights,
                                         int num_src_bins) {
    const int factor = std::max(1, num_src_bins / std::max(1, num_dst_bins));
    for (int d = 0; d < num_dst_bins; ++d) {
        double bin = 0.0, weight = 0.0;
        for (int i = 0; i < factor; ++i) {
            const int s = d * factor + i;
            if (s >= num_src_bins) break;
            bin += src_bins[s];
            weight += src_weights ? src_weights[s] : 1.0;
        }
        dst_bins[d] = (float)(bin / weight);
    }
}

// temporal -> distribution as VIAMD builds it from the frame mask (/root/reference/src/main.cpp:172-230); y_range = the
// Histogram's y_min / y_max (:212-229; untouched when no frame is set, as there)
extern "C" void vmd_compute_histogram_masked_y(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                               int dim, const uint8_t* frame_mask, int num_frames, bool aggregate, float* y_range) {
    const int hdim = aggregate ? 1 : dim;
    std::fill(bins, bins + (size_t)hdim * num_bins, 0.0f);
    const float ext = range_max - range_min;
    const float inv = ext > 0.0f ? 1.0f / ext : 0.0f;
    std::vector<int> count(hdim, 0);
    bool any = false;
    for (int f = 0; f < num_frames; ++f) {
        if (!frame_mask[f]) continue;
        any = true;
        for (int i = 0; i < dim; ++i) {
            const float v = values[(size_t)f * dim + i];
            if (v < range_min || range_max < v) continue;
            const int b = std::min(std::max((int)(((v - range_min) * inv) * num_bins), 0), num_bins - 1);
            const int row = aggregate ? 0 : i;
            bins[(size_t)row * num_bins + b] += 1.0f;
            count[row] += 1;
        }
    }
    if (!any || dim <= 0) return;
    float lo = FLT_MAX, hi = -FLT_MAX;
    const float width = ext / num_bins;
    for (int r = 0; r < hdim; ++r) {
        const float scl = 1.0f / (width * count[r]);
        for (int j = 0; j < num_bins; ++j) {
            float& v = bins[(size_t)r * num_bins + j];
            v *= scl;
            lo = lo < v ? lo : v;          // MIN(min_bin, val) / MAX(max_bin, val) as the reference's macros order them: a NaN bin
            hi = hi > v ? hi : v;          // (a row without samples: 0 * inf) REPLACES the running value
        }
    }
    if (y_range) { y_range[0] = lo; y_range[1] = hi; }
}

extern "C" void vmd_compute_histogram_masked(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                             int dim, const uint8_t* frame_mask, int num_frames, bool aggregate) {
    vmd_compute_histogram_masked_y(bins, num_bins, range_min, range_max, values, dim, frame_mask, num_frames, aggregate, nullptr);
}

// the unmasked form (/root/reference/src/main.cpp:139-170): normalised by 1 / (bin width x samples inside the range)
extern "C" void vmd_compute_histogram(float* bins, int num_bins, float range_min, float range_max, const float* values, int num_values,
                                      float* bin_val_min, float* bin_val_max) {
    std::fill(bins, bins + num_bins, 0.0f);
    const float ext = range_max - range_min;
    const float inv = 1.0f / ext;
    int count = 0;
    for (int i = 0; i < num_values; ++i) {
        if (values[i] < range_min || range_max < values[i]) continue;
        const int b = std::min(std::max((int)(((values[i] - range_min) * inv) * num_bins), 0), num_bins - 1);
        bins[b] += 1.0f;
        count += 1;
    }
    if (count == 0) {
        if (bin_val_min) *bin_val_min = 0;
        if (bin_val_max) *bin_val_max = 0;
        return;
    }
    float lo = FLT_MAX, hi = -FLT_MAX;
    const float width = ext / num_bins;
    const float scl = 1.0f / (width * count);
    for (int i = 0; i < num_bins; ++i) {
        bins[i] *= scl;
        lo = lo < bins[i] ? lo : bins[i];
        hi = hi > bins[i] ? hi : bins[i];
    }
    if (bin_val_min) *bin_val_min = lo;
    if (bin_val_max) *bin_val_max = hi;
}

// bins[i] /= weights[i] where the weight is not zero (/root/reference/src/main.cpp:252-261)
extern "C" void vmd_scale_histogram(float* bins, const float* weights, int num_bins) {
    for (int i = 0; i < num_bins; ++i) if (weights[i]) bins[i] /= weights[i];
}
