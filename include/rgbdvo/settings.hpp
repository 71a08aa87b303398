// Options of the rgbdvo front-end: same string keys and default values as the reference
// (rgbdvo/settings.hpp:8-52), parsed into the C ABI's rvo_params (stoi/stod like rgbdvo/helper.cpp:13-59).
#pragma once
#include <cstdio>
#include <map>
#include <string>

#include "../rvo.h"

namespace rgbdvo {

typedef std::map<std::string, std::string> Options;

namespace detail {
inline Options options_from_params(const rvo_params& p) {
  auto d = [](double v) { char b[64]; snprintf(b, sizeof b, "%.17g", v); return std::string(b); };
  return Options{{"reference-frame-change-threshold", std::to_string(p.reference_frame_change_threshold)},
                 {"feature-window-size", std::to_string(p.feature_window_size)},
                 {"num-pyramid-levels", std::to_string(p.num_pyramid_levels)},
                 {"fast-init-threshold", std::to_string(p.fast_init_threshold)},
                 {"fast-threshold-min", std::to_string(p.fast_threshold_min)},
                 {"fast-threshold-max", std::to_string(p.fast_threshold_max)},
                 {"fast-adaptive-gain", d(p.fast_adaptive_gain)},
                 {"target-pixels-per-feature", std::to_string(p.target_pixels_per_feature)},
                 {"grid-width", std::to_string(p.grid_width)},
                 {"grid-height", std::to_string(p.grid_height)},
                 {"max-keypoints-per-grid", std::to_string(p.max_keypoints_per_grid)},
                 {"inlier-max-reprojection-error", d(p.inlier_max_reprojection_error)},
                 {"clique-inlier-threshold", d(p.clique_inlier_threshold)},
                 {"min-feature-for-estimate", std::to_string(p.min_features_for_estimate)},
                 {"max-mean-reprojection-error", d(p.max_mean_reprojection_error)},
                 {"feature-search-window", d(p.feature_search_window)}};
}

// Missing keys keep the defaults in `p` and are reported on stderr like Option::Get* does (helper.cpp:17).
inline void params_from_options(const Options& o, rvo_params* p) {
  auto geti = [&](const char* k, int32_t* v) { auto it = o.find(k); if (it == o.end()) { fprintf(stderr, "Option %s not specified!\n", k); return; } *v = std::stoi(it->second); };
  auto getd = [&](const char* k, double* v) { auto it = o.find(k); if (it == o.end()) { fprintf(stderr, "Option %s not specified!\n", k); return; } *v = std::stod(it->second); };
  geti("reference-frame-change-threshold", &p->reference_frame_change_threshold);
  geti("feature-window-size", &p->feature_window_size);
  geti("num-pyramid-levels", &p->num_pyramid_levels);
  geti("fast-init-threshold", &p->fast_init_threshold);
  geti("fast-threshold-min", &p->fast_threshold_min);
  geti("fast-threshold-max", &p->fast_threshold_max);
  getd("fast-adaptive-gain", &p->fast_adaptive_gain);
  geti("target-pixels-per-feature", &p->target_pixels_per_feature);
  geti("grid-width", &p->grid_width);
  geti("grid-height", &p->grid_height);
  geti("max-keypoints-per-grid", &p->max_keypoints_per_grid);
  getd("inlier-max-reprojection-error", &p->inlier_max_reprojection_error);
  getd("clique-inlier-threshold", &p->clique_inlier_threshold);
  geti("min-feature-for-estimate", &p->min_features_for_estimate);
  getd("max-mean-reprojection-error", &p->max_mean_reprojection_error);
  getd("feature-search-window", &p->feature_search_window);
}
}  // namespace detail

inline Options GetRGBDVODefaultOptions() {  // rgbdvo/settings.hpp:10-30
  rvo_params p;
  rvo_default_vo_params(&p);
  return detail::options_from_params(p);
}

inline Options GetFrameMatcherDefaultOptions() {  // rgbdvo/settings.hpp:32-52
  rvo_params p;
  rvo_default_fm_params(&p);
  return detail::options_from_params(p);
}

}  // namespace rgbdvo
