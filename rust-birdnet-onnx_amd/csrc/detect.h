// Model-type detection from graph I/O shapes: the rules of the reference's
// src/detection.rs:15-174 and the ModelType constants of src/types.rs:14-44.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/birdnet_hip.h"

namespace bn {
// override_type < 0 means auto-detection.  On failure returns false and fills
// `reason` with the text the reference puts into Error::ModelDetection{reason}.
bool detect_model_type(const std::vector<int64_t> &input_shape, const std::vector<std::vector<int64_t>> &output_shapes,
                       int override_type, bn_model_config &cfg, std::string &reason);
uint32_t model_sample_rate(int model_type);
float model_segment_duration(int model_type);
uint64_t model_sample_count(int model_type);
const char *model_type_name(int model_type);  // Rust Debug names: BirdNetV24 / BirdNetV30 / PerchV2
}  // namespace bn
