#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {
hipError_t launch_text_classify(const uint8_t* buf, const RecordTable& t, uint32_t* text_w, uint32_t* lin_len,
                                hipStream_t st);
hipError_t launch_text_linearise(const uint8_t* buf, const RecordTable& t, const uint32_t* text_w,
                                 const uint64_t* lin_off, uint8_t* lin, hipStream_t st);
}  // namespace bsk
