#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "index.hpp"

namespace bsk {
hipError_t launch_text_classify(const uint8_t* buf, const RecordTable& t, uint32_t* text_w, uint32_t* lin_len,
                                hipStream_t st);
// lin_len[i] = bases of record i when its layout (RecordTable::text_w, from the index pass) is irregular, else 0
hipError_t launch_lin_len(const RecordTable& t, uint32_t* lin_len, hipStream_t st);
// flatten: every record that is not one line gets a linear copy (text_w_flat[i] = TEXT_IRREGULAR for those, else 0)
hipError_t launch_lin_len_all(const RecordTable& t, uint32_t* lin_len, uint32_t* text_w_flat, hipStream_t st);
// buf_n: bytes in the shard (0: unknown -- wide loads then stay inside each record's own text)
// long_list / long_count / long_max (launch_find_long on l_seq with long_thresh): records copied by whole blocks
hipError_t launch_text_flatten(const uint8_t* buf, uint64_t buf_n, const RecordTable& t, const uint64_t* lin_off, uint8_t* lin, hipStream_t st,
                               const uint32_t* long_list = nullptr, uint64_t long_count = 0, uint64_t long_max = 0, uint32_t long_thresh = 0);
hipError_t launch_text_linearise(const uint8_t* buf, const RecordTable& t, const uint32_t* text_w,
                                 const uint64_t* lin_off, uint8_t* lin, hipStream_t st);
}  // namespace bsk
