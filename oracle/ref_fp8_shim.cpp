// TEST INFRASTRUCTURE ONLY -- host-side C entry points over the reference's own
// software e4m3 FNUZ codec (csrc/quantization/fp8/amd/hip_float8.h:14-71, which
// the reference documents as host-compilable, :3-10).  Built by
// oracle/build_ref.py with the reference include directory on the command
// line; nothing from the reference is copied here.
#include <cstdint>
#include <cstring>
#include <iostream>

#include "quantization/fp8/amd/hip_float8.h"

extern "C" {

uint8_t ref_fnuz_encode_f32(float v) { return hip_fp8(v).data; }

float ref_fnuz_decode_f32(uint8_t b) {
  return static_cast<float>(hip_fp8(b, hip_fp8::from_bits()));
}

void ref_fnuz_encode_f32_n(const float* in, uint8_t* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = hip_fp8(in[i]).data;
}

void ref_fnuz_decode_f32_n(const uint8_t* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i)
    out[i] = static_cast<float>(hip_fp8(in[i], hip_fp8::from_bits()));
}
}
