// kernels.hpp -- launchers of kernels that live in their own translation units (compiled in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include "limbs.hpp"

namespace fecgpu {

// kernels_p256.hip: P-256 Curve::multiply, workgroup task scheduler.  out[i] = multiply(fixed ? points[0] : points[i], scalars[i])
void p256_launch_mul(bool fixed, const u32* scalars, const u32* points, u32* out, size_t n, hipStream_t s);

}  // namespace fecgpu
