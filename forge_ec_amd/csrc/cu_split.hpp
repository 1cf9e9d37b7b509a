// cu_split.hpp -- how two persistent P-256 launches that run side by side divide the chip.
//
// The P-256 scheduler kernel holds one workgroup per CU for its whole run (146 KiB of LDS), so two launches on two
// streams -- multiply(G, u1) beside multiply(Q, u2) in the ECDSA / Schnorr pipelines and the double multiplication --
// each own a fixed set of CUs, and whichever finishes first leaves its CUs idle.  Round 2 gave each half of the chip
// (halves): the fixed-base launch (affine addend: twelve products per addition instead of
// sixteen, and 24 of its 256 steps a table fetch) then finished 8 ms before the variable-base one.  The CUs are
// divided in proportion to the work instead: SchedEnv::cus is the number of CUs a launch may take, so the split is a
// SchedEnv per launch.
#pragma once
#include "../../include/fecgpu.h"
#include "kernels.hpp"

namespace fecgpu {

// ms per 2^20 elements on the whole chip (tools/p256_split_constants.py, round 4 kernels): variable base with projective
// base points / with affine ones (from_affine(public key): the twelve-product addition), fixed base, fixed base from the
// prefix table
constexpr double kP256VarMs = 23.3, kP256VarAffineMs = 20.9, kP256FixedMs = 20.9, kP256FixedPrefixMs = 19.1;

// env_fixed / env_var: `env` with the CUs of the fixed-base launch and of the variable-base launch(es) that run beside
// it on the other stream (`var_ms`: their summed cost per 2^20 elements, from the constants above).  Below 2^19
// elements the halves stay: a workgroup then holds about one fill of its 864 slots and its time is a ladder's latency,
// not its share of the elements (2^17: 6.4 ms in halves, 7.9 ms split 112 + 144).
inline void p256_cu_split(const SchedEnv& env, size_t n, double var_ms, SchedEnv& env_fixed, SchedEnv& env_var) {
  const unsigned cus = env.cus ? env.cus : 256u;
  const double f = env.gen_prefix[FEC_P256] != nullptr && env.gen_prefix_bits[FEC_P256] >= 16 ? kP256FixedPrefixMs : kP256FixedMs;
  unsigned cf = n >= ((size_t)1 << 19) ? (unsigned)((double)cus * f / (f + var_ms) + 0.5) : cus / 2u;
  // Workgroups go to the eight XCDs round-robin, and an XCD only has its own 32 CUs: both grids must be multiples of
  // eight, or one XCD is handed a workgroup more than it has CUs and that workgroup starts when another one has
  // finished -- measured with 115 + 141: 86 ms instead of 49 for the 2^20 ECDSA batch.
  if (cus % 8u == 0u && cus >= 16u) {
    cf = (cf + 4u) / 8u * 8u;
    if (cf < 8u) cf = 8u;
    if (cf > cus - 8u) cf = cus - 8u;
  }
  if (cf < 1u) cf = 1u;
  if (cus > 1u && cf > cus - 1u) cf = cus - 1u;
  env_fixed = env;
  env_var = env;
  if (cus > 1u) {
    env_fixed.cus = cf;
    env_var.cus = cus - cf;
  }
}

}  // namespace fecgpu
