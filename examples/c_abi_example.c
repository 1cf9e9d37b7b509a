/*
 * c_abi_example.c -- the C ABI of libfecgpu.so used from plain C, the way a Rust / Go / Java FFI would.
 *
 *   gcc -std=c11 -I include examples/c_abi_example.c -L forge_ec_amd -lfecgpu \
 *       -Wl,-rpath,'$ORIGIN/../forge_ec_amd' -Wl,-rpath-link,/opt/rocm/lib -o examples/c_abi_example
 *
 * 1. parity mode: out[i] = Curve::multiply(G, k[i]) for secp256k1, the bit pattern forge-ec's CPU code
 *    produces (fec_batch_mul_fixed), then Curve::to_affine and PointAffine::to_bytes on the GPU;
 * 2. canonical mode: the standard secp256k1 public keys of the same scalars (fec_canon_mul_base) --
 *    3*G.x is the BIP-340 test-vector-0 public key.
 * Prints one line per step and returns 0 when everything behaved.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "fecgpu.h"
#include "fecgpu_canon.h"

#define N 8

int main(void) {
  fec_ctx* ctx = NULL;
  int rc = fec_ctx_create(&ctx, 0);
  if (rc != FEC_OK) {
    printf("no usable gfx950 GPU: %s\n", fec_strerror(rc));
    return 2;  /* there is no CPU fallback by design */
  }
  uint64_t k[N][4];
  memset(k, 0, sizeof k);
  for (int i = 0; i < N; ++i) k[i][0] = (uint64_t)i + 1;  /* scalars 1..8 */

  /* ---- parity mode ---- */
  uint64_t g[12], out[N][12], xy[N][8];
  uint8_t inf[N], enc[N][33];
  rc = fec_generator(ctx, FEC_SECP256K1, g);
  if (rc == FEC_OK) rc = fec_batch_mul_fixed(ctx, FEC_SECP256K1, &k[0][0], g, &out[0][0], N);
  if (rc == FEC_OK) rc = fec_batch_to_affine(ctx, FEC_SECP256K1, &out[0][0], &xy[0][0], inf, N);
  if (rc == FEC_OK) rc = fec_batch_compress(ctx, FEC_SECP256K1, &xy[0][0], inf, &enc[0][0], N);
  if (rc != FEC_OK) {
    printf("parity path failed: %s\n", fec_strerror(rc));
    return 1;
  }
  printf("parity  multiply(G, 2) compressed: %02x", enc[1][0]);
  for (int b = 1; b < 9; ++b) printf("%02x", enc[1][b]);
  printf("...  (forge-ec's own arithmetic, reproduced bit for bit)\n");

  /* ---- canonical mode ---- */
  uint64_t pub[N][8];
  uint8_t st[N];
  rc = fec_canon_mul_base(ctx, FEC_SECP256K1, &k[0][0], &pub[0][0], st, N);
  if (rc != FEC_OK) {
    printf("canonical path failed: %s\n", fec_strerror(rc));
    return 1;
  }
  printf("canon   3*G.x = %016llx%016llx%016llx%016llx\n", (unsigned long long)pub[2][3], (unsigned long long)pub[2][2],
         (unsigned long long)pub[2][1], (unsigned long long)pub[2][0]);
  const uint64_t want[4] = {0x8601F113BCE036F9ULL, 0xB531C845836F99B0ULL, 0x49344F85F89D5229ULL, 0xF9308A019258C310ULL};
  int ok = memcmp(pub[2], want, sizeof want) == 0 && st[2] == FEC_CANON_FINITE;
  printf("%s\n", ok ? "c abi example ok" : "c abi example FAILED");
  fec_ctx_destroy(ctx);
  return ok ? 0 : 1;
}
