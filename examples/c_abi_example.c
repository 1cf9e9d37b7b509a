/*
 * c_abi_example.c -- the C ABI of libfecgpu.so used from plain C, the way a Rust / Go / Java FFI would.
 *
 *   gcc -std=c11 -I include examples/c_abi_example.c -L forge_ec_amd -lfecgpu \
 *       -Wl,-rpath,'$ORIGIN/../forge_ec_amd' -Wl,-rpath-link,/opt/rocm/lib -o examples/c_abi_example
 *
 * 1. parity mode: out[i] = Curve::multiply(G, k[i]) for secp256k1, the bit pattern forge-ec's CPU code
 *    produces (fec_batch_mul_fixed), then Curve::to_affine and PointAffine::to_bytes on the GPU;
 *    then round 4's calls: the prefix-table policy, schnorr::batch_verify::<Ed25519>, the multi-GPU calls' refusal;
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

  /* ---- round 4's additions, as an FFI would call them ---- */
  /* the prefix-table policy: how much of the free memory a table may take, a table built at a point of the caller's
   * choosing (refused memory is not an error: the bits say what there is), and the same products afterwards */
  uint64_t out2[N][12];
  rc = fec_ctx_set_fixed_prefix_budget(ctx, 10);
  if (rc == FEC_OK) rc = fec_ctx_set_fixed_prefix_bits(ctx, 12);
  if (rc == FEC_OK) rc = fec_ctx_build_fixed_prefix(ctx, FEC_SECP256K1);
  if (rc == FEC_OK) rc = fec_batch_mul_fixed(ctx, FEC_SECP256K1, &k[0][0], g, &out2[0][0], N);
  if (rc != FEC_OK) {
    printf("prefix-table calls failed: %s\n", fec_strerror(rc));
    return 1;
  }
  printf("parity  multiply(G, k) from a %d-bit prefix table: %s\n", fec_ctx_fixed_prefix_bits(ctx, FEC_SECP256K1),
         memcmp(out, out2, sizeof out) == 0 ? "identical" : "DIFFERENT");
  ok = ok && memcmp(out, out2, sizeof out) == 0;
  /* schnorr::batch_verify::<Ed25519, D> as a release build of the reference runs it; with every weight zero both folds
   * stay the identity (multiply's zero-scalar early-out), which the reference calls a valid batch, and no u128 sum of
   * its scalar Mul wraps */
  uint64_t pk2[2][8], r2[2][8], s2[2][4], a2[2][4], e2[2][4];
  memset(a2, 0, sizeof a2);
  for (int i = 0; i < 2; ++i) {
    for (int l = 0; l < 8; ++l) { pk2[i][l] = 0x1111111111111111ULL * (uint64_t)(l + 1 + i); r2[i][l] = 0x0101010101010101ULL * (uint64_t)(l + 3 + i); }
    pk2[i][3] &= 0x7FFFFFFFFFFFFFFFULL; pk2[i][7] &= 0x7FFFFFFFFFFFFFFFULL; r2[i][3] &= 0x7FFFFFFFFFFFFFFFULL; r2[i][7] &= 0x7FFFFFFFFFFFFFFFULL;
    for (int l = 0; l < 4; ++l) { s2[i][l] = 0xFFFFFFFFFFFFFFFFULL; e2[i][l] = (uint64_t)(7 + l + i); }
  }
  uint8_t verdict = 9, dbg = 9;
  rc = fec_schnorr_batch_verify_ed25519(ctx, &pk2[0][0], NULL, &r2[0][0], NULL, &s2[0][0], &a2[0][0], &e2[0][0], 2, &verdict, NULL, NULL, &dbg);
  if (rc != FEC_OK) {
    printf("fec_schnorr_batch_verify_ed25519 failed: %s\n", fec_strerror(rc));
    return 1;
  }
  printf("parity  schnorr::batch_verify::<Ed25519> with zero weights: result %u, a debug build would panic: %u\n", verdict, dbg);
  ok = ok && verdict == 1 && dbg == 0;
  /* a single-device ctx has fec_batch_*_dev; the device-resident multi-GPU calls say so */
  const size_t none = 0;
  const uint64_t* no_in[1] = {NULL};
  uint64_t* no_out[1] = {NULL};
  ok = ok && fec_multi_batch_mul_dev(ctx, FEC_P256, no_in, no_in, no_out, &none, NULL, 0, NULL) == FEC_E_UNSUPPORTED;
  printf("%s\n", ok ? "c abi example ok" : "c abi example FAILED");
  fec_ctx_destroy(ctx);
  return ok ? 0 : 1;
}
