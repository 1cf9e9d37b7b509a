//! Safe Rust API over `libfecgpu.so` (include/fecgpu.h): batched `Curve::multiply` and the calls next to
//! it on AMD MI355X, bit-identical to the CPU trait methods of forge-ec-curves.
//!
//! All `unsafe` of the integration lives here (`forge-ec-curves` is `#![forbid(unsafe_code)]`).
//! NOT COMPILED where it was written (no rustc in that image): see README.md; the `extern "C"` block
//! below is diffed against include/fecgpu.h by tests/test_rust_shim_signatures.py.
#![deny(missing_docs)]

use core::ffi::{c_char, c_double, c_float, c_int, c_uint, c_void};

use forge_ec_core::{Curve, Error, Result};
use forge_ec_curves::{ed25519, p256, secp256k1};

/// Opaque `fec_ctx`.
#[repr(C)]
pub struct FecCtx {
    _private: [u8; 0],
}

// ---- include/fecgpu.h, symbol for symbol (same order as the header) ----
#[link(name = "fecgpu")]
extern "C" {
    fn fec_point_limbs(curve: c_int) -> c_int;
    fn fec_ctx_create(out: *mut *mut FecCtx, device: c_int) -> c_int;
    fn fec_ctx_create_multi(out: *mut *mut FecCtx, devices: *const c_int, n_devices: c_int) -> c_int;
    fn fec_ctx_device_count(ctx: *mut FecCtx) -> c_int;
    fn fec_ctx_destroy(ctx: *mut FecCtx);
    fn fec_ctx_wipe(ctx: *mut FecCtx) -> c_int;
    fn fec_ctx_check(ctx: *mut FecCtx) -> c_int;
    fn fec_ctx_debug_force_fault(ctx: *mut FecCtx, enabled: c_int) -> c_int;
    fn fec_ctx_set_fixed_prefix_bits(ctx: *mut FecCtx, bits: c_uint) -> c_int;
    fn fec_ctx_fixed_prefix_bits(ctx: *mut FecCtx, curve: c_int) -> c_int;
    fn fec_ctx_build_fixed_prefix(ctx: *mut FecCtx, curve: c_int) -> c_int;
    fn fec_ctx_set_fixed_prefix_after(ctx: *mut FecCtx, elements: usize) -> c_int;
    fn fec_ctx_set_fixed_prefix_budget(ctx: *mut FecCtx, percent_of_free_memory: c_uint) -> c_int;
    fn fec_ctx_set_side_stream_max(ctx: *mut FecCtx, elements: usize) -> c_int;
    fn fec_generator(ctx: *mut FecCtx, curve: c_int, out: *mut u64) -> c_int;
    fn fec_generator_dev(ctx: *mut FecCtx, curve: c_int) -> *const u64;
    fn fec_batch_mul(ctx: *mut FecCtx, curve: c_int, scalars: *const u64, points: *const u64, out: *mut u64, n: usize) -> c_int;
    fn fec_batch_mul_fixed(ctx: *mut FecCtx, curve: c_int, scalars: *const u64, base: *const u64, out: *mut u64, n: usize) -> c_int;
    fn fec_batch_double_mul(ctx: *mut FecCtx, curve: c_int, u1: *const u64, u2: *const u64, q: *const u64, out: *mut u64, n: usize) -> c_int;
    fn fec_batch_to_affine(ctx: *mut FecCtx, curve: c_int, points: *const u64, xy: *mut u64, inf: *mut u8, n: usize) -> c_int;
    fn fec_multi_scalar_mul(ctx: *mut FecCtx, curve: c_int, scalars: *const u64, points: *const u64, out: *mut u64, n: usize) -> c_int;
    fn fec_batch_validate_point(ctx: *mut FecCtx, curve: c_int, xy: *const u64, inf: *const u8, ok: *mut u8, n: usize) -> c_int;
    fn fec_batch_validate_point_dev(ctx: *mut FecCtx, curve: c_int, d_xy: *const u64, d_inf: *const u8, d_ok: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_batch_ecdh(ctx: *mut FecCtx, curve: c_int, private_keys: *const u64, pk_xy: *const u64, pk_inf: *const u8, secrets: *mut u8, status: *mut u8, n: usize) -> c_int;
    fn fec_batch_ecdh_dev(ctx: *mut FecCtx, curve: c_int, d_private_keys: *const u64, d_pk_xy: *const u64, d_pk_inf: *const u8, d_secrets: *mut u8, d_status: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_ecdsa_batch_verify(ctx: *mut FecCtx, curve: c_int, digests: *const u8, r: *const u64, s: *const u64, pk_xy: *const u64, pk_inf: *const u8, a: *const u64, n: usize, result: *mut u8, detail: *mut u64) -> c_int;
    fn fec_eddsa_verify_ed25519(ctx: *mut FecCtx, r_xy: *const u64, r_inf: *const u8, pk_xy: *const u64, pk_inf: *const u8, s: *const u64, k: *const u64, status: *mut u8, n: usize) -> c_int;
    fn fec_ecdsa_verify_p256(ctx: *mut FecCtx, digests: *const u8, r: *const u64, s: *const u64, pk_xy: *const u64, pk_inf: *const u8, status: *mut u8, n: usize) -> c_int;
    fn fec_ecdsa_verify_secp256k1(ctx: *mut FecCtx, digests: *const u8, r: *const u64, s: *const u64, pk_xy: *const u64, pk_inf: *const u8, status: *mut u8, n: usize) -> c_int;
    fn fec_batch_compress(ctx: *mut FecCtx, curve: c_int, xy: *const u64, inf: *const u8, out: *mut u8, n: usize) -> c_int;
    fn fec_batch_decompress(ctx: *mut FecCtx, curve: c_int, r#in: *const u8, xy: *mut u64, inf: *mut u8, ok: *mut u8, n: usize) -> c_int;
    fn fec_batch_encode_uncompressed(ctx: *mut FecCtx, curve: c_int, xy: *const u64, inf: *const u8, out: *mut u8, n: usize) -> c_int;
    fn fec_batch_decode_uncompressed(ctx: *mut FecCtx, curve: c_int, r#in: *const u8, xy: *mut u64, inf: *mut u8, ok: *mut u8, n: usize) -> c_int;
    fn fec_schnorr_batch_verify_secp256k1(ctx: *mut FecCtx, pk_xy: *const u64, pk_inf: *const u8, r_xy: *const u64, r_inf: *const u8, s: *const u64, a: *const u64, e: *const u64, n: usize, result: *mut u8, sides_xy: *mut u64, sides_inf: *mut u8) -> c_int;
    fn fec_schnorr_batch_verify(ctx: *mut FecCtx, curve: c_int, pk_xy: *const u64, pk_inf: *const u8, r_xy: *const u64, r_inf: *const u8, s: *const u64, a: *const u64, e: *const u64, n: usize, result: *mut u8, sides_xy: *mut u64, sides_inf: *mut u8) -> c_int;
    fn fec_schnorr_batch_verify_ed25519(ctx: *mut FecCtx, pk_xy: *const u64, pk_inf: *const u8, r_xy: *const u64, r_inf: *const u8, s: *const u64, a: *const u64, e: *const u64, n: usize, result: *mut u8, sides_xy: *mut u64, sides_inf: *mut u8, debug_build_panics: *mut u8) -> c_int;
    fn fec_schnorr_verify(ctx: *mut FecCtx, curve: c_int, pk_xy: *const u64, pk_inf: *const u8, r_xy: *const u64, r_inf: *const u8, s: *const u64, e: *const u64, status: *mut u8, n: usize) -> c_int;
    fn fec_field_op(ctx: *mut FecCtx, curve: c_int, op: c_int, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    fn fec_point_op(ctx: *mut FecCtx, curve: c_int, op: c_int, p: *const u64, q: *const u64, out: *mut u64, n: usize) -> c_int;
    fn fec_batch_mul_dev(ctx: *mut FecCtx, curve: c_int, d_scalars: *const u64, d_points: *const u64, d_out: *mut u64, n: usize, stream: *mut c_void) -> c_int;
    fn fec_batch_mul_fixed_dev(ctx: *mut FecCtx, curve: c_int, d_scalars: *const u64, d_base: *const u64, d_out: *mut u64, n: usize, stream: *mut c_void) -> c_int;
    fn fec_batch_double_mul_dev(ctx: *mut FecCtx, curve: c_int, d_u1: *const u64, d_u2: *const u64, d_q: *const u64, d_out: *mut u64, n: usize, stream: *mut c_void) -> c_int;
    fn fec_multi_batch_mul_dev(ctx: *mut FecCtx, curve: c_int, scalars: *const *const u64, points: *const *const u64, out: *const *mut u64, counts: *const usize, gathered: *mut u64, consumer: c_int, streams: *const *mut c_void) -> c_int;
    fn fec_multi_batch_mul_fixed_dev(ctx: *mut FecCtx, curve: c_int, scalars: *const *const u64, bases: *const *const u64, out: *const *mut u64, counts: *const usize, gathered: *mut u64, consumer: c_int, streams: *const *mut c_void) -> c_int;
    fn fec_multi_batch_double_mul_dev(ctx: *mut FecCtx, curve: c_int, u1: *const *const u64, u2: *const *const u64, q: *const *const u64, out: *const *mut u64, counts: *const usize, gathered: *mut u64, consumer: c_int, streams: *const *mut c_void) -> c_int;
    fn fec_eddsa_verify_ed25519_dev(ctx: *mut FecCtx, d_r_xy: *const u64, d_r_inf: *const u8, d_pk_xy: *const u64, d_pk_inf: *const u8, d_s: *const u64, d_k: *const u64, d_status: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_ecdsa_verify_p256_dev(ctx: *mut FecCtx, d_digests: *const u8, d_r: *const u64, d_s: *const u64, d_pk_xy: *const u64, d_pk_inf: *const u8, d_status: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_ecdsa_verify_secp256k1_dev(ctx: *mut FecCtx, d_digests: *const u8, d_r: *const u64, d_s: *const u64, d_pk_xy: *const u64, d_pk_inf: *const u8, d_status: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_schnorr_verify_dev(ctx: *mut FecCtx, curve: c_int, d_pk_xy: *const u64, d_pk_inf: *const u8, d_r_xy: *const u64, d_r_inf: *const u8, d_s: *const u64, d_e: *const u64, d_status: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_batch_compress_dev(ctx: *mut FecCtx, curve: c_int, d_xy: *const u64, d_inf: *const u8, d_out: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_batch_to_affine_dev(ctx: *mut FecCtx, curve: c_int, d_points: *const u64, d_xy: *mut u64, d_inf: *mut u8, n: usize, stream: *mut c_void) -> c_int;
    fn fec_ctx_set_chunk(ctx: *mut FecCtx, elements: usize) -> c_int;
    fn fec_ctx_set_timing(ctx: *mut FecCtx, enabled: c_int) -> c_int;
    fn fec_ctx_last_kernel_ms(ctx: *mut FecCtx, ms: *mut c_float, kernel_name: *mut *const c_char) -> c_int;
    fn fec_measure_peak_mad32(ctx: *mut FecCtx, mad32_per_sec: *mut c_double) -> c_int;
    fn fec_ctx_device_info(ctx: *mut FecCtx, name: *mut c_char, name_len: usize, compute_units: *mut c_int, clock_khz: *mut c_int) -> c_int;
    fn fec_strerror(status: c_int) -> *const c_char;
}

/// Maps a non-zero `fec_status` to the reference's error type (`forge-ec-core/src/lib.rs:70-103`).
fn check(rc: c_int) -> Result<()> {
    match rc {
        0 => Ok(()),
        -1 => Err(Error::ValidationError),       // FEC_E_ARG
        -5 => Err(Error::UnsupportedOperation),  // FEC_E_UNSUPPORTED
        _ => Err(Error::GenericError),           // device / memory / launch / comm
    }
}

/// Text of a status code (`fec_strerror`).
pub fn status_text(rc: i32) -> &'static str {
    // SAFETY: fec_strerror returns a pointer to a static NUL-terminated string for every input.
    unsafe { core::ffi::CStr::from_ptr(fec_strerror(rc)) }.to_str().unwrap_or("?")
}

/// One `fec_ctx`: a GPU (or several, see [`GpuContext::new_multi`]).  `Send`, not `Sync`: calls on
/// one ctx are serialised by its owner, different ctxs are independent.
pub struct GpuContext {
    raw: *mut FecCtx,
}
// SAFETY: the ctx holds no thread-affine state; every entry point selects its device itself.
unsafe impl Send for GpuContext {}

impl GpuContext {
    /// `fec_ctx_create`: one MI355X.  Fails when no gfx950 GPU is usable -- there is no CPU fallback.
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = core::ptr::null_mut();
        // SAFETY: `raw` is a valid out-pointer; on failure the library leaves it null.
        check(unsafe { fec_ctx_create(&mut raw, device) })?;
        Ok(Self { raw })
    }

    /// `fec_ctx_create_multi`: the element-wise batch calls shard contiguously over `devices`
    /// (e.g. `&[0, 1, 2, 3, 4, 5, 6, 7]` on one 8-GPU node) and write into the caller's buffers.
    pub fn new_multi(devices: &[i32]) -> Result<Self> {
        let mut raw = core::ptr::null_mut();
        // SAFETY: pointer/length pair of a live slice.
        check(unsafe { fec_ctx_create_multi(&mut raw, devices.as_ptr(), devices.len() as c_int) })?;
        Ok(Self { raw })
    }

    /// Number of shard workers (1 for a single-device ctx).
    pub fn device_count(&self) -> usize {
        // SAFETY: self.raw is a live ctx.
        unsafe { fec_ctx_device_count(self.raw) as usize }
    }

    /// `fec_ctx_wipe`: zero every ctx-owned device buffer that can hold copies of caller data.
    pub fn wipe(&mut self) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_wipe(self.raw) })
    }

    /// `fec_ctx_check`: synchronise, then `Err` if a kernel launched through this ctx since the last check reported
    /// a fault (the outputs of those launches must not be used).  The host-pointer calls check by themselves; this
    /// is for callers of the `*_dev` entry points.
    pub fn check(&mut self) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_check(self.raw) })
    }

    /// Test hook (`fec_ctx_debug_force_fault`): scheduler kernels raise their fault word at once.
    pub fn debug_force_fault(&mut self, enabled: bool) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_debug_force_fault(self.raw, enabled as c_int) })
    }

    /// Size of the generator's fixed-base prefix tables (`fec_ctx_set_fixed_prefix_bits`): 2^bits entries per curve,
    /// 0 = off, default 24.  Results do not depend on it.
    pub fn set_fixed_prefix_bits(&mut self, bits: u32) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_set_fixed_prefix_bits(self.raw, bits as c_uint) })
    }

    /// Bits of the prefix table curve `C` has at this moment (0 = none).
    pub fn fixed_prefix_bits<C: GpuCurve>(&mut self) -> Result<u32> {
        // SAFETY: self.raw is a live ctx.
        let r = unsafe { fec_ctx_fixed_prefix_bits(self.raw, C::ID) };
        if r < 0 { check(r)?; }
        Ok(r as u32)
    }

    /// Attach or build the prefix table of curve `C`'s generator now (`fec_ctx_build_fixed_prefix`, synchronous).
    pub fn build_fixed_prefix<C: GpuCurve>(&mut self) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_build_fixed_prefix(self.raw, C::ID) })
    }

    /// A ctx left to its defaults builds a curve's table after this many multiplications by its generator.
    pub fn set_fixed_prefix_after(&mut self, elements: usize) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_set_fixed_prefix_after(self.raw, elements) })
    }

    /// Share of the device's free memory a prefix table may take, in percent (default 25).
    pub fn set_fixed_prefix_budget(&mut self, percent_of_free_memory: u32) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_set_fixed_prefix_budget(self.raw, percent_of_free_memory as c_uint) })
    }

    /// `u1*G` runs beside `u2*Q` on the ctx's second stream for launches of up to this many elements (measurement knob).
    pub fn set_side_stream_max(&mut self, elements: usize) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_set_side_stream_max(self.raw, elements) })
    }

    /// Device-resident shards of a multi-device ctx (`fec_multi_batch_mul_dev`): shard `g` -- `counts[g]` elements of
    /// curve `C`, raw limbs as `to_raw()` lays them out -- sits in the memory of the ctx's `g`-th device; results go to
    /// `out[g]` and, when `gathered` is given, are also copied over xGMI into that array on the `consumer`-th device.
    /// Synchronous.
    ///
    /// # Safety
    /// Every pointer must be a live HIP device allocation on the device it is listed for, 16-byte aligned and large
    /// enough for `counts[g]` elements (`gathered`: for the sum of the counts); the slices have one entry per device.
    pub unsafe fn multi_batch_multiply_dev<C: GpuCurve>(&mut self, scalars: &[*const u64], points: &[*const u64], out: &[*mut u64],
                                                        counts: &[usize], gathered: Option<*mut u64>, consumer: usize,
                                                        streams: Option<&[*mut c_void]>) -> Result<()> {
        let n = self.device_count();
        if scalars.len() != n || points.len() != n || out.len() != n || counts.len() != n || streams.map_or(false, |s| s.len() != n) {
            return Err(Error::ValidationError);
        }
        check(fec_multi_batch_mul_dev(self.raw, C::ID, scalars.as_ptr(), points.as_ptr(), out.as_ptr(), counts.as_ptr(),
                                      gathered.unwrap_or(std::ptr::null_mut()), consumer as c_int,
                                      streams.map_or(std::ptr::null(), |s| s.as_ptr())))
    }

    /// `fec_multi_batch_mul_fixed_dev`: as above with one base per device (`None` = the reference's `generator()`).
    ///
    /// # Safety
    /// As for [`GpuContext::multi_batch_multiply_dev`].
    pub unsafe fn multi_batch_multiply_fixed_dev<C: GpuCurve>(&mut self, scalars: &[*const u64], bases: Option<&[*const u64]>, out: &[*mut u64],
                                                              counts: &[usize], gathered: Option<*mut u64>, consumer: usize,
                                                              streams: Option<&[*mut c_void]>) -> Result<()> {
        let n = self.device_count();
        if scalars.len() != n || out.len() != n || counts.len() != n || bases.map_or(false, |b| b.len() != n) || streams.map_or(false, |s| s.len() != n) {
            return Err(Error::ValidationError);
        }
        check(fec_multi_batch_mul_fixed_dev(self.raw, C::ID, scalars.as_ptr(), bases.map_or(std::ptr::null(), |b| b.as_ptr()), out.as_ptr(),
                                            counts.as_ptr(), gathered.unwrap_or(std::ptr::null_mut()), consumer as c_int,
                                            streams.map_or(std::ptr::null(), |s| s.as_ptr())))
    }

    /// `fec_multi_batch_double_mul_dev`: `u1[i]*G + u2[i]*Q[i]` on device-resident shards.
    ///
    /// # Safety
    /// As for [`GpuContext::multi_batch_multiply_dev`].
    pub unsafe fn multi_batch_double_multiply_dev<C: GpuCurve>(&mut self, u1: &[*const u64], u2: &[*const u64], q: &[*const u64], out: &[*mut u64],
                                                               counts: &[usize], gathered: Option<*mut u64>, consumer: usize,
                                                               streams: Option<&[*mut c_void]>) -> Result<()> {
        let n = self.device_count();
        if u1.len() != n || u2.len() != n || q.len() != n || out.len() != n || counts.len() != n || streams.map_or(false, |s| s.len() != n) {
            return Err(Error::ValidationError);
        }
        check(fec_multi_batch_double_mul_dev(self.raw, C::ID, u1.as_ptr(), u2.as_ptr(), q.as_ptr(), out.as_ptr(), counts.as_ptr(),
                                             gathered.unwrap_or(std::ptr::null_mut()), consumer as c_int,
                                             streams.map_or(std::ptr::null(), |s| s.as_ptr())))
    }

    /// Elements per pipeline chunk of the host-pointer calls (tuning knob; results do not depend on it).
    pub fn set_chunk(&mut self, elements: usize) -> Result<()> {
        // SAFETY: self.raw is a live ctx.
        check(unsafe { fec_ctx_set_chunk(self.raw, elements) })
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        // SAFETY: created by fec_ctx_create*, destroyed exactly once.
        unsafe { fec_ctx_destroy(self.raw) }
    }
}

/// A curve the backend implements: maps the reference's value types to the ABI's limb arrays
/// (`to_raw()` layout: little-endian `[u64; 4]` per field element / scalar; a point = X, Y, Z(, T)).
pub trait GpuCurve: Curve {
    /// `fec_curve` value.
    const ID: c_int;
    /// `u64` limbs per projective point (`fec_point_limbs`).
    const LIMBS: usize;
    /// `Scalar::to_raw()`.
    fn scalar_limbs(s: &Self::Scalar) -> [u64; 4];
    /// Writes the point's raw coordinates into `out[..LIMBS]`.
    fn point_limbs(p: &Self::PointProjective, out: &mut [u64]);
    /// Rebuilds a point from `LIMBS` raw limbs (no validation, like `multiply`'s own return value).
    fn point_from_limbs(l: &[u64]) -> Self::PointProjective;
    /// Raw (x, y) and the infinity flag of an affine point.
    fn affine_limbs(a: &Self::PointAffine) -> ([u64; 8], bool);
    /// Rebuilds an affine point from raw (x, y) and the infinity flag.
    fn affine_from_limbs(xy: &[u64], infinity: bool) -> Self::PointAffine;
}

fn limb4(l: &[u64], i: usize) -> [u64; 4] {
    [l[4 * i], l[4 * i + 1], l[4 * i + 2], l[4 * i + 3]]
}

macro_rules! impl_weierstrass {
    ($curve:ty, $m:ident, $id:expr) => {
        impl GpuCurve for $curve {
            const ID: c_int = $id;
            const LIMBS: usize = 12;
            fn scalar_limbs(s: &$m::Scalar) -> [u64; 4] {
                s.to_raw()
            }
            fn point_limbs(p: &$m::ProjectivePoint, out: &mut [u64]) {
                for (i, c) in p.to_raw_coords().iter().enumerate() {
                    out[4 * i..4 * i + 4].copy_from_slice(c);
                }
            }
            fn point_from_limbs(l: &[u64]) -> $m::ProjectivePoint {
                $m::ProjectivePoint::from_raw_coords([limb4(l, 0), limb4(l, 1), limb4(l, 2)])
            }
            fn affine_limbs(a: &$m::AffinePoint) -> ([u64; 8], bool) {
                let (c, inf) = a.to_raw_coords();
                let mut o = [0u64; 8];
                o[..4].copy_from_slice(&c[0]);
                o[4..].copy_from_slice(&c[1]);
                (o, inf)
            }
            fn affine_from_limbs(xy: &[u64], infinity: bool) -> $m::AffinePoint {
                $m::AffinePoint::from_raw_coords([limb4(xy, 0), limb4(xy, 1)], infinity)
            }
        }
    };
}
impl_weierstrass!(secp256k1::Secp256k1, secp256k1, 0);
impl_weierstrass!(p256::P256, p256, 1);

impl GpuCurve for ed25519::Ed25519 {
    const ID: c_int = 2;
    const LIMBS: usize = 16;
    fn scalar_limbs(s: &ed25519::Scalar) -> [u64; 4] {
        s.to_raw()
    }
    fn point_limbs(p: &ed25519::ExtendedPoint, out: &mut [u64]) {
        for (i, c) in p.to_raw_coords().iter().enumerate() {
            out[4 * i..4 * i + 4].copy_from_slice(c);
        }
    }
    fn point_from_limbs(l: &[u64]) -> ed25519::ExtendedPoint {
        ed25519::ExtendedPoint::from_raw_coords([limb4(l, 0), limb4(l, 1), limb4(l, 2), limb4(l, 3)])
    }
    fn affine_limbs(a: &ed25519::AffinePoint) -> ([u64; 8], bool) {
        let (c, inf) = a.to_raw_coords();
        let mut o = [0u64; 8];
        o[..4].copy_from_slice(&c[0]);
        o[4..].copy_from_slice(&c[1]);
        (o, inf)
    }
    fn affine_from_limbs(xy: &[u64], infinity: bool) -> ed25519::AffinePoint {
        ed25519::AffinePoint::from_raw_coords([limb4(xy, 0), limb4(xy, 1)], infinity)
    }
}

fn pack_scalars<C: GpuCurve>(s: &[C::Scalar]) -> Vec<u64> {
    let mut k = vec![0u64; 4 * s.len()];
    for (i, x) in s.iter().enumerate() {
        k[4 * i..4 * i + 4].copy_from_slice(&C::scalar_limbs(x));
    }
    k
}

fn pack_points<C: GpuCurve>(p: &[C::PointProjective]) -> Vec<u64> {
    let mut v = vec![0u64; C::LIMBS * p.len()];
    for (i, x) in p.iter().enumerate() {
        C::point_limbs(x, &mut v[C::LIMBS * i..C::LIMBS * (i + 1)]);
    }
    v
}

fn unpack_points<C: GpuCurve>(v: &[u64]) -> Vec<C::PointProjective> {
    v.chunks_exact(C::LIMBS).map(C::point_from_limbs).collect()
}

/// `out[i] = C::multiply(&points[i], &scalars[i])`, bit-identical to the CPU trait method
/// (`secp256k1.rs:2635-2692`, `p256.rs:2120-2156`, `ed25519.rs:2062-2097`).
pub fn batch_multiply<C: GpuCurve>(ctx: &mut GpuContext, points: &[C::PointProjective], scalars: &[C::Scalar]) -> Result<Vec<C::PointProjective>> {
    if points.len() != scalars.len() {
        return Err(Error::ValidationError);
    }
    let (k, p) = (pack_scalars::<C>(scalars), pack_points::<C>(points));
    let mut out = vec![0u64; C::LIMBS * points.len()];
    // SAFETY: every pointer covers n elements of the documented layout; nothing is retained after return.
    check(unsafe { fec_batch_mul(ctx.raw, C::ID, k.as_ptr(), p.as_ptr(), out.as_mut_ptr(), points.len()) })?;
    Ok(unpack_points::<C>(&out))
}

/// `out[i] = C::multiply(base, &scalars[i])` -- key generation with `base = C::generator()`.
pub fn batch_multiply_fixed<C: GpuCurve>(ctx: &mut GpuContext, base: &C::PointProjective, scalars: &[C::Scalar]) -> Result<Vec<C::PointProjective>> {
    let k = pack_scalars::<C>(scalars);
    let b = pack_points::<C>(core::slice::from_ref(base));
    let mut out = vec![0u64; C::LIMBS * scalars.len()];
    // SAFETY: as above; `b` holds one point.
    check(unsafe { fec_batch_mul_fixed(ctx.raw, C::ID, k.as_ptr(), b.as_ptr(), out.as_mut_ptr(), scalars.len()) })?;
    Ok(unpack_points::<C>(&out))
}

/// `R[i] = C::multiply(&G, &u1[i]) + C::multiply(&q[i], &u2[i])` (`forge-ec-signature/src/ecdsa.rs:254-256`).
pub fn batch_double_multiply<C: GpuCurve>(ctx: &mut GpuContext, u1: &[C::Scalar], u2: &[C::Scalar], q: &[C::PointProjective]) -> Result<Vec<C::PointProjective>> {
    if u1.len() != u2.len() || u1.len() != q.len() {
        return Err(Error::ValidationError);
    }
    let (a, b, p) = (pack_scalars::<C>(u1), pack_scalars::<C>(u2), pack_points::<C>(q));
    let mut out = vec![0u64; C::LIMBS * q.len()];
    // SAFETY: as above.
    check(unsafe { fec_batch_double_mul(ctx.raw, C::ID, a.as_ptr(), b.as_ptr(), p.as_ptr(), out.as_mut_ptr(), q.len()) })?;
    Ok(unpack_points::<C>(&out))
}

/// `C::to_affine(&points[i])` with the reference's own field inversion (`secp256k1.rs:1342-1363`,
/// `p256.rs:1835-1857`, `ed25519.rs:1793-1811`).
pub fn batch_to_affine<C: GpuCurve>(ctx: &mut GpuContext, points: &[C::PointProjective]) -> Result<Vec<C::PointAffine>> {
    let n = points.len();
    let p = pack_points::<C>(points);
    let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
    // SAFETY: as above.
    check(unsafe { fec_batch_to_affine(ctx.raw, C::ID, p.as_ptr(), xy.as_mut_ptr(), inf.as_mut_ptr(), n) })?;
    Ok((0..n).map(|i| C::affine_from_limbs(&xy[8 * i..8 * i + 8], inf[i] != 0)).collect())
}

/// `PointAffine::to_bytes` of every point (33 bytes each; `secp256k1.rs:875-896`, `p256.rs:1558-1578`,
/// `ed25519.rs:1505-1525`).
pub fn batch_compress<C: GpuCurve>(ctx: &mut GpuContext, points: &[C::PointAffine]) -> Result<Vec<[u8; 33]>> {
    let n = points.len();
    let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
    for (i, a) in points.iter().enumerate() {
        let (l, f) = C::affine_limbs(a);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    let mut out = vec![0u8; 33 * n];
    // SAFETY: as above.
    check(unsafe { fec_batch_compress(ctx.raw, C::ID, xy.as_ptr(), inf.as_ptr(), out.as_mut_ptr(), n) })?;
    Ok(out.chunks_exact(33).map(|c| <[u8; 33]>::try_from(c).unwrap()).collect())
}

/// `PointAffine::from_bytes` of every 33-byte encoding (`secp256k1.rs:896-976`, `p256.rs:1580-1639`,
/// `ed25519.rs:1526-1582`): `None` exactly where the reference returns `None`.
pub fn batch_decompress<C: GpuCurve>(ctx: &mut GpuContext, encoded: &[[u8; 33]]) -> Result<Vec<Option<C::PointAffine>>> {
    let n = encoded.len();
    let (mut xy, mut inf, mut ok) = (vec![0u64; 8 * n], vec![0u8; n], vec![0u8; n]);
    // SAFETY: `encoded` is n contiguous 33-byte arrays; the outputs hold n elements each.
    check(unsafe { fec_batch_decompress(ctx.raw, C::ID, encoded.as_ptr().cast(), xy.as_mut_ptr(), inf.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
    Ok((0..n).map(|i| (ok[i] != 0).then(|| C::affine_from_limbs(&xy[8 * i..8 * i + 8], inf[i] != 0))).collect())
}

/// `UncompressedPoint::from_affine` (`forge-ec-encoding/src/point.rs:186-211`): 65 bytes per point.
pub fn batch_encode_uncompressed<C: GpuCurve>(ctx: &mut GpuContext, points: &[C::PointAffine]) -> Result<Vec<[u8; 65]>> {
    let n = points.len();
    let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
    for (i, a) in points.iter().enumerate() {
        let (l, f) = C::affine_limbs(a);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    let mut out = vec![0u8; 65 * n];
    // SAFETY: n elements behind every pointer.
    check(unsafe { fec_batch_encode_uncompressed(ctx.raw, C::ID, xy.as_ptr(), inf.as_ptr(), out.as_mut_ptr(), n) })?;
    Ok(out.chunks_exact(65).map(|c| <[u8; 65]>::try_from(c).unwrap()).collect())
}

/// `UncompressedPoint::to_affine` (`point.rs:214-281`) of every 65-byte encoding.
pub fn batch_decode_uncompressed<C: GpuCurve>(ctx: &mut GpuContext, encoded: &[[u8; 65]]) -> Result<Vec<Option<C::PointAffine>>> {
    let n = encoded.len();
    let (mut xy, mut inf, mut ok) = (vec![0u64; 8 * n], vec![0u8; n], vec![0u8; n]);
    // SAFETY: `encoded` is n contiguous 65-byte arrays; the outputs hold n elements each.
    check(unsafe { fec_batch_decode_uncompressed(ctx.raw, C::ID, encoded.as_ptr().cast(), xy.as_mut_ptr(), inf.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
    Ok((0..n).map(|i| (ok[i] != 0).then(|| C::affine_from_limbs(&xy[8 * i..8 * i + 8], inf[i] != 0))).collect())
}

/// `C::multi_scalar_multiply(points, scalars)` (`forge-ec-core/src/lib.rs:934-951`): the sum, folded
/// left to right from the identity exactly as the default method does.
pub fn multi_scalar_multiply<C: GpuCurve>(ctx: &mut GpuContext, points: &[C::PointProjective], scalars: &[C::Scalar]) -> Result<C::PointProjective> {
    if points.len() != scalars.len() {
        return Err(Error::ValidationError);
    }
    let (k, p) = (pack_scalars::<C>(scalars), pack_points::<C>(points));
    let mut out = vec![0u64; C::LIMBS];
    // SAFETY: as above; `out` holds one point.
    check(unsafe { fec_multi_scalar_mul(ctx.raw, C::ID, k.as_ptr(), p.as_ptr(), out.as_mut_ptr(), points.len()) })?;
    Ok(C::point_from_limbs(&out))
}

/// Outcome of one ECDSA verification as the reference computes it (`ecdsa.rs:213-281`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum VerifyStatus {
    /// `verify` returns `false`.
    Invalid,
    /// `verify` returns `true`.
    Valid,
    /// The reference panics (`CtOption::unwrap` on `None`: digest or affine x >= n as a scalar).
    ReferencePanics,
}

/// `Ecdsa::<Secp256k1, D>::verify` per element, everything after the hash on the GPU.
/// `digests[i] = D::digest(msg_i)` (32 bytes, as the hash emits them).
pub fn ecdsa_verify_batch_secp256k1(ctx: &mut GpuContext, digests: &[[u8; 32]], r: &[secp256k1::Scalar], s: &[secp256k1::Scalar], public_keys: &[secp256k1::AffinePoint]) -> Result<Vec<VerifyStatus>> {
    let n = digests.len();
    if r.len() != n || s.len() != n || public_keys.len() != n {
        return Err(Error::ValidationError);
    }
    type C = secp256k1::Secp256k1;
    let (rr, ss) = (pack_scalars::<C>(r), pack_scalars::<C>(s));
    let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
    for (i, a) in public_keys.iter().enumerate() {
        let (l, f) = C::affine_limbs(a);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    let mut status = vec![0u8; n];
    // SAFETY: `digests` is n contiguous 32-byte arrays; the other buffers hold n elements each.
    check(unsafe { fec_ecdsa_verify_secp256k1(ctx.raw, digests.as_ptr().cast(), rr.as_ptr(), ss.as_ptr(), xy.as_ptr(), inf.as_ptr(), status.as_mut_ptr(), n) })?;
    Ok(status.iter().map(|&v| match v { 1 => VerifyStatus::Valid, 2 => VerifyStatus::ReferencePanics, _ => VerifyStatus::Invalid }).collect())
}

/// `Ecdsa::<P256, D>::verify` per element, in the reference's own P-256 scalar arithmetic
/// (`p256.rs:924-1020`, `1409-1432`; range check = the default `Scalar::ct_lt`, core `lib.rs:497-531`).
pub fn ecdsa_verify_batch_p256(ctx: &mut GpuContext, digests: &[[u8; 32]], r: &[p256::Scalar], s: &[p256::Scalar], public_keys: &[p256::AffinePoint]) -> Result<Vec<VerifyStatus>> {
    let n = digests.len();
    if r.len() != n || s.len() != n || public_keys.len() != n {
        return Err(Error::ValidationError);
    }
    type C = p256::P256;
    let (rr, ss) = (pack_scalars::<C>(r), pack_scalars::<C>(s));
    let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
    for (i, a) in public_keys.iter().enumerate() {
        let (l, f) = C::affine_limbs(a);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    let mut status = vec![0u8; n];
    // SAFETY: `digests` is n contiguous 32-byte arrays; the other buffers hold n elements each.
    check(unsafe { fec_ecdsa_verify_p256(ctx.raw, digests.as_ptr().cast(), rr.as_ptr(), ss.as_ptr(), xy.as_ptr(), inf.as_ptr(), status.as_mut_ptr(), n) })?;
    Ok(status.iter().map(|&v| match v { 1 => VerifyStatus::Valid, 2 => VerifyStatus::ReferencePanics, _ => VerifyStatus::Invalid }).collect())
}

/// `C::validate_point(&points[i])` per element (`secp256k1.rs:2722-2726`, `p256.rs:2187-2191`; the trait default
/// `forge-ec-core/src/lib.rs:905-925` for Ed25519).
pub fn batch_validate_point<C: GpuCurve>(ctx: &mut GpuContext, points: &[C::PointAffine]) -> Result<Vec<bool>> {
    let n = points.len();
    let (mut xy, mut inf, mut ok) = (vec![0u64; 8 * n], vec![0u8; n], vec![0u8; n]);
    for (i, p) in points.iter().enumerate() {
        let (l, f) = C::affine_limbs(p);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    // SAFETY: every buffer holds n elements of the width the header states.
    check(unsafe { fec_batch_validate_point(ctx.raw, C::ID, xy.as_ptr(), inf.as_ptr(), ok.as_mut_ptr(), n) })?;
    Ok(ok.iter().map(|&v| v != 0).collect())
}

/// `KeyExchange::derive_shared_secret` per element (`secp256k1.rs:1884-1904`, `p256.rs:2281-2312`) for secp256k1 and
/// P-256: `Ok(secret)` or the reference's error (`InvalidPublicKey` from P-256's validation; `InvalidEncoding` /
/// `KeyExchangeError` when the product is the identity).  Reproduces reference behaviour; not a hardened ECDH.
pub fn batch_derive_shared_secret<C: GpuCurve>(ctx: &mut GpuContext, private_keys: &[C::Scalar], public_keys: &[C::PointAffine]) -> Result<Vec<Result<[u8; 32]>>> {
    let n = private_keys.len();
    if public_keys.len() != n {
        return Err(Error::ValidationError);
    }
    let kk = pack_scalars::<C>(private_keys);
    let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
    for (i, p) in public_keys.iter().enumerate() {
        let (l, f) = C::affine_limbs(p);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    let (mut secrets, mut status) = (vec![0u8; 32 * n], vec![0u8; n]);
    // SAFETY: every buffer holds n elements of the width the header states.
    check(unsafe { fec_batch_ecdh(ctx.raw, C::ID, kk.as_ptr(), xy.as_ptr(), inf.as_ptr(), secrets.as_mut_ptr(), status.as_mut_ptr(), n) })?;
    Ok((0..n).map(|i| match status[i] {
        0 => { let mut s = [0u8; 32]; s.copy_from_slice(&secrets[32 * i..32 * i + 32]); Ok(s) }
        1 => Err(Error::InvalidPublicKey),
        _ => Err(if C::ID == 0 { Error::InvalidEncoding } else { Error::KeyExchangeError }),
    }).collect())
}

/// `Ecdsa::<C, D>::batch_verify` (`forge-ec-signature/src/ecdsa.rs:287-391`) for `C` = secp256k1 or P-256 from
/// line 310 on: the caller hashes (`digests[i] = D::digest(msgs[i])`) and draws the weights `a` (302-306) with
/// the reference's own `Scalar::random`.
pub fn ecdsa_batch_verify<C: GpuCurve>(ctx: &mut GpuContext, digests: &[[u8; 32]], r: &[C::Scalar], s: &[C::Scalar], public_keys: &[C::PointAffine], a: &[C::Scalar]) -> Result<VerifyStatus> {
    let n = digests.len();
    if r.len() != n || s.len() != n || public_keys.len() != n || a.len() != n {
        return Err(Error::ValidationError);
    }
    let (rr, ss, aa) = (pack_scalars::<C>(r), pack_scalars::<C>(s), pack_scalars::<C>(a));
    let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
    for (i, p) in public_keys.iter().enumerate() {
        let (l, f) = C::affine_limbs(p);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    let mut result = 0u8;
    // SAFETY: every buffer holds n elements of the width the header states; detail may be null.
    check(unsafe { fec_ecdsa_batch_verify(ctx.raw, C::ID, digests.as_ptr().cast(), rr.as_ptr(), ss.as_ptr(), xy.as_ptr(), inf.as_ptr(), aa.as_ptr(), n, &mut result, core::ptr::null_mut()) })?;
    Ok(match result { 1 => VerifyStatus::Valid, 2 => VerifyStatus::ReferencePanics, _ => VerifyStatus::Invalid })
}

/// `Eddsa::<Ed25519, D>::verify` / `Ed25519::verify` per element from the point computation on
/// (`forge-ec-signature/src/eddsa.rs:174-211`, `430-447`).  The caller keeps the hashing and the message
/// special cases (157-170 / 361-374): `k[i] = Scalar::from_bytes_reduced(&hash_i[0..32])`, `sig_r[i]` and
/// `public_keys[i]` decoded with `PointAffine::from_bytes`.
pub fn eddsa_verify_batch_ed25519(ctx: &mut GpuContext, sig_r: &[ed25519::AffinePoint], sig_s: &[ed25519::Scalar], public_keys: &[ed25519::AffinePoint], k: &[ed25519::Scalar]) -> Result<Vec<VerifyStatus>> {
    let n = sig_r.len();
    if sig_s.len() != n || public_keys.len() != n || k.len() != n {
        return Err(Error::ValidationError);
    }
    type C = ed25519::Ed25519;
    let (ss, kk) = (pack_scalars::<C>(sig_s), pack_scalars::<C>(k));
    let (mut rxy, mut rinf, mut pxy, mut pinf) = (vec![0u64; 8 * n], vec![0u8; n], vec![0u64; 8 * n], vec![0u8; n]);
    for i in 0..n {
        let (l, f) = C::affine_limbs(&sig_r[i]);
        rxy[8 * i..8 * i + 8].copy_from_slice(&l);
        rinf[i] = f as u8;
        let (l, f) = C::affine_limbs(&public_keys[i]);
        pxy[8 * i..8 * i + 8].copy_from_slice(&l);
        pinf[i] = f as u8;
    }
    let mut status = vec![0u8; n];
    // SAFETY: every buffer holds n elements of the width the header states.
    check(unsafe { fec_eddsa_verify_ed25519(ctx.raw, rxy.as_ptr(), rinf.as_ptr(), pxy.as_ptr(), pinf.as_ptr(), ss.as_ptr(), kk.as_ptr(), status.as_mut_ptr(), n) })?;
    Ok(status.iter().map(|&v| match v { 1 => VerifyStatus::Valid, 2 => VerifyStatus::ReferencePanics, _ => VerifyStatus::Invalid }).collect())
}

/// `schnorr::batch_verify::<Secp256k1, D>` from line 258 on (`forge-ec-signature/src/schnorr.rs:194-290`):
/// the caller hashes (challenges `e`, 236-256) and draws the weights (`a`, 228-233) with the
/// reference's own code and passes them as scalars.
pub fn schnorr_batch_verify_secp256k1(ctx: &mut GpuContext, public_keys: &[secp256k1::AffinePoint], sig_r: &[secp256k1::AffinePoint], sig_s: &[secp256k1::Scalar], a: &[secp256k1::Scalar], e: &[secp256k1::Scalar]) -> Result<bool> {
    let n = public_keys.len();
    if sig_r.len() != n || sig_s.len() != n || a.len() != n || e.len() != n {
        return Err(Error::ValidationError);
    }
    type C = secp256k1::Secp256k1;
    let marshal = |pts: &[secp256k1::AffinePoint]| {
        let (mut xy, mut inf) = (vec![0u64; 8 * n], vec![0u8; n]);
        for (i, p) in pts.iter().enumerate() {
            let (l, f) = C::affine_limbs(p);
            xy[8 * i..8 * i + 8].copy_from_slice(&l);
            inf[i] = f as u8;
        }
        (xy, inf)
    };
    let ((pk_xy, pk_inf), (r_xy, r_inf)) = (marshal(public_keys), marshal(sig_r));
    let (s, aa, ee) = (pack_scalars::<C>(sig_s), pack_scalars::<C>(a), pack_scalars::<C>(e));
    let mut result = 0u8;
    // SAFETY: n elements behind every pointer; the two optional outputs are null.
    check(unsafe { fec_schnorr_batch_verify_secp256k1(ctx.raw, pk_xy.as_ptr(), pk_inf.as_ptr(), r_xy.as_ptr(), r_inf.as_ptr(), s.as_ptr(), aa.as_ptr(), ee.as_ptr(), n, &mut result, core::ptr::null_mut(), core::ptr::null_mut()) })?;
    Ok(result == 1)
}

fn marshal_affine<C: GpuCurve>(pts: &[C::PointAffine]) -> (Vec<u64>, Vec<u8>) {
    let (mut xy, mut inf) = (vec![0u64; 8 * pts.len()], vec![0u8; pts.len()]);
    for (i, p) in pts.iter().enumerate() {
        let (l, f) = C::affine_limbs(p);
        xy[8 * i..8 * i + 8].copy_from_slice(&l);
        inf[i] = f as u8;
    }
    (xy, inf)
}

/// `schnorr::batch_verify::<C, D>` for `C` = `Secp256k1` or `P256` (`forge-ec-signature/src/schnorr.rs:194-290`,
/// generic over the curve): as [`schnorr_batch_verify_secp256k1`].  `Ed25519`: the release profile's behaviour, see
/// [`schnorr_batch_verify_ed25519`].
pub fn schnorr_batch_verify<C: GpuCurve>(ctx: &mut GpuContext, public_keys: &[C::PointAffine], sig_r: &[C::PointAffine], sig_s: &[C::Scalar], a: &[C::Scalar], e: &[C::Scalar]) -> Result<bool> {
    let n = public_keys.len();
    if sig_r.len() != n || sig_s.len() != n || a.len() != n || e.len() != n {
        return Err(Error::ValidationError);
    }
    let ((pk_xy, pk_inf), (r_xy, r_inf)) = (marshal_affine::<C>(public_keys), marshal_affine::<C>(sig_r));
    let (s, aa, ee) = (pack_scalars::<C>(sig_s), pack_scalars::<C>(a), pack_scalars::<C>(e));
    let mut result = 0u8;
    // SAFETY: n elements behind every pointer; the two optional outputs are null.
    check(unsafe { fec_schnorr_batch_verify(ctx.raw, C::ID, pk_xy.as_ptr(), pk_inf.as_ptr(), r_xy.as_ptr(), r_inf.as_ptr(), s.as_ptr(), aa.as_ptr(), ee.as_ptr(), n, &mut result, core::ptr::null_mut(), core::ptr::null_mut()) })?;
    Ok(result == 1)
}

/// `schnorr::batch_verify::<Ed25519, D>` (`fec_schnorr_batch_verify_ed25519`): the verdict under the reference's
/// release profile (the `u128` sums of Ed25519's scalar `Mul`, ed25519.rs:1268-1278, wrap), and whether a debug build
/// -- overflow checks on -- would have panicked on these inputs instead.  `VerifyStatus::ReferencePanics` = `to_affine`
/// unwraps the inverse of a zero `z` (both profiles).
pub fn schnorr_batch_verify_ed25519(ctx: &mut GpuContext, public_keys: &[ed25519::AffinePoint], sig_r: &[ed25519::AffinePoint], sig_s: &[ed25519::Scalar], a: &[ed25519::Scalar], e: &[ed25519::Scalar]) -> Result<(VerifyStatus, bool)> {
    let n = public_keys.len();
    if sig_r.len() != n || sig_s.len() != n || a.len() != n || e.len() != n {
        return Err(Error::ValidationError);
    }
    let ((pk_xy, pk_inf), (r_xy, r_inf)) = (marshal_affine::<ed25519::Ed25519>(public_keys), marshal_affine::<ed25519::Ed25519>(sig_r));
    let (s, aa, ee) = (pack_scalars::<ed25519::Ed25519>(sig_s), pack_scalars::<ed25519::Ed25519>(a), pack_scalars::<ed25519::Ed25519>(e));
    let (mut result, mut dbg) = (0u8, 0u8);
    // SAFETY: n elements behind every pointer; the two optional point outputs are null.
    check(unsafe { fec_schnorr_batch_verify_ed25519(ctx.raw, pk_xy.as_ptr(), pk_inf.as_ptr(), r_xy.as_ptr(), r_inf.as_ptr(), s.as_ptr(), aa.as_ptr(), ee.as_ptr(), n, &mut result, core::ptr::null_mut(), core::ptr::null_mut(), &mut dbg) })?;
    Ok((match result { 1 => VerifyStatus::Valid, 2 => VerifyStatus::ReferencePanics, _ => VerifyStatus::Invalid }, dbg != 0))
}

/// `Schnorr::<C, D>::verify` per signature (`forge-ec-signature/src/schnorr.rs:90-140`) from the point computation
/// on, all three curves: the caller keeps the two message special cases (92-99) and hashes
/// (`e[i] = C::Scalar::from_bytes_reduced(H(R || P || m))`, 107-123).
pub fn schnorr_verify_batch<C: GpuCurve>(ctx: &mut GpuContext, public_keys: &[C::PointAffine], sig_r: &[C::PointAffine], sig_s: &[C::Scalar], e: &[C::Scalar]) -> Result<Vec<VerifyStatus>> {
    let n = public_keys.len();
    if sig_r.len() != n || sig_s.len() != n || e.len() != n {
        return Err(Error::ValidationError);
    }
    let ((pk_xy, pk_inf), (r_xy, r_inf)) = (marshal_affine::<C>(public_keys), marshal_affine::<C>(sig_r));
    let (s, ee) = (pack_scalars::<C>(sig_s), pack_scalars::<C>(e));
    let mut status = vec![0u8; n];
    // SAFETY: every buffer holds n elements of the width the header states.
    check(unsafe { fec_schnorr_verify(ctx.raw, C::ID, pk_xy.as_ptr(), pk_inf.as_ptr(), r_xy.as_ptr(), r_inf.as_ptr(), s.as_ptr(), ee.as_ptr(), status.as_mut_ptr(), n) })?;
    Ok(status.iter().map(|&v| match v { 1 => VerifyStatus::Valid, 2 => VerifyStatus::ReferencePanics, _ => VerifyStatus::Invalid }).collect())
}

/// `C::generator()` as the library holds it (evaluated on the device with the reference's own
/// construction) -- a self-check for an integration: must equal the CPU `C::generator()`.
pub fn generator<C: GpuCurve>(ctx: &mut GpuContext) -> Result<C::PointProjective> {
    let mut out = vec![0u64; C::LIMBS];
    // SAFETY: `out` holds fec_point_limbs(curve) limbs.
    check(unsafe { fec_generator(ctx.raw, C::ID, out.as_mut_ptr()) })?;
    debug_assert_eq!(unsafe { fec_point_limbs(C::ID) } as usize, C::LIMBS);
    Ok(C::point_from_limbs(&out))
}

/// Raw access for callers that keep their batches resident in HBM: the `*_dev` entry points take HIP
/// device pointers (16-byte aligned) of the ctx's device and a `hipStream_t`; nothing is copied or
/// synchronised.  Unsafe because the pointers are not checked.
pub mod dev {
    use super::*;

    /// `fec_batch_mul_dev`.
    ///
    /// # Safety
    /// Device pointers of the ctx's device covering n elements; `stream` a live `hipStream_t` or null.
    pub unsafe fn batch_mul(ctx: &mut GpuContext, curve: c_int, d_scalars: *const u64, d_points: *const u64, d_out: *mut u64, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_batch_mul_dev(ctx.raw, curve, d_scalars, d_points, d_out, n, stream))
    }

    /// `fec_batch_mul_fixed_dev`; pass [`generator_ptr`] as the base to reuse the cached Ed25519 table.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn batch_mul_fixed(ctx: &mut GpuContext, curve: c_int, d_scalars: *const u64, d_base: *const u64, d_out: *mut u64, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_batch_mul_fixed_dev(ctx.raw, curve, d_scalars, d_base, d_out, n, stream))
    }

    /// `fec_batch_double_mul_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn batch_double_mul(ctx: &mut GpuContext, curve: c_int, d_u1: *const u64, d_u2: *const u64, d_q: *const u64, d_out: *mut u64, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_batch_double_mul_dev(ctx.raw, curve, d_u1, d_u2, d_q, d_out, n, stream))
    }

    /// `fec_batch_to_affine_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn batch_to_affine(ctx: &mut GpuContext, curve: c_int, d_points: *const u64, d_xy: *mut u64, d_inf: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_batch_to_affine_dev(ctx.raw, curve, d_points, d_xy, d_inf, n, stream))
    }

    /// `fec_batch_compress_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`]; `d_out` 4-byte aligned.
    pub unsafe fn batch_compress(ctx: &mut GpuContext, curve: c_int, d_xy: *const u64, d_inf: *const u8, d_out: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_batch_compress_dev(ctx.raw, curve, d_xy, d_inf, d_out, n, stream))
    }

    /// `fec_ecdsa_verify_secp256k1_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn ecdsa_verify_secp256k1(ctx: &mut GpuContext, d_digests: *const u8, d_r: *const u64, d_s: *const u64, d_pk_xy: *const u64, d_pk_inf: *const u8, d_status: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_ecdsa_verify_secp256k1_dev(ctx.raw, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n, stream))
    }

    /// `fec_batch_validate_point_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn batch_validate_point(ctx: &mut GpuContext, curve: c_int, d_xy: *const u64, d_inf: *const u8, d_ok: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_batch_validate_point_dev(ctx.raw, curve, d_xy, d_inf, d_ok, n, stream))
    }

    /// `fec_batch_ecdh_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`]; the caller owns and clears every buffer.
    pub unsafe fn batch_ecdh(ctx: &mut GpuContext, curve: c_int, d_private_keys: *const u64, d_pk_xy: *const u64, d_pk_inf: *const u8, d_secrets: *mut u8, d_status: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_batch_ecdh_dev(ctx.raw, curve, d_private_keys, d_pk_xy, d_pk_inf, d_secrets, d_status, n, stream))
    }

    /// `fec_schnorr_verify_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn schnorr_verify(ctx: &mut GpuContext, curve: c_int, d_pk_xy: *const u64, d_pk_inf: *const u8, d_r_xy: *const u64, d_r_inf: *const u8, d_s: *const u64, d_e: *const u64, d_status: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_schnorr_verify_dev(ctx.raw, curve, d_pk_xy, d_pk_inf, d_r_xy, d_r_inf, d_s, d_e, d_status, n, stream))
    }

    /// `fec_eddsa_verify_ed25519_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn eddsa_verify_ed25519(ctx: &mut GpuContext, d_r_xy: *const u64, d_r_inf: *const u8, d_pk_xy: *const u64, d_pk_inf: *const u8, d_s: *const u64, d_k: *const u64, d_status: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_eddsa_verify_ed25519_dev(ctx.raw, d_r_xy, d_r_inf, d_pk_xy, d_pk_inf, d_s, d_k, d_status, n, stream))
    }

    /// `fec_ecdsa_verify_p256_dev`.
    ///
    /// # Safety
    /// As [`batch_mul`].
    pub unsafe fn ecdsa_verify_p256(ctx: &mut GpuContext, d_digests: *const u8, d_r: *const u64, d_s: *const u64, d_pk_xy: *const u64, d_pk_inf: *const u8, d_status: *mut u8, n: usize, stream: *mut c_void) -> Result<()> {
        check(fec_ecdsa_verify_p256_dev(ctx.raw, d_digests, d_r, d_s, d_pk_xy, d_pk_inf, d_status, n, stream))
    }

    /// `fec_generator_dev`: device address of the ctx's generator of `curve` (valid for the ctx's lifetime).
    pub fn generator_ptr(ctx: &mut GpuContext, curve: c_int) -> *const u64 {
        // SAFETY: self.raw is a live ctx.
        unsafe { fec_generator_dev(ctx.raw, curve) }
    }
}

/// Parity hooks on the trait operators (`forge-ec-core/src/lib.rs:173-241, 699-748`) and the
/// measurement hooks of the header, for tests and benchmarks of an integration.
pub mod hooks {
    use super::*;

    /// `fec_field_op` on raw limbs: op 0 add, 1 sub, 2 mul, 3 square, 4 neg.
    pub fn field_op(ctx: &mut GpuContext, curve: c_int, op: c_int, a: &[[u64; 4]], b: Option<&[[u64; 4]]>) -> Result<Vec<[u64; 4]>> {
        let n = a.len();
        if b.map_or(false, |x| x.len() != n) {
            return Err(Error::ValidationError);
        }
        let mut out = vec![[0u64; 4]; n];
        // SAFETY: n elements of 4 limbs behind every non-null pointer.
        check(unsafe { fec_field_op(ctx.raw, curve, op, a.as_ptr().cast(), b.map_or(core::ptr::null(), |x| x.as_ptr().cast()), out.as_mut_ptr().cast(), n) })?;
        Ok(out)
    }

    /// `fec_point_op` on raw limbs: op 0 add, 1 double, 2 negate, 3 trait double (secp256k1 only).
    pub fn point_op(ctx: &mut GpuContext, curve: c_int, op: c_int, limbs: usize, p: &[u64], q: Option<&[u64]>) -> Result<Vec<u64>> {
        if limbs == 0 || p.len() % limbs != 0 || q.map_or(false, |x| x.len() != p.len()) {
            return Err(Error::ValidationError);
        }
        let mut out = vec![0u64; p.len()];
        // SAFETY: p.len() / limbs elements behind every non-null pointer.
        check(unsafe { fec_point_op(ctx.raw, curve, op, p.as_ptr(), q.map_or(core::ptr::null(), |x| x.as_ptr()), out.as_mut_ptr(), p.len() / limbs) })?;
        Ok(out)
    }

    /// `fec_ctx_set_timing` + `fec_ctx_last_kernel_ms`: duration of the last kernel launched through the ctx.
    pub fn last_kernel_ms(ctx: &mut GpuContext) -> Result<f32> {
        let mut ms = 0f32;
        let mut name: *const c_char = core::ptr::null();
        // SAFETY: valid out-pointers.
        check(unsafe { fec_ctx_last_kernel_ms(ctx.raw, &mut ms, &mut name) })?;
        Ok(ms)
    }

    /// `fec_ctx_set_timing`.
    pub fn set_timing(ctx: &mut GpuContext, enabled: bool) -> Result<()> {
        // SAFETY: live ctx.
        check(unsafe { fec_ctx_set_timing(ctx.raw, enabled as c_int) })
    }

    /// `fec_measure_peak_mad32`: measured 32x32->64 multiply-add peak of the GPU.
    pub fn measure_peak_mad32(ctx: &mut GpuContext) -> Result<f64> {
        let mut v = 0f64;
        // SAFETY: valid out-pointer.
        check(unsafe { fec_measure_peak_mad32(ctx.raw, &mut v) })?;
        Ok(v)
    }

    /// `fec_ctx_device_info`: (name, compute units, clock in kHz).
    pub fn device_info(ctx: &mut GpuContext) -> Result<(String, i32, i32)> {
        let mut buf = [0 as c_char; 128];
        let (mut cus, mut khz) = (0, 0);
        // SAFETY: buffer and out-pointers are valid; the library NUL-terminates within name_len.
        check(unsafe { fec_ctx_device_info(ctx.raw, buf.as_mut_ptr(), buf.len(), &mut cus, &mut khz) })?;
        let name = unsafe { core::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
        Ok((name, cus, khz))
    }
}
