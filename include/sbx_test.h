/* include/sbx_test.h — TEST HOOKS of libsbx.  Not part of the drop-in surface.
 *
 * include/sbx.h is what a host binds (the replacement of the reference's per-pixel mainImage() loop, src/main.h:6-53).  The entry
 * points below exist so that tests/ and tools/ can (a) run the cross-check builds of the kernels against the default ones,
 * (b) evaluate single functions of the math spec on the device against the CPU oracle, (c) drive paths that ordinary frames
 * never reach (the fault path, the witness re-run, a store-exchange wait that times out).  They are exported by the same
 * libsbx.so, keep no compatibility promise across SBX_ABI_VERSION, and no reference interface corresponds to them.
 */
#ifndef SBX_TEST_H
#define SBX_TEST_H

#include "sbx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic knob: 0 = default kernels; 1 = the plain cross-check kernels where one exists: APP_CLOUDS with every
 * lane hashing its own lattice corners (no cache, no staging, no tables); APP_EGG / APP_SDF_AO / APP_VINYL with every
 * member of the SDF union evaluated everywhere (no culling); APP_PLANET without its exact skips.
 * 2 / 3 = the default kernels, except APP_EGG / APP_SDF_AO / APP_VINYL(_GPU): 2 = their square-root witness with the recording
 * edge raised to 1.0, so that the re-run path (csrc/sbx_sdf.h, Wit) executes on ordinary frames; 3 = the culled kernels with the
 * IEEE roots only.
 * All variants are specified to produce identical bits (tests/test_gpu_parity.py sweeps them against each other). */
int sbx_set_variant(sbx_ctx* ctx, int variant);

/* Device evaluation of the math spec, elementwise over device arrays (for parity tests):
 * fn in {"sin","cos","tan","exp","pow","acos","atan2","hash","div","div_rd","exp_h13","pow_h","sqrt_n","sqrt_ieee","exp_reg","exp_reg_plain","exp_reg64","exp_reg64_plain","exp_small","exp_small_plain","exp_reg4k","sin_b40","div3","sqrt_rs","divn","srgb_pow","pow_spec"}; b may be NULL
 * for unary fns ("exp_reg*": kernel-internal forms of exp — 32-entry table / degree 6 and 64-entry / degree 5, each with and without
 * the three-address asm — used by the regular-frame k_clouds and by k_atmosphere's density terms, equal to
 * "exp" for |x| <= 80; "exp_reg4k": the 4096-entry / degree-3 form of k_atmosphere's density terms, equal to "exp" for |x| <= 80;
 * "sin_b40": sin with a degree-15 polynomial, the hash passes' form, equal to "sin" for |x| <= 2^40;
 * "exp_small*": the degree-8 polynomial without argument reduction that the regular-frame k_clouds uses when
 * every argument lies in [-0.205, -0], equal to "exp" on that whole interval and at +0).
 * "div3" = a/b as q0 = a * RN(1/b), q = fma(fma(-q0, b, a), RN(1/b), q0): equal to "div" away from overflow and underflow;
 * "divn" = a/b through v_rcp_f32, one Newton step and div3's three instructions (equal to "div" away from overflow / underflow);
 * "pow_spec" = pow exactly as stated in the oracle ("pow" is the device's shorter instruction sequence for the same operations);
 * "srgb_pow" = pow(x, 1/2.2f) in the short form to_srgb uses on the device (equal to "pow" with b = 1/2.2f on all 2^32 arguments);
 * "sqrt_rs" = v_rsq_f32 and one corrected step: equal to "sqrt_ieee" for finite x >= 2^-102;
 * "div" = IEEE a/b, "div_rd" = the same quotient through the binary64 reciprocal of b (must be identical). */
int sbx_math_eval(sbx_ctx* ctx, const char* fn, const float* a, const float* b, float* out,
                  size_t n, void* stream);

int sbx_multi_set_variant(sbx_multi* m, int variant);

/* sbx_noise_eval (sbx.h) also answers three names that exist for the parity tests of the recorded-domain forms
 * (csrc/sbx_witness.h): "normalize" = v / length(v) in the IEEE form, "wit_normalize" = the fast form (sqrt_rs_, v_rcp_f32 + one
 * Newton step, three div3_), "wit_record" = out[0] 1 where that form's domain record fires. */

/* SampleLevel(linear, wrap, lod 0).r of a size^3 RGBA32F device volume at n points (xyz interleaved, device):
 * the texture-filter spec on its own, for parity tests. */
int sbx_tex3d_eval(sbx_ctx* ctx, int size, const float* rgba, const float* xyz, float* out, size_t n, void* stream);


/* One wave through the fault path of the hash cache (sbx_fault_status then reports SBX_ERR_FAULT until sbx_clear_fault). */
int sbx_debug_raise_fault(sbx_ctx* ctx, void* stream);

/* The store exchange's waits (sbx_shared_frame_begin / _end) give up after this many milliseconds and raise the device's fault
 * word (default 10 000); tests shorten it to see the fault. */
int sbx_shared_set_timeout_ms(sbx_shared* s, int ms);

/* MODEL of what RCCL's point-to-point receive kernels cost the frame's owner while the peers' slabs land (bench.py --emulate-ranks,
 * tools/strip_scaling.py; nothing in the product calls it): `workgroups` workgroups of 256 threads stay resident for
 * `duration_us` microseconds and copy `bytes` bytes from `src` to `dst` (device memory, 16-byte aligned) at that pace — the CUs
 * a grouped ncclRecv from N-1 peers holds for the time the links need, and the HBM writes of the landing. */
int sbx_model_landing(sbx_ctx* ctx, const void* src, void* dst, size_t bytes, int workgroups, float duration_us, void* stream);

/* The dispatch-order table of an app (csrc/sbx_capi.hip TileOrder: a full launch's tiles sorted by the cost earlier frames measured,
 * longest first — a hint about cost, never about pixels): tables built so far for the current launch shape (0 = the launches still
 * run in plain order), launches since the last one, and — if `table` is not NULL and one exists — the current table copied to the
 * host (`capacity` words; returns the number of tiles, or a negative sbx_status).  For the test that every table is a permutation. */
int sbx_debug_tile_order(sbx_ctx* ctx, int app, int* tables_built, int* launches_since, unsigned* table, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* SBX_TEST_H */
