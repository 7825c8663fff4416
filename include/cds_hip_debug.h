/*
 * cds_hip_debug.h -- diagnostic entry points of libcds_hip.so used by the CPU-side tests of the host
 * logic (tests/test_host_logic.py).  Not part of the plugin surface; they compute nothing on a GPU.
 *
 * The device encodes a target pixel as hue_code = class << 29 | bits(fl32(second) * rcp[max]) >> 1 and a
 * mask pixel as two inclusive code intervals (lo, len); the device predicate is
 *   (code - lo1 <= len1) || (code - lo2 <= len2)       (unsigned)
 * which must equal the reference's  max(target) > dataThreshold && calculatePixelGap(mask, target) <= zTolerance
 * (API/cds/PixelMatchColorDepthSearchAlgorithm.java:251-258).  `rcp_table` is any 256-entry reciprocal
 * table (on a GPU it is that device's v_rcp_f32, see cds_debug_device_rcp).
 */
#ifndef CDS_HIP_DEBUG_H
#define CDS_HIP_DEBUG_H
#include "cds_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
int cds_debug_intervals(const float *rcp_table, int32_t r, int32_t g, int32_t b, double z_tolerance, uint32_t *out4);
uint32_t cds_debug_hue_code(const float *rcp_table, int32_t r, int32_t g, int32_t b, int32_t threshold);
int cds_debug_device_rcp(const cds_context *ctx, float *out256);
/* How mask `mask_index` is laid out for the sparse scan: out8 = { slabs (0 = no device pixel, -1 = not representable),
 * mask rows per slab, LDS bytes of one workgroup, bounding-box columns, bounding-box rows, interval records + 1,
 * bytes per plane cell, interval table in LDS }. */
int cds_maskset_sparse_layout(const cds_maskset *ms, int32_t mask_index, int32_t *out8);
/* How the masks of a set were grouped for the multi-mask sparse scan: groups of consecutive masks, how many of them are
 * scanned by sparse_multi_kernel (the others fall back to one launch per mask) and how many masks those hold. */
int cds_maskset_sparse_groups(const cds_maskset *ms, int32_t *n_groups, int32_t *n_multi_groups, int32_t *n_multi_masks);
/* Stage 1 of the sparse scan tests one bit of a per-cell word: bit cds_debug_band_of(class 1..6, second, max) of the pixel
 * (its hue band, (class * 5 + floor(5 * ratio)) mod 32 with the ratio as the device computes it) against the union of
 * cds_debug_band_bits(lo, len) over the intervals of the mask pixels that can reach the cell.  The property that makes the
 * filter exact: a code inside [lo, lo + len] always has its band's bit set in cds_debug_band_bits(lo, len). */
int cds_debug_band_bits(uint32_t lo, uint32_t len, uint32_t *bits);
int32_t cds_debug_band_of(const float *rcp_table, int32_t cls, int32_t second, int32_t max);
/* Calibration of the instruction-issue roofline (bench.py): ns one SIMD needs per wave64 integer VALU instruction with 8
 * waves resident, measured on this device with independent v_add / v_xor chains (tools/valubench/valubench.hip, mode 0). */
int cds_debug_valu_issue_ns(cds_context *ctx, double *ns_per_wave_instruction_per_simd);
/* Measurement of the sparse scan's memory floor (bench.py): while `on` is non-zero, single-mask sparse searches made through THIS
 * CONTEXT launch the scan kernel's LOADS-ONLY twin -- same work distribution, run headers, record loads and prefetch distance, no
 * examination of the records; the scores it writes are meaningless.  Only for the one-workgroup-per-CU layout with one-byte
 * planes (CDS_ERR_UNSUPPORTED otherwise).  The flag lives in the context: searches through any other context (another thread's,
 * another GPU's, a mask set's pair slots) are never affected.  cds_debug_sparse_load_only_state reads it back. */
int cds_debug_sparse_load_only(cds_context *ctx, int32_t on);
int32_t cds_debug_sparse_load_only_state(const cds_context *ctx);
/* Which single-mask kernel the last sparse search of this mask set launched: 0 = over the 4-byte records, 1 = over the library's 16-bit
 * stage-1 stream (half the bytes per visited record), -1 = none yet; pilot_ratio = the share of visited records that reached an evaluation
 * round in the set's last pilot (the first 4,096 targets of the first search of >= 16,384 targets of a library holding the stream; the
 * stream is used when it is <= 0.05), -1 = no pilot yet.  CDS_SPARSE_SCAN16=1 / CDS_SPARSE_SCAN32=1 force a kernel. */
int cds_debug_maskset_scan_kernel(const cds_maskset *ms, int32_t *kernel, double *pilot_ratio);
/* Reference point of the same measurement: a plain streaming read of the library's whole record arena (16 bytes per lane, four loads in
 * flight, fastest of three launches) -- the rate THIS device's HBM delivers for the simplest pattern over the same memory. */
int cds_debug_sparse_stream_read(const cds_sparse_library *slib, double *gb_per_s, double *bytes_read);
#ifdef __cplusplus
}
#endif
#endif
