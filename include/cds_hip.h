/*
 * cds_hip.h -- C ABI of libcds_hip.so: the MI355X (gfx950) color-depth MIP search core.
 *
 * This is the drop-in boundary for ONE path of JaneliaSciComp/colormipsearch (v3.1.1): what its
 * `ColorDepthSearchAlgorithm` plugin surface computes.  A JVM host binds these entry points through
 * JNI / java.lang.foreign (INTEGRATION.md shows the stub); the Python mirror (colormipsearch_amd/cds.py)
 * binds them through ctypes; the C++ mirror is include/cds_hip.hpp.
 *
 * Citations: API/ = colormipsearch-api/src/main/java/org/janelia/colormipsearch/ of the reference.
 *
 * Conventions: plain C, opaque handles, int status returns (0 = CDS_OK), caller-allocated outputs, no
 * ownership transfer, no exceptions across the ABI.  cds_last_error() returns a thread-local message for
 * the last failing call on the calling thread.  Handles are immutable after creation and may be used
 * from several host threads at once (the reference shares one algorithm object between all worker
 * threads: tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:88-105).
 * Image sizes: 8..2000 columns, 1..1536 rows, at most 2,000,000 pixels (the brain alignment spaces' 1210 x 566 and the VNC ones' 573 x 1119 fit); the
 * create calls of libraries, mask sets and shards answer CDS_ERR_UNSUPPORTED beyond that.
 * There is NO CPU fallback: every call that computes fails with CDS_ERR_NO_DEVICE when no gfx950 GPU
 * is usable.
 *
 * Images: packed RGB, uint8, R,G,B byte order, row-major, row pitch 3*W -- the byte layout of the
 * reference's ColorImageArray (API/imageprocessing/ColorImageArray.java:4-38).  Gradient images:
 * uint16 row-major (the reference's ShortImageArray; widen a GRAY8 gradient on the host).
 */
#ifndef CDS_HIP_H
#define CDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDS_OK 0
#define CDS_ERR_INVALID_ARGUMENT (-1) /* the reference throws IllegalArgumentException here */
#define CDS_ERR_NO_DEVICE (-2)
#define CDS_ERR_HIP (-3)
#define CDS_ERR_OUT_OF_MEMORY (-4)
#define CDS_ERR_UNSUPPORTED (-5)

#define CDS_MAX_RECTS 8
#define CDS_N_TRANSFORMS 18 /* 9 xy-shifts x (plain, mirrored) */

typedef struct cds_context cds_context;   /* one GPU: device, default stream, tables          */
typedef struct cds_library cds_library;   /* a resident shard of target MIPs (+ variants)      */
typedef struct cds_sparse_library cds_sparse_library; /* a resident shard of targets kept as signal-pixel lists */
typedef struct cds_maskset cds_maskset;   /* n prepared PixelMatchColorDepthSearchAlgorithm    */
typedef struct cds_shapeset cds_shapeset; /* n prepared Shape2DMatchColorDepthSearchAlgorithm  */

/* Excluded ("label") region: half-open rectangle [x0,x1) x [y0,y1).  Replaces the predicate of
 * API/imageprocessing/ImageRegionDefinition.java:5-13; every caller in the reference builds a union of
 * rectangles (tools cmd/AbstractColorDepthMatchArgs.java:101-119). */
typedef struct cds_rect {
    int32_t x0, y0, x1, y1;
} cds_rect;

/* Parameters of ColorDepthSearchAlgorithmProviderFactory.createPixMatchCDSAlgorithmProvider
 * (API/cds/ColorDepthSearchAlgorithmProviderFactory.java:30-74) + the per-mask query threshold. */
typedef struct cds_pixmatch_params {
    int32_t mask_threshold; /* queryThreshold                                            */
    int32_t data_threshold; /* targetThreshold                                           */
    double z_tolerance;     /* pixColorFluctuation / 100  (Factory:55-56)                */
    int32_t xy_shift;       /* 0 or 2; odd -> CDS_ERR_INVALID_ARGUMENT (Factory:58-60);  */
                            /* >= 4 -> CDS_ERR_UNSUPPORTED (the reference overruns its   */
                            /* own array there, PixelMatch...Algorithm.java:114-125)     */
    int32_t mirror;         /* mirrorMask                                                */
} cds_pixmatch_params;

const char *cds_last_error(void);
const char *cds_version(void);

/* ---- context ------------------------------------------------------------------------------- */
/* A context owns a device, its streams and a cache of released device blocks: the device memory of mask sets, match lists and
 * per-call temporaries is recycled through the context (up to 6 GB idle) instead of going back to the driver, because hipFree
 * waits for every stream of the device -- a caller that pipelines mask sets would stall behind the running scan after each one.
 * cds_context_destroy returns the cache; mask sets, libraries and match lists of a context are destroyed BEFORE it. */
int cds_context_create(int device_id, cds_context **out);
void cds_context_destroy(cds_context *ctx);
/* Use an externally owned hipStream_t (e.g. torch's current stream) for all later launches; NULL is
 * HIP's legacy default stream.  cds_context_reset_stream goes back to the context's own stream. */
int cds_context_set_stream(cds_context *ctx, void *hip_stream);
int cds_context_reset_stream(cds_context *ctx);
int cds_context_synchronize(cds_context *ctx);
/* The idle blocks of the context's cache go back to the driver now (after the device has finished the work enqueued so far);
 * *freed_bytes (may be NULL) = what was returned.  cds_context_set_cache_limit caps what the cache may keep idle (bytes; 0 = keep
 * nothing: every release is a hipFree; default 6 GB, or CDS_CACHE_LIMIT_MB at context creation) and trims down to it.  Every device
 * allocation of the library that meets hipErrorOutOfMemory drops the caches of that device's contexts and tries once more, so the
 * retained memory is never the reason a library or mask set cannot be created. */
int cds_context_trim(cds_context *ctx, uint64_t *freed_bytes);
int cds_context_set_cache_limit(cds_context *ctx, uint64_t bytes);

/* ---- target library (replaces the host-side MIP cache, tools cmd/CachedMIPsUtils.java:60-100) -- */
/* Device bytes for `capacity` targets of W x H (padded row pitch, see DESIGN.md "data layout"). */
size_t cds_library_bytes(int32_t W, int32_t H, int64_t capacity);
int cds_library_create(cds_context *ctx, int32_t W, int32_t H, int64_t capacity, cds_library **out);
/* Adopt caller-owned device memory (e.g. a torch tensor) of at least cds_library_bytes() bytes; it need not be
 * initialised: every upload / generate call writes whole padded rows of the slots it fills. */
int cds_library_wrap(cds_context *ctx, int32_t W, int32_t H, int64_t capacity, void *device_ptr,
                     size_t device_bytes, cds_library **out);
void cds_library_destroy(cds_library *lib);
/* Image size and slot count of a library (any output may be NULL): what a binding checks its buffers against. */
int cds_library_dims(const cds_library *lib, int32_t *W, int32_t *H, int64_t *capacity);
/* Host -> device: `count` images, packed RGB pitch 3*W, into slots [first, first+count). */
int cds_library_upload(cds_library *lib, int64_t first, int64_t count, const uint8_t *rgb);
int cds_library_download(cds_library *lib, int64_t first, int64_t count, uint8_t *rgb);
/* Variant images of the same slots for the shape scorer (ComputeFileType.GradientImage / ZGapImage,
 * API/model/ComputeFileType.java:5-28).  Either pointer may be NULL to leave that variant unchanged. */
int cds_library_upload_variants(cds_library *lib, int64_t first, int64_t count, const uint16_t *grad,
                                const uint8_t *zgap_rgb);
int cds_library_download_variants(cds_library *lib, int64_t first, int64_t count, uint16_t *grad, uint8_t *zgap_rgb);
/* The reference's shape scorer takes a GRAY8 or a GRAY16 gradient image through ImageArray.get
 * (API/imageprocessing/ByteImageArray.java:3-16, ShortImageArray.java): an 8-bit gradient (pitch W) is widened into the u16
 * plane the scorer reads -- same values, same scores. */
int cds_library_upload_gradient_u8(cds_library *lib, int64_t first, int64_t count, const uint8_t *grad8);
/* Fill slots [first_slot, first_slot+count) on the device with synthetic images number
 * first_index.. of the stream `seed` (DESIGN.md "synthetic library").  kind: 0 sparse-lut target,
 * 1 dense-uniform target, 2 arbor mask.  with_variants != 0 also generates gradient + zgap (kind 0). */
int cds_library_generate_synthetic(cds_library *lib, uint64_t seed, int64_t first_index, int64_t first_slot,
                                   int64_t count, int32_t kind, int32_t with_variants);
/* The same for an arbitrary list of stream indices: slot first_slot + i gets image indices[i] (the synthetic stand-in for loading
 * just the selected targets of a gradient-score step). */
int cds_library_generate_synthetic_list(cds_library *lib, uint64_t seed, const int64_t *indices, int64_t first_slot,
                                        int64_t count, int32_t kind, int32_t with_variants);

/* ---- pixel-match scan (kernel 1) ----------------------------------------------------------------
 * cds_maskset_create == n x `new PixelMatchColorDepthSearchAlgorithm(...)`
 * (API/cds/PixelMatchColorDepthSearchAlgorithm.java:29-101 with getMaskPosArray
 * API/cds/AbstractColorDepthSearchAlgorithm.java:96-126).  masks_rgb: n_masks x (3*W*H) bytes. */
int cds_maskset_create(cds_context *ctx, int32_t n_masks, const uint8_t *masks_rgb, int32_t W, int32_t H,
                       const cds_pixmatch_params *params, const cds_rect *excluded, int32_t n_excluded,
                       cds_maskset **out);
void cds_maskset_destroy(cds_maskset *ms);
/* Masks in the set (bindings size their score arrays with it). */
int32_t cds_maskset_size(const cds_maskset *ms);
/* getQuerySize() per mask (API/cds/ColorDepthSearchAlgorithm.java:27). */
int cds_maskset_query_sizes(const cds_maskset *ms, int32_t *sizes);
/* Scan slots [first, first+count) of the library with every mask of the set ==
 * calculateMatchingScore per (mask,target) (PixelMatch...Algorithm.java:166-219).
 * scores[m*count + t] = matching pixels, mirrored[...] = bestScoreMirrored.  Host outputs; blocks. */
int cds_search(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count, int32_t *scores,
               uint8_t *mirrored);
/* Same, asynchronous on the context's stream, result left on the device:
 * d_scores[m*count + t] = score | (mirrored << 31)  (int32, device pointer). */
int cds_search_device(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count,
                      int32_t *d_scores);
/* Diagnostic: all CDS_N_TRANSFORMS match counts per pair, counts[(m*count+t)*18 + k]; k<9 plain
 * shifts in the reference's order (dx,dy)=(-2,-2),(-2,0),(-2,2),(0,-2),...,(2,2), k>=9 mirrored.
 * With xy_shift 0 only k=0 (and 9) are used. */
int cds_search_counts(const cds_maskset *ms, const cds_library *lib, int64_t first, int64_t count, int32_t *counts);
/* The reference's per-pair entry point, calculateMatchingScore(target) (API/cds/PixelMatchColorDepthSearchAlgorithm.java:
 * 166-219; size mismatch -> CDS_ERR_INVALID_ARGUMENT with the reference's message :171-175): one host image against mask
 * `mask_index`.  The mask set keeps one resident slot for it (0.16 ms per call at 1210x566). */
int cds_score_pair(const cds_maskset *ms, int32_t mask_index, const uint8_t *target_rgb, int32_t W, int32_t H,
                   int32_t *score, uint8_t *mirrored);
/* PixelMatchScore.getNormalizedScore (API/cds/PixelMatchScore.java:21-24) and
 * ColorMIPSearch.isMatch (API/cds/ColorMIPSearch.java:42-45). Pure host arithmetic. */
float cds_pixel_match_normalized(int32_t score, int32_t query_size);
int cds_is_match(int32_t score, int32_t query_size, double pct_positive_pixels);

/* ---- sparse target library (pixel-match scan only) -------------------------------------------------
 * A color-depth MIP is a few percent signal on black, and a pixel without a strict maximum channel has no hue
 * class and matches nothing (API/cds/AbstractColorDepthSearchAlgorithm.java:195-225, 227-232).  A sparse library
 * keeps per image only its signal pixels (4 bytes each: x, hue band, second and maximum channel -- independent of
 * every search parameter) in row-major order: ~4 % of the dense bytes, so a 1 M-image library fits one GPU.
 * It replaces the same host-side MIP cache (tools cmd/CachedMIPsUtils.java:60-100) for searches that need the
 * pixel-match score only; the shape scorer keeps reading dense slots + variants.  Results of cds_search_sparse*
 * are identical to cds_search* on the same images (tests/test_sparse_gpu.py).
 * capacity_pixels = total signal-pixel records (+ 1 per empty (strip, row) cell: at most 4 * H per image). */
int cds_sparse_library_create(cds_context *ctx, int32_t W, int32_t H, int64_t capacity_images, int64_t capacity_pixels,
                              cds_sparse_library **out);
void cds_sparse_library_destroy(cds_sparse_library *slib);
/* A GROWABLE library enlarges its arrays (by half, or to what the append needs) when an append or import finds a capacity lacking --
 * the reference's MIP cache sizes itself with use (tools cmd/CachedMIPsUtils.java:60-100), so a host that does not know its
 * library's size creates a small one and lets it grow (a growth step copies what is held device to device: transiently twice the
 * memory of the grown array).  cds_sparse_library_reserve grows to explicit capacities at once (never shrinks).  The derived 16-bit
 * stage-1 stream follows while records + stream stay under CDS_SPARSE_STREAM16_MAX_FRACTION (default 0.6) of the device's memory
 * and is dropped otherwise (searches then read the 4-byte records; results are the same). */
int cds_sparse_library_set_growable(cds_sparse_library *slib, int32_t on);
int cds_sparse_library_reserve(cds_sparse_library *slib, int64_t capacity_images, int64_t capacity_pixels);
int cds_sparse_library_capacity(const cds_sparse_library *slib, int64_t *capacity_images, int64_t *capacity_pixels, int32_t *has_stream16);
int cds_sparse_library_dims(const cds_sparse_library *slib, int32_t *W, int32_t *H);
/* Append slots [first, first+count) of a resident dense library (device -> device); the images get the next
 * indices size() .. size()+count-1.  An append call is a transaction: when it fails -- CDS_ERR_OUT_OF_MEMORY (a capacity
 * would be exceeded), CDS_ERR_UNSUPPORTED (density limit, below), a HIP error -- the library is as it was before the call. */
int cds_sparse_library_append(cds_sparse_library *slib, const cds_library *src, int64_t first, int64_t count);
/* Append `count` host images (packed RGB, pitch 3*W) through internal dense staging slots. */
int cds_sparse_library_append_host(cds_sparse_library *slib, int64_t count, const uint8_t *rgb);
/* Record layout: an image is stored in n_strips column strips of strip_width pixels (the last may be narrower), a strip
 * row by row, a row by ascending x; a (strip, row) cell without signal pixels holds one filler record. */
int cds_sparse_library_layout(const cds_sparse_library *slib, int32_t *n_strips, int32_t *strip_width);
/* Layout choice.  The sparse scan's cost grows with the records it visits; above about a quarter of the frame the dense
 * layout (cds_library + cds_search) is the faster one.  An append / import CALL whose images average more signal pixels than
 * `max_signal_fraction` of the frame (default 0.25) fails with CDS_ERR_UNSUPPORTED and a message saying so -- nothing of the
 * call is added, however it is cut into batches internally -- so that a dense library is never silently scanned the slow way;
 * >= 1.1 accepts everything. */
int cds_sparse_library_set_density_limit(cds_sparse_library *slib, double max_signal_fraction);
/* Minimum data threshold of the library (default 0).  The reference only looks at a target pixel whose brightest channel exceeds
 * the search's data threshold (PixelMatchColorDepthSearchAlgorithm.java:247-251), so a library whose searches never go below T need
 * not keep the pixels at or below T: images appended afterwards store signal pixels with max(r, g, b) > T only (on the reference's
 * LM test MIPs T = 20 drops 3 % of the records, T = 50 19 %, T = 100 39 %), and a search whose data_threshold is below the library's is refused with
 * CDS_ERR_INVALID_ARGUMENT -- results never change.  Can be raised, not lowered, once the library holds images. */
int cds_sparse_library_set_min_threshold(cds_sparse_library *slib, int32_t min_data_threshold);
int cds_sparse_library_min_threshold(const cds_sparse_library *slib, int32_t *min_data_threshold);
int64_t cds_sparse_library_size(const cds_sparse_library *slib);   /* images held  */
int64_t cds_sparse_library_pixels(const cds_sparse_library *slib); /* records held */
int cds_sparse_library_clear(cds_sparse_library *slib);
/* Diagnostic: the records of one image (bit 31 first of its cell | (x - strip start) << 22 | hue band << 16 | second << 8 | max,
 * band = class * 5 + floor(5 * second / max) in the device's arithmetic, 5..34) and its n_strips x (H+1) cell offsets; either
 * output may be NULL; *n_records is always set. */
int cds_sparse_library_image_records(const cds_sparse_library *slib, int64_t index, uint32_t *records, int64_t capacity,
                                     int64_t *n_records, uint32_t *row_offsets);
/* Diagnostic: offsets[count+1] = position of the first record of images first..first+count in the library's record
 * arena (absolute, 64-bit: a 1 M-image library holds > 2^32 records). */
int cds_sparse_library_image_offsets(const cds_sparse_library *slib, int64_t first, int64_t count, uint64_t *offsets);
/* Export / import of the record arrays, for persisting a converted library (colormipsearch_amd/mipio.py: sparse shard files).
 * Export images [first, first+count): image_offsets[count+1] (record offsets relative to the first exported image),
 * cell_offsets[count][n_strips][H+1] and the records themselves (any of the last two may be NULL; call once with NULLs to
 * size the buffers: the record count is image_offsets[count]).  Import appends `count` images given in that form; the
 * layout (n_strips, strip_width) must be this library's; the offset tables are validated on the host and the records
 * on the device (every cell non-empty, row flags, x order, classes, fillers): an inconsistent shard is refused.
 * min_data_threshold = cds_sparse_library_min_threshold of the library the records were exported from (their pixels at or below it
 * are missing): an import into a library whose own minimum data threshold is LOWER is refused with CDS_ERR_INVALID_ARGUMENT --
 * its searches between the two thresholds would silently miss pixels. */
int cds_sparse_library_export(const cds_sparse_library *slib, int64_t first, int64_t count, uint64_t *image_offsets,
                              uint32_t *cell_offsets, uint32_t *records, int64_t records_capacity);
int cds_sparse_library_import(cds_sparse_library *slib, int64_t count, const uint64_t *image_offsets, const uint32_t *cell_offsets,
                              const uint32_t *records, int32_t n_strips, int32_t strip_width, int32_t min_data_threshold);

/* ---- MIP files (host side) ------------------------------------------------------------------------
 * Decoding of the files the reference's pipeline reads -- replaces ImageArrayUtils.readImageArray
 * (API/imageprocessing/ImageArrayUtils.java:98-121; ImageJ's TIFF opener / LocalTiffDecoder.java, packBitsUncompress :229-258,
 * ImageIO for PNG).  Supported: baseline strip TIFF (8/16-bit, 1/3/4 samples, none / PackBits / LZW / Deflate, predictor 2)
 * and non-interlaced PNG (gray 8/16, RGB(A) 8, palette 8); anything else fails with a message.
 * Decoded layout: channels == 3 -> packed RGB u8 (alpha dropped, palette expanded); channels == 1 -> u8 or host-order u16.
 * Wherever an entry point below takes the PATH of an image file, `<archive>.zip!<entry>` (or `...zip!/<entry>`) names an entry of a zip archive --
 * the reference's FileData of type zipEntry (API/model/FileData.java:22-26, API/mips/NeuronMIPUtils.java:177-199): stored and deflated entries,
 * zip64, CRC checked; an entry not found under its full name is looked up by its file name alone, as the reference does (:163-170).  An
 * archive's directory is parsed once and kept (the 8 most recent archives). */
typedef struct cds_image_info {
    int32_t width, height, channels, bits;
} cds_image_info;
int cds_image_probe(const uint8_t *data, size_t size, cds_image_info *info);
int cds_image_decode(const uint8_t *data, size_t size, void *out, size_t out_bytes);
/* Append `count` image files (8-bit RGB, the library's size) to a sparse library: decoded on `threads` host threads
 * (<= 0: all cores) while the previous batch of 64 is uploaded and converted on the GPU.  Fails at the first file that
 * cannot be read / decoded (its path is in cds_last_error()); batches appended before it stay.  A density refusal
 * (CDS_ERR_UNSUPPORTED, tested per batch of 64 files) removes everything the call had appended. */
int cds_sparse_library_append_files(cds_sparse_library *slib, const char *const *paths, int64_t count, int32_t threads);
/* Kernel 2's inputs straight from files into resident dense slots [first, first + count): slot first + i gets the target image
 * target_paths[i] (8-bit RGB), its GradientImage gradient_paths[i] (8- or 16-bit gray; a NULL entry, or gradient_paths == NULL, leaves the
 * slot without a gradient: its pairs score (-1, -1), Shape2DMatchColorDepthSearchAlgorithm.java:155-158) and its ZGapImage zgap_paths[i]
 * (8-bit RGB; NULL: the slot has no zgap -- the caller may make one on the device, cds_library_make_zgap_variants, as the reference's
 * tests do).  What CalculateGradientScoresCmd does per match through NeuronMIPUtils.loadComputeFile (tools cmd/CalculateGradientScoresCmd
 * .java:541-586, API/mips/NeuronMIPUtils.java:36-103), for a whole batch: the files are decoded on `threads` host threads (<= 0: all
 * cores) into pinned memory, 64 slots at a time, while the previous 64 are uploaded.  Fails at the first file that cannot be read or
 * has another size (its path is in cds_last_error()); slots loaded before it stay. */
int cds_library_load_files(cds_library *lib, int64_t first, int64_t count, const char *const *target_paths,
                           const char *const *gradient_paths, const char *const *zgap_paths, int32_t threads);

/* ---- The variant shard: kernel 2's inputs of a library in one file, decoded on the GPU ---------------
 * CalculateGradientScoresCmd reads three image files per selected target through its variant suppliers (tools cmd/
 * CalculateGradientScoresCmd.java:541-586; NeuronMIPUtils.loadComputeFile, API/mips/NeuronMIPUtils.java:36-103) and inflates them on
 * host cores.  A shard holds the same three images of every target -- the MIP, its GradientImage, its ZGapImage -- LOSSLESSLY in codes a
 * wavefront expands (run lengths for the two RGB images; for the gradient, a distance ramp, its few non-zero second differences -- the image is
 * their 2-D prefix sum -- or two bits per pixel, raw 16-bit otherwise):
 * written once per library, then any subset of its targets is read into pinned memory and decoded into dense slots by two kernels.  The
 * slots are bit-equal to what cds_library_load_files makes of the files.  Layout: colormipsearch_amd/csrc/cds_variants.hip. */
typedef struct cds_variant_writer cds_variant_writer;
typedef struct cds_variant_shard cds_variant_shard;
#define CDS_VARIANT_HAS_GRADIENT 1u     /* per-target flags (cds_variant_shard_target_info) */
#define CDS_VARIANT_GRADIENT_DELTA2 2u  /* the gradient is stored as two bits per pixel (neither this nor DIFF2: raw) */
#define CDS_VARIANT_HAS_ZGAP 4u
#define CDS_VARIANT_RGB_SPANS 16u       /* the MIP is stored as the spans of its non-black pixels + their bytes (else run lengths) */
#define CDS_VARIANT_ZGAP_SPANS 32u      /* the same for the zgap image */
#define CDS_VARIANT_GRADIENT_DIFF2 8u   /* the gradient is stored as the non-zero second mixed differences (a distance ramp has few) */
int cds_variant_shard_create(const char *path, int32_t width, int32_t height, cds_variant_writer **out);
/* `n` targets from host memory: rgb [n][H][W][3]; gradients [n][H][W] u16 or NULL (none of them has one); zgaps like rgb or NULL.
 * Packed on `threads` host threads (<= 0: all cores), written in order: target numbers in the shard count up from 0 across calls. */
int cds_variant_shard_append(cds_variant_writer *w, int64_t n, const uint8_t *rgb, const uint16_t *gradients, const uint8_t *zgaps, int32_t threads);
/* The same from image files (formats of cds_image_decode; a NULL gradient / zgap entry, or a NULL array: that target has none).
 * Fails at the first file that cannot be read or has another size (its path is in cds_last_error()); nothing of the call is kept then. */
int cds_variant_shard_append_files(cds_variant_writer *w, int64_t n, const char *const *target_paths, const char *const *gradient_paths,
                                   const char *const *zgap_paths, int32_t threads);
/* Writes the index and the header, closes the file and releases `w` (also when it returns an error).  A file that was not finished
 * does not open. */
int cds_variant_shard_finish(cds_variant_writer *w);
int cds_variant_shard_open(const char *path, cds_variant_shard **out);
int cds_variant_shard_info(const cds_variant_shard *s, int32_t *width, int32_t *height, int64_t *n_targets, int64_t *file_bytes);
/* flags (CDS_VARIANT_*) and packed size of targets [first, first + count); either array may be NULL */
int cds_variant_shard_target_info(const cds_variant_shard *s, int64_t first, int64_t count, uint32_t *flags, uint32_t *bytes);
void cds_variant_shard_close(cds_variant_shard *s);
/* Dense slots [first_slot, first_slot + n) <- the shard's targets `targets[0..n)` (any order, repeats allowed): image, gradient and zgap
 * as stored; a target stored without a gradient / zgap leaves its slot without one (its pairs score (-1, -1) / the caller may make the
 * zgap on the device, as with cds_library_load_files).  Blobs are read on `threads` host threads (<= 0: all cores) into pinned memory, up
 * to 64 MB at a time, while the previous batch is copied and decoded.  Safe to call while another thread scores OTHER slots or another
 * library of the same context. */
int cds_library_load_variants(cds_library *lib, int64_t first_slot, const cds_variant_shard *s, const int64_t *targets, int64_t n, int32_t threads);
/* The same for the targets' IMAGES alone: of every blob only the MIP's section is read and decoded, the slots come out without gradient and zgap
 * -- what the ingest of a sparse library wants (cds_sparse_library_append from these slots: one shard then serves both kernels). */
int cds_library_load_variant_images(cds_library *lib, int64_t first_slot, const cds_variant_shard *s, const int64_t *targets, int64_t n, int32_t threads);

/* The scan over images [first, first+count) of a sparse library: same outputs as cds_search / cds_search_device /
 * cds_search_counts. */
int cds_search_sparse(const cds_maskset *ms, const cds_sparse_library *slib, int64_t first, int64_t count,
                      int32_t *scores, uint8_t *mirrored);
int cds_search_sparse_device(const cds_maskset *ms, const cds_sparse_library *slib, int64_t first, int64_t count,
                             int32_t *d_scores);
int cds_search_sparse_counts(const cds_maskset *ms, const cds_sparse_library *slib, int64_t first, int64_t count,
                             int32_t *counts);

/* The scan + ColorMIPSearch.isMatch (API/cds/ColorMIPSearch.java:42-45) + compaction, all on the device: the pairs the
 * reference keeps in findAllColorDepthMatches (tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:82-116,
 * AbstractColorMIPSearchProcessor.java:60-85), as a list ordered by mask, then target.  score = matching pixels | mirrored << 31.
 * A mask whose query size is 0 yields no pair.  Only the list crosses PCIe (a 1,024 x 1 M search has 4 GB of scores).
 * CDS_ERR_OUT_OF_MEMORY when the list is longer than `capacity` -- *n_matches then holds the length needed. */
typedef struct cds_match {
    int32_t mask, target, score;
} cds_match;
int cds_search_sparse_matches(const cds_maskset *ms, const cds_sparse_library *slib, int64_t first, int64_t count,
                              double pct_positive_pixels, cds_match *matches, int64_t capacity, int64_t *n_matches);
/* Per mask the k (<= 1024) best-scoring targets of the scan, selected on the device (histogram threshold + ordered pass +
 * sort: the "top-500 hits per mask" the gradient-score step starts from, cmd/CalculateGradientScoresCmd.java:323-379, before any
 * grouping by line / sample): targets[m * k + i] (index relative to `first`; -1 behind the last one when count < k) and
 * scores[m * k + i] = matching pixels | mirrored << 31, by score descending, equal scores by ascending target. */
int cds_search_sparse_topk(const cds_maskset *ms, const cds_sparse_library *slib, int64_t first, int64_t count, int32_t k,
                           int32_t *targets, int32_t *scores);
/* The selection alone, over scores already on the device (d_scores[m * count + t], the layout cds_search_sparse_device leaves). */
int cds_scores_topk(const cds_maskset *ms, const int32_t *d_scores, int64_t count, int32_t k, int32_t *targets, int32_t *scores);
/* The same in two steps, for callers that cannot guess the length: the list is built on the device (grown chunk by chunk to the
 * exact length: one scan whatever it turns out to be), its size returned, and fetched -- whole or in pieces -- afterwards. */
typedef struct cds_match_list cds_match_list;
int cds_search_sparse_match_list(const cds_maskset *ms, const cds_sparse_library *slib, int64_t first, int64_t count,
                                 double pct_positive_pixels, cds_match_list **out, int64_t *n_matches);
int64_t cds_match_list_size(const cds_match_list *list);
int cds_match_list_fetch(const cds_match_list *list, int64_t first, int64_t count, cds_match *matches);
/* The same rows as the columns a caller's result table is made of, written in one pass by the library's host threads (a 1,024-mask x
 * 1 M-target search lists tens of millions of rows: five numpy passes per mask set were the tail of the processor's pipeline):
 * mask[i] = row.mask + mask_offset, target[i], score[i] = matching pixels, mirrored[i], and -- when query_sizes (the mask set's,
 * cds_maskset_query_sizes) is given -- ratio[i] = (float)((double)score / max(query_sizes[row.mask], 1)), the reference's
 * matchingPixelsRatio (PixelMatchScore; ColorMIPSearch.java:42-45).  Any output pointer may be NULL. */
int cds_match_list_fetch_columns(const cds_match_list *list, int64_t first, int64_t count, const int32_t *query_sizes, int32_t n_masks,
                                 int32_t mask_offset, int32_t *mask, int64_t *target, int32_t *score, float *ratio, uint8_t *mirrored);
void cds_match_list_destroy(cds_match_list *list);

/* ---- several GPUs of one node: the target-sharded search --------------------------------------------------------
 * The reference's only parallel strategy is data parallelism over target partitions (host threads:
 * tools cmd/cdsprocess/LocalColorMIPSearchProcessor.java:93-116; Spark partitions + collect():
 * SparkColorMIPSearchProcessor.java:50-73).  Here: one cds_context per GPU, the library cut into one cds_sparse_library
 * per GPU (shard r through context r), the mask set replicated (one cds_maskset per GPU, same masks and parameters).
 * cds_search_sparse_sharded scans every shard on its own GPU at the same time, gathers the per-target score arrays to the
 * first GPU with ONE RCCL gather over xGMI and returns them in library order:
 *   scores[m * n_total + shard_offsets[r] + t], t = image t of shard r;  shard_offsets[n + 1] (optional output).
 * RCCL (librccl.so.1) is bound at run time by cds_comm_create; contexts that share a device (a one-GPU rehearsal; RCCL
 * refuses duplicate devices) are served by device-to-device copies -- cds_comm_uses_rccl tells which. */
typedef struct cds_comm cds_comm;
int cds_comm_create(cds_context *const *contexts, int32_t n, cds_comm **out);
void cds_comm_destroy(cds_comm *comm);
int32_t cds_comm_size(const cds_comm *comm);
int32_t cds_comm_uses_rccl(const cds_comm *comm);
int cds_search_sparse_sharded(cds_comm *comm, const cds_maskset *const *masksets, const cds_sparse_library *const *slibs,
                              int64_t *shard_offsets, int32_t *scores, uint8_t *mirrored);
/* The same search with ColorMIPSearch.isMatch and the compaction on every GPU (cds_search_sparse_match_list per shard, all GPUs at
 * once): the matches of all shards ordered by mask, then by library index (target = shard_offsets[r] + image of shard r), exactly the
 * list cds_search_sparse_matches gives over one library holding all shards -- what SparkColorMIPSearchProcessor.java:50-73 collect()s.
 * No collective on this path: list lengths differ per GPU and every GPU hands its list to the host over its own PCIe link.
 * CDS_ERR_OUT_OF_MEMORY when the list is longer than `capacity` (*n_matches then holds the length needed). */
int cds_search_sparse_sharded_matches(cds_comm *comm, const cds_maskset *const *masksets, const cds_sparse_library *const *slibs,
                                      double pct_positive_pixels, int64_t *shard_offsets, cds_match *matches, int64_t capacity,
                                      int64_t *n_matches);
/* Per mask the k (<= 1024) best-scoring targets over all shards (cds_search_sparse_topk on every GPU at once, the n * k candidates of a
 * mask merged on the host): targets[m * k + i] = library index (-1 behind the last one), scores[m * k + i] = matching pixels |
 * mirrored << 31, by score descending, equal scores by ascending library index. */
int cds_search_sparse_sharded_topk(cds_comm *comm, const cds_maskset *const *masksets, const cds_sparse_library *const *slibs, int32_t k,
                                   int64_t *shard_offsets, int32_t *targets, int32_t *scores);
/* Kernel 2 over the shards (declared here with the communicator; the scorer itself is further down): the GradientAreaGap scores of
 * n_pairs (mask index, LIBRARY index) pairs when the library's targets -- with their gradient / zgap variants -- are spread over
 * the GPUs as dense shards: shard r keeps library indices [shard_offsets[r], shard_offsets[r] + shard_sizes[r]) in slots
 * 0 .. shard_sizes[r] - 1 of libs[r] (shard_sizes NULL: every library's capacity); the shape set is replicated (shapesets[r] made
 * through context r from the same masks).  Every pair is scored on the GPU that owns its target -- the data parallelism over target
 * partitions of the reference (tools cmd/cdsprocess/SparkColorMIPSearchProcessor.java:50-73; the per-pair call is
 * cmd/CalculateGradientScoresCmd.java:541-586) --, all GPUs at once from one host thread, and the results come back in pair order
 * with the meaning of cds_gradscore (a slot without variants gives -1, -1).  No collective on this path. */
int cds_gradscore_sharded(cds_comm *comm, const cds_shapeset *const *shapesets, const cds_library *const *libs,
                          const int64_t *shard_sizes, int64_t n_pairs, const int32_t *pair_mask, const int64_t *pair_target,
                          int64_t *shard_offsets, int64_t *gradient_area_gap, int64_t *high_expression_area, uint8_t *mirrored);

/* Average device time (ms) of the scan kernel launches of the last cds_search*_ call on this mask
 * set, measured with hipEvents on the launch stream, and the number of launches. */
int cds_maskset_last_kernel_ms(const cds_maskset *ms, double *avg_ms, int32_t *launches);

/* ---- GradientAreaGap scorer (kernel 2) ----------------------------------------------------------
 * cds_shapeset_create == n x createShapeMatchCDSAlgorithmProvider(...).createColorDepthSearchAlgorithm
 * (API/cds/ColorDepthSearchAlgorithmProviderFactory.java:76-127) with queryBorderSize 0.
 * roi: optional W*H byte plane (non-zero = inside), NULL for none. */
int cds_shapeset_create(cds_context *ctx, int32_t n_masks, const uint8_t *masks_rgb, int32_t W, int32_t H,
                        int32_t query_threshold, int32_t mirror, const uint8_t *roi, const cds_rect *excluded,
                        int32_t n_excluded, cds_shapeset **out);
/* The same with queryBorderSize != 0 (Factory:103, fed by the CLI's --borderSize: tools cmd/AbstractColorDepthMatchArgs.java:25):
 * a frame of that many pixels is skipped by every fold over the query (API/imageprocessing/LImage.java:89-113) -- the query
 * mask and the high-expression mask are empty there and the gap sum leaves it out, while the max filters still see its pixels. */
int cds_shapeset_create_bordered(cds_context *ctx, int32_t n_masks, const uint8_t *masks_rgb, int32_t W, int32_t H,
                                 int32_t query_threshold, int32_t query_border_size, int32_t mirror, const uint8_t *roi,
                                 const cds_rect *excluded, int32_t n_excluded, cds_shapeset **out);
void cds_shapeset_destroy(cds_shapeset *ss);
int32_t cds_shapeset_size(const cds_shapeset *ss); /* masks in the set */
/* Sum of the query mask and of the high-expression mask per mask
 * (the values Shape2DMatch...Test.java:53-54 checks). */
int cds_shapeset_mask_stats(const cds_shapeset *ss, int64_t *mask_size, int64_t *high_expression_size);
/* Score n_pairs (mask index, library slot) pairs == Shape2DMatch...Algorithm.calculateMatchingScore
 * (API/cds/Shape2DMatchColorDepthSearchAlgorithm.java:150-186).  Slots must hold gradient and zgap
 * variants, otherwise that pair returns (-1,-1,false) as the reference does (:155-158).
 * Host outputs; blocks. */
int cds_gradscore(const cds_shapeset *ss, const cds_library *lib, int64_t n_pairs, const int32_t *pair_mask,
                  const int64_t *pair_slot, int64_t *gradient_area_gap, int64_t *high_expression_area,
                  uint8_t *mirrored);
/* Per-pair convenience with host images (grad/zgap may be NULL -> -1,-1). */
int cds_shape_score_pair(const cds_shapeset *ss, int32_t mask_index, const uint8_t *target_rgb,
                         const uint16_t *grad, const uint8_t *zgap_rgb, int32_t W, int32_t H,
                         int64_t *gradient_area_gap, int64_t *high_expression_area, uint8_t *mirrored);
/* GradientAreaGapUtils.calculate2DShapeScore / calculateNormalizedScore
 * (API/cds/GradientAreaGapUtils.java:199-235). Pure host arithmetic. */
int64_t cds_shape_score_value(int64_t gradient_area_gap, int64_t high_expression_area);
double cds_normalized_score(int32_t pixel_match_score, int64_t shape_score, int64_t max_pixel_match,
                            int64_t max_shape_score);
int cds_shapeset_last_kernel_ms(const cds_shapeset *ss, double *avg_ms, int32_t *launches);
/* How the last cds_gradscore / cds_shape_score_pair call on this set got S(z) of the zgap pixels: from the zgap bytes inside the
 * scorer (used_zslice_plane = 0: no conversion pass; chosen when the targets serve few pairs) or from the derived zslice plane
 * (1), in which case converted_slots images were converted by this call, in conversion_ms of device time.  Results are identical;
 * CDS_SHAPE_PAIRS_PER_CONVERSION (default 8 pairs per slot still to convert) moves the switch. */
int cds_shapeset_last_call_info(const cds_shapeset *ss, int32_t *used_zslice_plane, int64_t *converted_slots, double *conversion_ms);

/* Host helper used by tests and by callers that have no ZGapImage file: clearRegion -> mask(thr) ->
 * circular max filter (API/imageprocessing/ImageTransformation.java:353-572), computed on the GPU. */
int cds_make_zgap(cds_context *ctx, const uint8_t *target_rgb, int32_t W, int32_t H, int32_t threshold,
                  double radius, const cds_rect *excluded, int32_t n_excluded, uint8_t *zgap_rgb);
/* The same transformation for resident library slots, entirely on the device: fills the ZGapImage variant of slots
 * [first, first+count) from their targets (for libraries that come without ZGapImage files; the reference's tests do
 * this per image on the CPU, Shape2DMatch...Test.java:171-174). */
int cds_library_make_zgap_variants(cds_library *lib, int64_t first, int64_t count, int32_t threshold, double radius,
                                   const cds_rect *excluded, int32_t n_excluded);

#ifdef __cplusplus
}
#endif
#endif /* CDS_HIP_H */
