"""Host-side mirror of the reference's `cds` plugin surface, over the C ABI of libcds_hip.so.

Same names, argument meaning and error behaviour as the Java interfaces it replaces
(API/ = colormipsearch-api/src/main/java/org/janelia/colormipsearch/):

  ColorDepthSearchAlgorithm            API/cds/ColorDepthSearchAlgorithm.java:17-62
  ColorDepthSearchAlgorithmProvider    API/cds/ColorDepthSearchAlgorithmProvider.java:12-38
  ColorDepthSearchAlgorithmProviderFactory  API/cds/ColorDepthSearchAlgorithmProviderFactory.java:30-127
  ColorDepthSearchParams               API/cds/ColorDepthSearchParams.java:9-86
  ColorMIPSearch                       API/cds/ColorMIPSearch.java:12-47
  PixelMatchScore / ShapeMatchScore    API/cds/PixelMatchScore.java, API/cds/ShapeMatchScore.java
  ComputeFileType                      API/model/ComputeFileType.java:5-28 (the two members the path uses)

Images are numpy arrays: RGB uint8[H,W,3] (byte order of ColorImageArray), gradients uint16[H,W].
Java's IllegalArgumentException maps to capi.CdsInvalidArgument (a ValueError).

On top of the per-pair `calculateMatchingScore` the GPU-native granularity is exposed too:
`Library` (a resident shard of targets) and `searchLibrary` / `scoreLibraryPairs`, which is what a
`ColorMIPSearchProcessor` replacement would call (one launch per library, not one JNI call per pair).
Everything computes on the GPU through the C ABI; there is no CPU fallback in this package.
"""
import ctypes
import enum
import os
import threading

import numpy as np

from . import capi
from .capi import CdsError, CdsInvalidArgument  # noqa: F401  (re-exported)


class ComputeFileType(enum.Enum):
    InputColorDepthImage = "InputColorDepthImage"
    GradientImage = "GradientImage"
    ZGapImage = "ZGapImage"


# --------------------------------------------------------------------------------------------------
# excluded regions (ImageRegionDefinition) -- rectangles [x0,y0,x1,y1)
# --------------------------------------------------------------------------------------------------
def label_regions(width, has_name_label=True, has_color_scale_label=True):
    """tools cmd/AbstractColorDepthMatchArgs.java:101-119 (== test ImageTestUtils.java:12-24)."""
    rects = []
    if has_color_scale_label and width > 270:
        rects.append((width - 270, 0, width, 90))
    if has_name_label:
        rects.append((0, 0, 330, 100))
    return rects


def _resolve_regions(regions, width):
    if regions is None:
        return []
    if callable(regions):
        return list(regions(width))
    return list(regions)


# --------------------------------------------------------------------------------------------------
# context / library
# --------------------------------------------------------------------------------------------------
class Context:
    """One GPU (cds_context)."""

    def __init__(self, device=0):
        self._lib = capi.load()
        h = ctypes.c_void_p()
        capi.check(self._lib.cds_context_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cds_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """Launch on an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        capi.check(self._lib.cds_context_set_stream(self._h, ctypes.c_void_p(int(hip_stream) if hip_stream else None)))

    def reset_stream(self):
        capi.check(self._lib.cds_context_reset_stream(self._h))

    def synchronize(self):
        capi.check(self._lib.cds_context_synchronize(self._h))

    def trim(self):
        """Give the context's idle device blocks back to the driver; returns the bytes freed (cds_context_trim)."""
        freed = ctypes.c_uint64()
        capi.check(self._lib.cds_context_trim(self._h, ctypes.byref(freed)))
        return int(freed.value)

    def set_cache_limit(self, nbytes):
        """Cap the device memory the context may keep idle for recycling (default 6 GB; 0 = none)."""
        capi.check(self._lib.cds_context_set_cache_limit(self._h, ctypes.c_uint64(int(nbytes))))

    def device_rcp_table(self):
        out = np.zeros(256, dtype=np.float32)
        capi.check(self._lib.cds_debug_device_rcp(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out

    def valu_issue_ns(self):
        """ns one SIMD of this device needs per wave64 VALU instruction (calibration of the issue roofline, bench.py)."""
        v = ctypes.c_double()
        capi.check(self._lib.cds_debug_valu_issue_ns(self._h, ctypes.byref(v)))
        return v.value

    def sparse_load_only(self, on):
        """Diagnostic (bench.py): while on, single-mask sparse searches made through THIS context run the scan kernel's loads-only
        twin; other contexts are not affected (the flag lives in the context)."""
        capi.check(self._lib.cds_debug_sparse_load_only(self._h, int(bool(on))))

    def sparse_load_only_state(self):
        return bool(self._lib.cds_debug_sparse_load_only_state(self._h))

    def make_zgap(self, target, threshold, radius=10, excluded_regions=None):
        """clearRegion -> mask(threshold) -> unsafeMaxFilter(radius), on the GPU."""
        img = capi.rgb_images(target)[0]
        H, W, _ = img.shape
        r, n = capi.rect_array(_resolve_regions(excluded_regions, W))
        out = np.empty_like(img)
        capi.check(self._lib.cds_make_zgap(self._h, capi.as_u8p(img), W, H, int(threshold), float(radius), r, n,
                                           capi.as_u8p(out)))
        return out


class Comm:
    """Several GPUs of one node in ONE process (cds_comm): `contexts[r]` owns shard r of a target-sharded library.
    `searchSparseLibrarySharded` scans every shard on its own GPU, gathers the score arrays to the first GPU with one RCCL
    gather and returns them in library order (the reference: partitions of targets, SparkColorMIPSearchProcessor.java:50-73)."""

    def __init__(self, contexts):
        self._lib = capi.load()
        self.contexts = list(contexts)
        arr = (ctypes.c_void_p * len(self.contexts))(*[c._h for c in self.contexts])
        h = ctypes.c_void_p()
        capi.check(self._lib.cds_comm_create(arr, len(self.contexts), ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cds_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._lib.cds_comm_size(self._h))

    @property
    def uses_rccl(self):
        return bool(self._lib.cds_comm_uses_rccl(self._h))

    def searchSparseLibrarySharded(self, algorithms, sparse_libraries):
        """algorithms[r] / sparse_libraries[r]: the replicated mask set and the shard of context r.
        Returns (scores int32[n_masks, n_total], mirrored bool[n_masks, n_total], shard_offsets int64[n + 1])."""
        n = len(self.contexts)
        if len(algorithms) != n or len(sparse_libraries) != n:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "one mask set and one shard per context")
        ms = (ctypes.c_void_p * n)(*[a._h for a in algorithms])
        sl = (ctypes.c_void_p * n)(*[s._h for s in sparse_libraries])
        n_total = sum(len(s) for s in sparse_libraries)
        n_masks = algorithms[0].n_masks
        scores = np.zeros((n_masks, n_total), dtype=np.int32)
        mirrored = np.zeros((n_masks, n_total), dtype=np.uint8)
        offs = np.zeros(n + 1, dtype=np.int64)
        capi.check(self._lib.cds_search_sparse_sharded(self._h, ms, sl, capi.as_i64p(offs), capi.as_i32p(scores), capi.as_u8p(mirrored)))
        return scores, mirrored.astype(bool), offs

    def searchSparseLibraryShardedTopK(self, algorithms, sparse_libraries, k):
        """Per mask the k best-scoring targets over all shards (selected on every GPU at once, merged on the host):
        (targets int32[n_masks, k] as library indices with -1 behind the last one, scores int32[n_masks, k], mirrored bool[n_masks, k],
        shard_offsets int64[n + 1]); score descending, equal scores by ascending library index."""
        n = len(self.contexts)
        if len(algorithms) != n or len(sparse_libraries) != n:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "one mask set and one shard per context")
        ms = (ctypes.c_void_p * n)(*[a._h for a in algorithms])
        sl = (ctypes.c_void_p * n)(*[s._h for s in sparse_libraries])
        n_masks = algorithms[0].n_masks
        t = np.zeros((n_masks, int(k)), dtype=np.int32)
        s = np.zeros((n_masks, int(k)), dtype=np.int32)
        offs = np.zeros(n + 1, dtype=np.int64)
        capi.check(self._lib.cds_search_sparse_sharded_topk(self._h, ms, sl, int(k), capi.as_i64p(offs), capi.as_i32p(t), capi.as_i32p(s)))
        return t, (s & 0x7FFFFFFF).astype(np.int32), s < 0, offs

    def searchSparseLibraryShardedMatches(self, algorithms, sparse_libraries, pctPositivePixels):
        """The (mask, target) pairs for which ColorMIPSearch.isMatch holds, filtered and compacted on every GPU at once and handed to
        the host by each GPU directly: (mask int32[n], target int32[n] in library order, score int32[n], mirrored bool[n],
        shard_offsets int64[n + 1]), ordered by mask, then target -- the list searchSparseLibraryMatches gives over one library."""
        n = len(self.contexts)
        if len(algorithms) != n or len(sparse_libraries) != n:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "one mask set and one shard per context")
        ms = (ctypes.c_void_p * n)(*[a._h for a in algorithms])
        sl = (ctypes.c_void_p * n)(*[s._h for s in sparse_libraries])
        offs = np.zeros(n + 1, dtype=np.int64)
        cnt = ctypes.c_int64()
        cap = 1 << 16
        while True:
            buf = np.empty((cap, 3), dtype=np.int32)
            rc = self._lib.cds_search_sparse_sharded_matches(self._h, ms, sl, float(pctPositivePixels or 0.0), capi.as_i64p(offs),
                                                             buf.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(cnt))
            if rc == capi.CDS_ERR_OUT_OF_MEMORY and cnt.value > cap:     # (the list is longer than the first guess: once more, exactly sized)
                cap = cnt.value
                continue
            capi.check(rc)
            break
        buf = buf[:cnt.value]
        return buf[:, 0].copy(), buf[:, 1].copy(), (buf[:, 2] & 0x7FFFFFFF).astype(np.int32), (buf[:, 2] < 0), offs

    def scoreLibraryPairsSharded(self, shape_algorithms, libraries, pair_mask, pair_target, shard_sizes=None):
        """GradientAreaGap of (mask index, LIBRARY index) pairs over dense shards (cds_gradscore_sharded): shape_algorithms[r] /
        libraries[r] = the replicated shape set and the dense shard (targets + gradient / zgap variants in slots 0 .. size_r - 1) of
        context r; every pair is scored on the GPU that owns its target, all GPUs at once.  Returns (gradientAreaGap int64[n],
        highExpressionArea int64[n], mirrored bool[n], shard_offsets int64[len + 1]) in pair order -- what
        Shape2DMatchColorDepthSearchAlgorithm.scoreLibraryPairs gives over one library holding all shards."""
        n = len(self.contexts)
        if len(shape_algorithms) != n or len(libraries) != n:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "one shape set and one shard per context")
        ss = (ctypes.c_void_p * n)(*[a._h for a in shape_algorithms])
        lb = (ctypes.c_void_p * n)(*[l._h for l in libraries])
        pm = np.ascontiguousarray(pair_mask, dtype=np.int32)
        pt = np.ascontiguousarray(pair_target, dtype=np.int64)
        if pm.shape != pt.shape or pm.ndim != 1:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "pair_mask and pair_target must be 1-D arrays of one length")
        sizes = None if shard_sizes is None else np.ascontiguousarray(shard_sizes, dtype=np.int64)
        if sizes is not None and len(sizes) != n:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "one shard size per context")
        k = len(pm)
        gap, high, mir = np.zeros(k, dtype=np.int64), np.zeros(k, dtype=np.int64), np.zeros(k, dtype=np.uint8)
        offs = np.zeros(n + 1, dtype=np.int64)
        capi.check(self._lib.cds_gradscore_sharded(self._h, ss, lb, capi.as_i64p(sizes) if sizes is not None else None, k, capi.as_i32p(pm),
                                                   capi.as_i64p(pt), capi.as_i64p(offs), capi.as_i64p(gap), capi.as_i64p(high), capi.as_u8p(mir)))
        return gap, high, mir.astype(bool), offs


_default_ctx = {}
_ctx_lock = threading.Lock()


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    with _ctx_lock:
        if device not in _default_ctx:
            _default_ctx[device] = Context(device)
        return _default_ctx[device]


class Library:
    """A resident shard of target MIPs in HBM (cds_library); replaces the host-side MIP cache."""

    def __init__(self, width, height, capacity, context=None, device_buffer=None):
        self.ctx = context or default_context()
        self._lib = capi.load()
        self.width, self.height, self.capacity = int(width), int(height), int(capacity)
        h = ctypes.c_void_p()
        self._keepalive = device_buffer
        if device_buffer is None:
            capi.check(self._lib.cds_library_create(self.ctx._h, self.width, self.height, self.capacity, ctypes.byref(h)))
        else:  # e.g. a torch uint8 CUDA tensor
            ptr, nbytes = int(device_buffer.data_ptr()), int(device_buffer.numel() * device_buffer.element_size())
            capi.check(self._lib.cds_library_wrap(self.ctx._h, self.width, self.height, self.capacity,
                                                  ctypes.c_void_p(ptr), nbytes, ctypes.byref(h)))
        self._h = h

    @staticmethod
    def bytes_required(width, height, capacity):
        return int(capi.load().cds_library_bytes(int(width), int(height), int(capacity)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cds_library_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, first, images):
        imgs = capi.rgb_images(images)
        if imgs.shape[1:3] != (self.height, self.width):
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "image size does not match the library")
        capi.check(self._lib.cds_library_upload(self._h, int(first), imgs.shape[0], capi.as_u8p(imgs)))

    def download(self, first, count):
        out = np.empty((int(count), self.height, self.width, 3), dtype=np.uint8)
        capi.check(self._lib.cds_library_download(self._h, int(first), int(count), capi.as_u8p(out)))
        return out

    def upload_variants(self, first, gradients=None, zgaps=None):
        n = None
        gp = zp = None
        if gradients is not None and np.asarray(gradients).dtype == np.uint8:
            # a GRAY8 gradient (the reference accepts GRAY8 and GRAY16 through ImageArray.get): widened by the library
            g8 = np.ascontiguousarray(gradients)
            if g8.ndim == 2:
                g8 = g8[None]
            capi.check(self._lib.cds_library_upload_gradient_u8(self._h, int(first), g8.shape[0], capi.as_u8p(g8)))
            n = g8.shape[0]
            gradients = None
        if gradients is not None:
            gradients = np.ascontiguousarray(gradients, dtype=np.uint16)
            if gradients.ndim == 2:
                gradients = gradients[None]
            n = gradients.shape[0]
            gp = capi.as_u16p(gradients)
        if zgaps is not None:
            zgaps = capi.rgb_images(zgaps, n)
            n = zgaps.shape[0]
            zp = capi.as_u8p(zgaps)
        if n is None or (gp is None and zp is None):
            return
        capi.check(self._lib.cds_library_upload_variants(self._h, int(first), n, gp, zp))

    def load_files(self, first, target_paths, gradient_paths=None, zgap_paths=None, threads=0):
        """Slots first.. straight from files: the targets (8-bit RGB), their GradientImage (8- / 16-bit gray; None entry = none: the slot's
        pairs score (-1, -1)) and ZGapImage files (8-bit RGB; None entry = none), decoded natively on `threads` host threads while the
        previous 64 slots are uploaded (cds_library_load_files).  A path may be a pair (archive.zip, entry name)."""
        n = len(target_paths)
        capi.check(self._lib.cds_library_load_files(self._h, int(first), n, _path_array(target_paths, n), _path_array(gradient_paths, n),
                                                    _path_array(zgap_paths, n), int(threads)))

    def load_variants(self, first, shard, targets, threads=0, images_only=False):
        """Slots first.. <- the targets `targets` of a VariantShard: image, gradient and zgap as stored, decoded on the device
        (cds_library_load_variants).  Returns the targets' flags (VariantShard.HAS_GRADIENT / HAS_ZGAP bits) so that the caller knows
        which slots came without a gradient or a zgap.  `images_only`: the MIPs alone (cds_library_load_variant_images: only their
        sections of the blobs are read; the slots have no variants, the flags returned are 0)."""
        idx = np.ascontiguousarray(targets, dtype=np.int64)
        if images_only:
            capi.check(self._lib.cds_library_load_variant_images(self._h, int(first), shard._h, capi.as_i64p(idx), len(idx), int(threads)))
            return np.zeros(len(idx), dtype=np.uint32)
        capi.check(self._lib.cds_library_load_variants(self._h, int(first), shard._h, capi.as_i64p(idx), len(idx), int(threads)))
        return shard.flags[idx]

    def download_variants(self, first, count):
        g = np.empty((int(count), self.height, self.width), dtype=np.uint16)
        z = np.empty((int(count), self.height, self.width, 3), dtype=np.uint8)
        capi.check(self._lib.cds_library_download_variants(self._h, int(first), int(count), capi.as_u16p(g), capi.as_u8p(z)))
        return g, z

    def make_zgap_variants(self, first, count, threshold, radius=10, excluded_regions=None):
        """ZGapImage variants of resident slots, computed on the device: clearRegion -> mask(threshold) -> max filter."""
        r, n = capi.rect_array(_resolve_regions(excluded_regions, self.width))
        capi.check(self._lib.cds_library_make_zgap_variants(self._h, int(first), int(count), int(threshold), float(radius), r, n))

    def generate_synthetic(self, seed, first_index, first_slot, count, kind=0, with_variants=False):
        capi.check(self._lib.cds_library_generate_synthetic(self._h, ctypes.c_uint64(int(seed)), int(first_index),
                                                            int(first_slot), int(count), int(kind), int(bool(with_variants))))

    def generate_synthetic_list(self, seed, indices, first_slot=0, kind=0, with_variants=False):
        """Slot first_slot + i gets image indices[i] of the stream (the stand-in for loading just the selected targets' files)."""
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        capi.check(self._lib.cds_library_generate_synthetic_list(self._h, ctypes.c_uint64(seed), capi.as_i64p(idx), int(first_slot), len(idx),
                                                                 int(kind), int(bool(with_variants))))


def native_path(file_data):
    """A MIP's file_data as the native loaders take it: a path, or -- the reference's FileData of type zipEntry (API/model/FileData.java) --
    a pair (archive.zip, entry name), handed over as `archive.zip!entry`.  None stays None; anything else (bytes of an image, a file
    object, an archive that is not a .zip) is not a native path: ValueError."""
    if file_data is None:
        return None
    if isinstance(file_data, (tuple, list)) and len(file_data) == 2 and all(isinstance(x, (str, os.PathLike)) for x in file_data) \
            and os.fspath(file_data[0]).lower().endswith(".zip"):
        return os.fsencode(os.fspath(file_data[0]) + "!" + os.fspath(file_data[1]))
    if isinstance(file_data, (str, os.PathLike)):
        return os.fsencode(file_data)
    if isinstance(file_data, bytes) and b"\0" not in file_data and len(file_data) < 4096:
        return file_data
    raise ValueError("not a file path or (zip archive, entry) pair")


def is_native_path(file_data):
    try:
        native_path(file_data)
        return True
    except ValueError:
        return False


def _path_array(paths, n):
    if paths is None:
        return None
    if len(paths) != n:
        raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "one path (or None) per target")
    return (ctypes.c_char_p * max(1, n))(*[native_path(p) for p in paths])


class VariantShardWriter:
    """Writes a variant shard (cds_variant_shard_create / _append / _append_files / _finish): kernel 2's inputs of a library -- every
    target's MIP, GradientImage and ZGapImage -- losslessly in one file that Library.load_variants decodes on the GPU.  No GPU is needed
    to write one.  Use as a context manager, or call finish()."""

    def __init__(self, path, width, height):
        self._lib = capi.load()
        self.width, self.height = int(width), int(height)
        self.count = 0
        self._h = None
        h = ctypes.c_void_p()
        capi.check(self._lib.cds_variant_shard_create(os.fsencode(path), self.width, self.height, ctypes.byref(h)))
        self._h = h

    def append(self, images, gradients=None, zgaps=None, threads=0):
        images = capi.rgb_images(images)
        n = images.shape[0]
        gp = zp = None
        if gradients is not None:
            gradients = np.ascontiguousarray(gradients, dtype=np.uint16).reshape(n, self.height, self.width)
            gp = capi.as_u16p(gradients)
        if zgaps is not None:
            zgaps = capi.rgb_images(zgaps, n)
            zp = capi.as_u8p(zgaps)
        if images.shape[1:3] != (self.height, self.width):
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "image size differs from the shard's")
        capi.check(self._lib.cds_variant_shard_append(self._h, n, capi.as_u8p(images), gp, zp, int(threads)))
        self.count += n

    def append_files(self, target_paths, gradient_paths=None, zgap_paths=None, threads=0):
        n = len(target_paths)
        capi.check(self._lib.cds_variant_shard_append_files(self._h, n, _path_array(target_paths, n), _path_array(gradient_paths, n),
                                                            _path_array(zgap_paths, n), int(threads)))
        self.count += n

    def finish(self):
        if self._h is not None:
            h, self._h = self._h, None
            capi.check(self._lib.cds_variant_shard_finish(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.finish()
        return False

    def __del__(self):
        try:
            self.finish()
        except Exception:
            pass


class VariantShard:
    """A variant shard opened for reading (cds_variant_shard_open)."""
    HAS_GRADIENT, GRADIENT_DELTA2, HAS_ZGAP, GRADIENT_DIFF2, RGB_SPANS, ZGAP_SPANS = 1, 2, 4, 8, 16, 32

    def __init__(self, path):
        self._lib = capi.load()
        self._h = None
        h = ctypes.c_void_p()
        capi.check(self._lib.cds_variant_shard_open(os.fsencode(path), ctypes.byref(h)))
        self._h = h
        w, hh, n, b = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        capi.check(self._lib.cds_variant_shard_info(h, ctypes.byref(w), ctypes.byref(hh), ctypes.byref(n), ctypes.byref(b)))
        self.width, self.height, self.count, self.file_bytes = w.value, hh.value, n.value, b.value
        self.flags = np.zeros(self.count, dtype=np.uint32)
        self.packed_bytes = np.zeros(self.count, dtype=np.uint32)
        if self.count:
            capi.check(self._lib.cds_variant_shard_target_info(h, 0, self.count, self.flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                               self.packed_bytes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))

    def __len__(self):
        return self.count

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            self._lib.cds_variant_shard_close(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseLibrary:
    """A resident shard of target MIPs kept as signal-pixel lists (cds_sparse_library): ~4 % of the dense bytes,
    for searches that need the pixel-match score only (`searchSparseLibrary*`)."""

    def __init__(self, width, height, capacity_images=None, capacity_pixels=None, context=None, density_limit=None, min_threshold=None,
                 growable=None):
        """density_limit: images averaging more signal pixels than this share of the frame are refused at append time (default
        0.25: above it the dense layout's scan is faster); pass e.g. 2.0 to accept anything.
        min_threshold: pixels whose brightest channel is <= this are not stored (default 0: every signal pixel); searches with a
        smaller data threshold are refused -- results never change, the library is smaller and its scan visits fewer records.
        growable: appends enlarge the library when a capacity is lacking (the reference's MIP cache sizes itself with use,
        CachedMIPsUtils.java:60-100); default: growable exactly when no capacities are given (then it starts at 1,024 images)."""
        self.ctx = context or default_context()
        self._lib = capi.load()
        self.width, self.height = int(width), int(height)
        if growable is None:
            growable = capacity_images is None
        if capacity_images is None:
            capacity_images = 1024
        if capacity_pixels is None:
            capacity_pixels = int(capacity_images) * (20000 + 4 * self.height)
        h = ctypes.c_void_p()
        capi.check(self._lib.cds_sparse_library_create(self.ctx._h, self.width, self.height, int(capacity_images),
                                                       int(capacity_pixels), ctypes.byref(h)))
        self._h = h
        if growable:
            capi.check(self._lib.cds_sparse_library_set_growable(self._h, 1))
        ns, sw = ctypes.c_int32(), ctypes.c_int32()
        capi.check(self._lib.cds_sparse_library_layout(self._h, ctypes.byref(ns), ctypes.byref(sw)))
        self.n_strips, self.strip_width = ns.value, sw.value
        if density_limit is not None:
            self.set_density_limit(density_limit)
        if min_threshold is not None:
            self.set_min_threshold(min_threshold)

    def _capacity(self):
        ci, cp, s16 = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        capi.check(self._lib.cds_sparse_library_capacity(self._h, ctypes.byref(ci), ctypes.byref(cp), ctypes.byref(s16)))
        return ci.value, cp.value, bool(s16.value)

    @property
    def capacity_images(self):
        return self._capacity()[0]

    @property
    def capacity_pixels(self):
        return self._capacity()[1]

    @property
    def has_stream16(self):
        """Does the library hold the derived 16-bit stage-1 stream (dropped when records + stream would not stay resident)?"""
        return self._capacity()[2]

    def reserve(self, capacity_images, capacity_pixels):
        """Grow to at least these capacities at once (cds_sparse_library_reserve; never shrinks)."""
        capi.check(self._lib.cds_sparse_library_reserve(self._h, int(capacity_images), int(capacity_pixels)))

    def set_growable(self, on=True):
        capi.check(self._lib.cds_sparse_library_set_growable(self._h, 1 if on else 0))

    def set_min_threshold(self, min_data_threshold):
        capi.check(self._lib.cds_sparse_library_set_min_threshold(self._h, int(min_data_threshold)))

    @property
    def min_threshold(self):
        v = ctypes.c_int32()
        capi.check(self._lib.cds_sparse_library_min_threshold(self._h, ctypes.byref(v)))
        return v.value

    def stream_read_gbs(self):
        """Diagnostic (bench.py): GB/s of a plain streaming read of the whole record arena on this device, and the bytes read."""
        g, b = ctypes.c_double(), ctypes.c_double()
        capi.check(self._lib.cds_debug_sparse_stream_read(self._h, ctypes.byref(g), ctypes.byref(b)))
        return g.value, b.value

    def set_density_limit(self, max_signal_fraction):
        capi.check(self._lib.cds_sparse_library_set_density_limit(self._h, float(max_signal_fraction)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cds_sparse_library_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._lib.cds_sparse_library_size(self._h))

    @property
    def pixels(self):
        return int(self._lib.cds_sparse_library_pixels(self._h))

    def clear(self):
        capi.check(self._lib.cds_sparse_library_clear(self._h))

    def append_from(self, library, first=0, count=None):
        """Append dense resident slots [first, first+count) of `library` (device to device)."""
        count = library.capacity - first if count is None else int(count)
        capi.check(self._lib.cds_sparse_library_append(self._h, library._h, int(first), count))

    def append(self, images):
        """Append host images uint8[n,H,W,3]."""
        imgs = capi.rgb_images(images)
        if imgs.shape[1:3] != (self.height, self.width):
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "image size does not match the library")
        capi.check(self._lib.cds_sparse_library_append_host(self._h, imgs.shape[0], capi.as_u8p(imgs)))

    def append_from_shard(self, shard, targets=None, batch=1024, threads=0):
        """Append targets of a VariantShard (all of them, in the shard's order, unless `targets` lists some): their blobs are decoded on the
        device into `batch` dense staging slots and converted from there -- one file then serves both kernels, the pixel-match scan's
        resident library and the gradient-score step's variants.  (The records come out as from the image files: tests/test_mipio.py.)"""
        idx = np.arange(len(shard), dtype=np.int64) if targets is None else np.ascontiguousarray(targets, dtype=np.int64)
        if (shard.width, shard.height) != (self.width, self.height):
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "the shard's image size differs from the library's")
        if len(idx) == 0:
            return
        staging = Library(self.width, self.height, min(int(batch), len(idx)), context=self.ctx)
        for first in range(0, len(idx), staging.capacity):
            part = idx[first:first + staging.capacity]
            staging.load_variants(0, shard, part, threads=threads, images_only=True)
            self.append_from(staging, 0, len(part))

    def append_files(self, paths, threads=0):
        """Append 8-bit RGB image files (TIFF / PNG, the library's size): decoded natively on `threads` host threads
        (0 = all cores), two batches in flight (decode || upload + conversion).  A path may be a pair (archive.zip, entry name)."""
        capi.check(self._lib.cds_sparse_library_append_files(self._h, _path_array(paths, len(paths)), len(paths), int(threads)))

    def export(self, first=0, count=None):
        """(image_offsets uint64[count+1], cell_offsets uint32[count, n_strips, H+1], records uint32[n]) of images
        [first, first+count) -- the arrays `import_arrays` takes (and mipio.write_sparse_shard stores)."""
        count = len(self) - first if count is None else int(count)
        u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
        offs = np.zeros(count + 1, dtype=np.uint64)
        capi.check(self._lib.cds_sparse_library_export(self._h, int(first), count, offs.ctypes.data_as(u64p), None, None, 0))
        cells = np.zeros((count, self.n_strips, self.height + 1), dtype=np.uint32)
        rec = np.zeros(int(offs[count]), dtype=np.uint32)
        capi.check(self._lib.cds_sparse_library_export(self._h, int(first), count, offs.ctypes.data_as(u64p), cells.ctypes.data_as(u32p),
                                                       rec.ctypes.data_as(u32p), rec.size))
        return offs, cells, rec

    def import_arrays(self, image_offsets, cell_offsets, records, n_strips=None, strip_width=None, min_data_threshold=0):
        """Append images given as exported arrays (possibly memory-mapped)."""
        offs = np.ascontiguousarray(image_offsets, dtype=np.uint64)
        cells = np.ascontiguousarray(cell_offsets, dtype=np.uint32)
        rec = np.ascontiguousarray(records, dtype=np.uint32)
        count = len(offs) - 1
        if cells.size != count * self.n_strips * (self.height + 1) or (count >= 0 and rec.size < int(offs[count])):
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "array sizes do not match the image count / this library's layout")
        u64p, u32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
        capi.check(self._lib.cds_sparse_library_import(self._h, count, offs.ctypes.data_as(u64p), cells.ctypes.data_as(u32p), rec.ctypes.data_as(u32p),
                                                       self.n_strips if n_strips is None else int(n_strips),
                                                       self.strip_width if strip_width is None else int(strip_width), int(min_data_threshold or 0)))

    def image_offsets(self, first=0, count=None):
        """uint64[count + 1]: absolute position of the first record of images first..first+count in the record arena."""
        count = len(self) - first if count is None else int(count)
        offs = np.zeros(count + 1, dtype=np.uint64)
        capi.check(self._lib.cds_sparse_library_image_offsets(self._h, int(first), count, offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return offs

    def image_records(self, index):
        """(records uint32[n], cell offsets uint32[n_strips, H+1]) of one image -- diagnostic."""
        n = ctypes.c_int64()
        capi.check(self._lib.cds_sparse_library_image_records(self._h, int(index), None, 0, ctypes.byref(n), None))
        rec = np.zeros(n.value, dtype=np.uint32)
        rows = np.zeros((self.n_strips, self.height + 1), dtype=np.uint32)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        capi.check(self._lib.cds_sparse_library_image_records(self._h, int(index), rec.ctypes.data_as(u32p), n.value,
                                                              ctypes.byref(n), rows.ctypes.data_as(u32p)))
        return rec, rows


# --------------------------------------------------------------------------------------------------
# scores
# --------------------------------------------------------------------------------------------------
class PixelMatchScore:
    """API/cds/PixelMatchScore.java:3-31"""

    def __init__(self, matchingPixNum, matchingPixNumToMaskRatio, mirrored):
        self._n, self._ratio, self._mirrored = int(matchingPixNum), float(matchingPixNumToMaskRatio), bool(mirrored)

    def getScore(self):
        return self._n

    def getNormalizedScore(self):
        return float(np.float32(self._ratio))

    def isMirrored(self):
        return self._mirrored

    def __repr__(self):
        return f"PixelMatchScore(score={self._n}, ratio={self._ratio:.6f}, mirrored={self._mirrored})"


class CombinedMatchScore:
    """API/cds/CombinedMatchScore.java:5-30: the (pixel matches, shape score) pair the normalisation folds its maxima in
    (CalculateGradientScoresCmd.java:622-629)."""

    def __init__(self, pixelMatches, gradScore):
        self._px, self._grad = int(pixelMatches), int(gradScore)

    def getPixelMatches(self):
        return self._px

    def getGradScore(self):
        return self._grad

    def __repr__(self):
        return f"CombinedMatchScore(pixelMatches={self._px}, gradScore={self._grad})"


class ShapeMatchScore:
    """API/cds/ShapeMatchScore.java:5-65 (gradient-area-gap constructor)."""

    def __init__(self, gradientAreaGap, highExpressionArea, maxGradientAreaGap=-1, mirrored=False):
        self._gap, self._high, self._max, self._mirrored = int(gradientAreaGap), int(highExpressionArea), int(maxGradientAreaGap), bool(mirrored)

    def getGradientAreaGap(self):
        return self._gap

    def getHighExpressionArea(self):
        return self._high

    def getScore(self):
        v = int(capi.load().cds_shape_score_value(self._gap, self._high))
        return int(np.int64(v).astype(np.int32))  # (int) cast of ShapeMatchScore.java:29-33

    def getNormalizedScore(self):
        s = self.getScore()
        return float(np.float32(s) / np.float32(self._max)) if self._max > 0 else float(s)

    def isMirrored(self):
        return self._mirrored

    def __repr__(self):
        return f"ShapeMatchScore(gradientAreaGap={self._gap}, highExpressionArea={self._high}, mirrored={self._mirrored})"


def calculate2DShapeScore(gradientAreaGap, highExpressionArea):
    """GradientAreaGapUtils.calculate2DShapeScore (API/cds/GradientAreaGapUtils.java:199-207)."""
    if gradientAreaGap is None or highExpressionArea is None:
        return -1
    return int(capi.load().cds_shape_score_value(int(gradientAreaGap), int(highExpressionArea)))


def calculateNormalizedScore(pixelMatchScore, shapeScore, maxPixelMatch, maxShapeScore):
    """GradientAreaGapUtils.calculateNormalizedScore (API/cds/GradientAreaGapUtils.java:219-235)."""
    return float(capi.load().cds_normalized_score(int(pixelMatchScore), int(shapeScore), int(maxPixelMatch), int(maxShapeScore)))


# --------------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------------
class ColorDepthSearchParams:
    """API/cds/ColorDepthSearchParams.java:9-86"""

    def __init__(self, params=None):
        self._p = dict(params or {})

    def setParam(self, name, value):
        if value is not None:
            self._p[name] = value
        return self

    def asMap(self):
        return dict(self._p)

    def getParam(self, name, default=None):
        v = self._p.get(name)
        return default if v is None else v

    def getIntParam(self, name, default=None):
        v = self._p.get(name)
        return default if v is None else int(v)

    def getDoubleParam(self, name, default=None):
        v = self._p.get(name)
        return default if v is None else float(v)

    def getBoolParam(self, name, default=None):
        v = self._p.get(name)
        if v is None:
            return default
        if isinstance(v, str):
            return v.lower() == "true"
        return bool(v)


# --------------------------------------------------------------------------------------------------
# algorithms
# --------------------------------------------------------------------------------------------------
class ColorDepthSearchAlgorithm:
    """API/cds/ColorDepthSearchAlgorithm.java:17-62"""

    def getQueryImage(self):
        raise NotImplementedError

    def getQuerySize(self):
        raise NotImplementedError

    def getQueryFirstPixelIndex(self):
        raise NotImplementedError

    def getQueryLastPixelIndex(self):
        raise NotImplementedError

    def getRequiredTargetVariantTypes(self):
        raise NotImplementedError

    def calculateMatchingScore(self, targetImageArray, variantImageSuppliers=None):
        raise NotImplementedError


class PixelMatchColorDepthSearchAlgorithm(ColorDepthSearchAlgorithm):
    """API/cds/PixelMatchColorDepthSearchAlgorithm.java:20-265 for one mask (or a batch of masks that share
    the parameters: pass uint8[n,H,W,3]; the per-pair methods then address mask 0 unless told otherwise).
    The negative-query arguments of the reference constructor are not supported (the factory never passes
    them: ColorDepthSearchAlgorithmProviderFactory.java:65-67)."""

    def __init__(self, queryImage, queryThreshold, mirrorQuery, negQueryImage, negQueryThreshold, mirrorNegQuery,
                 targetThreshold, zTolerance, xyshift, excludedRegions=None, context=None):
        if negQueryImage is not None:
            raise CdsError(capi.CDS_ERR_UNSUPPORTED, "negative query images are not supported")
        self.ctx = context or default_context()
        self._lib = capi.load()
        self._masks = capi.rgb_images(queryImage)
        n, H, W, _ = self._masks.shape
        self.width, self.height, self.n_masks = W, H, n
        self.queryThreshold, self.targetThreshold = int(queryThreshold), int(targetThreshold)
        self.zTolerance, self.xyShift, self.mirrorQuery = float(zTolerance), int(xyshift), bool(mirrorQuery)
        self._rects = _resolve_regions(excludedRegions, W)
        params = capi.PixMatchParams(self.queryThreshold, self.targetThreshold, self.zTolerance, self.xyShift,
                                     int(self.mirrorQuery))
        r, nr = capi.rect_array(self._rects)
        h = ctypes.c_void_p()
        capi.check(self._lib.cds_maskset_create(self.ctx._h, n, capi.as_u8p(self._masks), W, H, ctypes.byref(params),
                                                r, nr, ctypes.byref(h)))
        self._h = h
        sizes = np.zeros(n, dtype=np.int32)
        capi.check(self._lib.cds_maskset_query_sizes(self._h, capi.as_i32p(sizes)))
        self._sizes = sizes

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cds_maskset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ColorDepthSearchAlgorithm ----
    def getQueryImage(self):
        return self._masks[0]

    def getQuerySize(self, mask_index=0):
        return int(self._sizes[mask_index])

    def getQuerySizes(self):
        return self._sizes.copy()

    def _positions(self, mask_index=0):
        m = self._masks[mask_index]
        sel = m.max(axis=2) > self.queryThreshold
        for (x0, y0, x1, y1) in self._rects:
            sel[max(0, y0):max(0, y1), max(0, x0):max(0, x1)] = False
        return np.flatnonzero(sel.reshape(-1))

    def _strip(self, mask_index=0):
        # PixelMatchColorDepthSearchAlgorithm.java:59-100 on top of the (multiplying) super methods
        # AbstractColorDepthSearchAlgorithm.java:76-84 -- informational only, quirks kept.
        W, H = self.width, self.height
        pos = self._positions(mask_index)
        if len(pos) == 0:
            first, last = H * W * W, 0
            return first, last
        xs, ys = pos % W, pos // W
        first = int(ys.min()) * W * int(xs.min())
        last = int(ys.max() + 1) * W * int(xs.max() + 1)
        shifts = [(0, 0)] if self.xyShift == 0 else [(dx, dy) for dx in (-2, 0, 2) for dy in (-2, 0, 2)]
        lists = []
        for dx, dy in shifts:
            x, y = xs + dx, ys + dy
            ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
            lists.append(np.where(ok, y * W + x, -1))
        if self.mirrorQuery:
            lists += [np.where(l == -1, -1, l + (W - 1) - 2 * (l % W)) for l in list(lists)]
        for l in lists:
            first = min(first, int(l[0]))
            last = max(last, int(l[-1]))
        return first, last

    def getQueryFirstPixelIndex(self):
        return self._strip()[0]

    def getQueryLastPixelIndex(self):
        return self._strip()[1]

    def getRequiredTargetVariantTypes(self):
        return set()

    def calculateMatchingScore(self, targetImageArray, variantImageSuppliers=None, mask_index=0):
        P = self.getQuerySize(mask_index)
        if P == 0:
            return PixelMatchScore(0, 0, False)
        t = np.ascontiguousarray(targetImageArray, dtype=np.uint8)
        if t.ndim != 3 or t.shape[2] != 3:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, f"expected an RGB image, got {t.shape}")
        s, m = ctypes.c_int32(), ctypes.c_uint8()
        capi.check(self._lib.cds_score_pair(self._h, int(mask_index), capi.as_u8p(t), t.shape[1], t.shape[0],
                                            ctypes.byref(s), ctypes.byref(m)))
        return PixelMatchScore(s.value, s.value / float(P), bool(m.value))

    # ---- GPU-native granularity ----
    def searchLibrary(self, library, first=0, count=None):
        """All masks x library slots [first, first+count): (scores int32[n_masks,count], mirrored bool[...])."""
        count = library.capacity - first if count is None else int(count)
        scores = np.zeros((self.n_masks, count), dtype=np.int32)
        mir = np.zeros((self.n_masks, count), dtype=np.uint8)
        capi.check(self._lib.cds_search(self._h, library._h, int(first), count, capi.as_i32p(scores), capi.as_u8p(mir)))
        return scores, mir.astype(bool)

    def searchLibraryCounts(self, library, first=0, count=None):
        count = library.capacity - first if count is None else int(count)
        counts = np.zeros((self.n_masks, count, capi.CDS_N_TRANSFORMS), dtype=np.int32)
        capi.check(self._lib.cds_search_counts(self._h, library._h, int(first), count, capi.as_i32p(counts)))
        return counts

    def searchLibraryDevice(self, library, first, count, d_scores_ptr):
        """Asynchronous; d_scores_ptr = device pointer to int32[n_masks*count] (score | mirrored<<31)."""
        capi.check(self._lib.cds_search_device(self._h, library._h, int(first), int(count), ctypes.c_void_p(int(d_scores_ptr))))

    def searchSparseLibrary(self, sparse_library, first=0, count=None):
        """searchLibrary over a SparseLibrary: identical results."""
        count = len(sparse_library) - first if count is None else int(count)
        scores = np.zeros((self.n_masks, count), dtype=np.int32)
        mir = np.zeros((self.n_masks, count), dtype=np.uint8)
        capi.check(self._lib.cds_search_sparse(self._h, sparse_library._h, int(first), count, capi.as_i32p(scores), capi.as_u8p(mir)))
        return scores, mir.astype(bool)

    def searchSparseLibraryCounts(self, sparse_library, first=0, count=None):
        count = len(sparse_library) - first if count is None else int(count)
        counts = np.zeros((self.n_masks, count, capi.CDS_N_TRANSFORMS), dtype=np.int32)
        capi.check(self._lib.cds_search_sparse_counts(self._h, sparse_library._h, int(first), count, capi.as_i32p(counts)))
        return counts

    def searchSparseLibraryDevice(self, sparse_library, first, count, d_scores_ptr):
        capi.check(self._lib.cds_search_sparse_device(self._h, sparse_library._h, int(first), int(count),
                                                      ctypes.c_void_p(int(d_scores_ptr))))

    def searchSparseLibraryMatches(self, sparse_library, pctPositivePixels, first=0, count=None):
        """The (mask, target) pairs of the scan for which ColorMIPSearch.isMatch holds, filtered and compacted on the device:
        (mask int32[n], target int32[n], score int32[n], mirrored bool[n]), ordered by mask, then target."""
        count = len(sparse_library) - first if count is None else int(count)
        h, n = ctypes.c_void_p(), ctypes.c_int64()
        capi.check(self._lib.cds_search_sparse_match_list(self._h, sparse_library._h, int(first), count, float(pctPositivePixels or 0.0),
                                                          ctypes.byref(h), ctypes.byref(n)))
        try:
            buf = np.empty((n.value, 3), dtype=np.int32)     # (the list is built on the device to its exact length: one scan)
            capi.check(self._lib.cds_match_list_fetch(h, 0, n.value, buf.ctypes.data_as(ctypes.c_void_p)))
        finally:
            self._lib.cds_match_list_destroy(h)
        return buf[:, 0].copy(), buf[:, 1].copy(), (buf[:, 2] & 0x7FFFFFFF).astype(np.int32), (buf[:, 2] < 0)

    def searchSparseLibraryMatchColumns(self, sparse_library, pctPositivePixels, first=0, count=None, mask_offset=0):
        """searchSparseLibraryMatches as the five columns of a result table, split and converted by the library's host threads while the
        list comes down (cds_match_list_fetch_columns): (mask + mask_offset int32[n], target int64[n], matching pixels int32[n],
        matchingPixelsRatio float32[n] = (float)((double) pixels / max(query size, 1)), mirrored bool[n])."""
        count = len(sparse_library) - first if count is None else int(count)
        h, n = ctypes.c_void_p(), ctypes.c_int64()
        capi.check(self._lib.cds_search_sparse_match_list(self._h, sparse_library._h, int(first), count, float(pctPositivePixels or 0.0),
                                                          ctypes.byref(h), ctypes.byref(n)))
        try:
            rows = n.value
            m, t, sc = np.empty(rows, np.int32), np.empty(rows, np.int64), np.empty(rows, np.int32)
            ratio, mir = np.empty(rows, np.float32), np.empty(rows, np.uint8)
            sizes = np.ascontiguousarray(self.getQuerySizes(), dtype=np.int32)
            vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            capi.check(self._lib.cds_match_list_fetch_columns(h, 0, rows, vp(sizes), len(sizes), int(mask_offset), vp(m), vp(t), vp(sc), vp(ratio), vp(mir)))
        finally:
            self._lib.cds_match_list_destroy(h)
        return m, t, sc, ratio, mir.view(bool)

    def searchSparseLibraryTopK(self, sparse_library, k, first=0, count=None):
        """Per mask the k best-scoring targets (score descending, equal scores by ascending target), selected on the device:
        (targets int32[n_masks, k] with -1 behind the last one, scores int32[n_masks, k], mirrored bool[n_masks, k])."""
        count = len(sparse_library) - first if count is None else int(count)
        t = np.zeros((self.n_masks, int(k)), dtype=np.int32)
        s = np.zeros((self.n_masks, int(k)), dtype=np.int32)
        capi.check(self._lib.cds_search_sparse_topk(self._h, sparse_library._h, int(first), count, int(k), capi.as_i32p(t), capi.as_i32p(s)))
        return t, (s & 0x7FFFFFFF).astype(np.int32), s < 0

    def topKOfDeviceScores(self, d_scores_ptr, count, k):
        """The same selection over a score matrix already on the device (int32[n_masks, count], as searchSparseLibraryDevice leaves it)."""
        t = np.zeros((self.n_masks, int(k)), dtype=np.int32)
        s = np.zeros((self.n_masks, int(k)), dtype=np.int32)
        capi.check(self._lib.cds_scores_topk(self._h, ctypes.c_void_p(int(d_scores_ptr)), int(count), int(k), capi.as_i32p(t), capi.as_i32p(s)))
        return t, (s & 0x7FFFFFFF).astype(np.int32), s < 0

    def lastScanKernel(self):
        """(kernel of the last single-mask sparse launch: 0 = 4-byte records, 1 = 16-bit stage-1 stream, -1 = none; share of visited records
        that reached a round in the last pilot, -1 = no pilot yet) -- diagnostic."""
        k, r = ctypes.c_int32(), ctypes.c_double()
        capi.check(self._lib.cds_debug_maskset_scan_kernel(self._h, ctypes.byref(k), ctypes.byref(r)))
        return k.value, r.value

    def sparseLayout(self, mask_index=0):
        """How the mask is laid out for the sparse scan (diagnostic): dict of slabs, rows per slab, LDS bytes, ..."""
        out = np.zeros(8, dtype=np.int32)
        capi.check(self._lib.cds_maskset_sparse_layout(self._h, int(mask_index), capi.as_i32p(out)))
        keys = ("slabs", "slab_rows", "lds_bytes", "box_cols", "box_rows", "interval_records", "plane_cell_bytes", "intervals_in_lds")
        return dict(zip(keys, (int(v) for v in out)))

    def sparseGroups(self):
        """(groups, groups scanned by the multi-mask kernel, masks in those) -- diagnostic of the multi-mask sparse scan."""
        a, b, c = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        capi.check(self._lib.cds_maskset_sparse_groups(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def lastKernelMs(self):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        capi.check(self._lib.cds_maskset_last_kernel_ms(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


class Shape2DMatchColorDepthSearchAlgorithm(ColorDepthSearchAlgorithm):
    """API/cds/Shape2DMatchColorDepthSearchAlgorithm.java:23-247 for one mask (or a batch), prepared as
    ColorDepthSearchAlgorithmProviderFactory.java:91-125 does; `queryBorderSize` is the frame every fold over the query skips
    (Factory:103, API/imageprocessing/LImage.java:89-113)."""

    def __init__(self, queryImage, queryThreshold, mirrorQuery, roiMask=None, excludedRegions=None, context=None, queryBorderSize=0):
        self.queryBorderSize = int(queryBorderSize or 0)
        self.ctx = context or default_context()
        self._lib = capi.load()
        self._masks = capi.rgb_images(queryImage)
        n, H, W, _ = self._masks.shape
        self.width, self.height, self.n_masks = W, H, n
        self.queryThreshold, self.mirrorQuery = int(queryThreshold), bool(mirrorQuery)
        self._rects = _resolve_regions(excludedRegions, W)
        r, nr = capi.rect_array(self._rects)
        roi_p = None
        if roiMask is not None:
            roi = np.ascontiguousarray(roiMask)
            if roi.ndim == 3:
                roi = roi.max(axis=2)
            roi = np.ascontiguousarray(roi != 0, dtype=np.uint8)
            if roi.shape != (H, W):
                raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "ROI mask size must match the query size")
            self._roi = roi
            roi_p = capi.as_u8p(roi)
        h = ctypes.c_void_p()
        capi.check(self._lib.cds_shapeset_create_bordered(self.ctx._h, n, capi.as_u8p(self._masks), W, H, self.queryThreshold,
                                                          self.queryBorderSize, int(self.mirrorQuery), roi_p, r, nr, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cds_shapeset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _cleared(self, mask_index=0):
        m = self._masks[mask_index].copy()
        for (x0, y0, x1, y1) in self._rects:
            m[max(0, y0):max(0, y1), max(0, x0):max(0, x1)] = 0
        return m

    def getQueryImage(self):
        return self._cleared()

    def _inner_signal(self, mask_index=0):
        """query pixels above the threshold that a fold visits: inside the border frame (LImage.java:89-113)"""
        sig = self._cleared(mask_index).max(axis=2) > self.queryThreshold
        b = self.queryBorderSize
        if b > 0:
            inner = np.zeros_like(sig)
            inner[b:self.height - b, b:self.width - b] = True
            sig &= inner
        return sig

    def getQuerySize(self, mask_index=0):
        return int(self._inner_signal(mask_index).sum())

    def getQueryFirstPixelIndex(self):
        nz = np.flatnonzero(self._inner_signal().reshape(-1))
        return int(nz[0]) if len(nz) else -1

    def getQueryLastPixelIndex(self):
        nz = np.flatnonzero(self._inner_signal().reshape(-1))
        return int(nz[-1]) if len(nz) else -1

    def getRequiredTargetVariantTypes(self):
        return {ComputeFileType.GradientImage, ComputeFileType.ZGapImage}

    def maskStats(self):
        """(sum of the query mask, sum of the high-expression mask) per mask."""
        a = np.zeros(self.n_masks, dtype=np.int64)
        b = np.zeros(self.n_masks, dtype=np.int64)
        capi.check(self._lib.cds_shapeset_mask_stats(self._h, capi.as_i64p(a), capi.as_i64p(b)))
        return a, b

    def calculateMatchingScore(self, targetImageArray, variantImageSuppliers=None, mask_index=0):
        sup = variantImageSuppliers or {}
        gs, zs = sup.get(ComputeFileType.GradientImage), sup.get(ComputeFileType.ZGapImage)
        grad = gs() if gs is not None else None
        zgap = zs() if zs is not None else None
        if grad is None or zgap is None:
            return ShapeMatchScore(-1, -1, -1, False)
        t = np.ascontiguousarray(targetImageArray, dtype=np.uint8)
        grad = np.ascontiguousarray(grad)
        if grad.dtype == np.uint8:  # GRAY8 gradient: ImageArray.get() widens it
            grad = grad.astype(np.uint16)
        grad = np.ascontiguousarray(grad, dtype=np.uint16)
        zgap = np.ascontiguousarray(zgap, dtype=np.uint8)
        if t.ndim != 3 or grad.shape != t.shape[:2] or zgap.shape != t.shape:
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "target, gradient and zgap sizes differ")
        g, h, m = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_uint8()
        capi.check(self._lib.cds_shape_score_pair(self._h, int(mask_index), capi.as_u8p(t), capi.as_u16p(grad),
                                                  capi.as_u8p(zgap), t.shape[1], t.shape[0], ctypes.byref(g),
                                                  ctypes.byref(h), ctypes.byref(m)))
        return ShapeMatchScore(g.value, h.value, -1, bool(m.value))

    def scoreLibraryPairs(self, library, pair_mask, pair_slot):
        """(gradientAreaGap int64[n], highExpressionArea int64[n], mirrored bool[n]) for (mask, slot) pairs."""
        pm = np.ascontiguousarray(pair_mask, dtype=np.int32)
        ps = np.ascontiguousarray(pair_slot, dtype=np.int64)
        n = len(pm)
        gap = np.zeros(n, dtype=np.int64)
        high = np.zeros(n, dtype=np.int64)
        mir = np.zeros(n, dtype=np.uint8)
        capi.check(self._lib.cds_gradscore(self._h, library._h, n, capi.as_i32p(pm), capi.as_i64p(ps), capi.as_i64p(gap),
                                           capi.as_i64p(high), capi.as_u8p(mir)))
        return gap, high, mir.astype(bool)

    def lastCallInfo(self):
        """(used the converted zslice plane?, zgap images converted by the last call, device ms of that conversion)"""
        mode, ms = ctypes.c_int32(), ctypes.c_double()
        n = np.zeros(1, dtype=np.int64)
        capi.check(self._lib.cds_shapeset_last_call_info(self._h, ctypes.byref(mode), capi.as_i64p(n), ctypes.byref(ms)))
        return bool(mode.value), int(n[0]), ms.value

    def lastKernelMs(self):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        capi.check(self._lib.cds_shapeset_last_kernel_ms(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


# --------------------------------------------------------------------------------------------------
# providers / factory / ColorMIPSearch
# --------------------------------------------------------------------------------------------------
class ColorDepthSearchAlgorithmProvider:
    """API/cds/ColorDepthSearchAlgorithmProvider.java:12-38"""

    def getDefaultCDSParams(self):
        raise NotImplementedError

    def createColorDepthSearchAlgorithm(self, queryImage, queryThreshold, queryBorderSize, cdsParams):
        raise NotImplementedError

    def createColorDepthQuerySearchAlgorithmWithDefaultParams(self, queryImage, queryThreshold, queryBorderSize):
        return self.createColorDepthSearchAlgorithm(queryImage, queryThreshold, queryBorderSize, ColorDepthSearchParams())


class _PixMatchProvider(ColorDepthSearchAlgorithmProvider):
    def __init__(self, mirrorMask, targetThreshold, pixColorFluctuation, xyShift, regions, context):
        self._defaults = (ColorDepthSearchParams().setParam("mirrorMask", mirrorMask).setParam("dataThreshold", targetThreshold)
                          .setParam("pixColorFluctuation", pixColorFluctuation).setParam("xyShift", xyShift))
        self._m, self._t, self._f, self._s, self._regions, self._ctx = mirrorMask, targetThreshold, pixColorFluctuation, xyShift, regions, context

    def getDefaultCDSParams(self):
        return self._defaults

    def forContext(self, context):
        """The same provider bound to another GPU's context (a replicated mask set per shard of a target-sharded library)."""
        return _PixMatchProvider(self._m, self._t, self._f, self._s, self._regions, context)

    def createColorDepthSearchAlgorithm(self, queryImage, queryThreshold, queryBorderSize, cdsParams):
        fl = cdsParams.getDoubleParam("pixColorFluctuation", self._f)
        zTolerance = 0.0 if fl is None else fl / 100      # Factory:55-56
        xyShift = cdsParams.getIntParam("xyShift", self._s)
        if (xyShift & 1) == 1:                              # Factory:58-60
            raise CdsInvalidArgument(capi.CDS_ERR_INVALID_ARGUMENT, "XY shift parameter must be an even number.")
        return PixelMatchColorDepthSearchAlgorithm(queryImage, queryThreshold, cdsParams.getBoolParam("mirrorMask", self._m),
                                                   None, 0, False, cdsParams.getIntParam("dataThreshold", self._t),
                                                   zTolerance, xyShift, self._regions, context=self._ctx)


class _ShapeMatchProvider(ColorDepthSearchAlgorithmProvider):
    def __init__(self, mirrorMask, roiMask, regions, context):
        self._defaults = ColorDepthSearchParams().setParam("mirrorMask", mirrorMask)
        self._m, self._roi, self._regions, self._ctx = mirrorMask, roiMask, regions, context

    def getDefaultCDSParams(self):
        return self._defaults

    def createColorDepthSearchAlgorithm(self, queryImage, queryThreshold, queryBorderSize, cdsParams):
        return Shape2DMatchColorDepthSearchAlgorithm(queryImage, queryThreshold, cdsParams.getBoolParam("mirrorMask", self._m),
                                                     self._roi, self._regions, context=self._ctx, queryBorderSize=queryBorderSize)


class ColorDepthSearchAlgorithmProviderFactory:
    """API/cds/ColorDepthSearchAlgorithmProviderFactory.java:30-127"""

    @staticmethod
    def createPixMatchCDSAlgorithmProvider(mirrorMask, targetThreshold, pixColorFluctuation, xyShiftParam,
                                           ignoredRegionsProvider=None, context=None):
        return _PixMatchProvider(mirrorMask, targetThreshold, pixColorFluctuation, xyShiftParam, ignoredRegionsProvider, context)

    @staticmethod
    def createShapeMatchCDSAlgorithmProvider(mirrorMask, roiMaskImageArray=None, excludedRegions=None, context=None):
        return _ShapeMatchProvider(mirrorMask, roiMaskImageArray, excludedRegions, context)


class ColorMIPSearch:
    """API/cds/ColorMIPSearch.java:12-47"""

    def __init__(self, pctPositivePixels, defaultQueryThreshold, cdsAlgorithmProvider):
        self.pctPositivePixels, self.defaultQueryThreshold, self.cdsAlgorithmProvider = pctPositivePixels, defaultQueryThreshold, cdsAlgorithmProvider

    def getCDSParameters(self):
        p = dict(self.cdsAlgorithmProvider.getDefaultCDSParams().asMap())
        p["pctPositivePixels"] = None if self.pctPositivePixels is None else str(self.pctPositivePixels)
        p["defaultMaskThreshold"] = None if self.defaultQueryThreshold is None else str(self.defaultQueryThreshold)
        return p

    def createQueryColorDepthSearchWithDefaultThreshold(self, queryImage):
        thr = 0 if self.defaultQueryThreshold is None else self.defaultQueryThreshold
        return self.cdsAlgorithmProvider.createColorDepthQuerySearchAlgorithmWithDefaultParams(queryImage, thr, 0)

    def createQueryColorDepthSearch(self, queryImage, queryThreshold, borderSize):
        return self.cdsAlgorithmProvider.createColorDepthQuerySearchAlgorithmWithDefaultParams(queryImage, queryThreshold, borderSize)

    def isMatch(self, pixelMatchScore):
        thr = self.pctPositivePixels / 100 if self.pctPositivePixels is not None else 0.0
        return pixelMatchScore.getScore() > 0 and pixelMatchScore.getNormalizedScore() > thr

    def isMatchArray(self, scores, query_size):
        """Vectorised isMatch for score arrays of one mask."""
        thr = self.pctPositivePixels / 100 if self.pctPositivePixels is not None else 0.0
        scores = np.asarray(scores)
        if query_size == 0:
            return np.zeros(scores.shape, dtype=bool)
        ratio = (scores.astype(np.float64) / float(query_size)).astype(np.float32).astype(np.float64)
        return (scores > 0) & (ratio > thr)
