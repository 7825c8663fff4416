// What cds_imageio.hip shares with the other host-side loaders of the library (cds_variants.hip): the native decoders behind one call
// and the pool of pinned staging buffers.  Internal; the C ABI is include/cds_hip.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace cds {

// The image file at `path` decoded into `dst`.  kind 0: an 8-bit RGB image of W x H (3 * W * H bytes, as NeuronMIPUtils.loadImageFromStream hands a
// MIP to the algorithms); kind 1: a grayscale image of 8 or 16 bits, widened to 16 (2 * W * H bytes: the GradientImage variant, ImageArray.get
// semantics).  false + `err` for anything else (unreadable, other size, other pixel type).  Thread-safe.
bool decode_file_into(const char *path, int W, int H, int kind, void *dst, std::string &err);

// Pinned host buffers kept between calls (pinning costs ~0.15 ms per MB).  get() returns nullptr when the host cannot pin that much.
uint8_t *pinned_get(size_t bytes);
void pinned_put(uint8_t *p);

}  // namespace cds
