// cds_hip.hpp -- header-only C++17 mirror of the reference's `cds` plugin surface over the C ABI (cds_hip.h).
//
// The reference is Java; this image has no JDK, so the compiled-language host side above the C ABI is C++.
// Same names / argument meaning / error behaviour as
//   ColorDepthSearchAlgorithm            API/cds/ColorDepthSearchAlgorithm.java:17-62
//   ColorDepthSearchAlgorithmProvider    API/cds/ColorDepthSearchAlgorithmProvider.java:12-38
//   ColorDepthSearchAlgorithmProviderFactory  API/cds/ColorDepthSearchAlgorithmProviderFactory.java:30-127
//   ColorMIPSearch                       API/cds/ColorMIPSearch.java:12-47
//   PixelMatchScore / ShapeMatchScore    API/cds/PixelMatchScore.java:3-31, API/cds/ShapeMatchScore.java:5-65
// (API/ = colormipsearch-api/src/main/java/org/janelia/colormipsearch/).  IllegalArgumentException maps to
// std::invalid_argument, every other failure to std::runtime_error.  Link with -lcds_hip.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <vector>

#include "cds_hip.h"

namespace cds {

inline void check(int rc) {
    if (rc == CDS_OK) return;
    std::string msg = cds_last_error();
    if (rc == CDS_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error("[cds " + std::to_string(rc) + "] " + msg);
}

enum class ComputeFileType { InputColorDepthImage, GradientImage, ZGapImage };  // API/model/ComputeFileType.java:5-28

// ImageArray<?> of the reference: type + width + height + pixels (API/imageprocessing/ImageArray.java:12-68).
struct ImageArray {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;    // packed R,G,B (ColorImageArray) -- or
    std::vector<uint16_t> gray;  // GRAY16 / widened GRAY8 (ShortImageArray / ByteImageArray)
    bool isColor() const { return !rgb.empty(); }
};

// ImageRegionDefinition as rectangles; labelRegions == AbstractColorDepthMatchArgs.getRegionGeneratorForTextLabels
// (colormipsearch-tools/.../cmd/AbstractColorDepthMatchArgs.java:101-119).
using Regions = std::vector<cds_rect>;
inline Regions labelRegions(int width, bool nameLabel = true, bool colorScaleLabel = true) {
    Regions r;
    if (colorScaleLabel && width > 270) r.push_back({width - 270, 0, width, 90});
    if (nameLabel) r.push_back({0, 0, 330, 100});
    return r;
}

class Context {
public:
    explicit Context(int device = 0) { check(cds_context_create(device, &h_)); }
    ~Context() { cds_context_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    cds_context *handle() const { return h_; }
    void setStream(void *hipStream) { check(cds_context_set_stream(h_, hipStream)); }
    void synchronize() { check(cds_context_synchronize(h_)); }
    uint64_t trim() { uint64_t freed = 0; check(cds_context_trim(h_, &freed)); return freed; }
    void setCacheLimit(uint64_t bytes) { check(cds_context_set_cache_limit(h_, bytes)); }
private:
    cds_context *h_ = nullptr;
};

// Kernel 2's inputs of a library in one file (cds_variant_shard_*): written once, from memory or from the library's image files (no GPU
// needed), then any subset of targets is decoded on the GPU by Library::loadVariants.
class VariantShardWriter {
public:
    VariantShardWriter(const char *path, int width, int height) { check(cds_variant_shard_create(path, width, height, &w_)); }
    ~VariantShardWriter() { if (w_) cds_variant_shard_finish(w_); }
    VariantShardWriter(const VariantShardWriter &) = delete;
    VariantShardWriter &operator=(const VariantShardWriter &) = delete;
    void append(int64_t n, const uint8_t *rgb, const uint16_t *gradients, const uint8_t *zgaps, int threads = 0) {
        check(cds_variant_shard_append(w_, n, rgb, gradients, zgaps, threads));
    }
    void appendFiles(const std::vector<const char *> &targets, const std::vector<const char *> &gradients, const std::vector<const char *> &zgaps,
                     int threads = 0) {
        check(cds_variant_shard_append_files(w_, (int64_t)targets.size(), targets.data(), gradients.empty() ? nullptr : gradients.data(),
                                             zgaps.empty() ? nullptr : zgaps.data(), threads));
    }
    void finish() {
        cds_variant_writer *w = w_;
        w_ = nullptr;
        if (w) check(cds_variant_shard_finish(w));
    }
private:
    cds_variant_writer *w_ = nullptr;
};

class VariantShard {
public:
    explicit VariantShard(const char *path) {
        check(cds_variant_shard_open(path, &s_));
        check(cds_variant_shard_info(s_, &w_, &h_, &n_, &bytes_));
    }
    ~VariantShard() { cds_variant_shard_close(s_); }
    VariantShard(const VariantShard &) = delete;
    VariantShard &operator=(const VariantShard &) = delete;
    const cds_variant_shard *handle() const { return s_; }
    int width() const { return w_; }
    int height() const { return h_; }
    int64_t size() const { return n_; }
    int64_t fileBytes() const { return bytes_; }
    uint32_t flags(int64_t target) const {   // CDS_VARIANT_HAS_GRADIENT | CDS_VARIANT_GRADIENT_DELTA2 | CDS_VARIANT_HAS_ZGAP
        uint32_t f = 0;
        check(cds_variant_shard_target_info(s_, target, 1, &f, nullptr));
        return f;
    }
private:
    cds_variant_shard *s_ = nullptr;
    int32_t w_ = 0, h_ = 0;
    int64_t n_ = 0, bytes_ = 0;
};

// Resident shard of targets (counterpart of the host-side MIP cache, CachedMIPsUtils.java:60-100).
class Library {
public:
    Library(Context &ctx, int width, int height, int64_t capacity) : w_(width), h_(height), cap_(capacity) {
        check(cds_library_create(ctx.handle(), width, height, capacity, &lib_));
    }
    ~Library() { cds_library_destroy(lib_); }
    Library(const Library &) = delete;
    Library &operator=(const Library &) = delete;
    cds_library *handle() const { return lib_; }
    int64_t capacity() const { return cap_; }
    void upload(int64_t first, int64_t count, const uint8_t *rgb) { check(cds_library_upload(lib_, first, count, rgb)); }
    void uploadVariants(int64_t first, int64_t count, const uint16_t *grad, const uint8_t *zgap) {
        check(cds_library_upload_variants(lib_, first, count, grad, zgap));
    }
    // targets + their GradientImage / ZGapImage files (either vector may be empty, single entries NULL) decoded natively into the slots
    void loadFiles(int64_t first, const std::vector<const char *> &targets, const std::vector<const char *> &gradients,
                   const std::vector<const char *> &zgaps, int threads = 0) {
        check(cds_library_load_files(lib_, first, (int64_t)targets.size(), targets.data(), gradients.empty() ? nullptr : gradients.data(),
                                     zgaps.empty() ? nullptr : zgaps.data(), threads));
    }
    // slots first.. <- the shard's targets `targets` (image, gradient, zgap as stored), decoded on the device
    void loadVariants(int64_t first, const VariantShard &shard, const std::vector<int64_t> &targets, int threads = 0) {
        check(cds_library_load_variants(lib_, first, shard.handle(), targets.data(), (int64_t)targets.size(), threads));
    }
    void loadVariantImages(int64_t first, const VariantShard &shard, const std::vector<int64_t> &targets, int threads = 0) {   // the MIPs alone
        check(cds_library_load_variant_images(lib_, first, shard.handle(), targets.data(), (int64_t)targets.size(), threads));
    }
    void uploadGradientU8(int64_t first, int64_t count, const uint8_t *grad8) {   // a GRAY8 gradient image (ByteImageArray.java:3-16)
        check(cds_library_upload_gradient_u8(lib_, first, count, grad8));
    }
    void generateSynthetic(uint64_t seed, int64_t firstIndex, int64_t firstSlot, int64_t count, int kind, bool variants = false) {
        check(cds_library_generate_synthetic(lib_, seed, firstIndex, firstSlot, count, kind, variants ? 1 : 0));
    }
private:
    cds_library *lib_ = nullptr;
    int w_, h_;
    int64_t cap_;
};

// Resident shard of targets kept as signal-pixel records (pixel-match scan only): ~4 % of the dense bytes.
class SparseLibrary {
public:
    SparseLibrary(Context &ctx, int width, int height, int64_t capacityImages, int64_t capacityPixels) {
        check(cds_sparse_library_create(ctx.handle(), width, height, capacityImages, capacityPixels, &lib_));
    }
    ~SparseLibrary() { cds_sparse_library_destroy(lib_); }
    SparseLibrary(const SparseLibrary &) = delete;
    SparseLibrary &operator=(const SparseLibrary &) = delete;
    cds_sparse_library *handle() const { return lib_; }
    int64_t size() const { return cds_sparse_library_size(lib_); }
    int64_t pixels() const { return cds_sparse_library_pixels(lib_); }
    void appendFrom(const Library &dense, int64_t first, int64_t count) { check(cds_sparse_library_append(lib_, dense.handle(), first, count)); }
    void append(int64_t count, const uint8_t *rgb) { check(cds_sparse_library_append_host(lib_, count, rgb)); }
    void clear() { check(cds_sparse_library_clear(lib_)); }
    // pixels at or below the searches' data threshold need not be resident (searches below it are refused)
    void setMinDataThreshold(int targetThreshold) { check(cds_sparse_library_set_min_threshold(lib_, targetThreshold)); }
    // appends enlarge the library when a capacity is lacking (the reference's MIP cache sizes itself with use, CachedMIPsUtils.java:60-100)
    void setGrowable(bool on = true) { check(cds_sparse_library_set_growable(lib_, on ? 1 : 0)); }
    void reserve(int64_t capacityImages, int64_t capacityPixels) { check(cds_sparse_library_reserve(lib_, capacityImages, capacityPixels)); }
    int64_t capacityImages() const { int64_t v = 0; check(cds_sparse_library_capacity(lib_, &v, nullptr, nullptr)); return v; }
    int64_t capacityPixels() const { int64_t v = 0; check(cds_sparse_library_capacity(lib_, nullptr, &v, nullptr)); return v; }
    bool hasStream16() const { int32_t v = 0; check(cds_sparse_library_capacity(lib_, nullptr, nullptr, &v)); return v != 0; }
private:
    cds_sparse_library *lib_ = nullptr;
};

struct ColorDepthMatchScore {  // API/cds/ColorDepthMatchScore.java:5-22
    virtual ~ColorDepthMatchScore() = default;
    virtual int getScore() const = 0;
    virtual float getNormalizedScore() const = 0;
    virtual bool isMirrored() const = 0;
};

class PixelMatchScore : public ColorDepthMatchScore {
public:
    PixelMatchScore(int n, double ratio, bool mirrored) : n_(n), ratio_(ratio), mirrored_(mirrored) {}
    int getScore() const override { return n_; }
    float getNormalizedScore() const override { return (float)ratio_; }
    bool isMirrored() const override { return mirrored_; }
private:
    int n_;
    double ratio_;
    bool mirrored_;
};

class ShapeMatchScore : public ColorDepthMatchScore {
public:
    ShapeMatchScore(int64_t gap, int64_t high, int64_t maxGap, bool mirrored) : gap_(gap), high_(high), max_(maxGap), mirrored_(mirrored) {}
    int getScore() const override { return (int)cds_shape_score_value(gap_, high_); }
    float getNormalizedScore() const override { long s = getScore(); return max_ > 0 ? s / (float)max_ : (float)s; }
    bool isMirrored() const override { return mirrored_; }
    int64_t getGradientAreaGap() const { return gap_; }
    int64_t getHighExpressionArea() const { return high_; }
private:
    int64_t gap_, high_, max_;
    bool mirrored_;
};

using VariantSuppliers = std::map<ComputeFileType, std::function<const ImageArray *()>>;

template <class S>
class ColorDepthSearchAlgorithm {  // API/cds/ColorDepthSearchAlgorithm.java:17-62
public:
    virtual ~ColorDepthSearchAlgorithm() = default;
    virtual const ImageArray &getQueryImage() const = 0;
    virtual int getQuerySize() const = 0;
    virtual std::set<ComputeFileType> getRequiredTargetVariantTypes() const = 0;
    virtual S calculateMatchingScore(const ImageArray &target, const VariantSuppliers &variantImageSuppliers) const = 0;
};

class PixelMatchColorDepthSearchAlgorithm : public ColorDepthSearchAlgorithm<PixelMatchScore> {
public:
    PixelMatchColorDepthSearchAlgorithm(Context &ctx, const ImageArray &query, int queryThreshold, bool mirrorQuery,
                                        int targetThreshold, double zTolerance, int xyShift, const Regions &excluded)
        : query_(query) {
        cds_pixmatch_params p{queryThreshold, targetThreshold, zTolerance, xyShift, mirrorQuery ? 1 : 0};
        check(cds_maskset_create(ctx.handle(), 1, query.rgb.data(), query.width, query.height, &p, excluded.data(),
                                 (int32_t)excluded.size(), &ms_));
        check(cds_maskset_query_sizes(ms_, &size_));
    }
    ~PixelMatchColorDepthSearchAlgorithm() override { cds_maskset_destroy(ms_); }
    const ImageArray &getQueryImage() const override { return query_; }
    int getQuerySize() const override { return size_; }
    std::set<ComputeFileType> getRequiredTargetVariantTypes() const override { return {}; }
    PixelMatchScore calculateMatchingScore(const ImageArray &target, const VariantSuppliers &) const override {
        if (size_ == 0) return PixelMatchScore(0, 0, false);
        int32_t s = 0;
        uint8_t m = 0;
        check(cds_score_pair(ms_, 0, target.rgb.data(), target.width, target.height, &s, &m));
        return PixelMatchScore(s, (double)s / (double)size_, m != 0);
    }
    // GPU-native granularity: the whole resident library in one launch
    void searchLibrary(const Library &lib, int64_t first, int64_t count, std::vector<int32_t> &scores, std::vector<uint8_t> &mirrored) const {
        scores.assign((size_t)count, 0);
        mirrored.assign((size_t)count, 0);
        check(cds_search(ms_, lib.handle(), first, count, scores.data(), mirrored.data()));
    }
    void searchLibrary(const SparseLibrary &lib, int64_t first, int64_t count, std::vector<int32_t> &scores, std::vector<uint8_t> &mirrored) const {
        scores.assign((size_t)count, 0);
        mirrored.assign((size_t)count, 0);
        check(cds_search_sparse(ms_, lib.handle(), first, count, scores.data(), mirrored.data()));
    }
    // scan + ColorMIPSearch.isMatch + compaction on the device: the pairs findAllColorDepthMatches keeps (mask, target, score | mirrored << 31)
    std::vector<cds_match> searchLibraryMatches(const SparseLibrary &lib, int64_t first, int64_t count, double pctPositivePixels) const {
        cds_match_list *list = nullptr;
        int64_t n = 0;
        check(cds_search_sparse_match_list(ms_, lib.handle(), first, count, pctPositivePixels, &list, &n));   // built on the device, exact length
        std::vector<cds_match> out((size_t)n);
        const int rc = cds_match_list_fetch(list, 0, n, out.data());
        cds_match_list_destroy(list);
        check(rc);
        return out;
    }
    // the same rows as the columns of a result table (cds_match_list_fetch_columns: split by the library's host threads while the list comes down)
    struct MatchColumns {
        std::vector<int32_t> mask, matchingPixels;
        std::vector<int64_t> target;
        std::vector<float> matchingPixelsRatio;
        std::vector<uint8_t> mirrored;
    };
    MatchColumns searchLibraryMatchColumns(const SparseLibrary &lib, int64_t first, int64_t count, double pctPositivePixels, int32_t maskOffset = 0) const {
        cds_match_list *list = nullptr;
        int64_t n = 0;
        check(cds_search_sparse_match_list(ms_, lib.handle(), first, count, pctPositivePixels, &list, &n));
        MatchColumns c;
        c.mask.resize((size_t)n), c.target.resize((size_t)n), c.matchingPixels.resize((size_t)n), c.matchingPixelsRatio.resize((size_t)n), c.mirrored.resize((size_t)n);
        std::vector<int32_t> sizes((size_t)cds_maskset_size(ms_));
        int rc = cds_maskset_query_sizes(ms_, sizes.data());
        if (rc == CDS_OK)
            rc = cds_match_list_fetch_columns(list, 0, n, sizes.data(), (int32_t)sizes.size(), maskOffset, c.mask.data(), c.target.data(), c.matchingPixels.data(),
                                              c.matchingPixelsRatio.data(), c.mirrored.data());
        cds_match_list_destroy(list);
        check(rc);
        return c;
    }
    // the k best-scoring targets, selected on the device (score descending, equal scores by ascending target)
    void searchLibraryTopK(const SparseLibrary &lib, int64_t first, int64_t count, int k, std::vector<int32_t> &targets, std::vector<int32_t> &packedScores) const {
        targets.assign((size_t)k, -1);
        packedScores.assign((size_t)k, 0);
        check(cds_search_sparse_topk(ms_, lib.handle(), first, count, k, targets.data(), packedScores.data()));
    }
    const cds_maskset *handle() const { return ms_; }   // (for ShardedSearch: one mask set per context)
private:
    ImageArray query_;
    cds_maskset *ms_ = nullptr;
    int32_t size_ = 0;
};

// The target-sharded search over the GPUs of one node: contexts[r] owns shard r (cds_comm; one RCCL gather per search).
class ShardedSearch {
public:
    explicit ShardedSearch(const std::vector<cds_context *> &contexts) { check(cds_comm_create(contexts.data(), (int32_t)contexts.size(), &comm_)); }
    ~ShardedSearch() { cds_comm_destroy(comm_); }
    ShardedSearch(const ShardedSearch &) = delete;
    ShardedSearch &operator=(const ShardedSearch &) = delete;
    int size() const { return cds_comm_size(comm_); }
    bool usesRccl() const { return cds_comm_uses_rccl(comm_) != 0; }
    // masksets[r] / shards[r] were created through contexts[r]; scores[m * total + offsets[r] + t]
    void search(const std::vector<const cds_maskset *> &masksets, const std::vector<const cds_sparse_library *> &shards, int nMasks,
                std::vector<int64_t> &offsets, std::vector<int32_t> &scores, std::vector<uint8_t> &mirrored) const {
        offsets.assign(shards.size() + 1, 0);
        int64_t total = 0;
        for (const cds_sparse_library *s : shards) total += cds_sparse_library_size(s);
        scores.assign((size_t)nMasks * (size_t)total, 0);
        mirrored.assign((size_t)nMasks * (size_t)total, 0);
        check(cds_search_sparse_sharded(comm_, masksets.data(), shards.data(), offsets.data(), scores.data(), mirrored.data()));
    }
    // the k best-scoring targets of every mask over all shards: targets[m * k + i] (library index, -1 behind the last), packedScores likewise
    void searchTopK(const std::vector<const cds_maskset *> &masksets, const std::vector<const cds_sparse_library *> &shards, int nMasks, int k,
                    std::vector<int64_t> &offsets, std::vector<int32_t> &targets, std::vector<int32_t> &packedScores) const {
        offsets.assign(shards.size() + 1, 0);
        targets.assign((size_t)nMasks * (size_t)k, -1);
        packedScores.assign((size_t)nMasks * (size_t)k, 0);
        check(cds_search_sparse_sharded_topk(comm_, masksets.data(), shards.data(), k, offsets.data(), targets.data(), packedScores.data()));
    }
    // isMatch + compaction on every GPU: the matches of all shards by mask, then library index (cds_match.target = offsets[r] + t)
    std::vector<cds_match> searchMatches(const std::vector<const cds_maskset *> &masksets, const std::vector<const cds_sparse_library *> &shards,
                                         double pctPositivePixels, std::vector<int64_t> &offsets) const {
        offsets.assign(shards.size() + 1, 0);
        std::vector<cds_match> out((size_t)1 << 16);
        for (;;) {
            int64_t n = 0;
            const int rc = cds_search_sparse_sharded_matches(comm_, masksets.data(), shards.data(), pctPositivePixels, offsets.data(), out.data(),
                                                             (int64_t)out.size(), &n);
            if (rc == CDS_ERR_OUT_OF_MEMORY && n > (int64_t)out.size()) {   // (longer than the first guess: once more, exactly sized)
                out.resize((size_t)n);
                continue;
            }
            check(rc);
            out.resize((size_t)n);
            return out;
        }
    }
    // kernel 2 over dense shards: GradientAreaGap of (mask index, LIBRARY index) pairs, every pair scored on the GPU owning its target
    // (shapesets[r] / libs[r] made through contexts[r]; shard r = library indices [offsets[r], offsets[r] + shardSizes[r]) in its slots 0..)
    void gradScores(const std::vector<const cds_shapeset *> &shapesets, const std::vector<const cds_library *> &libs,
                    const std::vector<int64_t> &shardSizes, const std::vector<int32_t> &pairMask, const std::vector<int64_t> &pairTarget,
                    std::vector<int64_t> &offsets, std::vector<int64_t> &gradientAreaGap, std::vector<int64_t> &highExpressionArea,
                    std::vector<uint8_t> &mirrored) const {
        offsets.assign(libs.size() + 1, 0);
        gradientAreaGap.assign(pairMask.size(), 0);
        highExpressionArea.assign(pairMask.size(), 0);
        mirrored.assign(pairMask.size(), 0);
        check(cds_gradscore_sharded(comm_, shapesets.data(), libs.data(), shardSizes.empty() ? nullptr : shardSizes.data(), (int64_t)pairMask.size(),
                                    pairMask.data(), pairTarget.data(), offsets.data(), gradientAreaGap.data(), highExpressionArea.data(), mirrored.data()));
    }
private:
    cds_comm *comm_ = nullptr;
};

class Shape2DMatchColorDepthSearchAlgorithm : public ColorDepthSearchAlgorithm<ShapeMatchScore> {
public:
    Shape2DMatchColorDepthSearchAlgorithm(Context &ctx, const ImageArray &query, int queryThreshold, bool mirrorQuery,
                                          const uint8_t *roiMask, const Regions &excluded)
        : query_(query), thr_(queryThreshold) {
        check(cds_shapeset_create(ctx.handle(), 1, query.rgb.data(), query.width, query.height, queryThreshold,
                                  mirrorQuery ? 1 : 0, roiMask, excluded.data(), (int32_t)excluded.size(), &ss_));
    }
    ~Shape2DMatchColorDepthSearchAlgorithm() override { cds_shapeset_destroy(ss_); }
    const ImageArray &getQueryImage() const override { return query_; }
    int getQuerySize() const override {
        int n = 0;
        for (size_t i = 0; i + 2 < query_.rgb.size(); i += 3)
            n += (query_.rgb[i] > thr_ || query_.rgb[i + 1] > thr_ || query_.rgb[i + 2] > thr_) ? 1 : 0;
        return n;
    }
    std::set<ComputeFileType> getRequiredTargetVariantTypes() const override { return {ComputeFileType::GradientImage, ComputeFileType::ZGapImage}; }
    ShapeMatchScore calculateMatchingScore(const ImageArray &target, const VariantSuppliers &sup) const override {
        auto g = sup.find(ComputeFileType::GradientImage), z = sup.find(ComputeFileType::ZGapImage);
        const ImageArray *grad = g != sup.end() ? g->second() : nullptr, *zgap = z != sup.end() ? z->second() : nullptr;
        if (!grad || !zgap) return ShapeMatchScore(-1, -1, -1, false);  // Shape2DMatch...Algorithm.java:155-158
        int64_t gap = 0, high = 0;
        uint8_t m = 0;
        check(cds_shape_score_pair(ss_, 0, target.rgb.data(), grad->gray.data(), zgap->rgb.data(), target.width, target.height, &gap, &high, &m));
        return ShapeMatchScore(gap, high, -1, m != 0);
    }
    void maskStats(int64_t &maskSize, int64_t &highExpressionSize) const { check(cds_shapeset_mask_stats(ss_, &maskSize, &highExpressionSize)); }
    const cds_shapeset *handle() const { return ss_; }   // (for ShardedSearch::gradScores: one shape set per context)
private:
    ImageArray query_;
    int thr_;
    cds_shapeset *ss_ = nullptr;
};

// ColorDepthSearchParams: the four keys the factory reads (Factory:39-43,55-68)
struct ColorDepthSearchParams {
    std::optional<bool> mirrorMask;
    std::optional<int> dataThreshold, xyShift;
    std::optional<double> pixColorFluctuation;
};

template <class S>
class ColorDepthSearchAlgorithmProvider {  // API/cds/ColorDepthSearchAlgorithmProvider.java:12-38
public:
    virtual ~ColorDepthSearchAlgorithmProvider() = default;
    virtual ColorDepthSearchParams getDefaultCDSParams() const = 0;
    virtual std::unique_ptr<ColorDepthSearchAlgorithm<S>> createColorDepthSearchAlgorithm(const ImageArray &query, int queryThreshold,
                                                                                         int queryBorderSize, const ColorDepthSearchParams &p) const = 0;
    std::unique_ptr<ColorDepthSearchAlgorithm<S>> createColorDepthQuerySearchAlgorithmWithDefaultParams(const ImageArray &query, int queryThreshold,
                                                                                                       int queryBorderSize) const {
        return createColorDepthSearchAlgorithm(query, queryThreshold, queryBorderSize, ColorDepthSearchParams{});
    }
};

class ColorDepthSearchAlgorithmProviderFactory {  // API/cds/ColorDepthSearchAlgorithmProviderFactory.java:30-127
public:
    static std::unique_ptr<ColorDepthSearchAlgorithmProvider<PixelMatchScore>> createPixMatchCDSAlgorithmProvider(
        Context &ctx, bool mirrorMask, int targetThreshold, double pixColorFluctuation, int xyShiftParam, const Regions &ignored) {
        struct P : ColorDepthSearchAlgorithmProvider<PixelMatchScore> {
            Context &ctx; bool m; int t; double f; int s; Regions r;
            P(Context &c, bool m_, int t_, double f_, int s_, Regions r_) : ctx(c), m(m_), t(t_), f(f_), s(s_), r(std::move(r_)) {}
            ColorDepthSearchParams getDefaultCDSParams() const override { return {m, t, s, f}; }
            std::unique_ptr<ColorDepthSearchAlgorithm<PixelMatchScore>> createColorDepthSearchAlgorithm(const ImageArray &q, int thr, int, const ColorDepthSearchParams &p) const override {
                double zTolerance = p.pixColorFluctuation.value_or(f) / 100;  // Factory:55-56
                int xy = p.xyShift.value_or(s);
                if ((xy & 1) == 1) throw std::invalid_argument("XY shift parameter must be an even number.");  // Factory:58-60
                return std::make_unique<PixelMatchColorDepthSearchAlgorithm>(ctx, q, thr, p.mirrorMask.value_or(m), p.dataThreshold.value_or(t), zTolerance, xy, r);
            }
        };
        return std::make_unique<P>(ctx, mirrorMask, targetThreshold, pixColorFluctuation, xyShiftParam, ignored);
    }
    static std::unique_ptr<ColorDepthSearchAlgorithmProvider<ShapeMatchScore>> createShapeMatchCDSAlgorithmProvider(
        Context &ctx, bool mirrorMask, const uint8_t *roiMask, const Regions &excluded) {
        struct P : ColorDepthSearchAlgorithmProvider<ShapeMatchScore> {
            Context &ctx; bool m; const uint8_t *roi; Regions r;
            P(Context &c, bool m_, const uint8_t *roi_, Regions r_) : ctx(c), m(m_), roi(roi_), r(std::move(r_)) {}
            ColorDepthSearchParams getDefaultCDSParams() const override { ColorDepthSearchParams p; p.mirrorMask = m; return p; }
            std::unique_ptr<ColorDepthSearchAlgorithm<ShapeMatchScore>> createColorDepthSearchAlgorithm(const ImageArray &q, int thr, int border, const ColorDepthSearchParams &p) const override {
                if (border != 0) throw std::runtime_error("queryBorderSize != 0 is not supported");
                return std::make_unique<Shape2DMatchColorDepthSearchAlgorithm>(ctx, q, thr, p.mirrorMask.value_or(m), roi, r);
            }
        };
        return std::make_unique<P>(ctx, mirrorMask, roiMask, excluded);
    }
};

class ColorMIPSearch {  // API/cds/ColorMIPSearch.java:12-47
public:
    ColorMIPSearch(std::optional<double> pctPositivePixels, std::optional<int> defaultQueryThreshold,
                   std::shared_ptr<ColorDepthSearchAlgorithmProvider<PixelMatchScore>> provider)
        : pct_(pctPositivePixels), thr_(defaultQueryThreshold), provider_(std::move(provider)) {}
    std::unique_ptr<ColorDepthSearchAlgorithm<PixelMatchScore>> createQueryColorDepthSearchWithDefaultThreshold(const ImageArray &query) const {
        return provider_->createColorDepthQuerySearchAlgorithmWithDefaultParams(query, thr_.value_or(0), 0);
    }
    bool isMatch(const PixelMatchScore &s) const {
        double t = pct_ ? *pct_ / 100 : 0.;
        return s.getScore() > 0 && s.getNormalizedScore() > t;
    }
private:
    std::optional<double> pct_;
    std::optional<int> thr_;
    std::shared_ptr<ColorDepthSearchAlgorithmProvider<PixelMatchScore>> provider_;
};

}  // namespace cds
