// Exercises the C++ mirror (include/cds_hip.hpp) the way the reference's JUnit tests exercise the Java plugin
// surface: provider factory -> algorithm -> calculateMatchingScore, plus the library-level entry point.
// Prints "OK <self score> <mirrored score> <mirrored flag>" on success.
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <cstdlib>

#include "cds_hip.hpp"

int main() {
    const int W = 1210, H = 566;
    try {
        cds::Context ctx(0);
        cds::Library lib(ctx, W, H, 4);
        lib.generateSynthetic(0xBADC0DEull, 0, 0, 1, 2);     // slot 0: the benchmark mask
        lib.generateSynthetic(0xC0FFEEull, 0, 1, 2, 0);      // slots 1,2: sparse targets
        cds::ImageArray mask;
        mask.width = W; mask.height = H; mask.rgb.resize((size_t)3 * W * H);
        cds::check(cds_library_download(lib.handle(), 0, 1, mask.rgb.data()));
        cds::ImageArray flipped = mask;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                for (int c = 0; c < 3; c++) flipped.rgb[3 * ((size_t)y * W + x) + c] = mask.rgb[3 * ((size_t)y * W + (W - 1 - x)) + c];
        lib.upload(3, 1, flipped.rgb.data());

        auto provider = cds::ColorDepthSearchAlgorithmProviderFactory::createPixMatchCDSAlgorithmProvider(ctx, true, 20, 1.0, 2, cds::labelRegions(W));
        auto alg = provider->createColorDepthQuerySearchAlgorithmWithDefaultParams(mask, 20, 0);
        cds::PixelMatchScore self = alg->calculateMatchingScore(mask, {});
        cds::PixelMatchScore mir = alg->calculateMatchingScore(flipped, {});
        if (self.getScore() <= 0 || self.getScore() > alg->getQuerySize() || self.isMirrored()) return 2;
        if (!mir.isMirrored() || mir.getScore() != self.getScore()) return 3;

        auto *pm = dynamic_cast<cds::PixelMatchColorDepthSearchAlgorithm *>(alg.get());
        std::vector<int32_t> scores;
        std::vector<uint8_t> mirrored;
        pm->searchLibrary(lib, 0, 4, scores, mirrored);
        if (scores[0] != self.getScore() || scores[3] != mir.getScore() || !mirrored[3] || mirrored[0]) return 4;

        // the same four targets as a sparse resident library: identical scores
        cds::SparseLibrary slib(ctx, W, H, 4, (int64_t)4 * W * H);
        slib.appendFrom(lib, 0, 3);
        slib.append(1, flipped.rgb.data());
        std::vector<int32_t> s2;
        std::vector<uint8_t> m2;
        pm->searchLibrary(slib, 0, 4, s2, m2);
        if (slib.size() != 4 || s2 != scores || m2 != mirrored) return 7;

        // match list and top-k on the device: the two targets that match (the mask itself and its mirror image)
        std::vector<cds_match> ml = pm->searchLibraryMatches(slib, 0, 4, 1.0);
        if (ml.size() != 2 || ml[0].target != 0 || ml[1].target != 3 || ml[0].score != scores[0] || ml[1].score != (int32_t)((uint32_t)scores[3] | 0x80000000u)) return 8;
        {
            auto mc = pm->searchLibraryMatchColumns(slib, 0, 4, 1.0, 10);   // the same two rows as table columns, mask offset 10
            if (mc.mask.size() != 2 || mc.mask[0] != 10 || mc.target[1] != 3 || mc.matchingPixels[0] != scores[0] || mc.mirrored[0] != 0 || mc.mirrored[1] != 1 ||
                mc.matchingPixelsRatio[0] <= 0.0f || mc.matchingPixelsRatio[0] > 1.0f) return 18;
        }
        std::vector<int32_t> tk, sk;
        pm->searchLibraryTopK(slib, 0, 4, 2, tk, sk);
        if (tk[0] != 0 || tk[1] != 3 || (sk[0] & 0x7fffffff) != scores[0]) return 9;
        // a growable library started far too small holds the same targets and scans the same (round 5: the reference's MIP cache sizes itself
        // with use); the context's block cache can be trimmed and capped
        {
            cds::SparseLibrary tiny(ctx, W, H, 1, 1000);
            tiny.setGrowable();
            tiny.appendFrom(lib, 0, 3);
            tiny.append(1, flipped.rgb.data());
            if (tiny.size() != 4 || tiny.pixels() != slib.pixels() || tiny.capacityImages() < 4 || tiny.capacityPixels() < tiny.pixels()) return 19;
            std::vector<int32_t> gs;
            std::vector<uint8_t> gm;
            pm->searchLibrary(tiny, 0, 4, gs, gm);
            if (gs != scores || gm != mirrored) return 20;
            (void)ctx.trim();
            ctx.setCacheLimit((uint64_t)6 << 30);
        }

        // the same library cut into two shards over two contexts (of the one GPU: the rehearsal path of cds_comm)
        {
            cds::Context ctx2(0);
            auto provider2 = cds::ColorDepthSearchAlgorithmProviderFactory::createPixMatchCDSAlgorithmProvider(ctx2, true, 20, 1.0, 2, cds::labelRegions(W));
            auto alg2 = provider2->createColorDepthQuerySearchAlgorithmWithDefaultParams(mask, 20, 0);
            auto *pm2 = dynamic_cast<cds::PixelMatchColorDepthSearchAlgorithm *>(alg2.get());
            cds::SparseLibrary sa(ctx, W, H, 4, (int64_t)4 * W * H), sb(ctx2, W, H, 4, (int64_t)4 * W * H);
            sa.appendFrom(lib, 0, 3);
            sb.append(1, flipped.rgb.data());
            cds::ShardedSearch sharded({ctx.handle(), ctx2.handle()});
            std::vector<int64_t> offs;
            std::vector<int32_t> s3;
            std::vector<uint8_t> m3;
            sharded.search({pm->handle(), pm2->handle()}, {sa.handle(), sb.handle()}, 1, offs, s3, m3);
            if (offs != std::vector<int64_t>({0, 3, 4}) || s3 != scores || m3 != mirrored) return 10;
            std::vector<cds_match> sm = sharded.searchMatches({pm->handle(), pm2->handle()}, {sa.handle(), sb.handle()}, 1.0, offs);
            if (sm.size() != ml.size() || sm[0].target != 0 || sm[1].target != 3 || sm[1].score != ml[1].score) return 11;

            // kernel 2 over the same two contexts (cds_gradscore_sharded): library indices 0..4 as dense shards of 3 + 2 targets with
            // their gradient / zgap variants; every pair goes to the context owning its target == cds_gradscore over one library
            cds::Library whole(ctx, W, H, 5), la(ctx, W, H, 3), lb(ctx2, W, H, 2);
            whole.generateSynthetic(0xC0FFEEull, 100, 0, 5, 0, true);
            la.generateSynthetic(0xC0FFEEull, 100, 0, 3, 0, true);
            lb.generateSynthetic(0xC0FFEEull, 103, 0, 2, 0, true);
            cds::Shape2DMatchColorDepthSearchAlgorithm sh1(ctx, mask, 20, true, nullptr, cds::labelRegions(W)), sh2(ctx2, mask, 20, true, nullptr, cds::labelRegions(W));
            const std::vector<int32_t> pmask = {0, 0, 0, 0, 0, 0};
            const std::vector<int64_t> ptarget = {4, 0, 3, 1, 2, 4};
            std::vector<int64_t> wg(6), wh(6), gg, gh;
            std::vector<uint8_t> wm(6), gm;
            cds::check(cds_gradscore(sh1.handle(), whole.handle(), 6, pmask.data(), ptarget.data(), wg.data(), wh.data(), wm.data()));
            sharded.gradScores({sh1.handle(), sh2.handle()}, {la.handle(), lb.handle()}, {}, pmask, ptarget, offs, gg, gh, gm);
            if (offs != std::vector<int64_t>({0, 3, 5}) || gg != wg || gh != wh || gm != wm || wg[0] <= 0) return 12;

            // the same five targets through a variant shard: packed from host memory, decoded on the device in another order, scored
            {
                const size_t px = (size_t)W * H;
                std::vector<uint8_t> rgb(5 * 3 * px), zg(5 * 3 * px);
                std::vector<uint16_t> gr(5 * px);
                cds::check(cds_library_download(whole.handle(), 0, 5, rgb.data()));
                cds::check(cds_library_download_variants(whole.handle(), 0, 5, gr.data(), zg.data()));
                char path[] = "/tmp/cds_mirror_shard_XXXXXX";
                const int fd = mkstemp(path);
                if (fd < 0) return 21;
                close(fd);
                {
                    cds::VariantShardWriter w(path, W, H);
                    w.append(3, rgb.data(), gr.data(), zg.data(), 2);
                    w.append(2, rgb.data() + 3 * 3 * px, gr.data() + 3 * px, zg.data() + 3 * 3 * px);
                    w.finish();
                }
                cds::VariantShard shard(path);
                std::remove(path);
                if (shard.size() != 5 || shard.width() != W || !(shard.flags(4) & CDS_VARIANT_HAS_GRADIENT) || !(shard.flags(0) & CDS_VARIANT_HAS_ZGAP)) return 22;
                cds::Library fromShard(ctx, W, H, 5);
                const std::vector<int64_t> order = {4, 0, 3, 1, 2};            // slot k holds target order[k]
                fromShard.loadVariants(0, shard, order, 2);
                std::vector<int64_t> slot(6), sg(6), shh(6);
                std::vector<uint8_t> smm(6);
                for (size_t i = 0; i < 6; i++)
                    for (size_t k = 0; k < 5; k++)
                        if (order[k] == ptarget[i]) slot[i] = (int64_t)k;
                cds::check(cds_gradscore(sh1.handle(), fromShard.handle(), 6, pmask.data(), slot.data(), sg.data(), shh.data(), smm.data()));
                if (sg != wg || shh != wh || smm != wm) return 23;
            }
        }

        cds::ColorMIPSearch search(1.0, 20, std::move(provider));
        if (!search.isMatch(self)) return 5;

        bool threw = false;
        try {
            auto bad = cds::ColorDepthSearchAlgorithmProviderFactory::createPixMatchCDSAlgorithmProvider(ctx, true, 20, 1.0, 3, cds::labelRegions(W));
            bad->createColorDepthQuerySearchAlgorithmWithDefaultParams(mask, 20, 0);
        } catch (const std::invalid_argument &) { threw = true; }
        if (!threw) return 6;
        std::printf("OK %d %d %d %d\n", self.getScore(), mir.getScore(), (int)mir.isMirrored(), alg->getQuerySize());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
