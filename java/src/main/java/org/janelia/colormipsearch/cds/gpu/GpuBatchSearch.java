package org.janelia.colormipsearch.cds.gpu;

import java.util.ArrayList;
import java.util.List;
import java.util.Map;

import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageArrayBytes;
import org.janelia.colormipsearch.imageprocessing.ImageRegionDefinition;

/**
 * Batch-level search: a resident target library + mask sets scanned against all of it, the match list filtered and compacted
 * on the device (cds_search_sparse_matches).  Parameters come from ColorMIPSearch.getCDSParameters() (cds/ColorMIPSearch.java:26-31:
 * mirrorMask, dataThreshold, pixColorFluctuation, xyShift, pctPositivePixels, defaultMaskThreshold).
 *
 * With several devices the library is cut into one shard per GPU -- shard r holds a contiguous range of library indices (library
 * index = the shard's offset + the index inside the shard, DESIGN.md 7); when the number of targets is known (expectTargets) the
 * arriving targets fill shard after shard in equal parts, otherwise they are dealt round-robin so that the shards stay balanced --,
 * the mask set is replicated, every GPU filters and compacts its own matches and the lists come back
 * merged by mask and library index (cds_search_sparse_sharded_matches) -- the counterpart of the reference's partitions of targets
 * whose results are collect()-ed (cdsprocess/SparkColorMIPSearchProcessor.java:50-73).
 */
public final class GpuBatchSearch<K> implements AutoCloseable {

    /** the matches of one mask set, ordered by mask, then target. */
    public static final class Matches {
        public int count;
        public int[] mask, target, score;
        public boolean[] mirrored;
        public int[] querySize;      // per mask of the set
    }

    private final long[] ctxs;       // one context per device
    private final long comm;         // 0 with one device
    private final boolean mirrorMask;
    private final int dataThreshold, xyShift, maskThreshold;
    private final double zTolerance, pctPositivePixels;
    private final ImageRegionDefinition excludedRegions;
    private final List<GpuTargetLibrary<K>> shards = new ArrayList<>();
    private long[] offsets = new long[]{0};   // library index of every shard's first target (set by targetsComplete)
    private int dealt;
    private long expectedTargets = -1;        // expectTargets(): the first guess of the library's size
    private int width = -1, height = -1;      // setImageSize(): the masks' size, which every target must have

    public GpuBatchSearch(int device, Map<String, Object> cdsParameters, ImageRegionDefinition excludedRegions) {
        this(new int[]{device}, cdsParameters, excludedRegions);
    }

    public GpuBatchSearch(int[] devices, Map<String, Object> cdsParameters, ImageRegionDefinition excludedRegions) {
        if (devices.length == 0) {
            throw new IllegalArgumentException("no GPU device given");
        }
        this.ctxs = new long[devices.length];
        for (int i = 0; i < devices.length; i++) {
            ctxs[i] = GpuContext.get(devices[i]);
        }
        this.comm = devices.length > 1 ? CdsHip.commCreate(ctxs) : 0;
        this.mirrorMask = Boolean.parseBoolean(String.valueOf(cdsParameters.getOrDefault("mirrorMask", "false")));
        this.dataThreshold = intParam(cdsParameters, "dataThreshold", 0);
        this.xyShift = intParam(cdsParameters, "xyShift", 0);
        this.maskThreshold = intParam(cdsParameters, "defaultMaskThreshold", 0);
        this.zTolerance = doubleParam(cdsParameters, "pixColorFluctuation", 0.) / 100;
        this.pctPositivePixels = doubleParam(cdsParameters, "pctPositivePixels", 0.);
        this.excludedRegions = excludedRegions;
        if ((xyShift & 1) == 1) {
            throw new IllegalArgumentException("XY shift parameter must be an even number.");
        }
    }

    private static int intParam(Map<String, Object> p, String name, int def) {
        Object v = p.get(name);
        return v == null ? def : (int) Double.parseDouble(v.toString());
    }

    private static double doubleParam(Map<String, Object> p, String name, double def) {
        Object v = p.get(name);
        return v == null ? def : Double.parseDouble(v.toString());
    }

    /** how many targets the caller is going to add: the library's first capacity (it grows when the guess was short). */
    public synchronized void expectTargets(long n) {
        expectedTargets = n;
    }

    /**
     * The image size of the run -- the masks' (PixelMatchColorDepthSearchAlgorithm.java:171-175 compares every target with the
     * query's size).  Without it the first target decides.
     */
    public synchronized void setImageSize(int width, int height) {
        if (shards.isEmpty()) {
            this.width = width;
            this.height = height;
        }
    }

    /**
     * @throws IllegalArgumentException when the image's size is not the run's: the reference's per-pair
     *         "Invalid image size" error -- the caller logs it and goes on with the next target
     *         (cmd/cdsprocess/AbstractColorMIPSearchProcessor.java:80-83)
     */
    public synchronized void addTarget(K key, ImageArray<?> image) {
        if (width < 0) {
            width = image.getWidth();
            height = image.getHeight();
        }
        if (image.getWidth() != width || image.getHeight() != height) {
            throw new IllegalArgumentException(String.format(
                    "Invalid image size - target's image size (%d, %d) must match query's image size: (%d, %d)",
                    image.getWidth(), image.getHeight(), width, height));
        }
        if (shards.isEmpty()) {
            // first capacities: the announced number of targets (else 4,096 per shard) at 20,000 records each -- real LM MIPs carry
            // ~36,000, synthetic ones ~18,000 --; the library is growable, so a short guess costs a copy, not the run.
            // -Dcds.gpu.library.images / .records override.  (Round 4 asked for 2^20 images x 40,000 records for ANY run: 168 GB.)
            long perShard = expectedTargets > 0 ? (expectedTargets + ctxs.length - 1) / ctxs.length : 4096;
            long images = Long.getLong("cds.gpu.library.images", perShard);
            long records = Long.getLong("cds.gpu.library.records", (20_000L + 4L * height) * images);
            for (long ctx : ctxs) {
                GpuTargetLibrary<K> shard = new GpuTargetLibrary<>(ctx, width, height, images, records, 64);
                shard.setMinDataThreshold(dataThreshold);   // the library lives for this run's searches: pixels the scan never looks at stay out
                shards.add(shard);
            }
        }
        int r;
        if (expectedTargets > 0) {   // contiguous ranges of the arrival order, equal parts (the last shard takes any surplus)
            long per = (expectedTargets + shards.size() - 1) / shards.size();
            r = (int) Math.min(shards.size() - 1, dealt / per);
        } else {
            r = dealt % shards.size();
        }
        dealt++;
        shards.get(r).add(key, image);
    }

    public int getWidth() {
        return width;
    }

    public int getHeight() {
        return height;
    }

    public synchronized void targetsComplete() {
        offsets = new long[shards.size() + 1];
        for (int r = 0; r < shards.size(); r++) {
            shards.get(r).flush();
            offsets[r + 1] = offsets[r] + shards.get(r).size();
        }
    }

    public int targetCount() {
        return (int) offsets[offsets.length - 1];
    }

    /** the target behind a library index of a match (shard offset + index inside the shard). */
    public K target(int index) {
        int r = 0;
        while (r + 1 < shards.size() && index >= offsets[r + 1]) {
            r++;
        }
        return shards.get(r).key((int) (index - offsets[r]));
    }

    /** one mask set (all images of the library's size) against every resident target. */
    public Matches findMatches(List<ImageArray<?>> maskImages) {
        Matches out = new Matches();
        int n = maskImages.size();
        out.querySize = new int[n];
        if (n == 0 || targetCount() == 0) {
            out.mask = out.target = out.score = new int[0];
            out.mirrored = new boolean[0];
            return out;
        }
        int w = maskImages.get(0).getWidth(), h = maskImages.get(0).getHeight();
        byte[] masks = new byte[n * 3 * w * h];
        for (int i = 0; i < n; i++) {
            ImageArray<?> m = maskImages.get(i);
            if (m.getWidth() != w || m.getHeight() != h) {
                throw new IllegalArgumentException("masks of one set must have the same size");
            }
            System.arraycopy(ImageArrayBytes.rgb(m), 0, masks, i * 3 * w * h, 3 * w * h);
        }
        long[] masksets = new long[ctxs.length];
        try {
            int[] regions = RegionRects.fromRegion(excludedRegions, maskImages.get(0));
            for (int r = 0; r < ctxs.length; r++) {   // the mask set is tiny: one replica per GPU
                masksets[r] = CdsHip.masksetCreate(ctxs[r], n, masks, w, h, maskThreshold, dataThreshold, zTolerance, xyShift, mirrorMask, regions);
            }
            CdsHip.masksetQuerySizes(masksets[0], out.querySize);
            long found;
            int[] triples;
            if (ctxs.length == 1) {
                long list = CdsHip.searchSparseMatchList(masksets[0], shards.get(0).nativeHandle(), 0, shards.get(0).size(), pctPositivePixels);
                try {
                    found = CdsHip.matchListSize(list);      // (exact: the list was built on the device, one scan)
                    triples = new int[(int) (3 * found)];
                    CdsHip.matchListFetch(list, 0, found, triples);
                } finally {
                    CdsHip.matchListDestroy(list);
                }
            } else {   // every GPU scans, filters and compacts its shard; the lists come back merged by mask and library index
                long[] slibs = new long[ctxs.length];
                for (int r = 0; r < ctxs.length; r++) {
                    slibs[r] = shards.get(r).nativeHandle();
                }
                triples = CdsHip.searchSparseShardedMatches(comm, masksets, slibs, pctPositivePixels);
                found = triples.length / 3;
            }
            out.count = (int) found;
            out.mask = new int[out.count];
            out.target = new int[out.count];
            out.score = new int[out.count];
            out.mirrored = new boolean[out.count];
            for (int i = 0; i < out.count; i++) {
                out.mask[i] = triples[3 * i];
                out.target[i] = triples[3 * i + 1];
                out.score[i] = triples[3 * i + 2] & 0x7FFFFFFF;
                out.mirrored[i] = triples[3 * i + 2] < 0;
            }
        } finally {
            for (long ms : masksets) {
                if (ms != 0) {
                    CdsHip.masksetDestroy(ms);
                }
            }
        }
        return out;
    }

    /**
     * GradientAreaGap scores of (mask index, library index) pairs when the selected targets and their gradient / zgap variants are
     * resident as one dense shard per GPU (cds_gradscore_sharded): every pair is scored on the GPU that owns its target, all GPUs at
     * once -- the gradient-score step (cmd/CalculateGradientScoresCmd.java:541-586) over the target partitions of
     * cdsprocess/SparkColorMIPSearchProcessor.java:50-73.  shapesets[r] / denseShards[r] belong to device r of this search;
     * returns {gradientAreaGap, highExpressionArea, mirrored (0 / 1)} in pair order.
     */
    public long[][] gradientScoresSharded(long[] shapesets, long[] denseShards, long[] shardSizes, int[] pairMask, long[] pairTarget) {
        int n = pairMask.length;
        long[] gap = new long[n], high = new long[n];
        byte[] mirrored = new byte[n];
        if (comm == 0) {   // one device: the plain call (slot = library index)
            CdsHip.gradscore(shapesets[0], denseShards[0], pairMask, pairTarget, gap, high, mirrored);
        } else {
            CdsHip.gradscoreSharded(comm, shapesets, denseShards, shardSizes, pairMask, pairTarget, gap, high, mirrored);
        }
        long[] mir = new long[n];
        for (int i = 0; i < n; i++) {
            mir[i] = mirrored[i];
        }
        return new long[][]{gap, high, mir};
    }

    @Override
    public void close() {
        for (GpuTargetLibrary<K> shard : shards) {
            shard.close();
        }
        shards.clear();
        if (comm != 0) {
            CdsHip.commDestroy(comm);
        }
    }
}
