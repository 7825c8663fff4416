package org.janelia.colormipsearch.cds.gpu;

import java.util.List;
import java.util.Map;

import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageArrayBytes;
import org.janelia.colormipsearch.imageprocessing.ImageRegionDefinition;

/**
 * Batch-level search: a resident target library + mask sets scanned against all of it, the match list filtered and compacted
 * on the device (cds_search_sparse_matches).  Parameters come from ColorMIPSearch.getCDSParameters() (cds/ColorMIPSearch.java:26-31:
 * mirrorMask, dataThreshold, pixColorFluctuation, xyShift, pctPositivePixels, defaultMaskThreshold).
 */
public final class GpuBatchSearch<K> implements AutoCloseable {

    /** the matches of one mask set, ordered by mask, then target. */
    public static final class Matches {
        public int count;
        public int[] mask, target, score;
        public boolean[] mirrored;
        public int[] querySize;      // per mask of the set
    }

    private final long ctx;
    private final boolean mirrorMask;
    private final int dataThreshold, xyShift, maskThreshold;
    private final double zTolerance, pctPositivePixels;
    private final ImageRegionDefinition excludedRegions;
    private GpuTargetLibrary<K> library;

    public GpuBatchSearch(int device, Map<String, Object> cdsParameters, ImageRegionDefinition excludedRegions) {
        this.ctx = GpuContext.get(device);
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

    public synchronized void addTarget(K key, ImageArray<?> image) {
        if (library == null) {   // sized for a million 1210 x 566 MIPs of a few percent signal; -Dcds.gpu.library.* override
            long images = Long.getLong("cds.gpu.library.images", 1L << 20);
            long records = Long.getLong("cds.gpu.library.records", 40_000L * images);
            library = new GpuTargetLibrary<>(ctx, image.getWidth(), image.getHeight(), images, records, 64);
            library.setMinDataThreshold(dataThreshold);   // the library lives for this run's searches: pixels the scan never looks at stay out
        }
        library.add(key, image);
    }

    public void targetsComplete() {
        if (library != null) {
            library.flush();
        }
    }

    public int targetCount() {
        return library == null ? 0 : library.size();
    }

    public K target(int index) {
        return library.key(index);
    }

    /** one mask set (all images of the library's size) against every resident target. */
    public Matches findMatches(List<ImageArray<?>> maskImages) {
        Matches out = new Matches();
        int n = maskImages.size();
        out.querySize = new int[n];
        if (n == 0 || library == null || library.size() == 0) {
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
        long maskset = CdsHip.masksetCreate(ctx, n, masks, w, h, maskThreshold, dataThreshold, zTolerance, xyShift, mirrorMask,
                RegionRects.fromRegion(excludedRegions, maskImages.get(0)));
        try {
            CdsHip.masksetQuerySizes(maskset, out.querySize);
            long list = CdsHip.searchSparseMatchList(maskset, library.nativeHandle(), 0, library.size(), pctPositivePixels);
            long found;
            int[] triples;
            try {
                found = CdsHip.matchListSize(list);      // (exact: the list was built on the device, one scan)
                triples = new int[(int) (3 * found)];
                CdsHip.matchListFetch(list, 0, found, triples);
            } finally {
                CdsHip.matchListDestroy(list);
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
            CdsHip.masksetDestroy(maskset);
        }
        return out;
    }

    @Override
    public void close() {
        if (library != null) {
            library.close();
            library = null;
        }
    }
}
