package org.janelia.colormipsearch.cds.gpu;

import java.util.ArrayList;
import java.util.List;
import java.util.function.BiPredicate;

import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageRegionDefinition;

/**
 * The C ABI takes excluded regions as at most 8 half-open rectangles; the reference describes them as a predicate
 * (imageprocessing/ImageRegionDefinition.java).  Every caller in the reference builds a union of label rectangles
 * (cmd/AbstractColorDepthMatchArgs.java:101-119), so the predicate is sampled over the image and decomposed into maximal
 * row runs merged over consecutive rows; a region that needs more than 8 rectangles is refused.
 */
final class RegionRects {
    static final int MAX_RECTS = 8;

    private RegionRects() {
    }

    static int[] fromRegion(ImageRegionDefinition regions, ImageArray<?> image) {
        if (regions == null) {
            return new int[0];
        }
        BiPredicate<Integer, Integer> inside = regions.getRegion(image);
        int w = image.getWidth(), h = image.getHeight();
        List<int[]> open = new ArrayList<>();     // rectangles still growing downwards: x0, y0, x1
        List<int[]> done = new ArrayList<>();
        for (int y = 0; y <= h; y++) {
            List<int[]> runs = new ArrayList<>();
            if (y < h) {
                int x = 0;
                while (x < w) {
                    if (inside.test(x, y)) {
                        int x0 = x;
                        while (x < w && inside.test(x, y)) {
                            x++;
                        }
                        runs.add(new int[]{x0, x});
                    } else {
                        x++;
                    }
                }
            }
            List<int[]> next = new ArrayList<>();
            for (int[] r : open) {
                boolean continued = false;
                for (int[] run : runs) {
                    if (run[0] == r[0] && run[1] == r[2]) {
                        continued = true;
                        run[0] = -1;               // consumed
                        break;
                    }
                }
                if (continued) {
                    next.add(r);
                } else {
                    done.add(new int[]{r[0], r[1], r[2], y});
                }
            }
            for (int[] run : runs) {
                if (run[0] >= 0) {
                    next.add(new int[]{run[0], y, run[1]});
                }
            }
            open = next;
        }
        if (done.size() > MAX_RECTS) {
            throw new UnsupportedOperationException("excluded regions need " + done.size() + " rectangles, the GPU library takes " + MAX_RECTS);
        }
        int[] out = new int[4 * done.size()];
        for (int i = 0; i < done.size(); i++) {
            System.arraycopy(done.get(i), 0, out, 4 * i, 4);
        }
        return out;
    }
}
