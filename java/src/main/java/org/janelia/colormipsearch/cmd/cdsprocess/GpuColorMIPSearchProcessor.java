package org.janelia.colormipsearch.cmd.cdsprocess;

import java.util.ArrayList;
import java.util.List;
import java.util.Map;
import java.util.Set;

import org.janelia.colormipsearch.cds.ColorMIPSearch;
import org.janelia.colormipsearch.cds.gpu.GpuBatchSearch;
import org.janelia.colormipsearch.cmd.CachedMIPsUtils;
import org.janelia.colormipsearch.imageprocessing.ImageArray;
import org.janelia.colormipsearch.imageprocessing.ImageRegionDefinition;
import org.janelia.colormipsearch.mips.NeuronMIP;
import org.janelia.colormipsearch.mips.NeuronMIPUtils;
import org.janelia.colormipsearch.model.AbstractNeuronEntity;
import org.janelia.colormipsearch.model.CDMatchEntity;
import org.janelia.colormipsearch.model.ComputeFileType;
import org.janelia.colormipsearch.model.ProcessingType;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;

/**
 * ColorMIPSearchProcessor on MI355X: the counterpart of LocalColorMIPSearchProcessor (:55-116) at the granularity a GPU
 * wants.  The reference makes one calculateMatchingScore call per (mask, target) pair from a thread pool; here
 *   1. the target MIPs are decoded once (through the reference's own loader / cache) and made resident as a sparse
 *      library on the GPU, partition by partition,
 *   2. the masks are prepared in sets of up to 128 (-Dcds.gpu.maskset, the size the measured pipeline uses) and every set scans the whole resident library in a few launches (groups of
 *      masks share one pass over the records),
 *   3. ColorMIPSearch.isMatch and the compaction run on the device: only the matches come back,
 * and each match becomes the CDMatchEntity findPixelMatch would have produced (same fields, same tags, same "mask is empty /
 * has no image: skip" rules).  The search parameters are read from ColorMIPSearch.getCDSParameters(), so the processor
 * takes the same ColorMIPSearch object the local one does.
 */
public class GpuColorMIPSearchProcessor<M extends AbstractNeuronEntity, T extends AbstractNeuronEntity> extends AbstractColorMIPSearchProcessor<M, T> {

    private static final Logger LOG = LoggerFactory.getLogger(GpuColorMIPSearchProcessor.class);
    private static final int MASKS_PER_SET = Integer.getInteger("cds.gpu.maskset", 128);   // a multiple of 8 (the multi-mask kernel's groups)

    private final ImageRegionDefinition excludedRegions;
    private final int[] devices;

    public GpuColorMIPSearchProcessor(Number cdsRunId,
                                      ColorMIPSearch colorMIPSearch,
                                      int localProcessingPartitionSize,
                                      ImageRegionDefinition excludedRegions,
                                      int device,
                                      Set<String> tags) {
        this(cdsRunId, colorMIPSearch, localProcessingPartitionSize, excludedRegions, new int[]{device}, tags);
    }

    /**
     * Several GPUs of the node: the targets are cut into one resident shard per device, every mask set is scanned by all of them at
     * once and the match lists come back merged (the reference's counterpart: partitions of targets on Spark executors,
     * SparkColorMIPSearchProcessor.java:50-73).
     */
    public GpuColorMIPSearchProcessor(Number cdsRunId,
                                      ColorMIPSearch colorMIPSearch,
                                      int localProcessingPartitionSize,
                                      ImageRegionDefinition excludedRegions,
                                      int[] devices,
                                      Set<String> tags) {
        super(cdsRunId, colorMIPSearch, localProcessingPartitionSize, tags);
        this.excludedRegions = excludedRegions;
        this.devices = devices.clone();
    }

    @Override
    public List<CDMatchEntity<M, T>> findAllColorDepthMatches(List<M> queryMIPs, List<T> targetMIPs) {
        long startTime = System.currentTimeMillis();
        LOG.info("Searching {} masks against {} targets on GPU(s) {}", queryMIPs.size(), targetMIPs.size(), java.util.Arrays.toString(devices));
        Map<String, Object> params = colorMIPSearch.getCDSParameters();
        List<CDMatchEntity<M, T>> results = new ArrayList<>();
        try (GpuBatchSearch<T> search = new GpuBatchSearch<>(devices, params, excludedRegions)) {
            // 0. the run's image size is the masks' (the reference compares every target with the QUERY's size,
            //    PixelMatchColorDepthSearchAlgorithm.java:171-175): the first mask that has an image decides
            for (M queryMIP : queryMIPs) {
                NeuronMIP<M> queryImage = NeuronMIPUtils.loadComputeFile(queryMIP, ComputeFileType.InputColorDepthImage);
                if (queryImage != null && !queryImage.hasNoImageArray()) {
                    search.setImageSize(queryImage.getImageArray().getWidth(), queryImage.getImageArray().getHeight());
                    break;
                }
            }
            search.expectTargets(targetMIPs.size());
            // 1. the targets, resident once for all masks; partitions bound the host memory held at a time.  A target that cannot be
            //    compared -- another image size, an unreadable file -- is the reference's per-pair error: logged, dropped, and the
            //    run goes on (AbstractColorMIPSearchProcessor.java:68-83: catch Throwable, setErrors; LocalColorMIPSearchProcessor
            //    .java:102-107: filter hasNoErrors)
            int targetErrors = 0;
            for (int p0 = 0; p0 < targetMIPs.size(); p0 += localProcessingPartitionSize) {
                for (T targetMIP : targetMIPs.subList(p0, Math.min(targetMIPs.size(), p0 + localProcessingPartitionSize))) {
                    try {
                        NeuronMIP<T> targetImage = CachedMIPsUtils.loadMIP(targetMIP, ComputeFileType.InputColorDepthImage);
                        if (NeuronMIPUtils.hasImageArray(targetImage)) {
                            search.addTarget(targetMIP, NeuronMIPUtils.getImageArray(targetImage));
                        }
                    } catch (Throwable e) {
                        targetErrors++;
                        LOG.warn("Error comparing the masks with {}", targetMIP, e);
                    }
                }
            }
            if (targetErrors > 0) {
                LOG.warn("{} of {} targets were dropped with errors", targetErrors, targetMIPs.size());
            }
            search.targetsComplete();
            LOG.info("{} of {} targets resident after {}s", search.targetCount(), targetMIPs.size(), (System.currentTimeMillis() - startTime) / 1000.);
            // 2. the masks, a set at a time
            List<M> setMIPs = new ArrayList<>();
            List<ImageArray<?>> setImages = new ArrayList<>();
            for (int qi = 0; qi <= queryMIPs.size(); qi++) {
                if (qi < queryMIPs.size()) {
                    M queryMIP = queryMIPs.get(qi);
                    NeuronMIP<M> queryImage = NeuronMIPUtils.loadComputeFile(queryMIP, ComputeFileType.InputColorDepthImage);   // no caching for the mask
                    if (queryImage == null || queryImage.hasNoImageArray()) {
                        LOG.error("No input color depth image found for mask {}", queryMIP);
                        continue;
                    }
                    ImageArray<?> queryArray = queryImage.getImageArray();
                    if (search.targetCount() > 0 && (queryArray.getWidth() != search.getWidth() || queryArray.getHeight() != search.getHeight())) {
                        // every pair of THIS mask is the reference's "Invalid image size" error; the other masks of the set are not touched
                        LOG.warn("Error comparing mask {} with the resident targets: Invalid image size - target's image size ({}, {}) must match query's image size: ({}, {})",
                                queryMIP, search.getWidth(), search.getHeight(), queryArray.getWidth(), queryArray.getHeight());
                        continue;
                    }
                    setMIPs.add(queryMIP);
                    setImages.add(queryArray);
                }
                if (setMIPs.size() == MASKS_PER_SET || (qi == queryMIPs.size() && !setMIPs.isEmpty())) {
                    collectMatches(search, setMIPs, setImages, results);
                    setMIPs.clear();
                    setImages.clear();
                }
            }
        }
        LOG.info("Finished all color depth searches {} masks with {} targets in {}s: {} matches", queryMIPs.size(), targetMIPs.size(),
                (System.currentTimeMillis() - startTime) / 1000., results.size());
        return results;
    }

    @SuppressWarnings("unchecked")
    private void collectMatches(GpuBatchSearch<T> search, List<M> maskMIPs, List<ImageArray<?>> maskImages, List<CDMatchEntity<M, T>> results) {
        GpuBatchSearch.Matches found;
        try {
            found = search.findMatches(maskImages);     // masks whose query size is 0 yield nothing ("No computation created ... because it is empty")
        } catch (RuntimeException e) {
            // (sizes were checked mask by mask before the set was built: what arrives here is a failure of the set's search itself.
            // One mask at a time then, so that a mask the library cannot prepare costs only its own pairs)
            if (maskMIPs.size() == 1) {
                LOG.warn("Error comparing mask {} with the resident targets", maskMIPs.get(0), e);
                return;
            }
            LOG.warn("Error searching a set of {} masks; retrying them one by one", maskMIPs.size(), e);
            for (int i = 0; i < maskMIPs.size(); i++) {
                collectMatches(search, maskMIPs.subList(i, i + 1), maskImages.subList(i, i + 1), results);
            }
            return;
        }
        for (int i = 0; i < found.count; i++) {
            M maskMIP = maskMIPs.get(found.mask[i]);
            T targetMIP = search.target(found.target[i]);
            CDMatchEntity<M, T> result = new CDMatchEntity<>();
            result.setMaskImage((M) maskMIP.addProcessedTags(ProcessingType.ColorDepthSearch, tags));
            result.setMatchedImage((T) targetMIP.addProcessedTags(ProcessingType.ColorDepthSearch, tags));
            result.setSessionRefId(cdsRunId);
            result.setMatchFound(true);
            result.setMatchingPixels(found.score[i]);
            result.setMatchingPixelsRatio((float) ((double) found.score[i] / (double) found.querySize[found.mask[i]]));
            result.setMirrored(found.mirrored[i]);
            result.addAllTags(tags);
            results.add(result);
        }
    }

    @Override
    public void terminate() {
        // nothing to do here: the native handles are released when findAllColorDepthMatches returns
    }
}
