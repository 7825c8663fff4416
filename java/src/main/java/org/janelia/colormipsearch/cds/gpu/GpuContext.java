package org.janelia.colormipsearch.cds.gpu;

import java.util.concurrent.ConcurrentHashMap;

/** One cds_context per GPU of the process, created on first use (device from -Dcds.gpu.device, default 0). */
public final class GpuContext {
    private static final ConcurrentHashMap<Integer, Long> CONTEXTS = new ConcurrentHashMap<>();

    private GpuContext() {
    }

    public static int defaultDevice() {
        return Integer.getInteger("cds.gpu.device", 0);
    }

    public static long get() {
        return get(defaultDevice());
    }

    public static long get(int device) {
        return CONTEXTS.computeIfAbsent(device, CdsHip::contextCreate);
    }
}
