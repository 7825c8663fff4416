package org.janelia.colormipsearch.imageprocessing;

/**
 * Access to the backing arrays of the reference's ImageArray classes for the GPU plugin.
 *
 * ImageArray.pixels / getPixels() and the ImageArray constructors are package-private in the reference
 * (imageprocessing/ImageArray.java, ColorImageArray.java), so this accessor lives in the same package.  The byte layout of a
 * ColorImageArray is already what libcds_hip.so takes: packed R,G,B, row-major, pitch 3 * width.
 */
public final class ImageArrayBytes {
    private ImageArrayBytes() {
    }

    /** packed R,G,B bytes of an RGB image (no copy). */
    public static byte[] rgb(ImageArray<?> a) {
        if (a instanceof ColorImageArray) {
            return ((ColorImageArray) a).pixels;
        }
        throw new IllegalArgumentException("RGB image expected, got " + a);
    }

    /** 16-bit gray values of a gradient image; an 8-bit image is widened (ImageArray.get semantics). */
    public static short[] gray16(ImageArray<?> a) {
        if (a instanceof ShortImageArray) {
            return ((ShortImageArray) a).pixels;
        }
        if (a instanceof ByteImageArray) {
            byte[] b = ((ByteImageArray) a).pixels;
            short[] s = new short[b.length];
            for (int i = 0; i < b.length; i++) {
                s[i] = (short) (b[i] & 0xFF);
            }
            return s;
        }
        throw new IllegalArgumentException("gray image expected, got " + a);
    }

    /** A ColorImageArray over packed R,G,B bytes (used by the plugin's own tests / tools). */
    public static ImageArray<?> colorImage(int width, int height, byte[] rgb) {
        return new ColorImageArray(ImageType.RGB, width, height, rgb);
    }
}
