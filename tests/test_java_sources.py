"""The reference-side plugin under java/ cannot be compiled here (no JDK).  What CAN be checked on CPU: the JNI C file
implements every native method CdsHip.java declares (names and parameter counts), and it only calls C-ABI functions that
include/cds_hip.h declares and libcds_hip.so exports."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "java", "src", "main", "java", "org", "janelia", "colormipsearch")


def test_every_native_method_has_a_jni_implementation():
    java = open(os.path.join(JAVA, "cds", "gpu", "CdsHip.java")).read()
    c = open(os.path.join(ROOT, "java", "src", "main", "c", "cds_hip_jni.c")).read()
    natives = re.findall(r"static native [\w\[\]]+\s+(\w+)\(([^)]*)\)", java)
    assert len(natives) >= 30
    impl = {name: args for name, args in re.findall(r"CLS\((\w+)\)\(JNIEnv \*env, jclass c(?:, )?([^)]*)\)", c)}
    for name, args in natives:
        assert name in impl, f"no JNI implementation of CdsHip.{name}"
        n_java = len([a for a in args.split(",") if a.strip()])
        n_c = len([a for a in impl[name].split(",") if a.strip()])
        assert n_java == n_c, (name, n_java, n_c)
    assert set(impl) == {n for n, _ in natives}


def test_jni_file_calls_only_declared_and_exported_entry_points():
    from colormipsearch_amd import capi
    c = open(os.path.join(ROOT, "java", "src", "main", "c", "cds_hip_jni.c")).read()
    header = open(os.path.join(ROOT, "include", "cds_hip.h")).read()
    called = set(re.findall(r"\b(cds_[a-z_0-9]+)\s*\(", c))
    assert len(called) >= 30
    lib = capi.load()
    for fn in called:
        assert re.search(r"\b" + fn + r"\s*\(", header), f"{fn} is not declared in include/cds_hip.h"
        assert hasattr(lib, fn), f"{fn} is not exported by libcds_hip.so"


def test_plugin_classes_name_the_reference_interfaces():
    px = open(os.path.join(JAVA, "cds", "gpu", "GpuPixelMatchColorDepthSearchAlgorithm.java")).read()
    sh = open(os.path.join(JAVA, "cds", "gpu", "GpuShape2DMatchColorDepthSearchAlgorithm.java")).read()
    pr = open(os.path.join(JAVA, "cmd", "cdsprocess", "GpuColorMIPSearchProcessor.java")).read()
    fa = open(os.path.join(JAVA, "cds", "gpu", "GpuColorDepthSearchAlgorithmProviderFactory.java")).read()
    assert "implements ColorDepthSearchAlgorithm<PixelMatchScore>" in px and "implements ColorDepthSearchAlgorithm<ShapeMatchScore>" in sh
    for m in ("getQueryImage", "getQuerySize", "getQueryFirstPixelIndex", "getQueryLastPixelIndex", "getRequiredTargetVariantTypes", "calculateMatchingScore"):
        assert m in px and m in sh
    assert "findAllColorDepthMatches(List<M> queryMIPs, List<T> targetMIPs)" in pr and "void terminate()" in pr
    assert "createPixMatchCDSAlgorithmProvider(" in fa and "createShapeMatchCDSAlgorithmProvider(" in fa


def test_jni_shim_parses_and_type_checks():
    """No JDK in the image: the JNI C file is at least parsed and type-checked by gcc against tests/jni_stub/jni.h (a minimal
    declaration set from the JNI specification -- test scaffolding, never linked) and the real C ABI header."""
    import subprocess
    src = os.path.join(ROOT, "java", "src", "main", "c", "cds_hip_jni.c")
    out = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                          "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "TEST SCAFFOLDING ONLY" in open(os.path.join(ROOT, "tests", "jni_stub", "jni.h")).read()


def test_jni_shim_never_blocks_inside_a_critical_section_and_checks_lengths():
    """The C ABI calls block on the GPU and are made from many Java threads: no GetPrimitiveArrayCritical (it would hold the GC lock across
    them); every native that takes a Java array or direct buffer validates its length before the call; local references of the
    path loop are released."""
    c = open(os.path.join(ROOT, "java", "src", "main", "c", "cds_hip_jni.c")).read()
    code = re.sub(r"/\*.*?\*/", "", c, flags=re.S)
    assert "GetPrimitiveArrayCritical" not in code
    assert "DeleteLocalRef" in code
    for fn in re.findall(r"JNIEXPORT [\w\s\*]+ JNICALL CLS\((\w+)\)\(([^)]*)\)", code):
        name, args = fn
        takes_array = re.search(r"\bj(?:byte|short|int|long|object)Array\b", args) or "jobject direct" in args
        if not takes_array:
            continue
        body = code[code.index("CLS(" + name + ")"):]
        body = body[:body.index("\nJNIEXPORT") if "\nJNIEXPORT" in body[1:] else len(body)]
        assert re.search(r"has_len|direct_bytes|handle_array|search_common|get_rects|GetArrayLength", body), f"CdsHip.{name} does not check its array lengths"


def _java_strip(mask, thr, rects, xy, mirror):
    """GpuPixelMatchColorDepthSearchAlgorithm.stripBounds restated line by line (the Java file cannot run here)."""
    import numpy as np
    H, W, _ = mask.shape
    sel = mask.max(axis=2) > thr
    for (x0, y0, x1, y1) in rects:
        sel[max(0, y0):max(0, y1), max(0, x0):max(0, x1)] = False
    pos = np.flatnonzero(sel.reshape(-1))
    minx, miny, maxx, maxy = W, H, 0, 0
    if len(pos):
        xs, ys = pos % W, pos // W
        minx, maxx, miny, maxy = int(xs.min()), int(xs.max()) + 1, int(ys.min()), int(ys.max()) + 1
    first, last = miny * W * minx, maxy * W * maxx
    if len(pos):
        def sh(p, dx, dy):
            x, y = p % W + dx, p // W + dy
            return y * W + x if 0 <= x < W and 0 <= y < H else -1
        step = max(xy, 1)
        for dx in range(-xy, xy + 1, step):
            for dy in range(-xy, xy + 1, step):
                a, b = sh(int(pos[0]), dx, dy), sh(int(pos[-1]), dx, dy)
                first, last = min(first, a), max(last, b)
                if mirror:
                    first = min(first, -1 if a == -1 else a + (W - 1) - 2 * (a % W))
                    last = max(last, -1 if b == -1 else b + (W - 1) - 2 * (b % W))
    return first, last


def test_java_first_last_pixel_index_equals_the_python_mirror():
    """getQueryFirstPixelIndex / getQueryLastPixelIndex: the Java class and cds.py must return the same (quirky) numbers the reference does
    (AbstractColorDepthSearchAlgorithm.java:76-84 multiplies row, width and column; PixelMatchColorDepthSearchAlgorithm.java:59-100)."""
    import types
    import numpy as np
    from colormipsearch_amd import cds
    java = open(os.path.join(JAVA, "cds", "gpu", "GpuPixelMatchColorDepthSearchAlgorithm.java")).read()
    assert "miny * w * minx" in java and "maxy * w * maxx" in java and "scanQuery" not in java
    rng = np.random.default_rng(5)
    W, H = 40, 30
    for case in range(12):
        mask = np.zeros((H, W, 3), dtype=np.uint8)
        if case:
            k = int(rng.integers(1, 12))
            ys, xs = rng.integers(0, H, k), rng.integers(0, W, k)
            if case % 3 == 0:
                ys[0], xs[0] = 0, 1          # a shift leaves the image: -1 entries
            if case % 4 == 0:
                ys[-1], xs[-1] = H - 1, W - 2
            mask[ys, xs] = rng.integers(21, 255, (k, 3))
        rects = [(0, 0, 5, 3)] if case % 2 else []
        for xy in (0, 2):
            for mirror in (False, True):
                fake = types.SimpleNamespace(_masks=mask[None], _rects=rects, width=W, height=H, xyShift=xy, mirrorQuery=mirror, queryThreshold=20)
                fake._positions = types.MethodType(cds.PixelMatchColorDepthSearchAlgorithm._positions, fake)
                want = cds.PixelMatchColorDepthSearchAlgorithm._strip(fake)
                assert _java_strip(mask.copy(), 20, rects, xy, mirror) == want, (case, xy, mirror)
