"""How far a conforming but different GLSL implementation lands from the strict oracle -- TEST INFRASTRUCTURE ONLY.

The strict oracle (librt_oracle.so) rounds every operation once, in source order, and evaluates pow by a fixed
sequence; the reference shader run by a real driver (say Mesa lavapipe, an LLVM JIT) may contract a multiply and
an add into one rounding and has its own pow.  Such differences change a float by an ulp, and an ulp can flip one of
the comparisons that steer the random stream (raytrace.comp.glsl:65, :83, :117, :125-129, :207, :219-221): from there
on that (pixel, frame) draws different samples.  This module renders the same frames with the strict oracle and
with the variant build (oracle/Makefile, target `variant`: -ffp-contract=fast -mfma, libm powf) and reports
  * the fraction of (pixel, frame) items whose final RNG word differs (streams that diverged), and
  * the RMSE of the converged radiance relative to its mean luminance (the criterion quoted for a comparison
    with the reference on lavapipe: <= 1e-3 at 4096 spp).
Used by tests/test_conforming_variant.py and tools/conforming_variant.py.
"""
import numpy as np

from . import pyoracle


def study(width=64, height=36, spp=64, frames=64, depth=16, aperture=0.05, nthreads=None):
    """Default scene and camera at the 4K configuration's settings (64 spp per frame, depth 16, aperture 0.05,
    64 frames = 4096 spp) on a small image.  Returns a dict of figures, or None where the variant cannot be
    built (CPU without FMA)."""
    variant = pyoracle.variant_lib()
    if variant is None:
        return None
    scene = pyoracle.default_scene()
    direction, focus = pyoracle.default_aim()
    acc = {"strict": None, "variant": None}
    diverged = 0
    for f in range(frames):
        params = pyoracle.make_params((13.0, 2.0, 3.0), direction, aperture, 20.0, focus, width, height,
                                      frame=f, spp=spp, max_depth=depth, sphere_count=5)
        words = pyoracle.params_words(params)
        a = pyoracle.render(words, scene, accum=acc["strict"], want_output=False, want_rng=True, nthreads=nthreads)
        b = pyoracle.render(words, scene, accum=acc["variant"], want_output=False, want_rng=True, nthreads=nthreads,
                            library=variant)
        acc["strict"], acc["variant"] = a["accum"], b["accum"]
        diverged += int(np.count_nonzero(a["rng"] != b["rng"]))
    total = float(spp * frames)
    x = acc["strict"][..., :3].astype(np.float64) / total
    y = acc["variant"][..., :3].astype(np.float64) / total
    mean = float(x.mean())
    rmse = float(np.sqrt(np.mean((x - y) ** 2)))
    identical_pixels = int(np.count_nonzero(np.all(acc["strict"].view(np.uint32) == acc["variant"].view(np.uint32), axis=2)))
    return {
        "image": "%dx%d, default scene and camera, %d spp x %d frames, depth %d, aperture %g" % (
            width, height, spp, frames, depth, aperture),
        "variant": "-ffp-contract=fast -mfma, pow = libm powf",
        "items": width * height * frames,
        "diverged_items": diverged,
        "diverged_fraction": diverged / float(width * height * frames),
        "pixels": width * height,
        "bit_identical_pixels": identical_pixels,
        "mean_luminance": mean,
        "rmse": rmse,
        "rmse_over_mean": rmse / mean,
    }
