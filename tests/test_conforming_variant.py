"""The strict oracle against the same restatement built as another conforming GLSL implementation would
evaluate it (contracted multiply-adds, the C library's pow): this is the nearest thing to a comparison with
the reference shader on a real Vulkan driver that can be run here.  Bit equality is not expected -- a
comparison that flips sends a (pixel, frame) down a different random stream -- but the converged image must
agree to the criterion quoted for the lavapipe comparison, RMSE <= 1e-3 of the mean luminance at 4096 spp."""
import numpy as np
import pytest

from oracle import pyoracle, variant_study


def test_variant_build_differs_in_arithmetic_only():
    v = pyoracle.variant_lib()
    if v is None:
        pytest.skip("no FMA on this CPU: the variant cannot be built")
    # integer parts are the same code
    assert int(v.rto_hash3(1, 2, 3)) == pyoracle.hash3(1, 2, 3) == 889768027
    # pow differs from the fixed sequence in the last bits for some arguments, and only there
    x = np.linspace(0.01, 0.99, 2000, dtype=np.float32)
    strict = np.array([pyoracle.lib().rto_powf(float(a), 5.0) for a in x], dtype=np.float32)
    other = np.array([v.rto_powf(float(a), 5.0) for a in x], dtype=np.float32)
    rel = np.abs(strict.astype(np.float64) - other) / other
    assert rel.max() <= 3e-6
    assert np.any(strict != other)


def test_converged_radiance_of_a_conforming_variant():
    r = variant_study.study()
    if r is None:
        pytest.skip("no FMA on this CPU: the variant cannot be built")
    print(r)
    # the streams do diverge here and there (otherwise the variant would not be exercising anything) ...
    assert 0 < r["diverged_items"] < 0.05 * r["items"]
    # ... and the converged image still agrees
    assert r["rmse_over_mean"] <= 1e-3
