"""Shared by the GPU parity suites (imported by tests/conftest.py and the test modules)."""
# The two evaluations of the body pairs (library option "winograd"): the direct sums (k_pair) and Winograd F(2,3) along the row
# (k_wino).  The library's default is auto — Winograd for well-conditioned weights, which the synthetic draw and (by its kappa) any
# sane model are — so the parity suites hold BOTH to the oracle at BASELINE's sizes: parametrize with EVALUATIONS and pass the
# value to `upscalers(..., evaluation=...)` / `pin_evaluation(up, ...)`.
EVALUATIONS = ("direct", "winograd")


def pin_evaluation(up, evaluation):
    """None: the library's default (auto).  The evaluation in force must be the one asked for."""
    if evaluation is not None:
        want = {"direct": 0, "winograd": 1}[evaluation]
        up.set_option("winograd", want)
        assert up.get_option("winograd") == want
    return up
