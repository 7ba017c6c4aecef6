"""A fixed batch of random small problems against the oracle (see tests/fuzz_vs_oracle.py)."""
import pytest

pytestmark = pytest.mark.gpu


# seeds >= 100 also vary the base weight and feed uint8 images; seeds >= 1000 draw filter radii up to 9 (guided,
# median, Gaussian, bilateral) and the exact median
@pytest.mark.parametrize("seed", [11, 12, 101, 102, 1001, 1002])
def test_random_small_cases_match_oracle(oracle, seed, capsys):
    import fuzz_vs_oracle

    failures = fuzz_vs_oracle.run_cases(60, seed, verbose=False)
    assert failures == 0, capsys.readouterr().out


@pytest.mark.parametrize("seed", [21, 22])
def test_random_operator_cases_match_oracle(oracle, seed, capsys):
    """Stand-alone operators: cost volume, integral image, box mean and medians bit-exact; guided,
    Gaussian, bilateral and aggregate_and_wta within 1e-5 (see tests/fuzz_ops_vs_oracle.py)."""
    import fuzz_ops_vs_oracle

    failures = fuzz_ops_vs_oracle.run_cases(150, seed, verbose=False)
    assert failures == 0, capsys.readouterr().out


def test_non_finite_pixels_through_the_operators(oracle, capsys):
    """NaN / inf inputs of integral image, box mean, guided, Gaussian, bilateral and exact median filters: same
    NaN / inf pattern as the oracle (see tests/fuzz_nonfinite_vs_oracle.py)."""
    import fuzz_nonfinite_vs_oracle

    failures = fuzz_nonfinite_vs_oracle.run_cases(300, 5)
    assert failures == 0, capsys.readouterr().out


def test_random_large_frames_with_and_without_tail_split(capsys):
    """Frames with more tiles than compute units: the launch that cuts its last tiles into disparity slices
    writes the same keys and band flags as the one that does not (see tests/fuzz_tail_split.py)."""
    import fuzz_tail_split

    failures = fuzz_tail_split.run(40, 3)
    assert failures == 0, capsys.readouterr().out
