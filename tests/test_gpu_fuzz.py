"""A fixed batch of random small problems against the oracle (see tests/fuzz_vs_oracle.py)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_small_cases_match_oracle(oracle, seed, capsys):
    import fuzz_vs_oracle

    failures = fuzz_vs_oracle.run_cases(60, seed, verbose=False)
    assert failures == 0, capsys.readouterr().out
