"""Randomised parity cases (tests/fuzz_parity.py): 18 random problems per run, three of every pattern family."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_patterns_against_oracle():
    import fuzz_parity
    worst = fuzz_parity.run(18, seed0=7000)
    assert len(worst) >= 12                      # every check ran on at least one case
    for k, (v, tag) in worst.items():
        assert v <= 1e-9, (k, v, tag)
