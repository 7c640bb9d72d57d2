"""Randomised parity sweep (tools/fuzz_parity.py): random decoder shapes, option flags, batch shapes, lt_mode and precision
(fp32 / split / bf16) against the float64 forward oracle and the autograd gradient oracle at the fp32 bar."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_meet_the_fp32_bar(seed):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    assert fuzz_parity.run(12, seed) == 0


def test_random_sampler_configurations_match_the_oracle_driver():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    assert fuzz_parity.run_beam(10, 21) == 0
