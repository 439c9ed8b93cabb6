"""A short run of the randomised differential script (tests/fuzz_parity.py: random format / link / regulariser / decay
mode / factor width / shared spaces / side tables / staging plan, HIP engine vs the C oracle, bit for bit) and of the ranker's
(tests/fuzz_ranker.py) inside the GPU suite; longer runs: python tests/fuzz_parity.py --iters 2500 --seed N."""
import sys

import pytest

import fuzz_parity
import fuzz_ranker  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 102])
def test_random_configurations_match_the_oracle(seed):
    stats = fuzz_parity.main(["--iters", "250", "--seed", str(seed)])
    assert stats["iters"] == 250 and stats["exact"] > 50


def test_random_ranker_streams_match_the_cpu_ranker(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["fuzz_ranker.py", "--iters", "60", "--seed", "5"])
    assert fuzz_ranker.main() == 0


def test_random_amd_gpus_handles_match_the_oracle_simulation(monkeypatch):
    import fuzz_multi
    monkeypatch.setattr(sys, "argv", ["fuzz_multi.py", "--iters", "80", "--seed", "7"])
    assert fuzz_multi.main() == 0
