import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


GOLDEN = os.path.join(ROOT, "tests", "golden")

GOLDEN_CASES = ["poisson5_6x5_sorted", "poisson5_6x5_shuffled", "ragged_dups_empty_rows",
                "symmetric_lower", "pattern_general", "pattern_symmetric", "skewed_one_long_row",
                "empty_matrix", "one_by_one", "random_70k",
                "array_sorting_kat"]      # the sort-by-key input of test/array_sorting.cpp:35-36, sorted by the reference itself


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
