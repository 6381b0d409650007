import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_OBSERVED = []


@pytest.fixture
def observed():
    """observed("text"): a line for the terminal summary — margins a passing test actually used (a regression inside
    a tolerance must be visible), shown under -q as well."""
    return _OBSERVED.append


def pytest_terminal_summary(terminalreporter):
    if _OBSERVED:
        terminalreporter.section("observed margins")
        for line in _OBSERVED:
            terminalreporter.write_line(line)
