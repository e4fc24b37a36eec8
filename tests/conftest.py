"""pytest configuration: the `gpu` marker + shared paths/fixtures."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
# the CLIs under test: MXG_BIN_DIR points at another build of them (the sanitizer builds of ntjoin_amd/csrc/Makefile)
BIN_DIR = os.environ.get("MXG_BIN_DIR") or os.path.join(REPO, "ntjoin_amd", "bin")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_cases():
    with open(os.path.join(GOLDEN, "cases", "index.json"), encoding="utf-8") as fh:
        return json.load(fh)


def load_case(name):
    with open(os.path.join(GOLDEN, "cases", name, "reference.json"), encoding="utf-8") as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    from tests import _oracle
    return _oracle.load()
