import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the reference (VMAS): /root/reference in the build container, "
                            "its byte-compiled build oracle/_ref (made by __graft_entry__.build()) on the GPU box")


def pytest_collection_modifyitems(config, items):
    from oracle import ref

    have_ref = ref.available()
    skip_ref = pytest.mark.skip(reason="neither /root/reference nor oracle/_ref present")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
