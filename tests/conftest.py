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


def pytest_sessionstart(session):
    # the parity tests append one line per fixture: start every session from an empty file
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "parity_allowance.jsonl")
    try:
        if os.path.exists(out) and not os.environ.get("PYTEST_XDIST_WORKER"):
            os.remove(out)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    """The golden parity tests accept |HIP - reference| <= 1e-5 (abs + rel) PLUS, outside the BASELINE fixtures, 8 x the
    measured sensitivity of the reference's own arithmetic to a 1-ulp input perturbation (tests/golden_util.py): how many
    values actually needed that allowance is printed here, per check, so that the test log itself shows it."""
    import json

    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "parity_allowance.jsonl")
    if not os.path.exists(out):
        return
    per_check, needing = {}, []
    with open(out) as f:
        for line in f:
            try:
                r = json.loads(line)
            except ValueError:
                continue
            c = per_check.setdefault(r["check"], [0, 0, 0.0, 0])
            c[0] += int(r["values"])
            c[1] += int(r["needed_allowance"])
            c[2] = max(c[2], float(r.get("max_abs_err_physical", r["max_abs_err"])))
            c[3] += int(r.get("blown_up_values", 0))
            if r["needed_allowance"]:
                needing.append(f"{r['fixture']}:{r['needed_allowance']}")
    tr = terminalreporter
    tr.write_sep("-", "parity allowance (values beyond 1e-5 abs+rel that needed the 8 x ulp-sensitivity term)")
    for check, (values, needed, worst, blown) in sorted(per_check.items()):
        tr.write_line(f"{check}: {needed} of {values} values needed it; max |HIP - reference| {worst:.3g} over the values of "
                      f"physical magnitude (|x| < 1e3); {blown} values of blown-up environments (the reference's own state is "
                      f"non-finite or beyond 1e3 there) are compared with the allowance but left out of that maximum")
    tr.write_line("fixtures with any: " + (", ".join(sorted(set(needing))) if needing else "none"))
