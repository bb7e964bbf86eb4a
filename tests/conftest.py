import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import mvf_oracle
    mvf_oracle.build()
    return mvf_oracle


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True, scope="session")
def _every_built_image_is_verifier_clean():
    """Every .mvf image a test builds through libmvf_host also has to pass the FlatBuffers verifier the reference reader
    runs (tests/fb_verify.py; src/reader.rs:64,:245): BuiltMvf.to_bytes / .save are wrapped for the whole session."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fb_verify
    from metrovector_amd import builder
    to_bytes, save = builder.BuiltMvf.to_bytes, builder.BuiltMvf.save
    seen = {"n": 0}

    def checked_to_bytes(self):
        img = to_bytes(self)
        fb_verify.verify_image(img)
        seen["n"] += 1
        return img

    def checked_save(self, path):
        save(self, path)
        with open(path, "rb") as fh:
            fb_verify.verify_image(fh.read())
        seen["n"] += 1

    builder.BuiltMvf.to_bytes, builder.BuiltMvf.save = checked_to_bytes, checked_save
    yield seen
    builder.BuiltMvf.to_bytes, builder.BuiltMvf.save = to_bytes, save
