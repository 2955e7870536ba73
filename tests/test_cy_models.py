"""The Cython boundary (common_amd/cy): the extension types behind model_descriptor.c_desc() -- this build's counterpart of
microscopes/_models.pxd:4-31 / _models.pyx:16-52 -- and a downstream module (tests/cy/downstream_probe.pyx) that
cimports `_base`, takes the shared_ptr[model] out of it and walks model -> hypers -> group through the C++ virtual API,
as mixturemodel / irm state objects do.  No device work here (constructing models, hypers and groups touches none)."""
import os
import sys

import numpy as np
import pytest

from common_amd import models, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe():
    from common_amd.cy.build import build_module
    build_module(os.path.join(ROOT, "common_amd", "cy", "_models.pyx"))
    build_module(os.path.join(ROOT, "tests", "cy", "downstream_probe.pyx"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "cy"))
    import downstream_probe
    return downstream_probe


def test_c_desc_returns_the_extension_types_of_the_reference_surface():
    from common_amd.cy import _models as cy
    names = {"bb": "_bb", "bnb": "_bnb", "gp": "_gp", "nich": "_nich", "bbnc": "_bbnc"}
    for d in (models.bb, models.bnb, models.gp, models.nich, models.bbnc):
        h = d.c_desc()
        assert type(h).__name__ == names[d.name()] and isinstance(h, cy._base)
        assert d.c_desc() is h                                      # one handle per descriptor, as upstream
        assert (h.family, h.dim) == (d.family, d.dim)
    assert type(models.dd(7).c_desc()).__name__ == "_dd" and models.dd(7).c_desc().dim == 7
    assert type(models.niw(3).c_desc()).__name__ == "_niw" and models.niw(3).c_desc().get_runtime_type() == (9, 3)
    assert type(models.dm(4).c_desc()).__name__ == "_dm" and models.dm(4).c_desc().get_runtime_type() == (5, 4)
    for bad in (lambda: cy._dd(0), lambda: cy._niw(-1), lambda: cy._dm(0)):
        with pytest.raises(ValueError):
            bad()
    with pytest.raises(RuntimeError):                              # C++ exceptions cross as Python ones (`except +`)
        cy._dd(129).default_hp_bytes()                             # distributions_hypers<DD128>(129) throws


def test_downstream_module_walks_model_hypers_group(probe):
    """cimport _base; desc.get() -> shared_ptr[model]; create_hypers(); get_runtime_type(); create_group(rng)"""
    want_type = {"bb": (0, 1, False), "bbnc": (0, 1, False), "gp": (6, 1, False), "bnb": (6, 1, False),
                 "nich": (9, 1, False), "dd": (5, 1, False), "niw": (9, 3, True), "dm": (5, 4, True)}
    for d in (models.bb, models.bbnc, models.gp, models.bnb, models.nich, models.dd(5), models.niw(3), models.dm(4)):
        r = probe.probe(d.c_desc())
        assert (r["type"], r["n"], r["vec"]) == want_type[d.name()], d.name()
        assert r["hp"] == r["hp_via_base"]                          # _base.create_hypers() == get().create_hypers()
        assert r["use_count"] >= 2                                  # the caller shares ownership with the descriptor
        # the bags are the wire format the Python descriptors speak
        hp = wire.loads(d.name() + ".shared", r["hp"])
        ss = wire.loads(d.name() + ".group", r["ss"])
        if d.name() == "nich":
            assert hp == {"mu": 0.0, "kappa": 1.0, "sigmasq": 1.0, "nu": 1.0}
            assert ss == {"count": 0, "mean": 0.0, "count_times_variance": 0.0}
        if d.name() == "dd":
            assert list(hp["alphas"]) == [1.0] * 5 and list(ss["counts"]) == [0] * 5
        if d.name() == "niw":
            assert hp["nu"] == 3.0 and len(hp["mu"]) == 3 and len(hp["psi"]) == 9 and ss["count"] == 0
