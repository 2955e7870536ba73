"""Cython boundary of the component-model plugin surface -- this build's own counterpart of the reference's
microscopes/_models.pxd / _models.pyx / _models_h.pxd (never copied, never run: Python-2 Cython against absent
libraries).  Downstream Cython state objects `cimport` the extension types from here

    from common_amd.cy._models cimport _base          # upstream: from microscopes._models cimport _base

and call `desc.c_desc().get()` / `.create_hypers()` exactly as they do upstream; the C++ objects behind the pointers
are include/microscopes/models/*.hpp, whose arithmetic runs on the device through the C ABI.

The extension is built in-tree by __graft_entry__.build() (common_amd/cy/build.py); importing it needs
libmicroscopes_hip.so but no GPU (constructing models / hypers / groups touches no device; scoring does).
"""
